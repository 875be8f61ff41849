cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pol3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python tools/time_policy.py > $O/times.txt 2>&1
python tools/rocpd_stats.py $O/t/t_results.db > $O/r3_policy_kernel_stats.md
find $O -name "*.db" -delete
grep pnpx $O/r3_policy_kernel_stats.md | head -24 | cut -c1-150
