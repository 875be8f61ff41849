cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/train3; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python tools/time_train.py 48 256 5 > $O/times.txt 2>&1
python tools/rocpd_stats.py $O/t/t_results.db > $O/r3_train_kernel_stats.md
find $O -name "*.db" -delete
head -24 $O/r3_train_kernel_stats.md | cut -c1-160
