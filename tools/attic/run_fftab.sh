cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/fftab; rm -rf $O; mkdir -p $O
for a in 0 1; do
  rocprofv3 --kernel-trace --stats -d $O/s$a -o t -- python tools/run_csmri.py $a 4 > $O/run$a.log 2>&1
  python tools/rocpd_stats.py $O/s$a/t_results.db | grep -i "fft_\|kernel |" > $O/stats$a.md
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$a -o p -- python tools/run_csmri.py $a 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w$a -o p -- python tools/run_csmri.py $a 2 > /dev/null 2>&1
  python tools/pmc_report.py $O/f$a/p_results.db $O/f$a/p_results.db $O/w$a/p_results.db > $O/pmc$a.md 2>&1
  grep -i "fft_" $O/pmc$a.md | head -8 > $O/pmc_fft$a.md
done
cat $O/run0.log $O/run1.log | grep checksum; cat $O/stats0.md; cat $O/stats1.md; cat $O/pmc_fft0.md; cat $O/pmc_fft1.md
find $O -name "*.db" -delete
