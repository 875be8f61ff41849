cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/ffttile; rm -rf $O; mkdir -p $O
for t in 512 1024 2048 4096; do
  FFT_TILE=$t rocprofv3 --kernel-trace --stats -d $O/s$t -o t -- python tools/run_csmri.py 1 4 > $O/run$t.log 2>&1
  echo "tile $t: $(grep checksum $O/run$t.log)"; python tools/rocpd_stats.py $O/s$t/t_results.db | grep -i "fft_" | cut -c1-60,130-190
done
find $O -name "*.db" -delete
