#!/bin/bash
# Round-5 profile collection on the GPU box (one gpurun call; counters in their own passes, never combined with other trace domains).
# usage: bash tools/profile_r5.sh [conv|tasks|all]      summaries under gpurun_out/prof_r5/ (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r5
mkdir -p $O
WHAT=${1:-all}
R="rocprofv3 --kernel-trace"
BOPT="--steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fp32-mode"
if [ $WHAT = conv ] || [ $WHAT = all ]; then
  # FETCH_SIZE / WRITE_SIZE calibration on known byte counts (16-byte loads, dword LDS-DMA streams, the halo gather)
  $R --pmc FETCH_SIZE -d $O/cal_fetch -o p -- tools/micro/fetch_calib > $O/calib.log 2>&1
  $R --pmc WRITE_SIZE -d $O/cal_write -o p -- tools/micro/fetch_calib >> $O/calib.log 2>&1
  python - > $O/r5_fetch_calibration.md <<PY
import sqlite3
print("# r5: FETCH_SIZE / WRITE_SIZE against known byte counts (tools/micro/fetch_calib.hip; rocprofv3 --pmc, one counter per pass)\n")
print(open("$O/calib.log").read().split("bytes:")[1].split("\n")[0].join(["bytes:", "\n"]))
print("| kernel | counter | value (KiB, summed over the XCDs) | value x 1 KiB / 1 GiB |\n|---|---|---|---|")
for db, cn in (("$O/cal_fetch/p_results.db", "FETCH_SIZE"), ("$O/cal_write/p_results.db", "WRITE_SIZE")):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? group by dispatch_id order by dispatch_id", (cn,)).fetchall()
    for did, kn, v in rows[-5:]:
        if ("write" in kn) == (cn == "WRITE_SIZE"):
            print(f"| {kn.split('(')[0]} | {cn} | {v:.0f} | {v * 1024 / 2**30:.3f} |")
PY
  $R --stats -d $O/bench -o bench -- python bench.py $BOPT > $O/bench.log 2>&1
  python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r5_bench_kernel_stats.md
  grep '^{' $O/bench.log | tail -1 > $O/r5_bench_profiled.json
  # the same half-split bench with ONE launch chain: with two (the default since r5) kernels of the two chains overlap pairwise and their
  # durations sum to about twice the wall time; this trace is the serial per-kernel view
  $R --stats -d $O/bench1 -o bench -- python bench.py $BOPT --ctx-option chains=1 > $O/bench1.log 2>&1
  python tools/rocpd_stats.py $O/bench1/bench_results.db > $O/r5_bench_kernel_stats_chains1.md
  grep '^{' $O/bench1.log | tail -1 > $O/r5_bench_profiled_chains1.json
  $R --stats -d $O/bench32 -o bench -- python bench.py $BOPT --ctx-option conv_mode=0 > $O/bench32.log 2>&1
  python tools/rocpd_stats.py $O/bench32/bench_results.db > $O/r5_bench_kernel_stats_fp32.md
  grep '^{' $O/bench32.log | tail -1 > $O/r5_bench_profiled_fp32.json
  for M in 0 1; do
    D="python tools/run_denoiser.py 48 256 1 $M"
    S=$([ $M = 0 ] && echo fp32 || echo hs)
    $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/den_sq$M -o p -- $D > $O/den_sq$M.log 2>&1
    $R --pmc FETCH_SIZE -d $O/den_fetch$M -o p -- $D > $O/den_fetch$M.log 2>&1
    $R --pmc WRITE_SIZE -d $O/den_write$M -o p -- $D > $O/den_write$M.log 2>&1
    if [ $M = 0 ]; then
      python tools/pmc_report.py $O/den_sq$M/p_results.db $O/den_fetch$M/p_results.db $O/den_write$M/p_results.db --json $O/r5_pmc_traffic_fp32.json --geom 48 256 256 --fetch-x2 'NO_KERNEL' --count 'conv3x3' > $O/r5_denoiser_pmc_fp32.md
    else
      python tools/pmc_report.py $O/den_sq$M/p_results.db $O/den_fetch$M/p_results.db $O/den_write$M/p_results.db --json $O/r5_pmc_traffic.json --geom 48 256 256 > $O/r5_denoiser_pmc_hs.md
    fi
  done
fi
if [ $WHAT = tasks ] || [ $WHAT = all ]; then
  T="python tools/bench_tasks.py"
  $R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
  $R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
  $R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
  python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r5_tasks_pmc_all.md
  python tools/pmc_tasks_summary.py $O/r5_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_" > $O/r5_tasks_pmc.md
  rm -f $O/r5_tasks_pmc_all.md
  cp $O/task_sq.log $O/r5_tasks_times.txt
  $R --stats -d $O/train -o t -- python tools/time_train.py 48 256 5 > $O/r5_train_times.txt 2>&1
  python tools/rocpd_stats.py $O/train/t_results.db > $O/r5_train_kernel_stats.md
  # the same training step in the fp32 family (r5: adjoint convolutions on the 8-wave Winograd kernel) and the DRUNet's two families
  python tools/time_train.py 48 256 5 0 --fused-only >> $O/r5_train_times.txt 2>&1
  python tools/time_drunet_modes.py 48 256 > $O/r5_drunet_times.txt 2>&1
fi
find $O -name "*.db" -delete
ls $O
