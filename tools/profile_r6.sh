#!/bin/bash
# Round-6 profile collection on the GPU box (one gpurun call; counters in their own passes, never combined with other trace domains).
# usage: bash tools/profile_r6.sh [conv|tasks|all]      summaries under gpurun_out/prof_r6/ (copy what is to be judged into profiles/)
# r6: the DEFAULT family is fp32 arithmetic (conv_mode 0): `bench.py` with no option profiles it; the half-split fast mode is `--ctx-option conv_mode=1`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r6
mkdir -p $O
WHAT=${1:-all}
R="rocprofv3 --kernel-trace"
BOPT="--steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fast-mode"
if [ $WHAT = conv ] || [ $WHAT = all ]; then
  # the headline leg (default family: fp32 Winograd, two launch chains) ...
  $R --stats -d $O/bench -o bench -- python bench.py $BOPT > $O/bench.log 2>&1
  python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r6_bench_kernel_stats.md
  grep '^{' $O/bench.log | tail -1 > $O/r6_bench_profiled.json
  # ... the same with ONE launch chain (with two, kernels of the two chains overlap pairwise and their durations sum to about twice the wall
  # time; this trace is the serial per-kernel view) ...
  $R --stats -d $O/bench1 -o bench -- python bench.py $BOPT --ctx-option fp32_chains=0 > $O/bench1.log 2>&1
  python tools/rocpd_stats.py $O/bench1/bench_results.db > $O/r6_bench_kernel_stats_chains1.md
  grep '^{' $O/bench1.log | tail -1 > $O/r6_bench_profiled_chains1.json
  # ... and the fast mode (half-split f16 x 3)
  $R --stats -d $O/benchhs -o bench -- python bench.py $BOPT --ctx-option conv_mode=1 > $O/benchhs.log 2>&1
  python tools/rocpd_stats.py $O/benchhs/bench_results.db > $O/r6_bench_kernel_stats_fast_mode.md
  grep '^{' $O/benchhs.log | tail -1 > $O/r6_bench_profiled_fast_mode.json
  for M in 0 1; do
    D="python tools/run_denoiser.py 48 256 1 $M"
    $R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/den_sq$M -o p -- $D > $O/den_sq$M.log 2>&1
    $R --pmc FETCH_SIZE -d $O/den_fetch$M -o p -- $D > $O/den_fetch$M.log 2>&1
    $R --pmc WRITE_SIZE -d $O/den_write$M -o p -- $D > $O/den_write$M.log 2>&1
    if [ $M = 0 ]; then
      python tools/pmc_report.py $O/den_sq$M/p_results.db $O/den_fetch$M/p_results.db $O/den_write$M/p_results.db --json $O/r6_pmc_traffic_fp32.json --geom 48 256 256 --fetch-x2 'NO_KERNEL' --count 'conv3x3' > $O/r6_denoiser_pmc_fp32.md
    else
      python tools/pmc_report.py $O/den_sq$M/p_results.db $O/den_fetch$M/p_results.db $O/den_write$M/p_results.db --json $O/r6_pmc_traffic.json --geom 48 256 256 > $O/r6_denoiser_pmc_hs.md
    fi
  done
  python tools/ab_ksplit.py 256 > $O/r6_ksplit.txt 2>&1
  python tools/drift_seeds.py 12 > $O/r6_drift_seeds.md 2> $O/drift.err
fi
if [ $WHAT = tasks ] || [ $WHAT = all ]; then
  T="python tools/bench_tasks.py"
  # UNPROFILED timing first (r5_tasks_times.txt was a profiled run's log)
  $T 2>/dev/null | grep -v amdgpu.ids > $O/r6_tasks_times.txt
  $R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
  $R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
  $R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
  python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r6_tasks_pmc_all.md
  python tools/pmc_tasks_summary.py $O/r6_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_|wino8_ksplit" > $O/r6_tasks_pmc.md
  rm -f $O/r6_tasks_pmc_all.md
  python tools/time_train.py 48 256 5 1 --fused-only 2>/dev/null | grep -v amdgpu.ids > $O/r6_train_times.txt
  python tools/time_train.py 48 256 5 0 --fused-only 2>/dev/null | grep -v amdgpu.ids >> $O/r6_train_times.txt
  python tools/time_drunet_modes.py 48 256 2>/dev/null | grep -v amdgpu.ids > $O/r6_drunet_times.txt
fi
find $O -name "*.db" -delete
ls $O
