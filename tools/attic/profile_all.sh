#!/bin/bash
# Round-2 profile collection on the GPU box (gpurun): kernel traces + separate --pmc passes (never combined with trace
# domains other than --kernel-trace).  Raw rocpd databases land under gpurun_out/prof_r2/; the summaries are copied to
# profiles/ by hand after inspection.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r2
rm -rf $O; mkdir -p $O
R="rocprofv3 --kernel-trace"
# 1. the bench command itself (kernel trace + stats)
$R --stats -d $O/bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fp32-mode > $O/bench.log 2>&1
# 2. one denoiser forward: SQ / GRBM, FETCH, WRITE passes
D="python tools/run_denoiser.py 48 256 1"
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/den_sq -o p -- $D > $O/den_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/den_fetch -o p -- $D > $O/den_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/den_write -o p -- $D > $O/den_write.log 2>&1
# 3. the prox / update kernels of all four tasks at the BASELINE sizes
T="python tools/bench_tasks.py"
$R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
# 4. the training path (fused ADMM forward + VJP, activation ring on / off, composed path)
$R --stats -d $O/train -o t -- python tools/time_train.py 48 256 5 > $O/r2_train_times.txt 2>&1
python tools/rocpd_stats.py $O/train/t_results.db > $O/r2_train_kernel_stats.md
# 5. power / clock telemetry while the denoiser loops
bash tools/power_probe.sh $O/r2_power_probe.txt > /dev/null 2>&1
# summaries (the raw databases exceed the 64 MiB that travel back)
python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r2_bench_kernel_stats.md
grep '^{' $O/bench.log | tail -1 > $O/r2_bench_profiled.json
python tools/pmc_report.py $O/den_sq/p_results.db $O/den_fetch/p_results.db $O/den_write/p_results.db --json $O/r2_pmc_traffic.json --geom 48 256 256 > $O/r2_denoiser_pmc_hs.md
python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r2_tasks_pmc_all.md
python tools/pmc_tasks_summary.py $O/r2_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_" > $O/r2_tasks_pmc.md
rm -f $O/r2_tasks_pmc_all.md
cp $O/task_sq.log $O/r2_tasks_times.txt
find $O -name "*.db" -delete
ls -la $O
tail -2 $O/bench.log | cut -c1-300
