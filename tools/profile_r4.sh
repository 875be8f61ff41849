#!/bin/bash
# Round-4 refresh: the bench command's kernel trace, the task kernels' counter passes, one training step.
# Same recipes as tools/profile_r3.sh sections 1 and 3 + tools/run_train_prof.sh; summaries under gpurun_out/prof_r4/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r4
rm -rf $O; mkdir -p $O
R="rocprofv3 --kernel-trace"
BOPT="--steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fp32-mode"
$R --stats -d $O/bench -o bench -- python bench.py $BOPT > $O/bench.log 2>&1
python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r4_bench_kernel_stats.md
grep '^{' $O/bench.log | tail -1 > $O/r4_bench_profiled.json
# the fp32 family (conv_mode 0: Winograd + direct fp32 MFMA kernels) through the same command
$R --stats -d $O/bench32 -o bench -- python bench.py $BOPT --ctx-option conv_mode=0 > $O/bench32.log 2>&1
python tools/rocpd_stats.py $O/bench32/bench_results.db > $O/r4_bench_kernel_stats_fp32.md
grep '^{' $O/bench32.log | tail -1 > $O/r4_bench_profiled_fp32.json
# one denoiser forward of the fp32 family: SQ / GRBM, FETCH, WRITE passes (the Winograd kernel's halo gathers are 4 B per lane and its
# weights stay in L2: no FETCH doubling); feeds fp32_mode.roofline.traffic
D="python tools/run_denoiser.py 48 256 1 0"
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/den_sq -o p -- $D > $O/den_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/den_fetch -o p -- $D > $O/den_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/den_write -o p -- $D > $O/den_write.log 2>&1
python tools/pmc_report.py $O/den_sq/p_results.db $O/den_fetch/p_results.db $O/den_write/p_results.db --json $O/r4_pmc_traffic_fp32.json --geom 48 256 256 --fetch-x2 'NO_KERNEL' --count 'conv3x3' > $O/r4_denoiser_pmc_fp32.md
T="python tools/bench_tasks.py"
$R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r4_tasks_pmc_all.md
python tools/pmc_tasks_summary.py $O/r4_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_" > $O/r4_tasks_pmc.md
rm -f $O/r4_tasks_pmc_all.md
cp $O/task_sq.log $O/r4_tasks_times.txt
$R --stats -d $O/train -o t -- python tools/time_train.py 48 256 5 > $O/r4_train_times.txt 2>&1
python tools/rocpd_stats.py $O/train/t_results.db > $O/r4_train_kernel_stats.md
find $O -name "*.db" -delete
tail -3 $O/r4_train_times.txt; head -12 $O/r4_tasks_pmc.md | cut -c1-200
