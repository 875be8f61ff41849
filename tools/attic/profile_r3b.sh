#!/bin/bash
# Round-3 refresh (end-of-round tree): the bench command's kernel trace, the task kernels' counter passes, one training step.
# Same recipes as tools/profile_r3.sh sections 1 and 3 + tools/run_train_prof.sh; summaries under gpurun_out/prof_r3b/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r3b
rm -rf $O; mkdir -p $O
R="rocprofv3 --kernel-trace"
BOPT="--steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fp32-mode"
$R --stats -d $O/bench -o bench -- python bench.py $BOPT > $O/bench.log 2>&1
python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r3_bench_kernel_stats.md
grep '^{' $O/bench.log | tail -1 > $O/r3_bench_profiled.json
T="python tools/bench_tasks.py"
$R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r3_tasks_pmc_all.md
python tools/pmc_tasks_summary.py $O/r3_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_" > $O/r3_tasks_pmc.md
rm -f $O/r3_tasks_pmc_all.md
cp $O/task_sq.log $O/r3_tasks_times.txt
$R --stats -d $O/train -o t -- python tools/time_train.py 48 256 5 > $O/r3_train_times.txt 2>&1
python tools/rocpd_stats.py $O/train/t_results.db > $O/r3_train_kernel_stats.md
find $O -name "*.db" -delete
tail -3 $O/r3_train_times.txt; head -12 $O/r3_tasks_pmc.md | cut -c1-200
