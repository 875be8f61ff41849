#!/bin/bash
# Round-3 profile collection on the GPU box (one gpurun call): kernel traces + separate --pmc passes (never combined with
# trace domains other than --kernel-trace).  Summaries land under gpurun_out/prof_r3/ and are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r3
rm -rf $O; mkdir -p $O
R="rocprofv3 --kernel-trace"
BOPT="--steps 2 --warmup 1 --no-cpu-baseline --no-batch-table --no-fp32-mode"
# 1. the bench command itself: default (half-split) and exact-fp32 convolution families
$R --stats -d $O/bench -o bench -- python bench.py $BOPT > $O/bench.log 2>&1
python tools/rocpd_stats.py $O/bench/bench_results.db > $O/r3_bench_kernel_stats.md
grep '^{' $O/bench.log | tail -1 > $O/r3_bench_profiled.json
$R --stats -d $O/bench32 -o bench -- python bench.py $BOPT --ctx-option conv_mode=0 > $O/bench32.log 2>&1
python tools/rocpd_stats.py $O/bench32/bench_results.db > $O/r3_bench_kernel_stats_fp32.md
grep '^{' $O/bench32.log | tail -1 > $O/r3_bench_profiled_fp32.json
# 2. one denoiser forward: SQ / GRBM, FETCH, WRITE passes
D="python tools/run_denoiser.py 48 256 1"
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/den_sq -o p -- $D > $O/den_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/den_fetch -o p -- $D > $O/den_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/den_write -o p -- $D > $O/den_write.log 2>&1
python tools/pmc_report.py $O/den_sq/p_results.db $O/den_fetch/p_results.db $O/den_write/p_results.db --json $O/r3_pmc_traffic.json --geom 48 256 256 > $O/r3_denoiser_pmc_hs.md
# 3. the prox / update kernels of all four tasks at the BASELINE sizes
T="python tools/bench_tasks.py"
$R --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/task_sq -o p -- $T > $O/task_sq.log 2>&1
$R --pmc FETCH_SIZE -d $O/task_fetch -o p -- $T > $O/task_fetch.log 2>&1
$R --pmc WRITE_SIZE -d $O/task_write -o p -- $T > $O/task_write.log 2>&1
python tools/pmc_report.py $O/task_sq/p_results.db $O/task_fetch/p_results.db $O/task_write/p_results.db --fetch-x2 'conv_hs' > $O/r3_tasks_pmc_all.md
python tools/pmc_tasks_summary.py $O/r3_tasks_pmc_all.md "conv_hs|conv3x3|conv_first|upsample2x|prep_input|maxpool|outc_" > $O/r3_tasks_pmc.md
rm -f $O/r3_tasks_pmc_all.md
cp $O/task_sq.log $O/r3_tasks_times.txt
# 4. DRUNet forward (B = 48, 256^2)
$R --stats -d $O/dru -o t -- python tools/time_drunet.py 48 256 3 > $O/r3_drunet_times.txt 2>&1
python tools/rocpd_stats.py $O/dru/t_results.db > $O/r3_drunet_kernel_stats.md
# 5. conv_hs ablations on real data (tuning build): HBM reads served from cache (16), + no stores (17), + no epilogue (18)
for a in 0 1 2 16 17 18; do
  echo "## abl=$a" >> $O/r3_conv_ablation_hbm.md
  PNPX_LIB=tfpnp_amd/libpnpx_tune.so python tools/profile_layers.py 48 256 1 --brief --abl=$a 2>/dev/null | grep -v amdgpu >> $O/r3_conv_ablation_hbm.md
done
# 6. launch chains / weights-in-registers A/B on whole forwards
python tools/ab_wall.py "chains=1,wreg=0;chains=1,wreg=2;chains=2,wreg=2" 48 256 2>/dev/null | grep forward > $O/r3_ab_wall.txt
python tools/ab_wall.py "chains=1;chains=2;chains=3" 6 256 2>/dev/null | grep forward >> $O/r3_ab_wall.txt
python tools/ab_wall.py "chains=1;chains=2" 24 256 2>/dev/null | grep forward >> $O/r3_ab_wall.txt
find $O -name "*.db" -delete
ls -la $O
tail -1 $O/bench.log | cut -c1-300
