#!/bin/bash
# One round's profile set on the GPU box (everything lands under gpurun_out/prof_$1/):
#   bench line + per-kernel times, rocprofv3 kernel-trace stats of the same command, ragged / eager bench lines,
#   SQ counters of the attention kernels, HBM counters of the attention backward kernels.
tag=${1:-r04}
out=gpurun_out/prof_$tag; mkdir -p $out
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python bench.py --kernel-times > $out/bench.json 2> $out/kernel_times.txt
python bench.py --no-cpu-baseline --no-legs --ragged > $out/bench_ragged.json 2>/dev/null
# the mid-size regime: per-kernel table at L = 4 096 on the batched schedule, and the line as the step runs it
MT_SPLIT_PASSES=0 python bench.py --patches 4096 --kernel-times --no-legs --no-cpu-baseline > $out/bench_L4096_batched.json 2> $out/L4096_kernel_times.txt
python bench.py --patches 4096 --no-legs --no-cpu-baseline > $out/bench_L4096.json 2>/dev/null
python bench.py --pathways real --no-legs --no-cpu-baseline > $out/bench_real_pathways.json 2>/dev/null
python bench.py --no-cpu-baseline --no-legs --eager > $out/bench_eager.json 2>/dev/null
# the data-parallel step on a ONE-rank RCCL communicator (every collective issued; three schedules captured and timed): toy and real pathways
python bench.py --dp-rehearsal --no-cpu-baseline > $out/bench_dp_rehearsal_rccl.json 2>/dev/null
python bench.py --dp-rehearsal --pathways real --no-cpu-baseline > $out/bench_dp_rehearsal_rccl_real_pathways.json 2>/dev/null
# kernel stats on the BATCHED schedule (every kernel alone on the chip: what `roofline` / `roofline_kernels` are taken on) ...
export MT_SPLIT_PASSES=0
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-legs > $out/stats.log 2>&1
unset MT_SPLIT_PASSES
# ... and of the default command (task passes as two groups: under the profiler the two graph branches run one after the other,
# so this prices the groups' smaller launches, not their overlap)
rocprofv3 --kernel-trace --stats -d $out/stats_groups -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-legs > $out/stats_groups.log 2>&1
find $out/stats_groups -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_pass_groups.csv \;
bash tools/pmc.sh ${tag}_fwd tools/fwd_microbench.py
bash tools/pmc.sh ${tag}_bwd tools/kv_microbench.py
bash tools/pmc_hbm.sh ${tag}_bwd tools/kv_microbench.py
python tools/pmc_summary.py gpurun_out/pmc_${tag}_fwd > $out/pmc_attn_fwd.txt
python tools/pmc_summary.py gpurun_out/pmc_${tag}_bwd > $out/pmc_attn_bwd.txt
python tools/pmc_summary.py gpurun_out/pmc_${tag}_bwd/f > $out/pmc_hbm_f.txt; python tools/pmc_summary.py gpurun_out/pmc_${tag}_bwd/w > $out/pmc_hbm_w.txt
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# the reference trainer's own loop on the drop-in nn.Module (3 model calls, torch loss / GradScaler / AdamW)
python bench.py --api module --optim fused > $out/bench_module.json 2>/dev/null
python bench.py --api module --optim torch > $out/bench_module_torch_adamw.json 2>/dev/null
MT_MODULE_GRAPH=0 python bench.py --api module --optim fused --no-cpu-baseline > $out/bench_module_eager_bridge.json 2>/dev/null
MT_MODULE_GRAPH=0 python bench.py --api module --optim torch --no-cpu-baseline > $out/bench_module_torch_adamw_eager_bridge.json 2>/dev/null
python tools/diag/module_phases.py 10000 fused host > $out/module_phases_host.txt 2>/dev/null
python tools/diag/module_phases.py > $out/module_phases.txt 2>/dev/null
# BASELINE config 4 (TITAN configuration, mixed bag lengths): bench line + rocprofv3 kernel-trace stats of the same command
python bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 > $out/titan_bench.json 2> $out/titan_bench.err
rocprofv3 --kernel-trace --stats -d $out/titan_stats -o s --output-format csv -- python3 bench.py --config titan --patches 4096 --ragged --steps 16 --warmup 8 --no-cpu-baseline > $out/titan_stats.log 2>&1
find $out/titan_stats -name "*kernel_stats.csv" -exec cp {} $out/titan_kernel_stats.csv \;
# SQ + HBM counters of the dense ALiBi attention kernels (TITAN configuration, N = 4097, 3 passes)
bash tools/pmc.sh ${tag}_dense tools/dense_microbench.py
bash tools/pmc_hbm.sh ${tag}_dense tools/dense_microbench.py
python tools/pmc_summary.py gpurun_out/pmc_${tag}_dense dense_attn > $out/pmc_dense_attn.txt
# GEMM: SQ counters over the five backbone shapes (persistent kernel on four of them), the same with the ping-pong kernel everywhere,
# cold / warm A/B against it, tile-order sweep, slice stamps
bash tools/pmc.sh ${tag}_gemm tools/gemm_microbench.py
python tools/pmc_summary.py gpurun_out/pmc_${tag}_gemm gemm_nt > $out/pmc_gemm_nt.txt
python tools/gemm_cold_bench.py > $out/gemm_cold_bench.txt 2>/dev/null
python tools/gemm_gc_sweep.py > $out/gemm_gc_sweep.txt 2>/dev/null
python tools/gemm_yard_cold.py > $out/gemm_hipblaslt_yard.txt 2>/dev/null
python tools/experiments/gemm_ps_stamp.py 3072 768 > $out/gemm_ps_stamp.txt 2>/dev/null
PS_COLD=1 python tools/experiments/gemm_ps_stamp.py 3072 768 >> $out/gemm_ps_stamp.txt 2>/dev/null
echo profile set done
