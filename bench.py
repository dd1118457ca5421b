#!/usr/bin/env python3
"""bench.py — slides/sec of the Modal-Adapter train step (BASELINE.json metric) on N MI355X.

A "step" = one slide exactly as train_modaltune.py:195-240: frozen text projector, 3 task passes of the full
model (frozen 12-layer LongNet backbone + Modal Adapter), KL distillation loss, backward, gradient all-reduce
(N > 1), AdamW.  Inputs are synthetic and resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)


def cpu_baseline_leg(cfg, sizes, L, seed, max_seconds=20.0):
    """Times the CPU oracle (a port of the reference arithmetic, fp32, all host cores) on a bounded sample of the
    workload: ONE frozen LongNet layer forward+backward at N = L+1 tokens, 1 task pass; a slide step is 36 such
    layer passes (3 tasks x 12 layers; >= 98 % of the step FLOPs) -> slides/s = 1 / (36 t)."""
    from oracle import modaltune_oracle as O     # CPU baseline leg: the oracle as the thing timed, nothing else
    from modaltune_amd import synth
    from modaltune_amd.config import segment_lengths
    cores = min(os.cpu_count() or 1, 64)      # threads actually used (more threads do not help these shapes)
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, sizes, seed).items() if k.startswith("encoder.layers.0.")}
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, L + 1, cfg.embed_dim, generator=g).requires_grad_(True)
    segs = segment_lengths(cfg.max_wsi_size, cfg.tile_size)
    times = []
    t_all = time.time()
    for it in range(12):                      # 1 warm-up + timed repeats, bounded to ~max_seconds of CPU work
        t0 = time.time()
        y = O.encoder_layer(x, sd, "encoder.layers.0", segs, (1, 2, 4, 8, 16))
        y.sum().backward()
        if it > 0:
            times.append(time.time() - t0)
        x.grad = None
        if time.time() - t_all > max_seconds and times:
            break
    t = min(times)
    return {"value": 1.0 / (36.0 * t), "unit": "slides/s", "cores": cores, "kind": "port",
            "sample": f"oracle (fp32 torch-CPU port, {cores} threads, best of {len(times)}), 1 LongNet layer fwd+bwd at N={L + 1}, 1 task pass: {t:.2f} s; "
                      f"step = 36 layer passes (3 tasks x 12 layers) -> {36 * t:.1f} s/slide"}


# HBM bytes per launch of the dominant kernel from PMC passes that cannot run inside this script (separate rocprofv3
# runs, one counter per pass): {(patches, tokens): bytes}.  FETCH_SIZE 373 139 KB, WRITE_SIZE 174 393 KB at L = 10 000.
RECORDED_KV_TRAFFIC = {(10000, 65): 2 * 331608 * 1024 + 174393 * 1024}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=10000)
    ap.add_argument("--pathways", default="6", help="number of toy pathways (sizes 5, 6, ...) or 'real': the reference's 331-pathway "
                                                    "grouping sizes (tests/golden/pathway_sizes_331.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true", help="run the step with Dropout / DropPath off (the parity configuration); "
                                                             "default: on, as model.train() leaves them in the reference")
    ap.add_argument("--kernel-times", action="store_true", help="print the per-kernel time table (stderr)")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("MT_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # MT_BENCH_DEVICE: rehearsals only
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)

    from modaltune_amd import ops, synth
    from modaltune_amd.config import ModelConfig, flops_per_slide_step
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep

    L = args.patches
    cfg = ModelConfig()                       # Prov-GigaPath ModalAdapter config (modaltune_gigapath_config.json)
    if args.pathways == "real":
        sizes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "pathway_sizes_331.json")))
    else:
        sizes = synth.toy_group_sizes(int(args.pathways))
    args.pathways = len(sizes)
    eng = Engine(cfg, sizes, dev)
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))      # identical weights on every rank
    eng.set_stochastic(not args.no_dropout, seed=20260 + rank)     # Dropout(0.25) / DropPath(<= 0.1): counter-based masks
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(0))
    # synthetic slides, distinct per rank, resident in HBM (2 alternating slides per rank)
    slides = []
    for j in range(2):
        inp = synth.synth_inputs(L, sizes, seed=1000 + 17 * rank + j, grid=128 if L <= 128 * 128 else 512)
        slides.append((torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), inp["coords"],
                       [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev)))

    def run(n, graphed=True):
        for i in range(n):
            x, coords, genes, text = slides[i % 2]
            if graphed and not args.eager:
                ts.step_graphed(x, coords, genes, text)      # hipGraph replay (2 eager warm-ups + capture happen in warm-up)
            else:
                ts.step(x, coords, genes, text, update=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    try:
        run(max(args.warmup, 3))      # >= 3: two eager steps + the capture step of the graph path
    except Exception as e:            # graph capture unavailable -> same arithmetic with eager launches
        if args.eager:                # (every rank runs the same code on the same hardware: the fallback is symmetric)
            raise
        print(f"[bench] hipGraph path failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
        args.eager = True
        ts._graphs, ts._gwarm, ts._gkey = None, 0, None
        torch.cuda.synchronize()
        run(max(args.warmup, 3))
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    loss = float(ts.loss)
    # per-kernel durations: HIP events on the launch stream around every launch of an eager, instrumented pass of the
    # same step (events cannot be recorded inside a replayed graph); not part of the timed region above
    prof_steps = min(args.steps, 3)
    ops.TIMER = {}
    run(prof_steps, graphed=False)
    barrier()
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    skipped = max(args.warmup, 3) + args.steps + prof_steps - int(ts.step_dev)

    if rank == 0:
        T = cfg.num_tokens
        fl = flops_per_slide_step(L, T)
        value = args.steps * world / dt
        summ = ops.timer_summary(timer)
        tot_ms = sum(v[1] for v in summ.values())
        if args.kernel_times:
            for k, (n, ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:28s} launches/step {n / prof_steps:7.1f}  ms/step {ms / prof_steps:9.3f}  ({100 * ms / tot_ms:5.1f} %)",
                      file=sys.stderr)
        # roofline of the dominant kernel (largest share of the step): the dK/dV kernel of the dilated-attention backward,
        # one launch per layer per step over all 5 branches and all 3 task passes.  Algorithmic FLOPs per launch: the four
        # products it owns (S = Q K^T, dP = dO V^T, dV = P^T dO, dK = dS^T Q) = 2 x the forward's two products
        # (SURVEY §8d counts the whole flash backward as 2.5 x forward; the dQ kernel carries the remaining 0.5 x).
        n_l, ms = summ["dilated_attn_bwd_kv"]
        launch_flops = 2.0 * 3 * fl["attn_layer"]
        achieved = launch_flops / (ms / n_l * 1e-3) / 1e12
        exec_flops = 2.0 * 3 * fl["attn_layer_executed"]      # zero-padded tiles are skipped, not computed
        out = {
            "metric": "slides/sec (train step) at 10k patches x 1536-d", "value": value, "unit": "slides/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Prov-GigaPath ModalAdapter train step, {L} patches x 1536-d, {args.pathways} pathways -> "
                                   f"{T - 1} gene tokens + 1 task token, 3 task passes, fp16 operands / fp32 accumulate, "
                                   f"1 slide per GPU per step, train mode: "
                                   + ("dropout / drop-path off (parity configuration)" if args.no_dropout else
                                      f"Dropout({cfg.dropout}) on the embedded input and both backbone branches, DropPath(0..{cfg.drop_path_rate}) "
                                      f"per layer and on the Extractor FFN (Philox masks regenerated in backward)"),
                       "patches": L, "tokens": T, "parallelism": f"dp{world}", "dropout": not args.no_dropout},
            "loss": loss, "skipped_steps": skipped,
            "step_tflops": fl["step"] / 1e12, "step_mfma_frac": fl["step"] * value / world / 1e12 / PEAK_F16_MFMA_TFLOPS,
            "roofline": {"kernel": "dilated_attn_bwd_kv_kernel", "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F16_MFMA_TFLOPS,
                         "traffic": RECORDED_KV_TRAFFIC.get((L, T)),
                         "avg_launch_ms": ms / n_l, "flops_per_launch": launch_flops,
                         "flops_executed_per_launch": exec_flops,
                         "mfma_frac_executed": exec_flops / (ms / n_l * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                         "traffic_source": ("recorded rocprofv3 --pmc passes (profiles/r01_pmc_hbm_attn_bwd.txt): 2 x FETCH_SIZE "
                                            "(gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, bytes per launch"
                                            if (L, T) in RECORDED_KV_TRAFFIC else None),
                         "measured": f"HIP events around each launch, eager instrumented pass of {prof_steps} steps after the timed region"},
            "launch": "eager" if args.eager else "hipGraph replay",
            "kernel_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:12]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(cfg, sizes, L, seed=0)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
