#!/usr/bin/env python3
"""bench.py — slides/sec of the Modal-Adapter train step (BASELINE.json metric) on N MI355X.

A "step" = one slide exactly as train_modaltune.py:195-240: frozen text projector, 3 task passes of the full
model (frozen 12-layer LongNet backbone + Modal Adapter), KL distillation loss, backward, bucketed gradient
all-reduce started from inside the backward (N > 1), AdamW.  Inputs are synthetic and resident in HBM before the
timed region.  One JSON line on rank 0.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself: N fresh child processes
(one per GPU, rendezvous on 127.0.0.1) are spawned BEFORE this process touches the GPU; under torchrun (RANK /
WORLD_SIZE set) the process is one of the ranks.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (MI355X_MICROARCH.md; 6.3 TB/s is what a copy kernel reaches)
# the reference itself, timed where it can be imported (the survey/build container, 8 cores)
CPU_REFERENCE = {"value": 0.0054, "unit": "slides/s", "cores": 8, "kind": "reference",
                 "sample": "BASELINE.md §2: reference modules (fp32, dropout 0, aten CPU flash stand-in), 1 timed step at L = 10 000 on the "
                           "8-core build container: 185.6 s/slide (the reference cannot travel to the GPU box)"}
# rocprofv3 --pmc passes of the dominant kernel cannot run inside this script (one counter group per run, gpurun refuses
# tracing + PMC together): the committed summary file is parsed at run time instead of a literal
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_hbm_attn_bwd.txt")
PMC_TRAFFIC_FALLBACK = os.path.join(ROOT, "profiles", "r05_pmc_hbm_attn_bwd.txt")
PMC_TRAFFIC_DENSE = os.path.join(ROOT, "profiles", "r06_pmc_dense_attn.txt")
PMC_TRAFFIC_DENSE_FALLBACK = os.path.join(ROOT, "profiles", "r05_pmc_dense_attn.txt")      # tools/dense_microbench.py geometry: N = 4097, 3 passes


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=10000)
    ap.add_argument("--pathways", default="6", help="number of toy pathways (sizes 5, 6, ...) or 'real': the reference's 331-pathway "
                                                    "grouping sizes (tests/golden/pathway_sizes_331.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the short secondary legs the default 1-GPU run appends as sub-records "
                                                           "(`module_api`, `pcie_inclusive`, `sizes`, `titan`)")
    ap.add_argument("--cpu-baseline-child", type=int, default=0, help=argparse.SUPPRESS)      # internal: the CPU leg's own process
    ap.add_argument("--no-dropout", action="store_true", help="run the step with Dropout / DropPath off (the parity configuration); "
                                                             "default: on, as model.train() leaves them in the reference")
    ap.add_argument("--kernel-times", action="store_true", help="print the per-kernel time table (stderr)")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--ragged", action="store_true", help="a different bag length every step (8 lengths in [0.5, 1] x --patches): "
                                                          "the steady state of real data; runs the eager schedule")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--api", default="trainstep", choices=["trainstep", "module"],
                    help="trainstep: the fused TrainStep (3 task passes batched, device-side loss / GradScaler / AdamW, hipGraph replay); "
                         "module: the reference trainer's own loop on the drop-in nn.Module (train_modaltune.py:172-177,225-238): "
                         "3x model(...) -> torch KL loss -> GradScaler.scale(loss).backward() -> torch.optim.AdamW.step()")
    ap.add_argument("--optim", default="fused", choices=["fused", "torch"],
                    help="--api module: modaltune_amd.optim.AdamW (the second import of INTEGRATION.md section 1: torch.optim.AdamW's surface, one "
                         "fused launch on the model's flat buffers) or torch.optim.AdamW itself, as the reference trainer builds it")
    ap.add_argument("--dp-rehearsal", action="store_true",
                    help="--gpus 1 only: a ONE-rank process group on --backend (default nccl = RCCL) with MT_DP_REHEARSE=1 -- every collective of "
                         "the data-parallel step is issued (per-bucket all-reduce from inside the backward, reduce-scatter / all-gather of the "
                         "sharded bucket, flag MAX), the three schedules are captured and timed and `comm` is reported: what the data-parallel "
                         "machinery costs on one GPU, and the only way to run the RCCL branches on a one-GPU box")
    ap.add_argument("--config", default="gigapath", choices=["gigapath", "titan"],
                    help="gigapath: BASELINE config 2 (the headline metric); titan: BASELINE config 4 (TITAN backbone configuration, "
                         "--patches foreground cells, --ragged: mixed bag lengths) -- a separate JSON line, never the headline")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """Self-launch: spawn --gpus fresh rank processes before any GPU call in this one; relay rank 0's JSON line."""
    import torch      # (importing torch and counting devices does not initialise the GPU)
    n = args.gpus
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback for the hot path)")
    if n > ndev and args.backend == "nccl":
        raise RuntimeError(f"--gpus {n} with the RCCL backend needs {n} GPUs, found {ndev} (rehearse with --backend gloo)")
    import tempfile
    import threading
    rc = 1
    for attempt in range(3):          # the bind/close port pick can lose a race with another process: retry on a fresh port
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        errlog = tempfile.TemporaryFile()
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=errlog if r == 0 else None))
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        # poll ALL children: the first non-zero exit (a rank that died in device / RCCL init would otherwise leave the others
        # hanging in the rendezvous until the c10d timeout) terminates the rest and becomes the exit code
        rc = 0
        live = set(range(n))
        while live and rc == 0:
            for r in list(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code
            if live and rc == 0:
                time.sleep(0.05)
        if rc != 0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
        reader.join(timeout=30)
        errlog.seek(0)
        err = errlog.read().decode(errors="replace")
        errlog.close()
        sys.stderr.write(err)
        if rc != 0 and attempt < 2 and ("EADDRINUSE" in err or "Address already in use" in err):
            continue
        break
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return rc


def cpu_baseline_child(L, pathways):
    """The CPU leg's own process (`bench.py --cpu-baseline-child L`): the oracle -- the fp32 torch-CPU restatement of the reference
    arithmetic, the thing TIMED here and nothing else -- runs ONE WHOLE train step of the bench workload (3 task passes through the
    12-layer backbone + adapters + gene encoder, KL loss, backward to every trainable tensor) on all host cores.
    A short single-layer sample is timed first so that the record can say what a layers-only extrapolation would have claimed."""
    import torch
    from oracle import modaltune_oracle as O
    from modaltune_amd import synth
    from modaltune_amd.config import ModelConfig, segment_lengths
    cfg = ModelConfig()
    sizes = json.load(open(os.path.join(ROOT, "tests", "golden", "pathway_sizes_331.json"))) if pathways == "real" else synth.toy_group_sizes(int(pathways))
    cores = min(os.cpu_count() or 1, 64)      # threads actually used (more threads do not help these shapes)
    torch.set_num_threads(cores)
    full = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, sizes, 0).items()}
    psd = {k: torch.from_numpy(v) for k, v in synth.projector_state(0).items()}
    segs = segment_lengths(cfg.max_wsi_size, cfg.tile_size)
    sd0 = {k: v for k, v in full.items() if k.startswith("encoder.layers.0.")}
    x = torch.randn(1, L + 1, cfg.embed_dim, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    tl = []
    for it in range(3):
        t0 = time.time()
        O.encoder_layer(x, sd0, "encoder.layers.0", segs, (1, 2, 4, 8, 16)).sum().backward()
        tl.append(time.time() - t0)
        x.grad = None
    inp = synth.synth_inputs(L, sizes, 0, grid=128 if L <= 128 * 128 else 512)
    a = (torch.from_numpy(inp["x"]), torch.from_numpy(inp["coords"]), [torch.from_numpy(v) for v in inp["genes"]], torch.from_numpy(inp["text"]))
    tr = synth.trainable_keys(cfg, sizes)
    t0 = time.time()
    _, loss, grads = O.train_step_loss_and_grads(full, cfg, tr, a[0], a[1], a[2], a[3], psd, segs)
    step_s = time.time() - t0
    print(json.dumps({"step_s": step_s, "layer_s": min(tl[1:]), "cores": cores, "loss": float(loss), "patches": L}))


def cpu_baseline_leg(L, pathways, max_seconds=170.0):
    """`cpu_baseline`: ONE whole oracle train step at the bench configuration, measured (not extrapolated) on this box's host
    cores, in a child process with a hard time limit (the exact PID is killed on overrun and the record says so)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(L), "--pathways", str(pathways)]
    t0 = time.time()
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max_seconds,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))      # the child never touches the GPU
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "slides/s", "cores": min(os.cpu_count() or 1, 64), "kind": "port",
                "sample": f"oracle whole train step at L = {L}: not finished within {max_seconds:.0f} s (child killed)"}
    line = [ln for ln in res.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not line:
        return {"value": None, "unit": "slides/s", "cores": None, "kind": "port",
                "sample": "oracle child failed: " + res.stderr.decode(errors="replace")[-300:]}
    r = json.loads(line[-1])
    return {"value": 1.0 / r["step_s"], "unit": "slides/s", "cores": r["cores"], "kind": "port",
            "sample": f"oracle (fp32 torch-CPU port of the reference arithmetic, {r['cores']} threads): ONE WHOLE train step of this workload, timed once "
                      f"cold: {L} patches x 1536-d, 3 task passes x 12 LongNet layers + adapters + gene encoder, KL loss, backward to all trainable "
                      f"tensors: {r['step_s']:.1f} s/slide (loss {r['loss']:.4f}); leg wall time {time.time() - t0:.0f} s",
            "layers_only_extrapolation": {"one_layer_fwd_bwd_s": round(r["layer_s"], 3), "x36_s": round(36 * r["layer_s"], 1),
                                          "note": "what 36 x (one LongNet layer fwd+bwd at N = L+1) would have claimed (rounds 1-3 reported this); "
                                                  "`value` is the measured whole step"}}


def recorded_traffic(kernel: str, L: int, T: int, paths=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (tools/pmc_hbm.sh + tools/pmc_summary.py
    at L = 10 000, T = 65): 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, KiB -> bytes."""
    if paths is None and (L, T) != (10000, 65):
        return None, None
    for path in (paths or (PMC_TRAFFIC_FILE, PMC_TRAFFIC_FALLBACK)):
        if not os.path.exists(path):
            continue
        vals, cur = {}, None
        for line in open(path):
            if not line.startswith(" "):
                cur = line.strip()
            elif cur is not None and kernel in cur:
                m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
                if m:
                    vals[m.group(1)] = float(m.group(2))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def kernel_rooflines(summ, prof_steps, fl, L, T, B=3):
    """Achieved rate of every big kernel against the roofline that bounds it.  MFMA-bound: algorithmic FLOPs per launch
    (2mnk GEMMs, attention products per DESIGN §4) / mean launch time; HBM-bound: algorithmic bytes per launch
     / mean launch time.  Keys as ops.TIMER names them."""
    N = L + 1
    M = B * N
    rows = []

    def add(key, bound, work, label=None):
        if key not in summ:
            return
        n, ms = summ[key]
        t = ms / n * 1e-3
        if bound == "mfma":
            ach, peak, unit = work / t / 1e12, PEAK_F16_MFMA_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work / t / 1e9, PEAK_HBM_GBS, "GB/s"
        rows.append({"kernel": label or key, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                     "frac": round(ach / peak, 4), "avg_launch_ms": round(ms / n, 4), "ms_per_step": round(ms / prof_steps, 3)})

    a = B * fl["attn_layer"]
    add("dilated_attn_fwd", "mfma", a)
    add("dilated_attn_bwd_kv", "mfma", 2.0 * a)
    add("dilated_attn_bwd_q", "mfma", 1.5 * a)
    for k in list(summ):
        m = re.match(r"gemm_nt\[(\d+)x(\d+)x(\d+)\]", k)
        if m and int(m.group(1)) >= 8192:
            mm, nn, kk = (int(v) for v in m.groups())
            add(k, "mfma", 2.0 * mm * nn * kk)
    D, F = 768, 3072
    add("add_layernorm_fwd[768]", "hbm", M * D * (4 + 2 + 4 + 2.0))           # x fp32 + branch fp16 read; h fp32 + y fp16 written
    add("layernorm_fwd[3072]", "hbm", M * F * (2 + 2.0))                      # a1 fp16 read, t16 fp16 written
    add("layernorm_bwd[3072]", "hbm", M * F * (2 + 2 + 2.0))                  # dy, a1 read; da1 written
    add("layernorm_bwd[768]", "hbm", M * D * (2 + 4 + 4 + 4 + 2.0))           # dy fp16, x fp32, dx fp32 read-modify-write, dx16
    add("dilated_mix_ln_fwd", "hbm", M * D * 2.0 * (1 + 31 / 16.0) + M * D * 2.0)   # covered branch rows (sum 1/r) + y
    add("dilated_mix_ln_bwd", "hbm", M * D * 2.0 * (1 + 31 / 16.0) + 2 * M * D * 2.0)
    return rows


# the keys of the reference's model_configs/modaltune_titan_config.json (values restated; no weights are loaded: pretrained False)
TITAN_JSON = dict(num_heads=12, output_dim=256, init_values=0.0, interaction_indexes=[[0, 1], [2, 3], [4, 5]],
                  geneclass_name="gene_mixer_group", with_cffn=True, cffn_ratio=0.25, add_prompt_feature=True, use_extra_extractor=True,
                  freeze_vit=True, with_cp=False, use_prompt_sa=True, prompt_dropout=0.0, prompt_agg="avg", token_agg="cat",
                  pretrained=False, drop_path_rate=0.2, clinfeat_dim=5)


def titan_cpu_baseline(vit_cpu, N, depth, passes, max_seconds=20.0):
    """The torch-CPU form of one frozen TITAN block forward + input-gradient backward at N tokens (the oracle's TITAN path runs
    the backbone module's own blocks: oracle.titan_model_forward), all host cores; a slide step = passes x depth of them."""
    import torch
    from oracle import modaltune_oracle as O     # CPU baseline leg only
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, N, 768, generator=g).requires_grad_(True)
    side = int(N ** 0.5) + 1
    cells = torch.stack([torch.arange(N - 1) // side, torch.arange(N - 1) % side], 1)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / 12) for i in range(12)])
    bias = O.alibi_bias_2d(cells, slopes).float().unsqueeze(0)
    mask = torch.ones(1, N, dtype=torch.bool)
    blk = vit_cpu.blocks.modules_list[0]
    times, t_all = [], time.time()
    for it in range(8):
        t0 = time.time()
        blk(x, bias, mask).sum().backward()
        if it > 0:
            times.append(time.time() - t0)
        x.grad = None
        if time.time() - t_all > max_seconds and times:
            break
    t = min(times)
    n = passes * depth
    return {"value": 1.0 / (n * t), "unit": "slides/s", "cores": cores, "kind": "port",
            "sample": f"torch-CPU fp32 ViT block (the stand-in module the oracle's TITAN path runs), {cores} threads, best of {len(times)}: 1 block "
                      f"fwd+bwd at N={N}, 1 pass: {t:.2f} s; step = {n} block passes ({passes} tasks x {depth} blocks) -> {n * t:.1f} s/slide"}


def run_titan(args, steps=None, warmup=None, patches=None, ragged=None, cpu_baseline=True):
    """BASELINE config 4: TITAN backbone configuration (model_configs/modaltune_titan_config.json), ~4k foreground cells, mixed bag
    lengths.  The TITAN snapshot is not in the reference tree: the backbone is a random-init ViT of TITAN's published geometry
    (768-d, 6 blocks, 12 heads x 64, MLP ratio 4, 2-D ALiBi, one-query attentional pooling: tests/golden/titan_standin.py) running
    on the native kernels -- backbone arithmetic parity against the real snapshot is UNPINNED and the line says so."""
    import torch
    if args.gpus != 1:
        raise RuntimeError("--config titan is a 1-GPU line (slides shard over ranks exactly as in the gigapath configuration)")
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    patches = args.patches if patches is None else patches
    ragged = args.ragged if ragged is None else ragged
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import titan_standin
    from modaltune_amd import ops, synth
    from modaltune_amd.config import flops_per_titan_step
    from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
    from modaltune_amd.trainer import TrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    vit = titan_standin.VisionTransformer(mlp_ratio=4.0)
    titan_standin.init_standin(vit, 0)
    vit_cpu = titan_standin.VisionTransformer(mlp_ratio=4.0)
    vit_cpu.load_state_dict(vit.state_dict())
    sizes = synth.toy_group_sizes(int(args.pathways) if args.pathways != "real" else 6)
    cfg = titan_model_config(TITAN_JSON, 3, False, 6)
    eng = TitanEngine(cfg, sizes, NativeBackbone(vit, dev), dev)
    report = eng.backbone.report
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
    eng.set_stochastic(not args.no_dropout, seed=20260)      # train mode: Extractor-FFN DropPath(0.2), gene-encoder dropouts
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(0))
    Lc = patches if patches != 10000 else 4096
    fr = (1.0, 0.625, 1.375, 0.75, 1.5, 0.5, 1.125, 0.875) if ragged else (1.0,)      # 2048 .. 6144 around 4096
    slides, cells_per = [], []
    for j, f in enumerate(fr):
        want = int(Lc * f)
        L = want + want // 15                    # a sixteenth of the patches share a cell with another (synth_inputs_titan)
        inp = synth.synth_inputs_titan(L, sizes, seed=2000 + j, grid=96)
        slides.append((torch.from_numpy(inp["x"]).to(dev).reshape(L, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev).reshape(L, 2),
                       [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev)))
        cells_per.append(L - L // 16)

    graphed = not args.eager
    ts.graph_cache_size = max(ts.graph_cache_size, len(slides))

    def run(n, first=0, eager=False):
        for i in range(first, first + n):
            x, coords, genes, text = slides[i % len(slides)]
            if graphed and not eager:
                ts.step_graphed(x, coords, genes, text)      # gridding + token-count read-back eager, the rest replayed per (patches, tokens)
            else:
                ts.step(x, coords, genes, text, update=True)

    nwarm = max(warmup, len(slides) * (ts.capture_after + 1 if graphed else 1))      # every bag length visited eagerly, then captured
    run(nwarm)
    eng.check_inputs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, first=nwarm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(ts.loss)
    prof_steps = len(slides)
    replays = ts.graph_replays
    ops.TIMER = {}
    run(prof_steps, first=0, eager=True)
    torch.cuda.synchronize()
    timer, ops.TIMER = ops.TIMER, None
    summ = ops.timer_summary(timer)
    T = cfg.num_tokens
    used = [cells_per[i % len(slides)] for i in range(nwarm, nwarm + steps)]
    step_flops = sum(flops_per_titan_step(c, T)["step"] for c in used) / len(used)
    value = steps / dt
    # dominant kernel: the dK / dV kernel of the dense attention backward, one launch per block per step over the 3 passes;
    # algorithmic FLOPs per launch = S, dP, dV, dK = 2 x the forward's two products, averaged over the profiled bag lengths
    n_l, ms = summ["dense_attn_bwd_kv"]
    kv_flops = sum(2.0 * 3 * flops_per_titan_step(c, T)["attn_layer"] for c in cells_per) / len(cells_per)
    achieved = kv_flops / (ms / n_l * 1e-3) / 1e12
    # HBM traffic of that kernel from the committed PMC summary: measured at N = 4097 tokens (the microbenchmark's geometry), which
    # is this run's mean bag; per launch like `achieved`
    traffic, traffic_file = recorded_traffic("dense_attn_bwd_kv_kernel", 0, 0, paths=(PMC_TRAFFIC_DENSE, PMC_TRAFFIC_DENSE_FALLBACK))
    table = []
    for key, mult in (("dense_attn_fwd", 1.0), ("dense_attn_bwd_kv", 2.0), ("dense_attn_bwd_q", 1.5)):
        if key in summ:
            n, m_ = summ[key]
            fl = sum(mult * 3 * flops_per_titan_step(c, T)["attn_layer"] for c in cells_per) / len(cells_per)
            a_ = fl / (m_ / n * 1e-3) / 1e12
            table.append({"kernel": key, "bound": "mfma", "achieved": round(a_, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(a_ / PEAK_F16_MFMA_TFLOPS, 4), "avg_launch_ms": round(m_ / n, 4), "ms_per_step": round(m_ / prof_steps, 3)})
    out = {
        "metric": "slides/sec (train step), TITAN backbone configuration, ~4k foreground cells", "value": value, "unit": "slides/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"BASELINE config 4: TITAN-geometry ViT (768-d, 6 blocks, 12 heads x 64, MLP 3072, 2-D ALiBi from one fp16 distance table per slide, attentional pooling; "
                               f"random init, stand-in for the absent MahmoodLab/TITAN snapshot: backbone parity UNPINNED) + Modal Adapter, "
                               f"{T - 1} gene tokens + 1 task token, 3 task passes batched, fp16 operands / fp32 accumulate, "
                               + ("train mode (DropPath 0.2 on the Extractor FFN, gene-encoder dropouts), " if not args.no_dropout else "dropout off, ") +
                               f"foreground cells per slide: " + "/".join(str(c) for c in cells_per) + (" in rotation (mixed bag lengths)" if ragged else ""),
                   "cells": cells_per, "tokens": T, "parallelism": "dp1", "backbone_impl": eng.backbone.kind, "self_check": report},
        "loss": loss, "step_tflops": step_flops / 1e12, "step_mfma_frac": step_flops * value / 1e12 / PEAK_F16_MFMA_TFLOPS,
        "roofline": {"kernel": "dense_attn_bwd_kv_kernel", "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F16_MFMA_TFLOPS, "traffic": traffic,
                     "traffic_source": (f"{traffic_file}: rocprofv3 --pmc passes at N = 4097 tokens x 3 passes (tools/dense_microbench.py), "
                                        "(2 x FETCH_SIZE + WRITE_SIZE) KiB" if traffic is not None else None),
                     "avg_launch_ms": ms / n_l, "flops_per_launch": kv_flops,
                     "measured": f"HIP events around each launch, eager instrumented pass over the {prof_steps} bag lengths after the timed region"},
        "roofline_kernels": table,
        "launch": ("hipGraph replay per (patches, tokens) geometry -- the gridding kernels and the token-count read-back stay eager; a bag length "
                   "that has not been seen three times runs the eager schedule" if graphed else "eager"), "graph_replays": replays,
        "kernel_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:14]},
        "pass_groups": {"by_tokens": {str(k): {"groups": bool(v), **ts.split_trials.get(k, {})} for k, v in sorted(ts.split_decisions.items())},
                        "how": "each recurring geometry captured both ways at capture time, the faster kept (TrainStep._trial_split)"},
    }
    if args.kernel_times:
        fam = {}
        for k, (n, m_) in summ.items():
            f = re.sub(r"\[.*", "", k)
            fam[f] = (fam.get(f, (0, 0.0))[0] + n, fam.get(f, (0, 0.0))[1] + m_)
        tot = sum(v[1] for v in fam.values())
        for k, (n, m_) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:28s} launches/step {n / prof_steps:7.1f}  ms/step {m_ / prof_steps:9.3f}  ({100 * m_ / tot:5.1f} %)", file=sys.stderr)
    if cpu_baseline and not args.no_cpu_baseline:
        out["cpu_baseline"] = titan_cpu_baseline(vit_cpu, int(sum(cells_per) / len(cells_per)) + 1, 6, 3)
    return out


def run_module(args, steps=None, warmup=None, optim=None):
    """The boundary north_star names, driven exactly as the reference trainer drives it (train_modaltune.py:123-149,172-177,
    195-240): Aggregator.create -> per step the frozen torch projector, three model(...) calls (one per task id), torch's
    KL-divergence loss under autocast, GradScaler.scale(loss).backward(), GradScaler.step(torch.optim.AdamW), update, zero_grad."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    if args.gpus != 1:
        raise RuntimeError("--api module is a 1-GPU line")
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    from modaltune_amd import synth
    from modaltune_amd.aggregators import Aggregator
    from modaltune_amd.config import flops_per_slide_step
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = args.patches
    sizes = json.load(open(os.path.join(ROOT, "tests", "golden", "pathway_sizes_331.json"))) if args.pathways == "real" else \
        synth.toy_group_sizes(int(args.pathways))
    groups = {i: ["g%d_%d" % (i, j) for j in range(n)] for i, n in enumerate(sizes)}
    from modaltune_amd.config import GIGAPATH_JSON
    kw = dict(GIGAPATH_JSON, pretrained=False)                     # TM:123-126: **json_config (no weight file exists offline)
    if args.no_dropout:
        kw.update(dropout=0.0, drop_path_rate=0.0)
    # TM:123-126 -- the model is used AS CONSTRUCTED (reference init families; pretrained False: no weight file exists offline)
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0, **kw).to(dev)
    cfg = model.cfg
    params = [{"params": list(filter(lambda p: p.requires_grad, model.parameters())), "lr": 1e-4 / 20}]             # TM:139-149
    optim = optim or args.optim
    if optim == "fused":
        from modaltune_amd.optim import AdamW
    else:
        AdamW = torch.optim.AdamW
    opt = AdamW(params, weight_decay=0.01, betas=(0.9, 0.999))
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 15)                                        # TM:107
    psd = {k: torch.from_numpy(v).to(dev) for k, v in synth.projector_state(0).items()}

    def projector(t):       # Projection_layer (TM:44-59): frozen random Conv1x1 -> LayerNorm -> ReLU -> Conv1x1, the trainer's own torch code
        h = F.linear(t, psd["conv1.0.weight"].flatten(1), psd["conv1.0.bias"])
        h = F.layer_norm(h, (h.shape[-1],), psd["conv1.1.weight"].flatten(), psd["conv1.1.bias"].flatten(), 1e-5)
        return F.linear(F.relu(h), psd["conv1.3.weight"].flatten(1), psd["conv1.3.bias"])
    loss_fn = nn.KLDivLoss(reduction="sum")
    eye = torch.eye(3, device=dev)
    slides = []
    for j in range(2):
        inp = synth.synth_inputs(L, sizes, seed=1000 + j, grid=128 if L <= 128 * 128 else 512)
        slides.append((torch.from_numpy(inp["x"]).to(dev), torch.from_numpy(inp["coords"]).to(dev),
                       {i: torch.from_numpy(a).to(dev) for i, a in enumerate(inp["genes"])}, torch.from_numpy(inp["text"]).to(dev)))
    model.train()
    last = {}

    def step(i):
        images, coords, gene_data, text = slides[i % len(slides)]
        text = projector(text)
        text = text / text.norm(dim=-1, keepdim=True)
        with torch.autocast("cuda", enabled=True):
            logit = torch.cat([model(x=images, coords=coords, genes=gene_data, clinical=[], task_token=eye[t]) for t in (0, 1, 2)], dim=0)
            logit = logit / logit.norm(dim=-1, keepdim=True)
            loss = loss_fn(F.log_softmax(logit, dim=1), F.softmax(text[[0, 1, 3], :], dim=1)) * 10
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        last["loss"] = loss.detach()

    # warm-up: one slide teaches the module the task-id pattern, then the batched pass is seen twice on the eager bridge, twice on the
    # priming visits of module_graph.ModuleReplay and captured on the next one -- the timed region is the steady state (replays)
    nwarm = max(warmup, 8)
    for i in range(nwarm):
        step(i)
    model.engine.check_inputs()
    torch.cuda.synchronize()
    replays_before = model._replay.replays
    t0 = time.perf_counter()
    for i in range(nwarm, nwarm + steps):
        step(i)
    host_dt = time.perf_counter() - t0          # when the host had ENQUEUED the last step (no sync inside the loop: below dt when the GPU is the bound)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    T = cfg.num_tokens
    fl = flops_per_slide_step(L, T)
    value = steps / dt
    out = {"metric": "slides/sec (train step) at 10k patches x 1536-d", "value": value, "unit": "slides/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16", "data": "synthetic",
           "config": {"workload": f"Prov-GigaPath ModalAdapter train step through the drop-in nn.Module exactly as train_modaltune.py drives it "
                                  f"(3 model calls, torch KL loss, GradScaler, {'modaltune_amd.optim.AdamW' if optim == 'fused' else 'torch.optim.AdamW'}), {L} patches x 1536-d, {len(sizes)} pathways -> {T - 1} gene "
                                  f"tokens + 1 task token, 1 slide per step, " + ("the engine's launches as two hipGraph replays per step (forward / backward), "
                                                                                     if replays_before is not None and model._replay.replays > replays_before else "eager, ") + ("dropout off" if args.no_dropout else "train mode (Dropout / DropPath on)"),
                      "api": "module", "patches": L, "tokens": T, "parallelism": "dp1"},
           "loss": float(last["loss"]), "step_tflops": fl["step"] / 1e12, "step_mfma_frac": fl["step"] * value / 1e12 / PEAK_F16_MFMA_TFLOPS,
           "launch": ("torch autograd + torch.optim around two hipGraph replays per step (module_graph.ModuleReplay)"
                      if model._replay.replays > replays_before else "eager (torch autograd + torch.optim)"),
           "graph_replays": model._replay.replays - replays_before, "optimizer": "modaltune_amd.optim.AdamW" if optim == "fused" else "torch.optim.AdamW",
           "optimizer_steps_fused": getattr(opt, "last_step_fused", None), "host_enqueue_ms_per_step": 1e3 * host_dt / steps,
           "task_tokens_read_back": model._nosync_rows is None}
    return out


def _leg(fn):
    """A secondary leg must never cost the headline line: failures become an error record."""
    t0 = time.time()
    try:
        r = fn()
    except Exception as e:
        import traceback
        r = {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().splitlines()[-3:]}
    r["leg_wall_s"] = round(time.time() - t0, 1)
    return r


def _brief(rec):
    """Sub-record form of a leg's full JSON line."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "loss", "step_tflops", "step_mfma_frac", "roofline",
            "roofline_kernels", "launch", "kernel_ms_per_step", "pass_groups", "graph_replays", "optimizer", "optimizer_steps_fused", "torch_adamw", "host_enqueue_ms_per_step", "task_tokens_read_back")
    return {k: rec[k] for k in keep if k in rec}


def leg_pcie(ts, eng, sizes, L, resident_value, steps=20):
    """The same train step with the slide arriving as the reference hands it over (train_modaltune.py:198-210): fp32 features
    [L, 1536] in HOST memory -> pinned staging -> async H2D on a copy stream -> fp16 cast on the device, one case ahead
    (modaltune_amd.data.CasePrefetcher), hipGraph replay.  Never the headline `value`."""
    import torch
    from modaltune_amd import data, synth
    host = []
    for j in range(3):
        inp = synth.synth_inputs(L, sizes, seed=3000 + j, grid=128 if L <= 128 * 128 else 512)
        host.append(dict(features=torch.from_numpy(inp["x"]).reshape(L, -1), coords=inp["coords"],
                         genes=[torch.from_numpy(a) for a in inp["genes"]], text=torch.from_numpy(inp["text"]), case_id=j))

    r0 = ts.graph_replays

    def timed(cases):
        def gen(n):
            for i in range(n):
                yield cases[i % len(cases)]
        for s in data.CasePrefetcher(gen(4)):
            ts.step_graphed(s.x, s.coords, s.genes, s.text)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in data.CasePrefetcher(gen(steps)):
            ts.step_graphed(s.x, s.coords, s.genes, s.text)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt = timed(host)
    # the fp16 shards SURVEY section 8 f3 describes (data.convert_to_f16_shard: half the bytes on the host side, on PCIe and no cast
    # kernel): the same leg with the features already fp16 in host memory -- attributes the gap to the resident step to bytes or not
    host16 = [dict(h, features=h["features"].half()) for h in host]
    dt16 = timed(host16)
    return {"metric": "slides/sec (train step), slide streamed from host memory per step (fp32 features, PCIe-inclusive)", "value": 1.0 / dt,
            "unit": "slides/s", "steps": steps, "warmup": 4, "ms_per_step": 1e3 * dt, "host_bytes_per_slide": L * 1536 * 4,
            "vs_resident": (1.0 / dt) / resident_value, "graph_replays": ts.graph_replays - r0,
            "fp16_shards": {"ms_per_step": 1e3 * dt16, "value": 1.0 / dt16, "host_bytes_per_slide": L * 1536 * 2,
                            "what": "the same with the features stored as fp16 shards (data.convert_to_f16_shard): half the host copy and the PCIe bytes, no cast kernel"},
            "how": "CasePrefetcher: pinned staging, H2D on a copy stream one case ahead of the running step, fp32 -> fp16 cast on the device"}


MFMA_BOUND = ("gemm_nt[", "dilated_attn_fwd", "dilated_attn_bwd_kv", "dilated_attn_bwd_q", "dense_attn")
HBM_BOUND = ("layernorm", "add_layernorm", "dilated_mix_ln", "dilated_attn_bwd_combine", "cast", "copy_rows", "dropout_f32", "inject_resid_bwd",
             "elementwise", "gelu_f16", "adamw")


def leg_overlap(ts, slide):
    """What the two concurrent pass groups actually hide: HIP events around EVERY launch of one eager step of the
    two-stream schedule (each group's events on its own stream; rocprofv3 serialises the two branches of the replayed graph and cannot
    show this), once with the groups overlapping and once with the same two groups one after the other (host sync between them).
    A kernel family is MFMA-bound (GEMMs, attention) or HBM-bound (LayerNorm family, branch mix, combine, casts, residual kernels);
    `hbm_ms_hidden` = HBM-bound kernel time of one group that ran while an MFMA-bound kernel of the OTHER group was running."""
    import torch
    from modaltune_amd import ops
    x, coords, genes, text = slide

    def instrumented(serial):
        ts._group_hook = (lambda gi: torch.cuda.synchronize()) if serial else None
        ops.TIMER, ops.TIMELINE = {}, []
        base = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        base.record()
        ts.step(x, coords, genes, text, update=True)
        torch.cuda.synchronize()
        tl, ops.TIMER, ops.TIMELINE = ops.TIMELINE, None, None
        ts._group_hook = None
        rows = [(k, base.elapsed_time(e0), base.elapsed_time(e1), st) for k, e0, e1, st in tl]
        return rows

    def cls(k):
        return "mfma" if k.startswith(MFMA_BOUND) else "hbm" if k.startswith(HBM_BOUND) else "other"

    for _ in range(2):
        ts.step(x, coords, genes, text, update=True)      # (eager warm-up of the split geometry)
    alone = instrumented(True)
    both = instrumented(False)
    streams = sorted({st for _, _, _, st in both}, key=lambda s_: -sum(1 for r in both if r[3] == s_))[:2]
    fam_alone, fam_both = {}, {}
    for rows, acc in ((alone, fam_alone), (both, fam_both)):
        for k, a, b, st in rows:
            acc[k] = acc.get(k, 0.0) + (b - a)
    iv = {st: [(a, b, cls(k)) for k, a, b, s2 in both if s2 == st] for st in streams}

    def overlap_ms(xs, ys):      # total length of [intervals of xs] intersected with [union of ys]; both sorted by start, ys disjoint on one stream
        tot, j = 0.0, 0
        for a, b in xs:
            while j < len(ys) and ys[j][1] <= a:
                j += 1
            i = j
            while i < len(ys) and ys[i][0] < b:
                tot += max(0.0, min(b, ys[i][1]) - max(a, ys[i][0]))
                i += 1
        return tot

    hidden = 0.0
    if len(streams) == 2:
        for sa, sb in ((streams[0], streams[1]), (streams[1], streams[0])):
            hb = sorted((a, b) for a, b, c in iv[sa] if c == "hbm")
            mf = sorted((a, b) for a, b, c in iv[sb] if c == "mfma")
            hidden += overlap_ms(hb, mf)
    hbm_both = sum(v for k, v in fam_both.items() if cls(k) == "hbm")
    hbm_alone = sum(v for k, v in fam_alone.items() if cls(k) == "hbm")
    mf_both = sum(v for k, v in fam_both.items() if cls(k) == "mfma")
    mf_alone = sum(v for k, v in fam_alone.items() if cls(k) == "mfma")
    span = lambda rows: max(b for _, _, b, _ in rows) - min(a for _, a, _, _ in rows)
    worst = sorted(((k, fam_both[k] - fam_alone.get(k, 0.0)) for k in fam_both), key=lambda kv: -kv[1])[:6]
    return {"span_ms_overlapped": round(span(both), 3), "span_ms_groups_one_after_the_other": round(span(alone), 3),
            "hbm_kernel_ms_alone": round(hbm_alone, 3), "hbm_kernel_ms_overlapped": round(hbm_both, 3), "hbm_ms_hidden": round(hidden, 3),
            "hbm_hidden_frac": round(hidden / max(hbm_both, 1e-9), 3),
            "mfma_kernel_ms_alone": round(mf_alone, 3), "mfma_kernel_ms_overlapped": round(mf_both, 3), "mfma_ms_slowed": round(mf_both - mf_alone, 3),
            "most_slowed_families_ms": {k: round(v, 3) for k, v in worst}, "launches": len(both),
            "how": "eager instrumented steps (HIP events around every launch, on the launching stream); spans include the event overhead of ~900 "
                   "launches, so they are longer than the replayed step -- compare the two spans with each other, not with ms_per_step"}


def leg_size(ts, eng, sizes, L, dropout, steps=8):
    """One more bag length through the SAME TrainStep (hipGraph replay after the usual eager visits): the reference's loader feeds anything
    from a few hundred to 25 000 patches per case (data_utils/datasets.py:274-281 `threshold`, scripts/submit_modaltune.sh:47-49)."""
    import torch
    from modaltune_amd import synth
    from modaltune_amd.config import flops_per_slide_step
    dev = eng.device
    slides = []
    for j in range(2):
        inp = synth.synth_inputs(L, sizes, seed=5000 + 7 * j + L, grid=128 if L <= 128 * 128 else 512)
        slides.append((torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev),
                       [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev)))
    r0 = ts.graph_replays
    for i in range(ts.capture_after + 2):
        ts.step_graphed(*slides[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ts.step_graphed(*slides[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = flops_per_slide_step(L, eng.cfg.num_tokens)
    return {"patches": L, "pathways": len(sizes), "tokens": eng.cfg.num_tokens, "ms_per_step": round(1e3 * dt, 3), "value": round(1.0 / dt, 3),
            "unit": "slides/s", "steps": steps, "step_tflops": round(fl["step"] / 1e12, 3),
            "step_mfma_frac": round(fl["step"] / dt / 1e12 / PEAK_F16_MFMA_TFLOPS, 4), "pass_groups": bool(ts._split_now(L)),
            "pass_groups_trial_ms": ts.split_trials.get(L),
            "graph_replays": ts.graph_replays - r0, "dropout": dropout}


def leg_real_pathways(args, L, steps=8):
    """The headline bag with the reference's REAL gene grouping (331 pathways, 33.5 M trainable parameters: the 134 MB gradient payload of
    the data-parallel step) instead of the 6-pathway toy grouping BASELINE's "6 pathway tokens" names."""
    import torch
    from modaltune_amd import synth
    from modaltune_amd.config import ModelConfig
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    dev = torch.device("cuda", torch.cuda.current_device())
    sizes = json.load(open(os.path.join(ROOT, "tests", "golden", "pathway_sizes_331.json")))
    cfg = ModelConfig()
    eng = Engine(cfg, sizes, dev)
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
    eng.set_stochastic(not args.no_dropout, seed=20261)
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(0))
    rec = leg_size(ts, eng, sizes, L, not args.no_dropout, steps=steps)
    rec["trainable_parameters"] = int(eng.store.n_flat)
    return rec


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.cpu_baseline_child, args.pathways)
    if args.config == "titan":
        return print(json.dumps(run_titan(args)))
    if args.api == "module":
        return print(json.dumps(run_module(args)))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np  # noqa: F401
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("MT_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # MT_BENCH_DEVICE: rehearsals only
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback for the hot path)")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if args.dp_rehearsal:
        if world != 1:
            raise SystemExit("--dp-rehearsal is a one-rank run (--gpus 1)")
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], os.environ["MT_DP_REHEARSE"] = "127.0.0.1", str(port_), "1"
    dp_on = world > 1 or args.dp_rehearsal
    if dp_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)

    from modaltune_amd import dp, ops, synth
    from modaltune_amd.config import ModelConfig, flops_per_slide_step
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep

    L = args.patches
    cfg = ModelConfig()                       # Prov-GigaPath ModalAdapter config (modaltune_gigapath_config.json)
    if args.pathways == "real":
        sizes = json.load(open(os.path.join(ROOT, "tests", "golden", "pathway_sizes_331.json")))
    else:
        sizes = synth.toy_group_sizes(int(args.pathways))
    args.pathways = len(sizes)
    eng = Engine(cfg, sizes, dev)
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))      # identical weights on every rank ...
    dp.broadcast_params_(eng.store.flat)                                 # ... and DDP's constructor broadcast on top
    eng.set_stochastic(not args.no_dropout, seed=20260 + rank)     # Dropout(0.25) / DropPath(<= 0.1): counter-based masks
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(0))
    # synthetic slides, distinct per rank, resident in HBM (2 alternating slides per rank; --ragged: 8 lengths)
    lengths = [L, L] if not args.ragged else [int(L * f) for f in (1.0, 0.62, 0.87, 0.5, 0.95, 0.71, 0.56, 0.8)]
    slides = []
    for j, Lj in enumerate(lengths):
        inp = synth.synth_inputs(Lj, sizes, seed=1000 + 17 * rank + j, grid=128 if Lj <= 128 * 128 else 512)
        slides.append((torch.from_numpy(inp["x"]).to(dev).half().reshape(Lj, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev),
                       [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev)))
    if args.ragged:
        ts.capture_after = 1 << 30      # every length is "new": the eager schedule is the steady state being measured

    def run(n, graphed=True, first=0):
        for i in range(first, first + n):
            x, coords, genes, text = slides[i % len(slides)]
            if graphed and not args.eager:
                ts.step_graphed(x, coords, genes, text)      # eager visits + capture happen in warm-up; then hipGraph replay
            else:
                ts.step(x, coords, genes, text, update=True)

    def barrier():
        torch.cuda.synchronize()
        if dp_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    nwarm = max(args.warmup, len(slides) if args.ragged else 3)   # 2 eager visits + the capture step (same geometry)
    warm_fallback = None
    try:
        run(nwarm)
    except Exception as e:            # graph capture unavailable -> same arithmetic with eager launches
        if dp_on and not args.eager and ts.dp_schedule != "batched" and "MT_DP_SCHEDULE" not in os.environ:
            # the two-stream data-parallel schedule could not be captured here (it has never run over RCCL before the driver's node):
            # the same code fails the same way on every rank, so all of them fall back to the batched schedule -- and the line says so
            warm_fallback = f"{ts.dp_schedule} failed in warm-up ({type(e).__name__}: {e}"[:300] + "); batched schedule"
            print(f"[bench] {warm_fallback}", file=sys.stderr)
            os.environ["MT_DP_SCHEDULE"] = "batched"
            ts.dp_schedule = "batched"
            ts._gcache.clear(); ts._cap = None
            ts.reducer.pending.clear(); ts.reducer.started.clear()
            torch.cuda.synchronize()
            run(nwarm)
        elif args.eager or dp_on:   # multi-rank: a capture failure must not pass unnoticed (ranks could also diverge on it):
            raise                     # the run fails with a non-zero exit code; `--eager` is the explicit way to time eager launches
        else:
            warm_fallback = f"hipGraph path failed ({type(e).__name__}: {e})"[:300] + "; eager launches"
    if warm_fallback is not None and warm_fallback.endswith("eager launches"):
        print(f"[bench] {warm_fallback}", file=sys.stderr)
        args.eager = True
        ts._gcache.clear()
        ts._cap = None
        ts.reducer.pending.clear(); ts.reducer.started.clear()
        torch.cuda.synchronize()
        run(nwarm)
    eng.check_inputs()
    comm = None
    extra_steps = 0
    if dp_on:
        # the collective path is real before anything is timed: an all-reduce of ones must come back as the world size
        probe = torch.ones(1, device=dev) if args.backend == "nccl" else torch.ones(1)
        torch.distributed.all_reduce(probe)
        comm = {"ranks_seen_by_all_reduce": int(round(float(probe))), "backend": torch.distributed.get_backend(),
                "bucket_bytes": [4 * sum(n for _, n in bk) for bk in ts.reducer.buckets],
                "last_bucket_sharded": bool(ts.reducer.sharded), "warmup_fallback": warm_fallback}
        if comm["ranks_seen_by_all_reduce"] != world:
            raise RuntimeError(f"all_reduce of ones returned {float(probe)} on a world of {world}")
        # Which data-parallel schedule of a long bag is fastest HERE (interconnect, payload, bag length)?  TrainStep.dp_schedule:
        # "groups_joined" (two pass groups, every bucket summed and started at its own join), "groups_exposed" (round 5: the whole
        # reduction behind the groups' final join), "batched" (one B = 3 pass, buckets started from inside its backward).  Each is
        # captured and timed for a few steps; the choice is the minimum of the MAX-over-ranks times and is the same on every rank.
        cands = ["groups_joined", "groups_exposed", "batched"] if (ts.split_passes and L >= ts.split_min_patches and not (args.eager or args.ragged)) \
            else [ts.dp_schedule]
        forced = os.environ.get("MT_DP_SCHEDULE")
        if forced:
            cands = [forced]
        timings = {}
        for sched in cands:
            ts.dp_schedule = sched
            steps_before = int(ts.step_dev)
            try:
                run(3, first=nwarm)                  # eager visits + capture of this schedule's graphs
                ts.comm_events = []
                barrier()
                t0 = time.perf_counter()
                run(4, first=nwarm)
                barrier()
            except Exception as e:                   # (the same code on every rank: a schedule that cannot be captured here fails everywhere;
                if len(cands) == 1:                  # the others are still timed.  Nothing is hidden: the record names the error)
                    raise
                timings[sched] = {"error": f"{type(e).__name__}: {e}"[:300]}
                ts._gcache.clear(); ts._cap = None
                ts.reducer.pending.clear(); ts.reducer.started.clear()
                ts.comm_events = None
                torch.cuda.synchronize()
                extra_steps += int(ts.step_dev) - steps_before
                continue
            extra_steps += 7
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if args.backend != "nccl" else dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            ev, ts.comm_events = ts.comm_events, None
            exposed = [a.elapsed_time(b) for k, a, b in ev if k == "grad"]
            timings[sched] = {"ms_per_step": round(1e3 * float(tt) / 4, 3), "comm_exposed_ms_this_rank": round(sum(exposed) / max(1, len(exposed)), 3)}
        ok = {k: v for k, v in timings.items() if "ms_per_step" in v}
        if not ok:
            raise RuntimeError(f"no data-parallel schedule could be timed: {timings}")
        chosen = min(ok, key=lambda k: ok[k]["ms_per_step"])
        ts.dp_schedule = chosen
        comm["schedule_chosen"], comm["schedule_timings"] = chosen, timings
        comm["schedule_how"] = ("each candidate captured, then 4 replayed steps between barriers, MAX over ranks; the timed region below runs the "
                                "fastest" + (" (MT_DP_SCHEDULE forced the choice)" if forced else ""))
        run(2, first=nwarm)
        extra_steps += 2
        comm["buckets_started_before_the_backward_ended"] = int(ts.buckets_started_early) if chosen == "groups_joined" else None
        ts.comm_events = []
    barrier()
    t0 = time.perf_counter()
    run(args.steps, first=nwarm)
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    if comm is not None:
        ev, ts.comm_events = ts.comm_events, None
        for kind, key in (("grad", "comm_exposed_ms"), ("param", "param_gather_exposed_ms")):
            ms = [a.elapsed_time(b) for k, a, b in ev if k == kind]
            comm[key] = sum(ms) / max(1, args.steps)      # per step: GPU time on the compute stream between the two marks
            comm[key.replace("_ms", "_events")] = len(ms)
    loss = float(ts.loss)
    replays, eager_steps = ts.graph_replays, ts.eager_steps
    # per-kernel durations: HIP events on the launch stream around every launch of an eager, instrumented pass of the
    # same step (events cannot be recorded inside a replayed graph); not part of the timed region above
    # The timed region above runs the task passes of a long bag as two concurrent groups (TrainStep.split_passes): kernels of the two
    # groups overlap there, so a per-launch duration would be that of a kernel sharing the chip.  The per-kernel table and the roofline
    # are taken on the BATCHED schedule (one B = 3 pass, every kernel alone on the chip): comparable with rounds 1-4 and with
    # `rocprofv3 --kernel-trace --stats` of `MT_SPLIT_PASSES=0 python3 bench.py` (profiles/).
    pass_groups = {"on": bool(ts._split_now(L)), "groups": [list(g) for g in ts._groups], "min_patches": ts.split_min_patches,
                   "decided_by": ("trial at capture time (ms per replayed step, both schedules captured; training state restored behind it): " + json.dumps(ts.split_trials[L])) if L in ts.split_trials
                   else "threshold", "kernel_table": "batched schedule (one kernel at a time)"}
    split_was, ts.split_passes = ts.split_passes, False
    if pass_groups["on"]:
        run(2, graphed=False, first=nwarm + args.steps)          # (the batched geometry's workspace and gradient arena: untimed)
    prof_steps = min(args.steps, 3)
    ops.TIMER = {}
    run(prof_steps, graphed=False, first=nwarm + args.steps)
    barrier()
    timer, ops.TIMER = ops.TIMER, None
    skipped = nwarm + args.steps + prof_steps + (2 if pass_groups["on"] else 0) + extra_steps - int(ts.step_dev)
    # the token side on its own: every token-side launch of ONE step recorded in order, captured as a hipGraph on the same buffers and
    # replayed -- what the ~220 small dependent launches cost INSIDE the replayed step (HIP events around each launch of the eager pass
    # above add the event overhead to every one of them: that table's `token_side` row is an upper bound)
    token_graph = None
    if not dp_on and not args.ragged:
        try:
            ops.RECORD, ops.RECORD_KEEP[:] = [], []
            run(1, graphed=False, first=nwarm + args.steps + prof_steps)
            rec, ops.RECORD = ops.RECORD, None
            torch.cuda.synchronize()
            gs = torch.cuda.Stream()
            with torch.cuda.stream(gs):
                tg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(tg, stream=gs):
                    for fn, a, k in rec:
                        fn(*a, **k)
                tg.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(gs)
                for _ in range(10):
                    tg.replay()
                e1.record(gs); torch.cuda.synchronize()
            token_graph = {"launches": len(rec), "ms": e0.elapsed_time(e1) / 10,
                           "how": "all token-side launches of one step (LayerNorms and fp32 products of <= 1024 rows, adds, copies, DropPath rows, "
                                  "T x T attention, pathway networks, loss head) re-issued in order inside one hipGraph on the step's own buffers, "
                                  "10 replays"}
            del tg, rec
        except Exception as e:       # a measurement aid must never cost the headline line
            token_graph = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ops.RECORD = None
            ops.RECORD_KEEP[:] = []
    ts.split_passes = split_was
    if dp_on:
        host = args.backend != "nccl"
        tt = torch.tensor([dt, comm["comm_exposed_ms"], comm["param_gather_exposed_ms"]], device="cpu" if host else dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt, comm["comm_exposed_ms"], comm["param_gather_exposed_ms"] = (float(v) for v in tt)
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, round(1e3 * dt_own / args.steps, 3))
        comm["per_rank_ms_per_step"] = per_rank
        comm["how"] = ("comm_exposed_ms: HIP events on the compute stream from the end of the backward to the point the optimiser may start "
                       "(bucket all-reduces launched from inside the backward + the last bucket's reduce-scatter + the 4-byte found_inf MAX), "
                       "per step, MAX over ranks; param_gather_exposed_ms: the next step's wait for the sharded parameter all-gather, same rule")

    if rank == 0:
        T = cfg.num_tokens
        fl = flops_per_slide_step(L, T)
        value = args.steps * world / dt
        summ = ops.timer_summary(timer)
        tot_ms = sum(v[1] for v in summ.values())
        if args.kernel_times:
            for k, (n, ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:32s} launches/step {n / prof_steps:7.1f}  ms/step {ms / prof_steps:9.3f}  ({100 * ms / tot_ms:5.1f} %)",
                      file=sys.stderr)
        # roofline of the dominant kernel (largest share of the step): the dK/dV kernel of the dilated-attention backward,
        # one launch per layer per step over all 5 branches and all 3 task passes.  Algorithmic FLOPs per launch: the four
        # products it owns (S = Q K^T, dP = dO V^T, dV = P^T dO, dK = dS^T Q) = 2 x the forward's two products
        # (SURVEY §8d counts the whole flash backward as 2.5 x forward; the dQ kernel carries the remaining 0.5 x).
        n_l, ms = summ["dilated_attn_bwd_kv"]
        launch_flops = 2.0 * 3 * fl["attn_layer"]
        achieved = launch_flops / (ms / n_l * 1e-3) / 1e12
        exec_flops = 2.0 * 3 * fl["attn_layer_executed"]      # zero-padded tiles are skipped, not computed
        traffic, traffic_file = recorded_traffic("dilated_attn_bwd_kv_kernel", L, T) if not args.ragged else (None, None)
        table = kernel_rooflines(summ, prof_steps, fl, L, T) if not args.ragged else []
        meas = ops.measured_mfma_peak_tflops()
        peak_meas = {"value": meas, "unit": "TFLOP/s", "vendor_peak": PEAK_F16_MFMA_TFLOPS,
                     "pool_typical": 1760.0, "box_speed_vs_pool_typical": round(meas / 1760.0, 3),
                     "box_note": "the same bare MFMA loop measured 1750-1783 TFLOP/s on most boxes of this pool in rounds 4-6 and 1630-1670 on its slow "
                                 "ones, where every MFMA-bound kernel (and ms_per_step) is 6-8 % slower with identical binaries (DESIGN section 5): compare "
                                 "lines of different boxes through roofline_frac_of_measured / step_frac_of_measured",
                     "roofline_frac_of_measured": achieved / meas, "step_frac_of_measured": fl["step"] * value / world / 1e12 / meas,
                     "how": "bare v_mfma_f32_32x32x16_f16 loop, 4 independent accumulators, pseudo-random register operands, 1024 workgroups x 4 "
                            "waves, best of 3 (mt_mfma_probe); `roofline.frac` stays priced against the 2.5 PF/s vendor figure"}
        worst = min((r for r in table if r["ms_per_step"] >= 0.4), key=lambda r: r["frac"], default=None)
        out = {
            "metric": "slides/sec (train step) at 10k patches x 1536-d", "value": value, "unit": "slides/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Prov-GigaPath ModalAdapter train step, {L} patches x 1536-d, {args.pathways} pathways -> "
                                   f"{T - 1} gene tokens + 1 task token, 3 task passes, fp16 operands / fp32 accumulate, "
                                   f"1 slide per GPU per step"
                                   + (", task passes as two concurrent groups (B = 2 | 1 on two HIP streams)" if pass_groups["on"] else "")
                                   + ", train mode: "
                                   + ("dropout / drop-path off (parity configuration)" if args.no_dropout else
                                      f"Dropout({cfg.dropout}) on the embedded input and both backbone branches, DropPath(0..{cfg.drop_path_rate}) "
                                      f"per layer and on the Extractor FFN (Philox masks regenerated in backward)")
                                   + ("; ragged: bag lengths " + "/".join(str(v) for v in lengths) + " in rotation" if args.ragged else ""),
                       "patches": L, "tokens": T, "parallelism": f"dp{world}", "dropout": not args.no_dropout, "pass_groups": pass_groups,
                       "backend": args.backend if dp_on else None,
                       **({"dp_rehearsal": "ONE rank: every collective of the data-parallel step issued on a one-rank communicator "
                                           "(MT_DP_REHEARSE=1); `value` is NOT a scaling point"} if args.dp_rehearsal else {})},
            "loss": loss, "skipped_steps": skipped,
            "step_tflops": fl["step"] / 1e12, "step_mfma_frac": fl["step"] * value / world / 1e12 / PEAK_F16_MFMA_TFLOPS,
            "roofline": {"kernel": "dilated_attn_bwd_kv_kernel", "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F16_MFMA_TFLOPS,
                         "frac_basis": "algorithmic FLOPs (SURVEY §8d: the reference computes on zero-padded segments); the kernel "
                                       "skips the padding -- mfma_frac_executed is the fraction of peak the MFMA pipe itself runs at",
                         "traffic": traffic,
                         "avg_launch_ms": ms / n_l, "flops_per_launch": launch_flops,
                         "flops_executed_per_launch": exec_flops,
                         "mfma_frac_executed": exec_flops / (ms / n_l * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                         "traffic_source": (f"{traffic_file}: rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs), "
                                            "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 tallies 128-B read requests at 64 B)"
                                            if traffic is not None else None),
                         "measured": f"HIP events around each launch, eager instrumented pass of {prof_steps} steps after the timed region"
                                     + (", on the batched schedule (one B = 3 pass: each kernel alone on the chip; the timed region overlaps two pass groups)"
                                        if pass_groups["on"] else "")},
            "mfma_peak_measured": peak_meas,
            "roofline_worst": worst,
            "roofline_kernels": table,
            "launch": "eager" if (args.eager or args.ragged) else "hipGraph replay",
            "graph_replays": replays, "eager_steps": eager_steps, "comm": comm, "token_side_in_graph": token_graph,
            "kernel_ms_per_step": {k: round(v[1] / prof_steps, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:12]},
        }
        from modaltune_amd import _lib
        bi = _lib.build_info()       # which binary this line timed: the sha256 of (csrc/*, include/*, flags) it was built from
        out["build_id"], out["build_id_matches_tree"], out["lib"] = bi["build_id"], bi["build_id_matches_tree"], bi["lib"]
        default_line = world == 1 and not (args.ragged or args.eager or args.dp_rehearsal)
        if default_line and not args.no_legs:
            # secondary configurations as short sub-records of the SAME driver-observed line (each <= ~10 s; never the headline)
            out["pcie_inclusive"] = _leg(lambda: leg_pcie(ts, eng, sizes, L, value))
            if pass_groups["on"]:
                pass_groups["overlap"] = _leg(lambda: leg_overlap(ts, slides[0]))
            # the bag lengths the reference's loader actually feeds (BASELINE config 1's 512, 4 096, the `threshold` 25 000), same model
            out["sizes"] = [_leg(lambda Lx=Lx: leg_size(ts, eng, sizes, Lx, not args.no_dropout)) for Lx in (512, 4096, 25000) if Lx != L]
            del ts, eng, slides
            torch.cuda.empty_cache()
            out["sizes"].append(_leg(lambda: leg_real_pathways(args, L)))
            torch.cuda.empty_cache()
            def module_leg():
                rec = run_module(args, steps=8, warmup=4, optim="fused")
                torch.cuda.empty_cache()
                ref = run_module(args, steps=6, warmup=4, optim="torch")      # the trainer's loop with NOTHING but the model import replaced
                rec["torch_adamw"] = {"ms_per_step": ref["ms_per_step"], "value": ref["value"], "optimizer": ref["optimizer"]}
                return _brief(rec)
            out["module_api"] = _leg(module_leg)
            torch.cuda.empty_cache()
            out["titan"] = _leg(lambda: _brief(run_titan(args, steps=16, warmup=8, patches=4096, ragged=True, cpu_baseline=False)))
        if world == 1 and not args.no_cpu_baseline and not args.dp_rehearsal:
            out["cpu_baseline"] = cpu_baseline_leg(L, "real" if args.pathways == 331 else args.pathways)
            out["cpu_baseline_reference"] = CPU_REFERENCE
        print(json.dumps(out))
    if dp_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
