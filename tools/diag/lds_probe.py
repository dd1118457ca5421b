"""Which product kernels disturb the LDS reads of a co-resident workgroup of ANOTHER kernel (two HIP streams)?
The victim is tools/diag/lds_probe.hip (fills its LDS with a pattern, re-reads it with b128 / b64 / b32 reads, counts mismatches);
the aggressors are all ops.* launches of one train step of a depth-2 model, recorded once and replayed per op name on a second stream.
    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o build_variants/lds_probe/liblds_probe.so tools/diag/lds_probe.hip
    python tools/diag/lds_probe.py [L] [rounds] [titan]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from modaltune_amd import ops, synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402
from modaltune_amd.trainer import TrainStep  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
probe = ctypes.CDLL(os.path.join(ROOT, "build_variants", "lds_probe", "liblds_probe.so"))
probe.lds_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
probe.lds_probe_sweep_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
probe.lds_probe_count_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
PROBE = os.environ.get("PROBE", "count")
SWEEP = PROBE == "sweep"

if len(sys.argv) > 3 and sys.argv[3] == "titan":
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import titan_standin
    import bench
    from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
    vit = titan_standin.VisionTransformer(mlp_ratio=4.0)
    titan_standin.init_standin(vit, 0)
    sizes = synth.toy_group_sizes(6)
    cfg = titan_model_config(bench.TITAN_JSON, 3, False, 6)
    eng = TitanEngine(cfg, sizes, NativeBackbone(vit, torch.device("cuda", 0)), torch.device("cuda", 0))
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
    ts = TrainStep(eng, lr=0.0, weight_decay=0.0, split_passes=False)
    Lt = 4096 + 4096 // 15
    inp = synth.synth_inputs_titan(Lt, sizes, seed=2000, grid=96)
    x = torch.from_numpy(inp["x"]).cuda().reshape(Lt, -1).contiguous()
    coords = torch.from_numpy(inp["coords"]).cuda().reshape(Lt, 2)
else:
    cfg = ModelConfig(depth=2, interaction_indexes=((0, 0), (1, 1)))
    sizes = synth.toy_group_sizes(6)
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, 3))
    ts = TrainStep(eng, lr=0.0, weight_decay=0.0, split_passes=False)
    inp = synth.synth_inputs(L, sizes, 3, grid=128)
    x = torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1)
    coords = inp["coords"]
ts.set_projector(synth.projector_state(3))
genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
text = torch.from_numpy(inp["text"]).cuda()
ops.TIMER = {}                     # launch-by-launch form of the backbone layers (every kernel its own ops.* call)
ts.step(x, coords, genes, text, update=False)
torch.cuda.synchronize()

REC, KEEP = [], []
skip = ("check", "make_plan", "dropout_spec", "struct_of", "sgemm_problem", "make_dense_plan", "timer_summary", "measured_mfma_peak_tflops",
        "dilated_attn_bwd_workspace_bytes", "alibi_dist_halves", "pool_attn_workspace_floats", "rowmap", "mfma_probe", "build_info")
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not isinstance(getattr(ops, n), type) and not n.startswith("_") and n not in skip]
orig = {n: getattr(ops, n) for n in names}


def keep(o):
    if torch.is_tensor(o):
        KEEP.append(o)
    elif isinstance(o, (list, tuple)):
        for v in o:
            keep(v)
    elif isinstance(o, dict):
        for v in o.values():
            keep(v)


depth = [0]
for n in names:
    def mk(n):
        def w(*a, **k):
            if depth[0] == 0:
                REC.append((n, a, k))
                keep(a); keep(k)
            depth[0] += 1
            try:
                return orig[n](*a, **k)
            finally:
                depth[0] -= 1
        return w
    setattr(ops, n, mk(n))
from modaltune_amd import tape as tape_mod  # noqa: E402
_new, _zl, _empty = tape_mod.Tape.new, tape_mod.Tape.zeros_like, torch.empty
tape_mod.Tape.new = lambda self, *s: (lambda t: (KEEP.append(t), t)[1])(_new(self, *s))
tape_mod.Tape.zeros_like = lambda self, t: (lambda z: (KEEP.append(z), z)[1])(_zl(self, t))
torch.empty = lambda *a, **k: (lambda t: (KEEP.append(t), t)[1])(_empty(*a, **k))
ts.step(x, coords, genes, text, update=False)
torch.cuda.synchronize()
torch.empty = _empty
ops.TIMER = None
for n in names:
    setattr(ops, n, orig[n])
by_name = {}
for n, a, k in REC:
    key = n
    if n == "gemm_nt":
        key = f"gemm_nt[{a[3]}x{a[4]}x{a[5]}]"
    elif n in ("dilated_attn_bwd_phases", "dense_attn_bwd") or n.startswith("_"):
        key = n
    by_name.setdefault(key, []).append((n, a, k))
print("recorded", len(REC), "launches of", len(by_name), "kinds", flush=True)

res = torch.zeros(24, dtype=torch.int64, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


ONLY = os.environ.get("ONLY")


def trial(name, calls, threads=512, lds=25 * 1024, grid=12, iters=8, launches=3000):
    """The victim is launched `launches` times back to back on stream A (a small grid, as the token-side kernels are) while the
    aggressor's recorded launches cycle on stream B (one every fourth victim launch): the victim's workgroups start on CUs that the
    aggressor's waves already occupy."""
    res.zero_()
    torch.cuda.synchronize()
    ci = 0
    with torch.cuda.stream(sa):
        st = torch.cuda.current_stream().cuda_stream
        for i in range(launches):
            if PROBE == "count":
                probe.lds_probe_count_launch(1, grid, threads, 65, iters * 8, res.data_ptr(), st)
                probe.lds_probe_count_launch(0, grid, threads, 65, iters * 8, res.data_ptr() + 64, st)
            elif SWEEP:
                probe.lds_probe_sweep_launch(grid, threads, 65, iters, res.data_ptr(), st)
            else:
                probe.lds_probe_launch(grid, threads, lds, iters * 50, res.data_ptr(), st)
            if calls and i % 4 == 0:
                with torch.cuda.stream(sb):
                    n, a, k = calls[ci % len(calls)]
                    orig[n](*a, **k)
                    ci += 1
    torch.cuda.synchronize()
    r_ = res.tolist()
    if PROBE == "count":
        print(f"{name:34s} x{len(calls):3d}: counted wait released EARLY: mixed 4-byte + 16-byte reads {r_[1]} (16-byte result missing {r_[0]}; by 16-lane group {r_[4:8]})"
              f" | 8-byte + 16-byte reads {r_[9]}", flush=True)
        return
    if SWEEP:
        print(f"{name:34s} x{len(calls):3d}: wrong sweeps by dword of the 16-byte read {r_[0:4]}, by 16-lane group {r_[4:8]}", flush=True)
        return
    tag = ""
    if r_[3]:
        first, who = r_[4] >> 32, r_[4] & 0xffffffff
        tag = f"  e.g. kind {('b128', 'b64', 'b32', 'b128-bcast')[(first >> 30) & 3]} dword {first & 0xfffffff} elem {(first >> 28) & 3} tid {who & 0xfff} (lane {who & 63}) got {r_[5] >> 32:#x} want {r_[5] & 0xffffffff:#x}"
    print(f"{name:34s} x{len(calls):3d}: bad words b128 {r_[0]} b64 {r_[1]} b32 {r_[2]} | broadcast b128 {r_[6]} b64 {r_[7]} b32 {r_[8]} (threads with errors {r_[3]}){tag}", flush=True)


trial("(alone)", [])
for key, calls in sorted(by_name.items()):
    if ONLY and not key.startswith(ONLY):
        continue
    trial(key, calls)
    if ONLY:
        trial(key + " probe 256 thr", calls, threads=256)
        trial(key + " probe 64 KB LDS", calls, lds=64 * 1024)
