"""Runs tools/experiments/issue_probe.hip: time per loop iteration of each instruction mix at 1 / 2 / 3 waves per SIMD."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "build_variants", "issue_probe.so"))
lib.issue_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(16, device="cuda")
ncu = torch.cuda.get_device_properties(0).multi_processor_count
names = ["mfma28", "exp32+valu58", "mfma28|exp32+valu58 phased", "same, scheduler free", "exp32", "valu58", "mfma28|valu58", "mfma28|exp32", "fwd: mfma14", "fwd: exp32+valu52", "fwd: phased", "fwd: free"]
iters = 4000
clk = 2.4e9
for occ in (1, 2, 3):
    row = []
    for which in range(len(names)):
        st = torch.cuda.current_stream().cuda_stream
        lib.issue_probe(which, sink.data_ptr(), ncu * occ, 200, st); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = lib.issue_probe(which, sink.data_ptr(), ncu * occ, iters, st); e1.record(); torch.cuda.synchronize()
            assert rc == 0
            best = min(best, e0.elapsed_time(e1))
        # per SIMD: occ waves each ran `iters` iterations -> SIMD cycles per wave-iteration (at a nominal 2.4 GHz)
        row.append(best * 1e-3 * clk / iters / occ)
    print(f"{occ} wave(s)/SIMD  cycles per wave-iteration @2.4GHz: " + "  ".join(f"{n}: {v:.0f}" for n, v in zip(names, row)), flush=True)
