import ctypes, torch
lib = ctypes.CDLL("build_variants/mfma_tick_probe.so")
out = torch.zeros(256, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
src = torch.zeros(1 << 29, dtype=torch.float16, device="cuda")
names = {0: "48 MFMAs alone", 1: "+ 14 ds_read_b128 (one per 3 MFMAs), consumed after the loop", 2: "+ 7 LDS-DMA pieces (one per 7 MFMAs)", 3: "+ 14 ds_read_b128 consumed by the last 14 MFMAs", 4: "48 MFMAs, operands rotating through 14 fragment registers as in the GEMM slice"}
for mode in (0, 1, 2, 3, 4):
    for rep in range(2):
        lib.run_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(src.data_ptr()), 1000, 256, mode)
        torch.cuda.synchronize()
    print(f"mode {mode} ({names[mode]}): {out.double().mean().item() / 1000:.1f} ticks per iteration (min {out.min().item() / 1000:.1f}, max {out.max().item() / 1000:.1f})")
