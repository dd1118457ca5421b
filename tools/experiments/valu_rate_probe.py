import ctypes, torch
lib = ctypes.CDLL("build_variants/valu_rate_probe.so")
out = torch.zeros(256 * 12, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
names = {0: "64 v_exp_f32", 1: "64 v_fma_f32", 2: "64 v_exp_f32 + 64 v_fma_f32 interleaved", 3: "64 v_pk_fma_f32", 4: "64 v_sqrt_f32",
         5: "64 v_cvt_pk_f16_f32", 6: "64 v_fma_mix_f32", 7: "64 v_exp_f32 + 16 MFMA 32x32x16", 8: "16 MFMA 32x32x16", 9: "64 v_fma_f32 + 16 MFMA",
         10: "64 v_max3_f32", 11: "64 v_mov_b32", 12: "64 v_exp_f32 + 192 v_fma_f32 (1:3)",
         13: "64 v_cvt_pkrtz_f16_f32", 14: "64 v_pk_mul_f32", 15: "64 v_pk_add_f32", 16: "64 v_mul_f32", 17: "64 v_perm_b32"}
for waves in (1, 2, 3):
    for mode in range(18):
        for rep in range(2):
            out.zero_()
            lib.run_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sink.data_ptr()), 500, 256, mode, waves)
            torch.cuda.synchronize()
        v = out[: 256 * 4 * waves].double()
        print(f"{waves} wave(s)/SIMD  mode {mode:2d} ({names[mode]}): {v.mean().item() / 500:.1f} ticks per iteration per wave")
