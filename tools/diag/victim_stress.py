"""Which kernel, running on ANOTHER stream, corrupts the prompt self-attention (mt_token_mha_fwd) -- or any small victim kernel?
One backbone layer's launches (forward and backward, launch by launch: csrc/layer.hip's list) are the aggressors, one at a time, in
a loop on stream B; the victim loops on stream A and counts outputs that differ from its solo result.
    python tools/diag/victim_stress.py [L] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops, synth  # noqa: E402
from modaltune_amd._lib import rowmap  # noqa: E402
from modaltune_amd.config import DILATED_RATIOS, ModelConfig, branch_table  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
B, D, Fd = 2, 768, 3072
N = L + 1
M = B * N
cfg = ModelConfig(depth=1, interaction_indexes=((0, 0),))
sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, "cuda")
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=31))
eng._build_caches()
ws = eng._workspace(B, L)
plan = ops.make_plan(branch_table(N, eng.seg_lengths, DILATED_RATIOS), N, B)
eng._ctx = dict(B=B, L=L, N=N, M=M, Mp=B * L, ws=ws, plan=plan, patch_map=rowmap(L, N, 1))
eng._drop_now = False
ops.TIMER = {}                     # launch-by-launch form of the layer
tape = eng.tape
tape.grad_enabled = True
tape.reset()
g = torch.Generator(device="cuda").manual_seed(1)
ws["hin0"].copy_(torch.randn(M, D, generator=g, device="cuda"))
eng._layer(0, ws["hout0"], None, defer=False)
ws["dh"].copy_(torch.randn(M, D, generator=g, device="cuda") * 64)
eng._ctx["dh16_valid"] = False
bwd = tape.back[-1]
bwd()
torch.cuda.synchronize()
ops.TIMER = None
t, f16 = eng.store.tensors, eng._frozen16
p = "encoder.layers.0."
u16, t16, br16, qkv, obr, lsebr, lsetot, a1 = ws["u16"], ws["t16"], ws["br16"], ws["qkv0"], ws["obr0"], ws["lsebr0"], ws["lsetot0"], ws["a1_0"]
hin, hmid, st1, stin, st2, stf = ws["hin0"], ws["hmid0"], ws["st1_0"], ws["stin_0"], ws["st2_0"], ws["stf_0"]
dh, dy16, dt16, da1 = ws["dh"], ws["dy16"], ws["dt16"], ws["da1"]
AGG = {
    "ln_fwd": lambda: ops.layernorm_fwd(hin, t[p + "self_attn_layer_norm.weight"], t[p + "self_attn_layer_norm.bias"], u16, st1, M, D),
    "gemm_qkv(ps)": lambda: ops.gemm_nt(u16, f16[p + "qkv"].w, qkv, M, 3 * D, D, bias=f16[p + "bqkv"], epilogue=ops.EPI_QKV_HM),
    "attn_fwd": lambda: ops.dilated_attn_fwd(qkv, plan, obr, lsebr),
    "mix_ln_fwd": lambda: ops.dilated_mix_ln_fwd(obr, lsebr, plan, t[p + "self_attn.inner_attn_ln.weight"], t[p + "self_attn.inner_attn_ln.bias"], u16, stin, lsetot),
    "gemm_out(768^2)": lambda: ops.gemm_nt(u16, f16[p + "out"].w, br16, M, D, D, bias=t[p + "self_attn.out_proj.bias"]),
    "add_ln_fwd": lambda: ops.add_layernorm_fwd(hin, br16, t[p + "final_layer_norm.weight"], t[p + "final_layer_norm.bias"], hmid, u16, st2, M, D),
    "gemm_fc1(ps)": lambda: ops.gemm_nt(u16, f16[p + "fc1"].w, a1, M, Fd, D, bias=t[p + "ffn.fc1.bias"]),
    "ln_gelu_fwd": lambda: ops.layernorm_fwd(a1, t[p + "ffn.ffn_layernorm.weight"], t[p + "ffn.ffn_layernorm.bias"], t16, stf, M, Fd, gelu_in=True),
    "gemm_fc2(pp)": lambda: ops.gemm_nt(t16, f16[p + "fc2"].w, br16, M, D, Fd, bias=t[p + "ffn.fc2.bias"]),
    "gemm_dfc2": lambda: ops.gemm_nt(dy16, f16[p + "fc2"].wt, dt16, M, Fd, D),
    "ln_gelu_bwd": lambda: ops.layernorm_bwd(dt16, a1, t[p + "ffn.ffn_layernorm.weight"], stf, da1, M, Fd, gelu_in=True),
    "gemm_dfc1(pp)": lambda: ops.gemm_nt(da1, f16[p + "fc1"].wt, dy16, M, D, Fd),
    "ln_bwd": lambda: ops.layernorm_bwd(dy16, hmid, t[p + "final_layer_norm.weight"], st2, ws["scratch32"].view(-1)[:M * D].view(M, D) if ws["scratch32"].numel() >= M * D else dh, M, D),
    "mix_ln_bwd": lambda: ops.dilated_mix_ln_bwd(u16, obr, lsebr, lsetot, plan, t[p + "self_attn.inner_attn_ln.weight"], stin, ws["dmixed"], ws["delta"]),
    "attn_bwd(kv+q+combine)": lambda: ops.dilated_attn_bwd(qkv, ws["dmixed"], lsetot, ws["delta"], plan, ws["attn_ws"], ws["dqkv16"]),
    "gemm_dqkv": lambda: ops.gemm_nt(ws["dqkv16"], f16[p + "qkv"].wt, dy16, M, D, 3 * D),
    "copy(hbm)": lambda: ws["dt16"].copy_(ws["t16"]),
}

# victim: the prompt self-attention at the step's geometry (one pass, 65 tokens, 12 heads x 16)
T, E, H = 65, 192, 12
q, k, v = (torch.randn(1, T, E, generator=g, device="cuda") for _ in range(3))
out, probs = torch.empty(1, T, E, device="cuda"), torch.empty(1 * H * T * T, device="cuda")
ops.token_mha_fwd(q, k, v, out, probs, 1, T, E, H)
torch.cuda.synchronize()
ref_out, ref_probs = out.clone(), probs.clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(2, dtype=torch.int64, device="cuda")


def trial(name, agg, iters):
    bad.zero_()
    torch.cuda.synchronize()
    stop_after = iters
    with torch.cuda.stream(sb):
        for _ in range(max(1, iters // 20)):
            agg()
    with torch.cuda.stream(sa):
        for i in range(stop_after):
            ops.token_mha_fwd(q, k, v, out, probs, 1, T, E, H)
            bad[0] += (out != ref_out).any()
            bad[1] += (probs != ref_probs).any()
            if i % 20 == 0:
                with torch.cuda.stream(sb):
                    agg()
    torch.cuda.synchronize()
    print(f"{name:28s}: victim wrong out {int(bad[0])} / probs {int(bad[1])} of {iters}", flush=True)


trial("(alone)", lambda: None, ITERS)
for name, agg in AGG.items():
    trial(name, agg, ITERS)
