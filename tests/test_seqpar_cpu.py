"""Sequence-parallel DilatedAttention: the oracle's restatement against the golden recorded from the REFERENCE's
own gathering / gather_kv / scattering run on W simulated ranks (tests/golden/make_golden.py seqpar), outputs and all
gradients; plus the host-side planning of modaltune_amd.seqpar (which branches stay local, which gather over which ranks)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import unit_inputs  # noqa: E402


@pytest.mark.parametrize("name", sorted(unit_inputs.SEQPAR_CASES))
def test_oracle_sequence_parallel_attention_vs_reference_golden(name):
    from oracle import modaltune_oracle as O
    g = np.load(os.path.join(HERE, "golden", "unit_seqpar.npz"))
    W, B, L, segs, ratios = unit_inputs.SEQPAR_CASES[name]
    q, k, v, dy = (torch.from_numpy(a) for a in unit_inputs.seqpar_inputs(int(g["seed"]), name))
    qs, ks, vs = ([t[r].clone().requires_grad_(True) for r in range(W)] for t in (q, k, v))
    mixed = O.dilated_attention_core_sp(qs, ks, vs, segs, ratios)
    sum((mixed[r] * dy[r]).sum() for r in range(W)).backward()

    def err(a, key):        # the fixture is stored in fp32
        b = torch.from_numpy(g[key]).double()
        return float((a.detach() - b).abs().max() / b.abs().max())
    for r in range(W):
        assert err(mixed[r], f"{name}_attn{r}") < 2e-7
        assert err(qs[r].grad, f"{name}_dq{r}") < 2e-7
        assert err(ks[r].grad, f"{name}_dk{r}") < 2e-7
        assert err(vs[r].grad, f"{name}_dv{r}") < 2e-7


def test_oracle_sequence_parallel_single_rank_is_the_plain_attention():
    from oracle import modaltune_oracle as O
    r = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(2, 40, 16, 48, generator=r, dtype=torch.float64) for _ in range(3))
    a = O.dilated_attention_core_sp([q], [k], [v], [8, 20, 64], [1, 2, 4])[0]
    assert torch.equal(a, O.dilated_attention_core(q, k, v, [8, 20, 64], [1, 2, 4]))


def test_sequence_parallel_planning_groups_and_workspace():
    """Host-side planning only (no GPU): which branches stay local, which gather over which ranks, the query limits of the
    group plans, and that the workspace offsets used for the row copies are the library's own layout."""
    from modaltune_amd import ops
    from modaltune_amd.seqpar import SeqParallelAttention
    segs, ratios, B, L = [160, 320, 640, 5120], [1, 1, 2, 4], 2, 320
    for rank in range(4):
        sp = SeqParallelAttention(segs, ratios, B, L, device="cpu", rank=rank, world=4)
        assert sp.nb_loc == 2 and sp.nb == 4
        assert [(g.first, g.size) for g in sp.groups] == [(rank // 2 * 2, 2), (0, 4)]
        assert [b.seg for b in sp.branches] == [160, 320, 320, 320] and [b.nseg for b in sp.branches] == [2, 1, 1, 1]
        pair, allr = sp.groups
        assert pair.plan.N == 640 and pair.plan.n[0] == 320 and pair.plan.qlimit[0] == 160
        assert allr.plan.N == 1280 and allr.plan.n[0] == 320 and allr.plan.qlimit[0] == 80
        assert sp.ws_off[-1] * 2 == ops.dilated_attn_bwd_workspace_bytes(sp.plan_full)
        assert pair.ws_off[-1] * 2 == pair.ws_bytes and allr.ws_off[-1] * 2 == allr.ws_bytes
        assert sp.pay == B * 16 * (160 + 80) * 144
    one = SeqParallelAttention(segs, ratios, B, L, device="cpu", rank=0, world=1)      # one rank: nothing gathers
    assert one.nb_loc == 4 and not one.groups
    with pytest.raises(ValueError):
        SeqParallelAttention([100, 500], [1, 2], 1, 320, device="cpu", rank=0, world=2)     # 500 % 320 != 0 (DA:63)
    with pytest.raises(ValueError):
        SeqParallelAttention([100, 660], [1, 4], 1, 330, device="cpu", rank=0, world=2)     # 330 % 4 != 0
