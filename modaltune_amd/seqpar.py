"""LongNet sequence parallelism for the dilated-attention core.

Reference: `DilatedAttention` with `args.seq_parallel` (torchscale/component/dilated_attention.py:61-111,212-255) and its
`Allgather` autograd function (torchscale/component/utils.py:43-82).  Each rank holds one chunk of `Lloc` tokens of the
sequence.  A branch (sl, dr) whose segment fits the chunk (sl <= Lloc) is purely local -- segments are counted from the
chunk's first row.  A longer segment (sl > Lloc, sl % Lloc == 0) spans `sl // Lloc` consecutive ranks: the chunk is one
segment, every rank sparsifies its own K / V, the group's sparse K / V are concatenated (all-gather) and the rank's own
queries attend over them; in the backward a rank's dK / dV is the sum over its group (reduce-scatter).

MI355X form (one process per GPU, RCCL over xGMI; what differs from the reference's call pattern):
  * ONE all-gather per layer: the dense head-major k | v slab of the chunk (the reference gathers the sparse K and V of every
    long branch separately: 2 collectives per branch), and ONE reduce-scatter per layer for all long branches' dK / dV;
  * the long branches run through the SAME kernels as the local ones: the group's K / V are laid out as one sequence of
    G * Lloc rows (chunks of Lloc rows keep their dilation residues because Lloc % dr == 0; key order is irrelevant to a
    non-causal softmax), the rank's queries sit in its first Lloc rows, and the plan's `qlimit` tells the kernels that only
    the first Lloc / dr sparse entries act as queries -- no redundant query work, no second kernel family;
  * branch outputs / LSEs are moved between the two row spaces by strided row copies (`mt_copy_rows_f32` with row maps), so the
    branch mix, the inner LayerNorm and the final combine see ordinary local tensors.
Everything between the collectives is launched on the current stream through the C ABI; torch only allocates and runs the
collectives -- the same calls on RCCL and on gloo (the rehearsal backend stages through host copies of the same fp16 payloads).
NOT routed through `Engine`: ModalTune never enables seq_parallel (config.py:60), so this is the standalone op.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops
from .config import Branch
from .ops import rowmap

H, HD, DM = 16, 48, 768
F16, F32 = torch.float16, torch.float32


def _f32(t: torch.Tensor) -> torch.Tensor:
    """fp16 storage seen as fp32 words (row copies move bits; 48 halves = 24 words)."""
    return t.view(torch.float32)


class _Group:
    """The long branches that share one set of ranks [first, first + G)."""

    def __init__(self, first: int, size: int):
        self.first, self.size = first, size
        self.ranks = list(range(first, first + size))
        self.branches: List[int] = []          # indices into the ordered branch list
        self.plan = None
        self.ws_bytes = 0


class SeqParallelAttention:
    """q, k, v (head-major fp16 slab [3][16][B * Lloc][48], q pre-scaled by MT_QK_SCALE_LOG2) -> LN(mix of the branch outputs)
    for ONE layer, forward and backward, with the sequence sharded over the ranks of `group`."""

    def __init__(self, seg_lengths: Sequence[int], ratios: Sequence[int], B: int, Lloc: int, group=None, device="cuda",
                 rank: Optional[int] = None, world: Optional[int] = None):
        """rank / world default to the process group's; passing them explicitly builds the plan of another rank (tests)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.W = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.B, self.L, self.M = B, Lloc, B * Lloc
        self.dev = torch.device(device)
        local, long_ = [], []
        for sl, dr in zip(seg_lengths, ratios):
            sl, dr = int(sl), int(dr)
            if self.W > 1 and sl > Lloc:                     # DA:90-95
                if sl % Lloc:
                    raise ValueError(f"segment length {sl} is not a multiple of the local sequence length {Lloc} (DA:63)")
                if Lloc % dr:
                    raise ValueError(f"local length {Lloc} must be a multiple of the dilation {dr} of a gathered branch "
                                     "(the reference would insert zero keys inside the gathered sequence)")
                nrps = sl // Lloc
                first = self.rank // nrps * nrps
                long_.append((sl, dr, first, min(self.W, first + nrps) - first))
            else:
                local.append((sl, dr))
        long_.sort(key=lambda t: (t[3], t[2]))
        self.nb_loc, self.nb = len(local), len(local) + len(long_)
        # all branches in LOCAL geometry (what the mix / combine kernels see); the local ones first, so that the per-branch
        # strides of o_br / lse_br / delta_br / the workspace are the same for the prefix plan of the local launch
        br = []
        for sl, dr in local + [(t[0], t[1]) for t in long_]:
            s = min(sl, Lloc)
            br.append(Branch(seg=s, ratio=dr, nseg=-(-Lloc // s), n=-(-s // dr)))
        self.branches = br
        self.plan_full = ops.make_plan(br, Lloc, B)
        self.plan_loc = ops.make_plan(br[:self.nb_loc], Lloc, B) if self.nb_loc else None
        self.ws_off = [0]
        for b in br:
            self.ws_off.append(self.ws_off[-1] + B * Lloc * 3 * (H // b.ratio) * HD)   # halves, token-major (attn_common.h: make_plan)
        self.groups: List[_Group] = []
        for i, (sl, dr, first, size) in enumerate(long_):
            g = next((x for x in self.groups if (x.first, x.size) == (first, size)), None)
            if g is None:
                g = _Group(first, size)
                self.groups.append(g)
            g.branches.append(self.nb_loc + i)
        for g in self.groups:
            Ng = g.size * Lloc
            gb = [Branch(seg=Ng, ratio=br[i].ratio, nseg=1, n=Ng // br[i].ratio) for i in g.branches]
            g.plan = ops.make_plan(gb, Ng, B, qlimit=[Lloc // br[i].ratio for i in g.branches])
            g.ws_bytes = ops.dilated_attn_bwd_workspace_bytes(g.plan)
            g.ws_off = [0]
            for b in gb:
                g.ws_off.append(g.ws_off[-1] + B * Ng * 3 * (H // b.ratio) * HD)
        # exchange payload: per long branch the compact token rows [B][Lloc][q|k|v][16 / dr][48] of one chunk
        self.pay_off = [0]
        for g in self.groups:
            for i in g.branches:
                self.pay_off.append(self.pay_off[-1] + B * Lloc * 3 * (H // br[i].ratio) * HD)
        self.pay = self.pay_off[-1]

    # ------------------------------------------------------------------ helpers
    def _new(self, *shape, dtype=F16, zero=False):
        return (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)

    def _to_group_rows(self, src, dst, segs: int, words: int, Ng: int, chunk: int = 0):
        """rows [seg * Lloc + p] of src -> rows [seg * Ng + chunk * Lloc + p] of dst (fp32 words per row)."""
        ops.copy_rows(src, dst, segs * self.L, words, dmap=rowmap(self.L, Ng, chunk * self.L))

    def _from_group_rows(self, src, dst, segs: int, words: int, Ng: int):
        ops.copy_rows(src, dst, segs * self.L, words, smap=rowmap(self.L, Ng, 0))

    # -- collectives: the SAME torch.distributed calls on every backend (all_gather_into_tensor, all_to_all_single, fp16
    # payloads).  RCCL ("nccl") takes the device tensors as they are; gloo (the CPU / one-GPU rehearsals) takes host copies.
    def _host_staged(self) -> bool:
        return self.dist.get_backend(self.group) == "gloo"

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.W,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # flat views: every backend accepts the concatenated form [W * n] <- [n]
        if self._host_staged():
            h = torch.empty(out.numel(), dtype=t.dtype)
            self.dist.all_gather_into_tensor(h, t.reshape(-1).cpu(), group=self.group)
            out.view(-1).copy_(h)
        else:
            self.dist.all_gather_into_tensor(out.view(-1), t.reshape(-1), group=self.group)
        return out

    def _all_to_all(self, send: torch.Tensor, sizes: List[int]) -> torch.Tensor:
        """Symmetric all-to-all of one flat payload (`sizes[r]` elements to and from rank r)."""
        recv = torch.empty_like(send)
        if self._host_staged():
            hs, hr = send.cpu(), torch.empty(send.shape, dtype=send.dtype)
            self.dist.all_to_all_single(hr, hs, sizes, sizes, group=self.group)
            recv.copy_(hr)
        else:
            self.dist.all_to_all_single(recv, send, sizes, sizes, group=self.group)
        return recv

    def _exchange_plan(self):
        """Who gets which slices of the dK / dV payload: rank rc receives, for every long branch k whose group holds both of
        us, my partial sums for ITS chunk.  Groups are symmetric, so what I send to rc is as long as what rc sends to me."""
        if getattr(self, "_xplan", None) is None:
            per_dst = [[] for _ in range(self.W)]
            k = 0
            for g in self.groups:
                for _ in g.branches:
                    for rc in g.ranks:
                        per_dst[rc].append(k)
                    k += 1
            sizes = [sum(self.pay_off[k + 1] - self.pay_off[k] for k in ks) for ks in per_dst]
            self._xplan = (per_dst, sizes)
        return self._xplan

    def _exchange_sum(self, contrib: torch.Tensor) -> torch.Tensor:
        """contrib fp16 [W, pay] (row rc = my partial dK / dV for rank rc's chunk) -> fp32 [pay]: the sum over the ranks of my
        groups of THEIR row for me (Allgather.backward, TS/component/utils.py:60-80, is a reduce-scatter; here an all-to-all
        of the fp16 partials restricted to the group members -- half the wire bytes of an fp32 reduce-scatter over all W
        ranks, nothing sent to ranks outside the segment -- followed by an fp32 sum on the receiver)."""
        per_dst, sizes = self._exchange_plan()
        send = torch.cat([contrib[rc, self.pay_off[k]:self.pay_off[k + 1]] for rc in range(self.W) for k in per_dst[rc]])
        recv = self._all_to_all(send, sizes)
        red = self._new(self.pay, dtype=F32, zero=True)
        tmp = self._new(self.pay, dtype=F32)
        off = 0
        for src in range(self.W):
            for k in per_dst[src]:
                n = self.pay_off[k + 1] - self.pay_off[k]
                ops.cast_f16_to_f32(recv[off:off + n], tmp[:n], n)
                dst = red[self.pay_off[k]:self.pay_off[k + 1]]
                ops.axpy(dst, tmp[:n], 1.0, dst, n)
                off += n
        return red

    # ------------------------------------------------------------------ forward
    def forward(self, qkv_hm: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor):
        B, L, M, nb = self.B, self.L, self.M, self.nb
        assert qkv_hm.dtype == F16 and qkv_hm.numel() == 3 * H * M * HD
        o_br = self._new(nb, M, DM)
        lse_br = self._new(nb, M, H, dtype=F32)
        if self.plan_loc is not None:
            ops.dilated_attn_fwd(qkv_hm, self.plan_loc, o_br, lse_br)
        slabs = []
        if self.groups:
            kv_all = self._all_gather(qkv_hm.view(3, H * M * HD)[1:].contiguous())        # [W, 2, 16 * M * 48]
            for g in self.groups:
                Ng = g.size * L
                S = self._new(3, H, B * Ng, HD, zero=True)
                self._to_group_rows(_f32(qkv_hm.view(3, -1)[0]), _f32(S[0]), H * B, HD // 2, Ng)
                for c, rc in enumerate(g.ranks):
                    self._to_group_rows(_f32(kv_all[rc]), _f32(S[1:]), 2 * H * B, HD // 2, Ng, chunk=c)
                o_g = self._new(len(g.branches), B * Ng, DM)
                lse_g = self._new(len(g.branches), B * Ng, H, dtype=F32)
                ops.dilated_attn_fwd(S, g.plan, o_g, lse_g)
                for j, i in enumerate(g.branches):
                    self._from_group_rows(_f32(o_g[j]), _f32(o_br[i]), B, DM // 2, Ng)
                    self._from_group_rows(lse_g[j], lse_br[i], B, H, Ng)
                slabs.append(S)
        y = self._new(M, DM)
        stats = self._new(M, 2, dtype=F32)
        lse_tot = self._new(M, H, dtype=F32)
        ops.dilated_mix_ln_fwd(o_br, lse_br, self.plan_full, ln_w, ln_b, y, stats, lse_tot)
        return y, (qkv_hm, o_br, lse_br, lse_tot, stats, slabs, ln_w)

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dy: torch.Tensor) -> torch.Tensor:
        """dy fp16 [M, 768] (gradient of the LN output) -> dqkv fp16 [M, 2304] (q columns: gradient of the pre-scaled q)."""
        qkv_hm, o_br, lse_br, lse_tot, stats, slabs, ln_w = ctx
        B, L, M, nb = self.B, self.L, self.M, self.nb
        dmixed = self._new(H, M, HD)
        delta_br = self._new(nb, M, H, dtype=F32)
        ops.dilated_mix_ln_bwd(dy, o_br, lse_br, lse_tot, self.plan_full, ln_w, stats, dmixed, delta_br)
        ws = self._new(self.ws_off[-1])
        dqkv = self._new(M, 3 * DM)
        if self.plan_loc is not None:
            ops.dilated_attn_bwd_phases(qkv_hm, dmixed, lse_tot, delta_br, self.plan_loc, ws, dqkv, ops.ATTN_BWD_KV | ops.ATTN_BWD_Q)
        if self.groups:
            contrib = self._new(self.W, self.pay, zero=True)                     # fp16 entries, summed in fp32 by the receiver
            k = 0
            for g, S in zip(self.groups, slabs):
                Ng = g.size * L
                dm_g = self._new(H, B * Ng, HD, zero=True)
                self._to_group_rows(_f32(dmixed), _f32(dm_g), H * B, HD // 2, Ng)
                lt_g = self._new(B * Ng, H, dtype=F32, zero=True)
                self._to_group_rows(lse_tot, lt_g, B, H, Ng)
                dl_g = self._new(len(g.branches), B * Ng, H, dtype=F32, zero=True)
                for j, i in enumerate(g.branches):
                    self._to_group_rows(delta_br[i], dl_g[j], B, H, Ng)
                ws_g = self._new(g.ws_bytes // 2, zero=True)
                ops.dilated_attn_bwd_phases(S, dm_g, lt_g, dl_g, g.plan, ws_g, ws_g, ops.ATTN_BWD_KV | ops.ATTN_BWD_Q)
                for j, i in enumerate(g.branches):
                    wr = 3 * (H // self.branches[i].ratio) * HD // 2             # fp32 words per compact token row [q|k|v][16/dr][48]
                    ent = _f32(ws_g[g.ws_off[j]:g.ws_off[j + 1]])                # token rows [B][Ng] of the group space
                    # dq of this rank's queries (its rows are the first Lloc of every pass) + its own dk / dv share -> local workspace
                    ops.copy_rows(ent, _f32(ws[self.ws_off[i]:self.ws_off[i + 1]]), B * L, wr, smap=rowmap(L, Ng, 0))
                    # dk / dv of every chunk of the group -> the owner's slot of the exchange payload
                    for c, rc in enumerate(g.ranks):
                        ops.copy_rows(ent, _f32(contrib[rc, self.pay_off[k]:self.pay_off[k + 1]]), B * L, wr, smap=rowmap(L, Ng, c * L))
                    k += 1
            red = self._exchange_sum(contrib)                                    # [pay] fp32: sum over the ranks of my groups
            red16 = self._new(self.pay)
            ops.cast_f32_to_f16(red, red16)
            k = 0
            for g in self.groups:
                for i in g.branches:
                    wp = (H // self.branches[i].ratio) * HD // 2                 # words of one of q | k | v in a compact token row
                    src = _f32(red16[self.pay_off[k]:self.pay_off[k + 1]])
                    dst = _f32(ws[self.ws_off[i]:self.ws_off[i + 1]])
                    # the k and v thirds of every token row (the q third of the local workspace keeps this rank's own dq)
                    ops.copy_rows(src[wp:], dst[wp:], B * L, 2 * wp, lds=3 * wp, ldd=3 * wp)
                    k += 1
        ops.dilated_attn_bwd_phases(qkv_hm, dmixed, lse_tot, delta_br, self.plan_full, ws, dqkv, ops.ATTN_BWD_COMBINE)
        return dqkv
