"""Explicit forward / backward schedule of the Modal-Adapter train step over the C-ABI HIP kernels.

Mirrors LongNetGeneAdapter.forward (reference models/aggregators/longvit_adapter.py:205-347) for B task
passes batched (multitask_forward, train_modaltune.py:156-179, batched as B = 3; mathematically identical
with dropout off): frozen LongNet backbone (torchscale encoder.py:121-175) interleaved with the Injector /
Extractor adapters (vitadapter/adapter_modules.py:296-369,459-523), the gene encoder
(genomic_utils/gene_encoder.py:194-223) and the fusion head (longvit_adapter.py:309-347).

torch is the allocator and the stream owner only: every arithmetic op below is a HIP kernel from
include/modaltune_hip.h.  The patch side (B*N rows) runs fp16 operands / fp32 accumulation with an fp32
residual stream; the token side (T <= 66 rows) runs fp32.  Backward is selective: activation gradients flow
through all frozen layers, weight gradients exist only for adapter / gene / head parameters.
"""
from __future__ import annotations


import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import rowmap
from .config import ModelConfig, branch_table, coords_to_rowcol, segment_lengths, sincos_1d_table, DILATED_RATIOS
from .synth import param_specs
from .tape import Param, Tape, Var

H16 = torch.float16
F32 = torch.float32


class ParamStore:
    """All model tensors under the reference's state_dict names.  Trainable tensors are views of one flat fp32
    buffer (16-byte aligned slots, state_dict order) with a matching flat gradient buffer, so AdamW and the
    data-parallel gradient all-reduce are one launch / one collective each."""

    def __init__(self, cfg: ModelConfig, group_sizes: Sequence[int], device):
        self.cfg, self.group_sizes, self.device = cfg, list(group_sizes), device
        self.specs = param_specs(cfg, group_sizes)
        off = 0
        self.slots: Dict[str, tuple] = {}
        for k, shape, kind, train in self.specs:
            if train:
                n = int(np.prod(shape))
                self.slots[k] = (off, n, shape)
                off += (n + 3) // 4 * 4
        self.n_flat = off
        self.sync = None                  # set by TrainStep: waits for a sharded parameter all-gather in flight (dp.GradReducer.wait_params)
        self.flat = torch.zeros(off, dtype=F32, device=device)
        self.flat_grad = torch.zeros(off, dtype=F32, device=device)
        self.tensors: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        for k, shape, kind, train in self.specs:
            if train:
                o, n, _ = self.slots[k]
                self.tensors[k] = self.flat[o:o + n].view(shape)
                self.grads[k] = self.flat_grad[o:o + n].view(shape)
            else:
                self.tensors[k] = torch.zeros(shape, dtype=F32, device=device)

    def load(self, state: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        missing = [k for k in self.tensors if k not in state]
        extra = [k for k in state if k not in self.tensors]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} unexpected {extra[:5]}")
        for k, t in self.tensors.items():
            if k in state:
                v = state[k]
                v = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v
                if tuple(v.shape) != tuple(t.shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(t.shape)}")
                t.copy_(v.to(device=self.device, dtype=F32))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        if self.sync is not None:
            self.sync()               # a sharded optimiser step's parameter all-gather may still be in flight (dp.GradReducer)
        return {k: v.detach().clone() for k, v in self.tensors.items()}

    def param(self, k: str) -> Param:
        return Param(self.tensors[k], self.grads.get(k))

    def new_grad_set(self):
        """A second flat gradient buffer with its per-tensor views: (flat, views).  TrainStep's pass groups each accumulate into their
        own (their token-side weight gradients are read-modify-write, not atomic) and the sets are summed before the optimiser."""
        flat = torch.zeros(self.n_flat, dtype=F32, device=self.device)
        return flat, {k: flat[o:o + n].view(shape) for k, (o, n, shape) in self.slots.items()}

    def use_grad_set(self, flat, views):
        """Point the gradient side of the store at (flat, views) -- every Param / closure created from now on binds to it -- and
        return the previous pair (hand it back the same way)."""
        old = (self.flat_grad, self.grads)
        self.flat_grad, self.grads = flat, views
        return old


class _Lease:
    """Hands a fresh call's workspace store back to the engine's pool when the last holder (workspace dict, tape, share dict) dies."""
    __slots__ = ("pool", "store")

    def __init__(self, pool: list, store: dict):
        self.pool, self.store = pool, store

    def __del__(self):
        try:
            if len(self.pool) < 8:
                self.pool.append(self.store)
        except Exception:       # interpreter shutdown
            pass


QK_SCALE_LOG2 = 0.14433756729740643 * 1.4426950408889634      # 48^-1/2 * log2(e) (include/modaltune_hip.h: MT_QK_SCALE_LOG2)


def _scaled_copy(t: torch.Tensor, alpha: float) -> torch.Tensor:
    """alpha * t as a new fp32 tensor (mt_axpy_dev; the scale factor of the attention's q projection)."""
    out = torch.empty_like(t)
    a = torch.full((1,), float(alpha), dtype=F32, device=t.device)
    ops.axpy_dev(None, t.contiguous(), a, out)
    return out


class _W16:
    """fp16 caches of one nn.Linear weight: as stored [N,K] (forward) and transposed [K,N] (dX GEMM)."""

    def __init__(self, srcs: Sequence[torch.Tensor], device, need_t: bool = True):
        self.srcs = list(srcs)
        self.N = sum(s.shape[0] for s in srcs)
        self.K = srcs[0].shape[1]
        self.w = torch.empty(self.N, self.K, dtype=H16, device=device)
        self.wt = torch.empty(self.K, self.N, dtype=H16, device=device) if need_t else None
        self.refresh()

    def refresh(self):
        r = 0
        for s in self.srcs:
            n = s.shape[0]
            ops.pack_weight(s, self.w[r:r + n], n, self.K, transpose=False)
            r += n
        if self.wt is not None:
            if len(self.srcs) == 1:
                ops.pack_weight(self.srcs[0], self.wt, self.N, self.K, transpose=True)
            else:   # concatenated rows -> columns of the transposed copy
                full = torch.empty(self.N, self.K, dtype=F32, device=self.w.device)
                r = 0
                for s in self.srcs:
                    full[r:r + s.shape[0]].copy_(s)
                    r += s.shape[0]
                ops.pack_weight(full, self.wt, self.N, self.K, transpose=True)


class Engine:
    def __init__(self, cfg: ModelConfig, group_sizes: Sequence[int], device="cuda"):
        cfg.validate()
        self.cfg, self.device = cfg, torch.device(device)
        self.group_sizes = list(group_sizes)
        self.store = ParamStore(cfg, group_sizes, self.device)
        self.tape = Tape(self.device)
        self._fresh_arenas = {"need": 0, "pool": []}      # gradient arenas of the per-call tapes (module API), see Tape.__init__
        self.tape.on_realloc = self._bump_generation
        self._main_tape = self.tape
        self.seg_lengths = segment_lengths(cfg.max_wsi_size, cfg.tile_size)
        self.pos_table = torch.from_numpy(sincos_1d_table(cfg.slide_ngrids, cfg.embed_dim // 2)).to(self.device)
        self._frozen16: Dict[str, _W16] = {}
        self._train16: Dict[str, _W16] = {}
        self._ws: Dict[tuple, Dict[str, torch.Tensor]] = {}          # (B, L, slot) -> views of the flat storage
        self._ws_store: Dict[tuple, dict] = {}                        # (B, slot) -> {cap, flat buffers}
        self._caches_ready = False
        # Bumped whenever storage that a captured hipGraph may point to is replaced (workspace growth, rebuilt fp16 weight
        # caches, the stochastic toggle's extra buffer): graph owners (TrainStep, EmbeddingExtractor) key their captures on it.
        self.generation = 0
        # backward stages whose parameter gradients are final (data-parallel bucket boundaries): set by TrainStep
        self.grad_ready_hook = None
        self.record_markers = False           # mark the blocks on the tape even without a hook (TrainStep's pass groups run their backwards stage by stage)
        self._coord_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._coord_err_host = None           # pinned mirror of _coord_err + the event of its last async read-back:
        self._coord_err_event = None          # bad coords raise at the next call WITHOUT a host sync (poll_inputs)
        self._pin_ring: List[tuple] = []      # pinned staging slots for host-side coords (no pageable-copy sync per step)
        self._pin_next = 0
        self.collect_taps = False      # tests: keep cls / token states after every interaction block
        self.taps: Dict[str, torch.Tensor] = {}
        self.T = cfg.num_tokens
        self._mean_w = torch.full((max(self.T, 2),), 1.0 / max(1, cfg.gene.final_groups), device=self.device)
        # pathway networks (gene_networks.{i}.{0,1}.0.*): offset tables into the flat parameter / gradient buffers
        sl, G = self.store.slots, len(self.group_sizes)
        offs = [[sl[f"gene_encoder.gene_networks.{i}.{a}.0.{w}"][0] for a, w in ((0, "weight"), (0, "bias"), (1, "weight"), (1, "bias"))]
                for i in range(G)]
        goff = np.concatenate([[0], np.cumsum(self.group_sizes)[:-1]]).astype(np.int64)
        self._gene_offs = torch.tensor(offs, dtype=torch.int64, device=self.device)
        self._gene_sizes = torch.tensor(self.group_sizes, dtype=torch.int32, device=self.device)
        self._gene_goff = torch.from_numpy(goff).to(self.device)
        self._gene_total = int(sum(self.group_sizes))
        # train-mode stochastic ops (fact 3 of the survey: the frozen backbone stays in train mode): counter-based masks
        # from a device-side {seed_lo, seed_hi, step, 0}; off by default (parity is defined at p = 0)
        self.stochastic = False
        self.rng = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._drop_now = False
        self._fresh_calls, self._site_base = 0, 0
        self._fresh_tick = 0
        self._fresh_pool: Dict[int, list] = {}      # B -> free workspace stores of the module API's per-call workspaces
        dpr = np.linspace(0.0, float(cfg.drop_path_rate), cfg.depth) if cfg.depth > 1 else np.zeros(1)
        self._layer_path_p = [float(v) for v in dpr]          # ENC:37-41
        self.leaf_stream = os.environ.get("MT_LEAF_STREAM", "0") == "1"
        self._side, self._side_refs, self._side_busy = None, [], False

    def _bump_generation(self):
        self.generation += 1

    # -- weight-gradient LEAVES of the adapters' big-M linears (mt_gemm_tn_f16: atomics / L2 bound, 0.8 ms per step): nothing reads
    # their result before the optimiser, so they may run on a second stream beside the MFMA-bound kernels that follow.  Off by
    # default (MT_LEAF_STREAM=1 / engine.leaf_stream = True): same-box A/B in hipGraph replay 41.73 / 41.99 ms without, 42.21 / 42.30 with
    #  -- the cross-stream edges of the replayed graph cost more than the overlap returns.
    def _leaf(self, fn, *keep):
        if not self.leaf_stream:
            fn()
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))      # the operands were produced on the main stream
        with torch.cuda.stream(self._side):
            fn()
        self._side_refs.extend(keep)          # the operands stay allocated until the join (the allocator knows nothing of the side stream)
        self._side_busy = True

    def _leaf_join(self):
        if self._side_busy:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._side_refs.clear()
            self._side_busy = False

    def set_stochastic(self, on: bool, seed: int = 0):
        """Dropout(cfg.dropout) on the embedded input and after out_proj / fc2, per-layer DropPath on both backbone
        branches, DropPath on the Extractor FFN branch -- as model.train() leaves them in the reference."""
        if bool(on) != self.stochastic:
            self.generation += 1          # the input-dropout buffer comes / goes with the toggle
        self.stochastic = bool(on)
        self.rng.copy_(torch.tensor([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF, 0, 0], dtype=torch.int32))

    def _drop(self, site: int, p: float, path_site: int = 0, path_p: float = 0.0, rows_per_pass: int = 1):
        if not self._drop_now or (p <= 0.0 and path_p <= 0.0):
            return None
        off = self._site_base          # distinct masks for every forward call that owns its tape (module-style loops)
        return ops.dropout_spec(self.rng, site + off, p, path_site + off, path_p, rows_per_pass)

    def _layer_drops(self, l: int, N: int):
        """(attention branch, FFN branch) masks of backbone layer l (ENC:149-152, FFN:142 + ENC:169-170)."""
        if l < 0:
            return None, None
        p, pp = float(self.cfg.dropout), self._layer_path_p[l]
        return (self._drop(16 + 4 * l, p, 17 + 4 * l, pp, N), self._drop(18 + 4 * l, p, 19 + 4 * l, pp, N))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state, strict=True):
        self.store.load(state, strict)
        self._caches_ready = False
        self.generation += 1              # captured graphs read the fp16 caches that are about to be rebuilt

    def _build_caches(self):
        t, dev, cfg = self.store.tensors, self.device, self.cfg
        self._frozen16 = {"patch": _W16([t["patch_embed.proj.weight"]], dev, need_t=False)}
        for l in range(cfg.depth):
            p = f"encoder.layers.{l}."
            # the attention kernels take q pre-multiplied by 48^-1/2 log2(e): baked into the frozen q rows / bias here, in
            # fp32, so q is still rounded to fp16 exactly once (by the QKV GEMM's epilogue)
            self._frozen16[p + "qkv"] = _W16([_scaled_copy(t[p + "self_attn.q_proj.weight"], QK_SCALE_LOG2),
                                              t[p + "self_attn.k_proj.weight"], t[p + "self_attn.v_proj.weight"]], dev)
            self._frozen16[p + "out"] = _W16([t[p + "self_attn.out_proj.weight"]], dev)
            self._frozen16[p + "fc1"] = _W16([t[p + "ffn.fc1.weight"]], dev)
            self._frozen16[p + "fc2"] = _W16([t[p + "ffn.fc2.weight"]], dev)
            self._frozen16[p + "bqkv"] = torch.cat([_scaled_copy(t[p + "self_attn.q_proj.bias"], QK_SCALE_LOG2),
                                                    t[p + "self_attn.k_proj.bias"], t[p + "self_attn.v_proj.bias"]]).contiguous()
        # pointer tables of the composite layer launchers (mt_longnet_layer_fwd / _bwd: one C call per layer and direction)
        self._layer_w = []
        for l in range(cfg.depth):
            p = f"encoder.layers.{l}."
            f16 = self._frozen16
            self._layer_w.append(ops.struct_of(
                ops.MtLongNetLayerWeights, ln1_w=t[p + "self_attn_layer_norm.weight"], ln1_b=t[p + "self_attn_layer_norm.bias"],
                inner_ln_w=t[p + "self_attn.inner_attn_ln.weight"], inner_ln_b=t[p + "self_attn.inner_attn_ln.bias"],
                ln2_w=t[p + "final_layer_norm.weight"], ln2_b=t[p + "final_layer_norm.bias"], ffn_ln_w=t[p + "ffn.ffn_layernorm.weight"],
                ffn_ln_b=t[p + "ffn.ffn_layernorm.bias"], b_qkv=f16[p + "bqkv"], b_out=t[p + "self_attn.out_proj.bias"],
                b_fc1=t[p + "ffn.fc1.bias"], b_fc2=t[p + "ffn.fc2.bias"], w_qkv=f16[p + "qkv"].w, w_out=f16[p + "out"].w, w_fc1=f16[p + "fc1"].w,
                w_fc2=f16[p + "fc2"].w, wt_qkv=f16[p + "qkv"].wt, wt_out=f16[p + "out"].wt, wt_fc1=f16[p + "fc1"].wt, wt_fc2=f16[p + "fc2"].wt))
        self._train16 = {}
        self._pack_table = None
        for pref in self._cross_attn_prefixes():
            self._train16[pref + "q_proj"] = _W16([t[pref + "q_proj.weight"]], dev)
            self._train16[pref + "q_in"] = _W16([t[pref + "multihead_attn.q_proj_weight"]], dev)
            self._train16[pref + "kv"] = _W16([t[pref + "multihead_attn.k_proj_weight"], t[pref + "multihead_attn.v_proj_weight"]], dev)
            self._train16[pref + "out_in"] = _W16([t[pref + "multihead_attn.out_proj.weight"]], dev)
            self._train16[pref + "output_proj"] = _W16([t[pref + "output_proj.weight"]], dev)
        self._caches_ready = True
        self.generation += 1              # new fp16 cache tensors: graphs captured against the old ones are stale

    def refresh_trainable_caches(self):
        """Re-derive the fp16 copies of the trainable big-M weights after an optimiser step."""
        if not self._caches_ready:
            self._build_caches()
            return
        # one grouped launch for all of them (records: include/modaltune_hip.h, mt_pack_weights_f16)
        if getattr(self, "_pack_table", None) is None:
            recs = []
            for w in self._train16.values():
                r = 0
                for src in w.srcs:
                    assert src.is_contiguous() and src.shape[1] == w.K
                    recs.append([src.data_ptr(), w.w.data_ptr(), 0 if w.wt is None else w.wt.data_ptr(), src.shape[0], w.K, r,
                                 w.K, w.N])
                    r += src.shape[0]
            self._pack_table = torch.tensor(recs, dtype=torch.int64, device=self.device)
        ops.pack_weights(self._pack_table, self._pack_table.shape[0])

    def _cross_attn_prefixes(self) -> List[str]:
        out = []
        nint = len(self.cfg.interaction_indexes)
        for i in range(nint):
            out.append(f"interactions.{i}.injector.attn.")
            out.append(f"interactions.{i}.extractor.attn.")
            if i == nint - 1 and self.cfg.use_extra_extractor:
                out += [f"interactions.{i}.extra_extractors.{j}.attn." for j in range(2)]
        return out

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B: int, L: int, fresh: bool = False, slot: int = 0) -> Dict[str, torch.Tensor]:
        """Activation / gradient buffers of one (B, L) geometry.  Bags are ragged (one slide per step, any L): the
        storage is ONE set of flat buffers sized for the largest bag seen so far (grown by >= 25 % when a bigger one
        arrives) and each geometry gets views of their heads -- no allocator traffic per step, no growth with the number
        of distinct lengths.  fresh=True (module API: several forwards alive at once) allocates privately.
        slot: concurrent users of the SAME B (TrainStep's pass groups when both hold one pass) get storage of their own."""
        key = (B, L, slot)
        if not fresh and key in self._ws:
            return self._ws[key]
        dev = self.device

        def spec(Lx):       # name -> (dtype, shape) at bag length Lx
            return self._ws_spec(B, Lx)

        def numel(shape):
            n = 1
            for d in shape:
                n *= d
            return n

        if fresh:
            # A private workspace, RECYCLED: exact-size allocations per call (every slide has its own length) fragment the caching
            # allocator -- the reserved memory crept up by ~1 GiB per 100 slides under the reference loop (tools/soak_ragged.py).
            # A call leases one flat store sized for the largest bag seen (grown by >= 25 %); the lease rides on the call's tape,
            # on the slide's `share` dict and on the workspace itself, and hands the store back when the last of them is gone.
            want = spec(L)
            self._fresh_tick += 1
            for pl in self._fresh_pool.values():          # stores nobody asked for in the last 64 calls go back to the allocator
                pl[:] = [st for st in pl if self._fresh_tick - st.get("tick", 0) <= 64]      # (e.g. the B = 1 stores of the step that
            pool = self._fresh_pool.setdefault(B, [])                                          # taught the speculative batching its rows)
            fit = [st for st in pool if st["cap"] >= L and all(k in st["flat"] for k in want)]
            if fit:
                store = min(fit, key=lambda st: st["cap"])
                pool.remove(store)
            else:
                small = [st for st in pool if all(k in st["flat"] for k in want)]
                cap = max([L] + [st["cap"] + st["cap"] // 4 for st in small])
                if small:                       # replace the largest of the too-small stores instead of piling up
                    pool.remove(max(small, key=lambda st: st["cap"]))
                store = {"cap": cap, "flat": {k: torch.empty(numel(shape), dtype=dt, device=dev) for k, (dt, shape) in spec(cap).items()}}
            store["tick"] = self._fresh_tick
            w = {k: store["flat"][k][:numel(shape)].view(shape) for k, (dt, shape) in want.items()}
            w["_lease"] = _Lease(pool, store)
            return w
        store = self._ws_store.get((B, slot))
        if store is None or L > store["cap"] or (self.stochastic and "x0d" not in store["flat"]):
            cap = L if store is None else max(L, store["cap"] + store["cap"] // 4)
            self._ws_store.pop((B, slot), None)
            self._ws.clear()              # views of the old storage die with it
            self.generation += 1          # ... and so do the graphs captured on them
            store = {"cap": cap, "flat": {k: torch.empty(numel(shape), dtype=dt, device=dev) for k, (dt, shape) in spec(cap).items()}}
            self._ws_store[(B, slot)] = store
        if len(self._ws) > 64:
            self._ws.clear()
        w = {k: store["flat"][k][:numel(shape)].view(shape) for k, (dt, shape) in spec(L).items()}
        self._ws[key] = w
        return w

    def _ws_spec(self, B: int, Lx: int) -> Dict[str, tuple]:
        """name -> (dtype, shape) of every activation / gradient buffer at bag length Lx (overridden by the TITAN engine, whose
        frozen blocks save different tensors)."""
        cfg = self.cfg
        D, Fd = cfg.embed_dim, cfg.ffn_dim
        nb = len(self.seg_lengths)
        N = Lx + 1
        M, Mp = B * N, B * Lx
        sp = {"x16": (H16, (Lx, cfg.in_chans)), "x0": (F32, (Lx, D)), "prow": (torch.int32, (Lx,)), "pcol": (torch.int32, (Lx,))}
        if self.stochastic:
            sp["x0d"] = (F32, (B * Lx, D))      # per-pass input dropout (ENC:339) of the shared patch embedding
        for l in range(cfg.depth):
            sp[f"hin{l}"] = (F32, (M, D)); sp[f"hmid{l}"] = (F32, (M, D)); sp[f"qkv{l}"] = (H16, (M, 3 * D))
            sp[f"obr{l}"] = (H16, (nb, M, D)); sp[f"lsebr{l}"] = (F32, (nb, M, 16)); sp[f"lsetot{l}"] = (F32, (M, 16))
            sp[f"a1_{l}"] = (H16, (M, Fd))
            for st in ("st1", "stin", "st2", "stf"):
                sp[f"{st}_{l}"] = (F32, (M, 2))
        for i in range(len(cfg.interaction_indexes)):
            sp[f"hout{i}"] = (F32, (M, D))
        if cfg.first_interaction_layer > 0:
            sp["hpre"] = (F32, (M, D))          # [cls; patches] after the plain layers in front of the first interaction (LVA:269-281)
        # transients shared by all layers
        for nm in ("u16", "br16", "dy16", "dh16", "dmixed"):
            sp[nm] = (H16, (M, D))
        for nm in ("t16", "dt16", "da1"):
            sp[nm] = (H16, (M, Fd))
        sp["dh"] = (F32, (M, D)); sp["delta"] = (F32, (nb, M, 16)); sp["dqkv16"] = (H16, (M, 3 * D))
        sp["scratch32"] = (F32, (Mp, D))
        plan = ops.make_plan(branch_table(N, self.seg_lengths, DILATED_RATIOS), N, B)
        sp["attn_ws"] = (H16, (ops.dilated_attn_bwd_workspace_bytes(plan) // 2,))
        return sp

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, coords, genes: Sequence[torch.Tensor], task_onehots: torch.Tensor,
                need_grad: bool = True, fresh: bool = False, staged: bool = False, geometry=None,
                clinical: Optional[torch.Tensor] = None, share: Optional[dict] = None, tape: Optional[Tape] = None,
                site_group: int = 0, ws_slot: int = 0) -> torch.Tensor:
        """x [L, in_chans] (or [1,L,in]); coords [L,2] (host or device); genes: list of [1, n_i]; task_onehots [B, num_tasks].
        Returns logits [B, output_dim] (fp32, device).  fresh=True gives this call its own tape and workspace so that
        several forwards can precede one backward (the reference calls the model 3x before loss.backward(), TM:175-177);
        `self.last_call` is the handle `backward(..., call=)` takes.  staged=True: x / coords were already uploaded with
        stage_inputs() into this geometry's workspace (hipGraph replay path: no host work inside the step).
        share (fresh calls of the module API): a dict owned by the caller for the calls of ONE slide -- the first call leaves the
        task-independent patch embedding in it (`x0`: input cast + patch-embed GEMM + positional term, LVA:232-242), the later
        ones take it from there instead of recomputing it (the reference recomputes it for every task id)."""
        cfg, dev, t = self.cfg, self.device, self.store.tensors
        if self.store.sync is not None:
            self.store.sync()         # parameters complete on this rank before anything reads them (no-op when nothing is pending)
        if not self._caches_ready:
            self._build_caches()
        if staged:
            B, L = geometry
        else:
            x = x.reshape(-1, x.shape[-1])
            L = x.shape[0]
            B = task_onehots.shape[0]
        if L < 1 or B < 1:
            raise ValueError(f"empty bag: {L} patches, {B} task passes (a slide needs at least one patch embedding; the reference's "
                             "datasets never produce an empty one, data_utils/datasets.py:213-285)")
        N, D, Fd, E, T = L + 1, cfg.embed_dim, cfg.ffn_dim, cfg.adapter_dim, self.T
        M, Mp = B * N, B * L
        ws = self._workspace(B, L, fresh=fresh, slot=ws_slot)
        # fresh: this call owns its tape (several forwards alive at once); the engine's long-lived tape -- whose gradient
        # arena captured graphs point into -- is put back before returning
        # (tape=: a long-lived tape of the caller's -- TrainStep runs the task passes of a step as two concurrent groups, each with
        # its own tape, workspace geometry (B differs) and, through site_group, its own dropout masks)
        self.tape = tape if tape is not None else (Tape(self.device, shared=self._fresh_arenas) if fresh else self._main_tape)
        tape = self.tape
        if fresh:
            tape.lease = ws["_lease"]             # the backward closures read views of the leased store
            if share is not None:
                share.setdefault("_leases", []).append(ws["_lease"])      # (x0 of the first call serves the later calls of the slide)
        tape.grad_enabled = need_grad
        tape.reset()
        self._drop_now = bool(self.stochastic and need_grad)
        self._ext_calls = 0
        if fresh and self._drop_now:     # several forwards may precede one backward (TM:175-177): the masks of a call are
            self._fresh_calls += 1       # tied to the call, not to the device step counter alone
        self._site_base = 4096 * (self._fresh_calls & 0xFFFFF) if fresh else 1024 * int(site_group)
        self._ctx = dict(B=B, L=L, N=N, M=M, Mp=Mp, ws=ws)
        self._ctx["plan"] = self._attention_plan(N, B)
        P = self.store.param
        patch_map = rowmap(L, N, 1)     # patch rows of a [B, N, D] buffer
        self._ctx["patch_map"] = patch_map

        # ---- patch embedding + positional table + cls (LVA:232-242); shared by the B passes
        if share is not None and "x0" in share and tuple(share["x0"].shape) == (L, D):
            ws["x0"] = share["x0"]            # read-only from here on (block 0's input carries no gradient)
        else:
            self._embed_patches(x, coords, ws, staged, L)
            if share is not None:
                share["x0"] = ws["x0"]

        # ---- token side: gene encoder (one pass when dropout is off, own masks per task pass otherwise) + task token per pass
        # (LVA:257-266)
        gene = self._gene_encoder(genes, passes=int(task_onehots.shape[0]))  # Var [1, G64, P, D]
        c = self._assemble_tokens(gene, task_onehots, clinical)            # Var [B, T, D]
        pe = P("gene_pe")

        # ---- interaction blocks (LVA:294-307, AM:484-523)
        nint = len(cfg.interaction_indexes)
        src, src_map = ws["x0"], rowmap(L, 0, 0)      # injector 0 reads the shared patch embedding (broadcast)
        d_in = self._drop(1, float(cfg.dropout), rows_per_pass=L)
        if d_in is not None:                          # Encoder.prepare_forward's Dropout: one mask per task pass
            if "x0d" not in ws:
                ws["x0d"] = torch.empty(B * L, D, dtype=F32, device=dev)
            ops.dropout_f32(ws["x0"], ws["x0d"], B * L, D, d_in, xmap=rowmap(L, 0, 0))
            src, src_map = ws["x0d"], rowmap(L, L, 0)
        a0 = cfg.first_interaction_layer
        if a0 > 0:
            # interaction_indexes[0][0] != 0 (LVA:269-281): the layers below the first interaction run on [cls; embedded slide] as plain
            # backbone layers.  Nothing trainable sits in front of them, so they run forward-only (no tape entries); with Dropout /
            # DropPath on, every task pass draws its own masks there too.
            hin0 = ws["hin0"]
            ops.copy_rows(self._cls_source().view(1, D), hin0, B, D, smap=rowmap(1, 0, 0), dmap=rowmap(1, N, 0))
            ops.copy_rows(src, hin0, B * L, D, smap=src_map, dmap=patch_map)
            grad_was, tape.grad_enabled = tape.grad_enabled, False
            pend = None
            for l in range(a0):
                pend = self._layer(l, ws[f"hin{l + 1}"] if l < a0 - 1 else ws["hpre"], pend, defer=(l < a0 - 1))
            tape.grad_enabled = grad_was
            src, src_map = ws["hpre"], patch_map
        for i, (la, lb) in enumerate(cfg.interaction_indexes):
            # Everything recorded from here on belongs to interaction block i (and the blocks above): when the backward
            # reaches this marker, the gradients of interactions.{i}.* / prompt_selfattention.{i}.* (and of the head, for the
            # last block) are final -- the data-parallel reducer starts their all-reduce while the blocks below still run.
            if need_grad and (self.grad_ready_hook is not None or self.record_markers):
                tape.record_marker(i, lambda i=i: self.grad_ready_hook is not None and (self._leaf_join(), self.grad_ready_hook(i)))
            if i > 0 and cfg.use_prompt_sa:
                c = self._prompt_self_attention(c, pe, f"prompt_selfattention.{i}.")
            hin = ws[f"hin{la}"]
            # cls row of every pass: cls_token (+ pos_embed[0] = 0) for block 0, else carried from the previous block
            if i == 0 and a0 > 0:
                ops.copy_rows(ws["hpre"], hin, B, D, smap=rowmap(1, N, 0), dmap=rowmap(1, N, 0))
            elif i == 0:
                ops.copy_rows(self._cls_source().view(1, D), hin, B, D, smap=rowmap(1, 0, 0), dmap=rowmap(1, N, 0))
            else:
                ops.copy_rows(ws[f"hout{i - 1}"], hin, B, D, smap=rowmap(1, N, 0), dmap=rowmap(1, N, 0))
            self._injector(i, c, pe, src, src_map, hin, first=(i == 0))
            pend = None
            for l in range(la, lb + 1):
                out = ws[f"hin{l + 1}"] if l < lb else ws[f"hout{i}"]
                pend = self._layer(l, out, pend, defer=(l < lb))
            c = self._extractor(f"interactions.{i}.extractor.", c, pe, ws[f"hout{i}"])
            if i == nint - 1 and cfg.use_extra_extractor:
                for j in range(2):
                    c = self._extractor(f"interactions.{i}.extra_extractors.{j}.", c, pe, ws[f"hout{i}"])
            src, src_map = ws[f"hout{i}"], patch_map
            if self.collect_taps:
                self.taps[f"cls{i}"] = ws[f"hout{i}"].view(B, N, D)[:, 0].clone()
                self.taps[f"c{i}"] = c.data.clone()
                self.taps[f"x{i}_head"] = ws[f"hout{i}"].view(B, N, D)[:, 1:9].clone()
        # ---- fusion head (LVA:309-347)
        logits = self._head(c, ws[f"hout{nint - 1}"])
        self._logits = logits
        self.last_call = (tape, logits)
        self.tape = self._main_tape
        return logits.data

    def prepare_shared(self, x, coords, share: dict):
        """The task-independent part of a slide's forward (input cast, grid indices, patch-embed GEMM + positional term) into buffers
        of its own, left in `share` for the calls that follow: callers that run those calls on SEVERAL streams (the module bridge's
        pass groups) do this once on the stream they fork from."""
        cfg, dev = self.cfg, self.device
        x = x.reshape(-1, x.shape[-1])
        L = x.shape[0]
        ws = {"x16": torch.empty(L, cfg.in_chans, dtype=H16, device=dev), "x0": torch.empty(L, cfg.embed_dim, dtype=F32, device=dev),
              "prow": torch.empty(L, dtype=torch.int32, device=dev), "pcol": torch.empty(L, dtype=torch.int32, device=dev)}
        if not self._caches_ready:
            self._build_caches()
        self.stage_inputs(x, coords, ws)
        Engine._embed_patches(self, None, None, ws, True, L)
        share["x0"], share["_x0_keep"] = ws["x0"], ws

    # -- overridable pieces of the image side (modaltune_amd/titan.py plugs the TITAN backbone in here)
    def _attention_plan(self, N: int, B: int):
        return ops.make_plan(branch_table(N, self.seg_lengths, DILATED_RATIOS), N, B)

    def _embed_patches(self, x, coords, ws, staged: bool, L: int):
        cfg, t = self.cfg, self.store.tensors
        if not staged:
            self.stage_inputs(x, coords, ws)
        ops.gemm_nt(ws["x16"], self._frozen16["patch"].w, ws["x0"], L, cfg.embed_dim, cfg.in_chans, epilogue=ops.EPI_POSEMB,
                    bias=t["patch_embed.proj.bias"], pos_table=self.pos_table, pos_row=ws["prow"], pos_col=ws["pcol"])

    def _cls_source(self) -> torch.Tensor:
        return self.store.tensors["cls_token"]      # (+ pos_embed[0] = zeros)

    def _image_token(self, hout: torch.Tensor) -> Var:
        """The image-side feature of the fusion head: the cls row of every pass (LVA:309-312).  Records the closure that
        starts the patch-side backward (zeroes dh, scatters the cls gradient)."""
        ctx, tape, D = self._ctx, self.tape, self.cfg.embed_dim
        B, N, ws = ctx["B"], ctx["N"], ctx["ws"]
        cls = Var(tape.new(B, D))
        if self.cfg.global_pool:             # img_outcome = x.mean(dim=1) over the PATCH rows (LVA:309-310; x excludes cls there)
            L = N - 1
            w = torch.full((L,), 1.0 / L, dtype=F32, device=self.device)
            ops.sgemm(w, (0, 1), hout.view(-1)[D:], (1, D), cls.data, (D, 1), 1, D, L, batch=B, b_bs=N * D, c_bs=D)
        else:
            ops.copy_rows(hout, cls.data, B, D, smap=rowmap(1, N, 0))

        def bwd():
            dh = ws["dh"]
            dh.zero_()                       # start of the patch-side backward: only the cls rows (or, pooled, every patch row) carry gradient
            ctx["dh16_valid"] = False
            if cls.grad is None:
                return
            if self.cfg.global_pool:         # d x[b, l, :] = d img[b, :] / L
                L = N - 1
                w = torch.full((L,), 1.0 / L, dtype=F32, device=self.device)
                ops.sgemm(w, (1, 0), cls.grad, (1, 0), dh.view(-1)[D:], (D, 1), L, D, 1, accumulate=True, batch=B, b_bs=D, c_bs=N * D)
            else:
                ops.copy_rows(cls.grad, dh, B, D, dmap=rowmap(1, N, 0))
        tape.record(bwd)
        return cls

    def stage_inputs(self, x: torch.Tensor, coords, ws: Optional[Dict[str, torch.Tensor]] = None, B: Optional[int] = None):
        """Input boundary: grid indices from coords (slide_encoder.py:198-211) and the fp16 copy of the patch embeddings,
        written into the workspace's static buffers.  Nothing here synchronises the stream: device coords are binned by a
        kernel (out-of-range cells raise at the next check_inputs()), host coords are checked on the host and travel
        through a small ring of pinned slots."""
        cfg = self.cfg
        x = x.reshape(-1, x.shape[-1])
        L = x.shape[0]
        if L < 1:
            raise ValueError("empty bag: 0 patches (a slide needs at least one patch embedding)")
        if ws is None:
            ws = self._workspace(B, L)
        if torch.is_tensor(coords) and coords.is_cuda:
            c = coords.reshape(-1, 2).to(F32).contiguous()
            if c.shape[0] != L:
                raise ValueError(f"coords has {c.shape[0]} rows for {L} patches")
            self.poll_inputs()                # raises for a bad slide of an EARLIER call once its flag has landed (no sync)
            ops.coords_to_grid(c, L, float(cfg.tile_size), cfg.slide_ngrids, ws["prow"], ws["pcol"], self._coord_err)
            if not torch.cuda.is_current_stream_capturing():
                if self._coord_err_host is None:
                    self._coord_err_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
                    self._coord_err_event = torch.cuda.Event()
                self._coord_err_host.copy_(self._coord_err, non_blocking=True)
                self._coord_err_event.record()
        else:
            coords_np = coords.detach().numpy() if torch.is_tensor(coords) else np.asarray(coords)
            prow, pcol = coords_to_rowcol(coords_np.reshape(-1, 2), float(cfg.tile_size))
            if prow.shape[0] != L:
                raise ValueError(f"coords has {prow.shape[0]} rows for {L} patches")
            if int(prow.max()) >= cfg.slide_ngrids or int(pcol.max()) >= cfg.slide_ngrids or int(min(prow.min(), pcol.min())) < 0:
                raise ValueError("coords outside the slide_ngrids x slide_ngrids positional grid")
            slot = self._pinned_slot(2 * L)
            slot[:L].copy_(torch.from_numpy(prow.astype(np.int32)))
            slot[L:2 * L].copy_(torch.from_numpy(pcol.astype(np.int32)))
            ws["prow"].copy_(slot[:L], non_blocking=True)
            ws["pcol"].copy_(slot[L:2 * L], non_blocking=True)
            self._pin_ring[self._pin_last][1].record()
        if x.dtype == H16 and x.is_cuda:
            ws["x16"].copy_(x)
        else:
            ops.cast_f32_to_f16(x.to(self.device, F32).contiguous(), ws["x16"])

    def _pinned_slot(self, n: int) -> torch.Tensor:
        """Next slot of a 4-deep ring of pinned int32 buffers; a slot is reused only after the copy that read it has run."""
        if len(self._pin_ring) < 4:
            self._pin_ring.append([torch.empty(max(n, 1 << 15), dtype=torch.int32, pin_memory=True), torch.cuda.Event()])
            self._pin_last = len(self._pin_ring) - 1
        else:
            self._pin_last = self._pin_next
            self._pin_next = (self._pin_next + 1) % 4
            buf, ev = self._pin_ring[self._pin_last]
            ev.synchronize()
            if buf.numel() < n:
                self._pin_ring[self._pin_last][0] = torch.empty(n + n // 4, dtype=torch.int32, pin_memory=True)
        return self._pin_ring[self._pin_last][0]

    def check_inputs(self):
        """Raises if a device-side coords binning since the last call met a cell outside the positional grid or a
        non-finite coordinate (host sync: call it where the host reads the loss / logits anyway)."""
        if int(self._coord_err) != 0:
            self._coord_err.zero_()
            if self._coord_err_host is not None:
                self._coord_err_host.zero_()
            raise ValueError("coords outside the slide_ngrids x slide_ngrids positional grid (or not finite)")

    def poll_inputs(self):
        """check_inputs() without the sync: looks at the pinned mirror of the error flag if its last asynchronous read-back
        has completed.  Called at the top of every device-side binning, so a slide with bad / NaN coords (which trains
        with clamped positions, SE:198-211 would index out of bounds) raises at the latest one call later."""
        ev = self._coord_err_event
        if ev is not None and ev.query() and int(self._coord_err_host[0]) != 0:
            self._coord_err.zero_()
            self._coord_err_host.zero_()
            raise ValueError("coords outside the slide_ngrids x slide_ngrids positional grid (or not finite) in an earlier slide")

    # ------------------------------------------------------------------ token-side pieces
    def _gene_encoder(self, genes: Sequence[torch.Tensor], passes: int = 1) -> Var:
        """GeneEncoder_Group.gene_encode (gene_encoder.py:194-215) of one slide for `passes` task passes: returns the gene tokens
        [1, G64, P, D] with P = passes in train mode with dropout (own masks per pass), else 1 (shared)."""
        tape, P, g = self.tape, self.store.param, self.cfg.gene
        G = len(self.group_sizes)
        if not torch.is_tensor(genes) and len(genes) != G:
            raise ValueError(f"expected {G} gene groups, got {len(genes)}")
        # all pathway networks in one launch per direction (the reference loops over 2 G nn.Linear modules)
        if torch.is_tensor(genes):
            gflat = genes.to(self.device, F32).reshape(-1)
        else:
            gflat = torch.cat([gi.reshape(-1) for gi in genes]).to(self.device, F32)
        if gflat.numel() != self._gene_total:
            raise ValueError(f"expected {self._gene_total} gene values in {G} groups, got {gflat.numel()}")
        # train mode: AlphaDropout after each ELU (GE:178-181) and Dropout in the mixer.  The reference calls the model once per
        # task, so every task pass draws its own masks: with dropout on the encoder carries a pass axis [1, G, P, C] behind the
        # group axis (weights streamed once for all passes; the counter-based masks differ because the element offsets do);
        # with dropout off the passes are identical and ONE is computed and broadcast.
        gp = float(g.dropout)
        adrop = self._drop(300, gp)
        Pn = passes if adrop is not None else 1
        z = Var(tape.new(1, G, Pn, g.latent_dim))
        a1, a2 = tape.new(G, g.latent_dim), tape.new(G, Pn, g.latent_dim)
        st = self.store
        ops.gene_snn_fwd(st.flat, self._gene_offs, self._gene_sizes, self._gene_goff, gflat, G, g.latent_dim, a1, a2, z.data,
                         alpha_drop=adrop, passes=Pn)
        z0 = z          # (closures bind late: `z` is rebound by the mixer loop below)

        fgrad = st.flat_grad      # (bound NOW: the store's gradient side may point elsewhere by the time the backward runs)

        def bwd_networks():
            if z0.grad is None:
                return
            ops.gene_snn_bwd(st.flat, fgrad, self._gene_offs, self._gene_sizes, self._gene_goff, gflat, G, g.latent_dim,
                             a1, a2, z0.grad, alpha_drop=adrop, passes=Pn)
        tape.record(bwd_networks)
        for k in range(g.depth):
            p = f"gene_encoder.mlp_mixer.{k}."
            n1 = tape.layernorm(z, P(p + "0.norm.weight"), P(p + "0.norm.bias"))
            # FeedForward = dense, GELU, Dropout, dense, Dropout (GE:184-192)
            # (activation, Dropout and the residual add ride on each product's epilogue: 4 launches per mixer block)
            m1 = tape.axis_linear(n1, P(p + "0.fn.0.weight"), P(p + "0.fn.0.bias"), act=ops.ACT_GELU, drop=self._drop(310 + 4 * k, gp))
            z = tape.axis_linear(m1, P(p + "0.fn.3.weight"), P(p + "0.fn.3.bias"), drop=self._drop(311 + 4 * k, gp), resid=z)
            n2 = tape.layernorm(z, P(p + "1.norm.weight"), P(p + "1.norm.bias"))
            f1 = tape.linear(n2, P(p + "1.fn.0.weight"), P(p + "1.fn.0.bias"), act=ops.ACT_GELU, drop=self._drop(312 + 4 * k, gp))
            z = tape.linear(f1, P(p + "1.fn.3.weight"), P(p + "1.fn.3.bias"), drop=self._drop(313 + 4 * k, gp), resid=z)
        p = "gene_encoder.mlp_mixer."
        z = tape.layernorm(z, P(p + f"{g.depth}.weight"), P(p + f"{g.depth}.bias"))
        z = tape.linear(z, P(p + f"{g.depth + 1}.weight"), P(p + f"{g.depth + 1}.bias"))          # [1, G, P, D]
        return tape.axis_linear(z, P("gene_encoder.pathway_compression.weight"), P("gene_encoder.pathway_compression.bias"))

    def _assemble_tokens(self, gene: Var, onehots: torch.Tensor, clinical: Optional[torch.Tensor] = None) -> Var:
        """c[b] = cat([clinical_mlp(clinical)], task_weight(onehot_b), gene_embedding) (LVA:263-266; 572-580 for the
        clinical variant: the clinical token goes first)."""
        tape, P, D, T = self.tape, self.store.param, self.cfg.embed_dim, self.T
        B = onehots.shape[0]
        G64 = gene.data.shape[1]
        c = Var(tape.new(B, T, D))
        nt, ncl, ngc = int(self.cfg.is_multi), int(self.cfg.clinical), int(self.cfg.has_gene_cls)
        task = clin = None
        gcls = P("gene_cls") if ngc else None      # prompt_agg == "cls": a learned token in front of the gene tokens (LVA:259-261)
        if ngc:
            ops.copy_rows(gcls.data.view(1, D), c.data, B, D, smap=rowmap(1, 0, 0), dmap=rowmap(1, T, ncl + nt))
        if ncl:
            if clinical is None:
                raise ValueError("this model variant needs `clinical` features [1, clinfeat_dim]")
            cv = Var(clinical.to(self.device, F32).reshape(1, -1).contiguous(), needs_grad=False)
            h1 = tape.linear(cv, P("clinical_mlp.0.weight"), P("clinical_mlp.0.bias"), act=ops.ACT_RELU)
            clin = tape.layernorm(tape.linear(h1, P("clinical_mlp.2.weight"), P("clinical_mlp.2.bias")),
                                  P("clinical_mlp.3.weight"), P("clinical_mlp.3.bias"))                  # [1, D], shared by the passes
            ops.copy_rows(clin.data, c.data, B, D, smap=rowmap(1, 0, 0), dmap=rowmap(1, T, 0))
        if nt:
            oh = Var(onehots.to(self.device, F32).contiguous(), needs_grad=False)
            task = tape.layernorm(tape.linear(oh, P("task_weight.0.weight"), P("task_weight.0.bias")),
                                  P("task_weight.1.weight"), P("task_weight.1.bias"))                  # [B, D]
            ops.copy_rows(task.data, c.data, B, D, dmap=rowmap(1, T, ncl))
        Pg = gene.data.shape[2]              # 1 (one encoder pass shared by the task passes) or B (own dropout masks per pass)
        assert Pg in (1, B)
        gsrc = gene.data.view(G64 * Pg, D)
        for b in range(B):                   # rows (g, b) of the pathway-major gene tokens -> pass b's gene slots
            ops.copy_rows(gsrc, c.data[b], G64, D, smap=rowmap(1, Pg, b if Pg > 1 else 0), dmap=rowmap(G64, T, ncl + nt + ngc))

        def bwd():
            if c.grad is None:
                return
            if task is not None:
                ops.copy_rows(c.grad, task.g(), B, D, smap=rowmap(1, T, ncl), accumulate=True)
            gg = gene.g().view(G64 * Pg, D)
            for b in range(B):      # d gene tokens of pass b (summed over the passes when they share one encoder pass)
                ops.copy_rows(c.grad, gg, G64, D, smap=rowmap(G64, T, b * T + ncl + nt + ngc), dmap=rowmap(1, Pg, b if Pg > 1 else 0),
                              accumulate=True)
                if gcls is not None and gcls.grad is not None:
                    ops.copy_rows(c.grad, gcls.grad.view(1, D), 1, D, smap=rowmap(1, T, b * T + ncl + nt), accumulate=True)
                if clin is not None:
                    ops.copy_rows(c.grad, clin.g(), 1, D, smap=rowmap(1, T, b * T), accumulate=True)
        tape.record(bwd)
        return c

    def _mha_in(self, pref: str):
        """Views of nn.MultiheadAttention's packed parameters (adapter_modules.py:157-164)."""
        P, E = self.store.param, self.cfg.adapter_dim
        b = self.store.tensors[pref + "in_proj_bias"]
        gb = self.store.grads[pref + "in_proj_bias"]
        bq, bk, bv = (Param(b[i * E:(i + 1) * E], gb[i * E:(i + 1) * E]) for i in range(3))
        return P(pref + "q_proj_weight"), P(pref + "k_proj_weight"), P(pref + "v_proj_weight"), bq, bk, bv

    def _prompt_self_attention(self, c: Var, pe: Param, pref: str) -> Var:
        """SelfAttentionLayer.forward_pre (AM:81-94)."""
        tape, P = self.tape, self.store.param
        tn = tape.layernorm(c, P(pref + "norm.weight"), P(pref + "norm.bias"))
        kin = tape.add_rows_param(tn, pe)
        Wq, Wk, Wv, bq, bk, bv = self._mha_in(pref + "self_attn.")
        q1, k, v = tape.linear_group([(kin, P(pref + "q_proj.weight"), P(pref + "q_proj.bias")), (kin, Wk, bk), (tn, Wv, bv)])
        q = tape.linear(q1, Wq, bq)
        a = tape.token_mha(q, k, v, self.cfg.num_heads)
        o = tape.linear(a, P(pref + "self_attn.out_proj.weight"), P(pref + "self_attn.out_proj.bias"))
        return tape.linear(o, P(pref + "output_proj.weight"), P(pref + "output_proj.bias"), resid=c)

    # ------------------------------------------------------------------ injector (A.1)
    def _injector(self, i: int, c: Var, pe: Param, src: torch.Tensor, src_map, hin: torch.Tensor, first: bool):
        cfg, ctx, tape, P, t = self.cfg, self._ctx, self.tape, self.store.param, self.store.tensors
        B, L, N, Mp, D, E, T = ctx["B"], ctx["L"], ctx["N"], ctx["Mp"], cfg.embed_dim, cfg.adapter_dim, self.T
        pref = f"interactions.{i}.injector."
        ap = pref + "attn."
        pm = ctx["patch_map"]
        w16 = self._train16
        dev = self.device
        # token side: k, v from LN_kq(c) + pe (AM:218,227)
        chat = tape.layernorm(c, P(ap + "norm_kq.weight"), P(ap + "norm_kq.bias"), add_rows=pe)
        _, Wk, Wv, _, bk, bv = self._mha_in(ap + "multihead_attn.")
        k, v = tape.linear_group([(chat, Wk, bk), (chat, Wv, bv)])
        # patch side
        xhat = torch.empty(Mp, D, dtype=H16, device=dev)
        st = torch.empty(Mp, 2, dtype=F32, device=dev)
        ops.layernorm_fwd(src, t[ap + "norm.weight"], t[ap + "norm.bias"], xhat, st, Mp, D, xmap=src_map)
        q1 = torch.empty(Mp, E, dtype=H16, device=dev)
        ops.gemm_nt(xhat, w16[ap + "q_proj"].w, q1, Mp, E, D, bias=t[ap + "q_proj.bias"])
        q2 = torch.empty(Mp, E, dtype=H16, device=dev)
        bqi = t[ap + "multihead_attn.in_proj_bias"][:E]
        ops.gemm_nt(q1, w16[ap + "q_in"].w, q2, Mp, E, E, bias=bqi)
        a = torch.empty(Mp, E, dtype=H16, device=dev)
        alse = torch.empty(Mp, 12, dtype=F32, device=dev)
        ops.inject_attn_fwd(q2, k.data, v.data, a, Mp, L, T, lse=alse)
        o1 = torch.empty(Mp, E, dtype=H16, device=dev)
        ops.gemm_nt(a, w16[ap + "out_in"].w, o1, Mp, E, E, bias=t[ap + "multihead_attn.out_proj.bias"])
        ops.gemm_nt(o1, w16[ap + "output_proj"].w, hin, Mp, D, E, cmap=pm, epilogue=ops.EPI_INJECT, bias=t[ap + "output_proj.bias"],
                    resid=src, ldr=D, rmap=src_map, colscale=t[pref + "gamma"])
        g = self.store.grads
        ws = ctx["ws"]

        def bwd():
            dh = ws["dh"]
            proj = torch.empty(Mp, D, dtype=H16, device=dev)
            ops.gemm_nt(o1, w16[ap + "output_proj"].w, proj, Mp, D, E, bias=t[ap + "output_proj.bias"])
            dproj = torch.empty(Mp, D, dtype=H16, device=dev)
            # residual path: dh_patch <- (1+gamma) dh_patch (in place; block 0's input x0 needs no gradient)
            ops.inject_resid_bwd(dh, src, proj, t[pref + "gamma"], dh if not first else ws["scratch32"], dproj,
                                 g[pref + "gamma"], Mp, D, dymap=pm, xmap=src_map, dxmap=pm if not first else None)
            self._leaf(lambda: ops.gemm_tn(dproj, o1, g[ap + "output_proj.weight"], Mp, D, E, colsum=g[ap + "output_proj.bias"]), dproj, o1)
            do1 = torch.empty(Mp, E, dtype=H16, device=dev)
            ops.gemm_nt(dproj, w16[ap + "output_proj"].wt, do1, Mp, E, D)
            self._leaf(lambda: ops.gemm_tn(do1, a, g[ap + "multihead_attn.out_proj.weight"], Mp, E, E, colsum=g[ap + "multihead_attn.out_proj.bias"]),
                       do1, a)
            da = torch.empty(Mp, E, dtype=H16, device=dev)
            ops.gemm_nt(do1, w16[ap + "out_in"].wt, da, Mp, E, E)
            dq2 = torch.empty(Mp, E, dtype=H16, device=dev)
            ops.inject_attn_bwd(q2, a, alse, da, k.data, v.data, dq2, k.g(), v.g(), Mp, L, T)
            self._leaf(lambda: ops.gemm_tn(dq2, q1, g[ap + "multihead_attn.q_proj_weight"], Mp, E, E, colsum=g[ap + "multihead_attn.in_proj_bias"][:E]),
                       dq2, q1)
            dq1 = torch.empty(Mp, E, dtype=H16, device=dev)
            ops.gemm_nt(dq2, w16[ap + "q_in"].wt, dq1, Mp, E, E)
            self._leaf(lambda: ops.gemm_tn(dq1, xhat, g[ap + "q_proj.weight"], Mp, E, D, colsum=g[ap + "q_proj.bias"]), dq1, xhat)
            dxhat = torch.empty(Mp, D, dtype=H16, device=dev)
            ops.gemm_nt(dq1, w16[ap + "q_proj"].wt, dxhat, Mp, D, E)
            if first:
                ops.layernorm_bwd(dxhat, src, t[ap + "norm.weight"], st, ws["scratch32"], Mp, D, xmap=src_map,
                                  dw=g[ap + "norm.weight"], db=g[ap + "norm.bias"])
            else:
                ops.layernorm_bwd(dxhat, src, t[ap + "norm.weight"], st, dh, Mp, D, xmap=src_map, dxmap=pm, accumulate=True,
                                  dw=g[ap + "norm.weight"], db=g[ap + "norm.bias"])
        tape.record(bwd)

    # ------------------------------------------------------------------ one frozen LongNet layer (A.4)
    def _layer(self, l: int, out: torch.Tensor, pend=None, defer: bool = False):
        """One frozen LongNet layer (ENC:78-125).  The residual adds ride on the LayerNorm that consumes the sum
        (mt_add_layernorm_fwd): the out_proj / fc2 GEMMs write their fp16 branch with a plain epilogue.  `pend` is the
        (stream, branch, dropout) of the layer below whose fc2 add is still outstanding; with `defer` this layer leaves
        its own fc2 add to the layer above and returns such a triple instead of writing `out`."""
        cfg, ctx, t = self.cfg, self._ctx, self.store.tensors
        M, D, Fd, ws, plan = ctx["M"], cfg.embed_dim, cfg.ffn_dim, ctx["ws"], ctx["plan"]
        p = f"encoder.layers.{l}."
        f16 = self._frozen16
        hin, hmid, qkv, obr, lsebr, lsetot, a1 = (ws[f"hin{l}"], ws[f"hmid{l}"], ws[f"qkv{l}"], ws[f"obr{l}"], ws[f"lsebr{l}"],
                                                  ws[f"lsetot{l}"], ws[f"a1_{l}"])
        st1, stin, st2, stf = ws[f"st1_{l}"], ws[f"stin_{l}"], ws[f"st2_{l}"], ws[f"stf_{l}"]
        u16, t16 = ws["u16"], ws["t16"]
        N_tok = ctx["N"]
        d_attn, d_ffn = self._layer_drops(l, N_tok)
        br16 = ws["br16"]
        feeds_lower = all(l != a for a, _ in cfg.interaction_indexes)
        d_lower_ffn = self._layer_drops(l - 1, N_tok)[1] if feeds_lower else None
        if ops.TIMER is None:
            # one C call per direction (csrc/layer.hip enqueues exactly the launches spelled out below); the instrumented
            # pass of bench.py (ops.TIMER) keeps the launch-by-launch form so that every kernel gets its own HIP events
            cb = ws.get(("_cb", l))
            if cb is None:
                cb = ws[("_cb", l)] = ops.struct_of(
                    ops.MtLongNetLayerBuffers, hin=hin, hmid=hmid, qkv=qkv, o_br=obr, lse_br=lsebr, lse_tot=lsetot, a1=a1, st1=st1, stin=stin,
                    st2=st2, stf=stf, u16=u16, br16=br16, t16=t16, dh=ws["dh"], dy16=ws["dy16"], dh16=ws["dh16"], dt16=ws["dt16"],
                    da1=ws["da1"], dmixed=ws["dmixed"], dqkv16=ws["dqkv16"], delta=ws["delta"], attn_ws=ws["attn_ws"])
            lw = self._layer_w[l]
            ops.longnet_layer_fwd(lw, cb, plan, M, D, Fd, out, pend=pend, defer=defer, drop_attn=d_attn, drop_ffn=d_ffn)

            def bwd_c():
                ops.longnet_layer_bwd(lw, cb, plan, M, D, Fd, bool(ctx.get("dh16_valid")), feeds_lower, drop_attn=d_attn, drop_ffn=d_ffn,
                                      drop_lower_ffn=d_lower_ffn)
                ctx["dh16_valid"] = feeds_lower
            self.tape.record(bwd_c)
            return (hmid, br16, d_ffn) if defer else None
        if pend is None:
            ops.layernorm_fwd(hin, t[p + "self_attn_layer_norm.weight"], t[p + "self_attn_layer_norm.bias"], u16, st1, M, D)
        else:       # hin = hmid(l-1) + drop(fc2 branch of l-1)
            ops.add_layernorm_fwd(pend[0], pend[1], t[p + "self_attn_layer_norm.weight"], t[p + "self_attn_layer_norm.bias"],
                                  hin, u16, st1, M, D, drop=pend[2])
        ops.gemm_nt(u16, f16[p + "qkv"].w, qkv, M, 3 * D, D, bias=f16[p + "bqkv"], epilogue=ops.EPI_QKV_HM)   # head-major q|k|v
        ops.dilated_attn_fwd(qkv, plan, obr, lsebr)
        ops.dilated_mix_ln_fwd(obr, lsebr, plan, t[p + "self_attn.inner_attn_ln.weight"], t[p + "self_attn.inner_attn_ln.bias"],
                               u16, stin, lsetot)
        ops.gemm_nt(u16, f16[p + "out"].w, br16, M, D, D, bias=t[p + "self_attn.out_proj.bias"])
        ops.add_layernorm_fwd(hin, br16, t[p + "final_layer_norm.weight"], t[p + "final_layer_norm.bias"], hmid, u16, st2, M, D,
                              drop=d_attn)       # hmid = hin + drop(out_proj branch)
        ops.gemm_nt(u16, f16[p + "fc1"].w, a1, M, Fd, D, bias=t[p + "ffn.fc1.bias"])
        ops.layernorm_fwd(a1, t[p + "ffn.ffn_layernorm.weight"], t[p + "ffn.ffn_layernorm.bias"], t16, stf, M, Fd, gelu_in=True)
        if defer:
            ops.gemm_nt(t16, f16[p + "fc2"].w, br16, M, D, Fd, bias=t[p + "ffn.fc2.bias"])
            nxt = (hmid, br16, d_ffn)
        else:
            ops.gemm_nt(t16, f16[p + "fc2"].w, out, M, D, Fd, epilogue=ops.EPI_BIAS_RESID, bias=t[p + "ffn.fc2.bias"], resid=hmid, ldr=D,
                        drop=d_ffn)
            nxt = None

        # (feeds_lower: nothing else touches dh between two layers of one interaction block: the lower layer can take fp16(dh)
        # from here)
        def bwd():
            dh, dy16, dt16, da1 = ws["dh"], ws["dy16"], ws["dt16"], ws["da1"]
            # FFN: out = hmid + fc2(LN(gelu(fc1(LN(hmid)))))
            if ctx.get("dh16_valid"):             # the layer above left fp16(dh) behind (LN backward's second output)
                src16 = ws["dh16"]
            else:
                ops.cast_f32_to_f16(dh, dy16, drop=d_ffn, D=D)        # gradient of the (dropped) FFN branch output
                src16 = dy16
            ops.gemm_nt(src16, f16[p + "fc2"].wt, dt16, M, Fd, D)
            ops.layernorm_bwd(dt16, a1, t[p + "ffn.ffn_layernorm.weight"], stf, da1, M, Fd, gelu_in=True)
            ops.gemm_nt(da1, f16[p + "fc1"].wt, dy16, M, D, Fd)
            ops.layernorm_bwd(dy16, hmid, t[p + "final_layer_norm.weight"], st2, dh, M, D, accumulate=True, dx16=ws["dh16"],
                              dx16_drop=d_attn)
            # attention: hmid = hin + out_proj(LN(mix(dilated(qkv(LN(hin))))))
            ops.gemm_nt(ws["dh16"], f16[p + "out"].wt, u16, M, D, D)
            ops.dilated_mix_ln_bwd(u16, obr, lsebr, lsetot, plan, t[p + "self_attn.inner_attn_ln.weight"], stin, ws["dmixed"], ws["delta"])
            ops.dilated_attn_bwd(qkv, ws["dmixed"], lsetot, ws["delta"], plan, ws["attn_ws"], ws["dqkv16"])
            ops.gemm_nt(ws["dqkv16"], f16[p + "qkv"].wt, dy16, M, D, 3 * D)
            ops.layernorm_bwd(dy16, hin, t[p + "self_attn_layer_norm.weight"], st1, dh, M, D, accumulate=True,
                              dx16=ws["dh16"] if feeds_lower else None, dx16_drop=d_lower_ffn if feeds_lower else None)
            ctx["dh16_valid"] = feeds_lower
        self.tape.record(bwd)
        return nxt

    # ------------------------------------------------------------------ extractor (A.2)
    def _extractor(self, pref: str, c: Var, pe: Param, hout: torch.Tensor) -> Var:
        cfg, ctx, tape, P, t, g = self.cfg, self._ctx, self.tape, self.store.param, self.store.tensors, self.store.grads
        B, L, N, Mp, D, E, T = ctx["B"], ctx["L"], ctx["N"], ctx["Mp"], cfg.embed_dim, cfg.adapter_dim, self.T
        ap = pref + "attn."
        pm, dev, w16, ws = ctx["patch_map"], self.device, self._train16, ctx["ws"]
        # patch side: k | v of LN_kq(x)
        xk = torch.empty(Mp, D, dtype=H16, device=dev)
        st = torch.empty(Mp, 2, dtype=F32, device=dev)
        ops.layernorm_fwd(hout, t[ap + "norm_kq.weight"], t[ap + "norm_kq.bias"], xk, st, Mp, D, xmap=pm)
        kv = torch.empty(Mp, 2 * E, dtype=H16, device=dev)
        bkv = t[ap + "multihead_attn.in_proj_bias"][E:]
        ops.gemm_nt(xk, w16[ap + "kv"].w, kv, Mp, 2 * E, D, bias=bkv)
        # token side: q = in_proj_q(q_proj(LN(c) + pe))
        t2 = tape.layernorm(c, P(ap + "norm.weight"), P(ap + "norm.bias"), add_rows=pe)
        q1 = tape.linear(t2, P(ap + "q_proj.weight"), P(ap + "q_proj.bias"))
        Wq, _, _, bq, _, _ = self._mha_in(ap + "multihead_attn.")
        q2 = tape.linear(q1, Wq, bq)
        out = Var(tape.new(B, T, E))
        lse = tape.new(B, T, 12)
        kps = -(-(-(-L // max(1, min(64, L // 256))) ) // 64) * 64      # keys per split: multiple of the 64-key tile
        nsplit = -(-L // kps)                                           # every split owns >= 1 key
        pa = tape.new(B * 12 * nsplit * T * 16)
        pml = tape.new(B * 12 * nsplit * T * 2)
        ops.extract_attn_fwd(q2.data, kv, out.data, lse, pa, pml, B, T, L, nsplit)

        def bwd_core():
            if out.grad is None:
                return
            dkv = torch.empty(Mp, 2 * E, dtype=H16, device=dev)
            ops.extract_attn_bwd(q2.data, kv, out.data, lse, out.grad, q2.g(), dkv, B, T, L)
            self._leaf(lambda: ops.gemm_tn(dkv, xk, g[ap + "multihead_attn.k_proj_weight"], Mp, 2 * E, D,       # k | v weights are adjacent
                                           colsum=g[ap + "multihead_attn.in_proj_bias"][E:]), dkv, xk)
            dxk = torch.empty(Mp, D, dtype=H16, device=dev)
            ops.gemm_nt(dkv, w16[ap + "kv"].wt, dxk, Mp, D, 2 * E)
            ops.layernorm_bwd(dxk, hout, t[ap + "norm_kq.weight"], st, ws["dh"], Mp, D, xmap=pm, dxmap=pm, accumulate=True,
                              dw=g[ap + "norm_kq.weight"], db=g[ap + "norm_kq.bias"])
        tape.record(bwd_core)
        o = tape.linear(out, P(ap + "multihead_attn.out_proj.weight"), P(ap + "multihead_attn.out_proj.bias"))
        # c1 = query + (tgt + output_proj(.)) with tgt == query (AM:231,324): 2 c + output_proj(.) on the product's epilogue
        c1 = tape.linear(o, P(ap + "output_proj.weight"), P(ap + "output_proj.bias"), resid=c, resid_scale=2.0)
        fp = pref + "ffn."
        tn = tape.layernorm(c1, P(fp + "norm.weight"), P(fp + "norm.bias"))
        f1 = tape.linear(tn, P(fp + "linear1.weight"), P(fp + "linear1.bias"), act=ops.ACT_RELU)
        # query + drop_path(ffn(query)) (AM:319,327; one Bernoulli per task pass): DropPath and the add ride on linear2's epilogue,
        # the backward regenerates the factor as the dX / dW products load dy
        self._ext_calls = getattr(self, "_ext_calls", 0) + 1
        d_path = self._drop(0, 0.0, 200 + self._ext_calls, float(cfg.drop_path_rate), T)
        return tape.linear(f1, P(fp + "linear2.weight"), P(fp + "linear2.bias"), drop=d_path, resid=c1)

    # ------------------------------------------------------------------ fusion head (LVA:309-347)
    def _head(self, c: Var, hout: torch.Tensor) -> Var:
        cfg, ctx, tape, P = self.cfg, self._ctx, self.tape, self.store.param
        B, N, D, T, ws = ctx["B"], ctx["N"], cfg.embed_dim, self.T, ctx["ws"]
        nt, ncl, ngc = int(cfg.is_multi), int(cfg.clinical), int(cfg.has_gene_cls)
        off = ncl + nt                              # token order: [clinical], [task], [gene_cls], genes (LVA:259-266,572-580)
        G64 = T - off - ngc
        cls = self._image_token(hout)
        gene = Var(tape.new(B, D))
        if ngc:                                     # prompt_agg "cls": the gene_cls slot is the gene outcome (LVA:316-320 / 620-629)
            ops.copy_rows(c.data, gene.data, B, D, smap=rowmap(1, T, off))
        else:       # "avg": mean over the gene tokens: gene[b, d] = sum_t (1/G64) c[b, off + t, d]
            ops.sgemm(self._mean_w, (0, 1), c.data[:, off:], (1, D), gene.data, (D, 1), 1, D, G64, batch=B, b_bs=T * D, c_bs=D)
        task = clin = None
        if nt:
            task = Var(tape.new(B, D))
            ops.copy_rows(c.data, task.data, B, D, smap=rowmap(1, T, ncl))
        if ncl:
            clin = Var(tape.new(B, D))
            ops.copy_rows(c.data, clin.data, B, D, smap=rowmap(1, T, 0))

        def bwd_gather():
            cg = c.g()
            if gene.grad is not None and ngc:
                ops.copy_rows(gene.grad, cg, B, D, dmap=rowmap(1, T, off), accumulate=True)
            elif gene.grad is not None:   # dc[b, off + t, :] += dgene[b, :] / G64
                ops.sgemm(self._mean_w, (1, 0), gene.grad, (1, 0), cg[:, off:], (D, 1), G64, D, 1, accumulate=True, batch=B,
                          b_bs=D, c_bs=T * D)
            if task is not None and task.grad is not None:
                ops.copy_rows(task.grad, cg, B, D, dmap=rowmap(1, T, ncl), accumulate=True)
            if clin is not None and clin.grad is not None:
                ops.copy_rows(clin.grad, cg, B, D, dmap=rowmap(1, T, 0), accumulate=True)
        tape.record(bwd_gather)
        if cfg.token_agg == "sum":                  # LVA:328-333 / 645-653
            outc = tape.add(cls, gene)
            if task is not None:
                outc = tape.add(outc, task)
            if clin is not None:
                outc = tape.add(outc, clin)
        else:                                       # cat order: img, task, gene, clinical (LVA:334-340 / 654-664)
            parts = [cls] + ([task] if task is not None else []) + [gene] + ([clin] if clin is not None else [])
            outc = Var(tape.new(B, D * len(parts)))
            for j, pv in enumerate(parts):
                ops.copy_rows(pv.data, outc.data[:, j * D:], B, D, ldd=D * len(parts))

            def bwd_cat():
                if outc.grad is None:
                    return
                for j, pv in enumerate(parts):
                    ops.copy_rows(outc.grad[:, j * D:], pv.g(), B, D, lds=D * len(parts), accumulate=True)
            tape.record(bwd_cat)
        n = tape.layernorm(outc, P("final_norm.weight"), P("final_norm.bias"))
        return tape.linear(n, P("final_project.weight"), P("final_project.bias"))

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits: torch.Tensor, call=None):
        """Accumulates into ParamStore.flat_grad; dlogits [B, output_dim] fp32 (already loss-scaled if desired)."""
        tape, logits = call if call is not None else (self.tape, self._logits)
        logits.grad = dlogits.to(self.device, F32).contiguous()
        tape.run_backward()
        self._leaf_join()                      # (side-stream weight-gradient leaves, if enabled: final before anything reads the flat gradient)

    def backward_begin(self, dlogits: torch.Tensor, call):
        """Stage-wise backward (TrainStep's pass groups with per-bucket joins): seed the gradient, then call backward_stage(call)
        until it returns None."""
        _, logits = call
        logits.grad = dlogits.to(self.device, F32).contiguous()

    def backward_stage(self, call):
        """Runs `call`'s backward down to the next interaction-block marker and returns that block's index (the gradients of
        interactions.{i}.* and of everything above are final on the current stream), or None when the backward has finished."""
        tape, _ = call
        blk = tape.run_backward_stage()
        self._leaf_join()
        return blk
