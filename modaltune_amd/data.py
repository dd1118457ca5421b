"""Input pipeline for the train / eval step: case assembly, patch subsampling, host -> HBM staging.

Reference: data_utils/datasets.py:213-285 (`FeaturesGeneTextDataset.__getitem__`, case-wise mode) over the on-disk format
written by data_utils/TCGA_extract_feats_GIGAPATH.py:107-110 (`torch.save({"features": [L, 1536] fp32, "coords": [L, 2]})`):
  * the slides of one case are concatenated; slide i's coords get `+ [0, offset_i]` with
    `offset_i = max_y(coords of slide i-1 as stored) + 1500` (DS:236-238 -- the previous slide's OWN maximum, not a running sum);
  * above `threshold` patches (25 000 GigaPath / 15 000 TITAN) a uniformly random subset of `threshold` patches is kept in
    ascending index order (`torch.randperm(n)[:threshold].sort()`, DS:274-281);
  * genes: dict pathway index -> float32 vector, in pathway order.
The reference does this in DataLoader worker processes and moves fp32 tensors to the GPU synchronously inside the step
(`images.to(device)`, TM:199-206).  Here a `CasePrefetcher` keeps the next case's upload in flight on a copy stream while
the current step runs: pinned fp32 staging -> async H2D -> fp16 cast on the device (the GEMM operand dtype), gated by an
event.  PyTorch is used for files, pinned memory, streams and events only.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

Y_GAP = 1500      # DS:238


def concat_case(slides: Sequence[Dict[str, torch.Tensor]]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Features [L, C] and coords [L, 2] of one case from its slides' `{features, coords}` dicts (DS:231-240)."""
    imgs, coords = [], []
    offset = 0
    for s in slides:
        imgs.append(s["features"])
        coords.append(s["coords"] + torch.tensor([0, offset]))
        offset = s["coords"].max(dim=0)[0][1].item() + Y_GAP
    return torch.cat(imgs), torch.cat(coords)


def subsample_indices(n: int, threshold: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
    """Sorted random subset of `threshold` patch indices when n > threshold, else None (DS:274-281; same RNG call)."""
    if n <= threshold:
        return None
    idx = torch.randperm(n, generator=generator)[:threshold]
    return idx.sort()[0]


def load_case(paths: Sequence[str], threshold: int = 25000, generator: Optional[torch.Generator] = None):
    """torch.load the slides of a case, assemble and subsample: (features fp32 [L, C], coords [L, 2]) on the host."""
    feats, coords = concat_case([torch.load(p, map_location="cpu") for p in paths])
    idx = subsample_indices(len(feats), threshold, generator)
    if idx is not None:
        feats, coords = feats[idx, :], coords[idx, :]
    return feats, coords


def flatten_genes(genes) -> torch.Tensor:
    """dict / list of per-pathway vectors -> one flat fp32 vector in pathway order (what the grouped kernels take)."""
    if torch.is_tensor(genes):
        return genes.reshape(-1).float()
    vals = [genes[k] for k in sorted(genes.keys())] if isinstance(genes, dict) else list(genes)
    return torch.cat([torch.as_tensor(v, dtype=torch.float32).reshape(-1) for v in vals])


def convert_to_f16_shard(pt_path: str, out_prefix: str) -> Tuple[str, str]:
    """Re-encode a reference `.pt` slide as two memory-mappable arrays: `<prefix>.features.f16.npy` (half the bytes the
    trainer has to read and upload; the GEMM consumes fp16) and `<prefix>.coords.npy`."""
    d = torch.load(pt_path, map_location="cpu")
    fp, cp = out_prefix + ".features.f16.npy", out_prefix + ".coords.npy"
    np.save(fp, d["features"].numpy().astype(np.float16))
    np.save(cp, d["coords"].numpy())
    return fp, cp


def load_case_shards(prefixes: Sequence[str], threshold: int = 25000, generator: Optional[torch.Generator] = None):
    """Same assembly from fp16 shards (memory-mapped; only the kept rows are touched after subsampling)."""
    slides = [{"features": torch.from_numpy(np.load(p + ".features.f16.npy", mmap_mode="r")[:]),
               "coords": torch.from_numpy(np.load(p + ".coords.npy"))} for p in prefixes]
    feats, coords = concat_case(slides)
    idx = subsample_indices(len(feats), threshold, generator)
    if idx is not None:
        feats, coords = feats[idx, :], coords[idx, :]
    return feats, coords


class StagedCase:
    __slots__ = ("x", "coords", "genes", "text", "clinical", "case_id", "ready")

    def __init__(self, x, coords, genes, text, clinical, case_id, ready):
        self.x, self.coords, self.genes, self.text, self.clinical, self.case_id, self.ready = x, coords, genes, text, clinical, case_id, ready


def _host_copy(dst: torch.Tensor, src: torch.Tensor):
    """Pageable -> pinned staging copy on the CALLING thread (one memcpy through numpy).  torch's own CPU copy fans a 61 MB slide out over
    every core the machine reports (128 OpenMP threads on a GPU box whose cgroup grants 16): the workers then spin on the cores the launch
    thread needs, and the step that should hide the staging got 3-6 ms longer (tools/pipeline_bench.py: 45-48 ms against 42.5 resident)."""
    if src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous() and not src.requires_grad:
        np.copyto(dst.numpy(), src.numpy())
    else:
        dst.copy_(src)


class CasePrefetcher:
    """Iterates host-side cases `{features|x, coords, genes, text, clinical?, case_id?}` and yields them resident in HBM
    (x as fp16 [L, C]) with the NEXT case's upload already in flight.  `depth` staging slots; every slot owns its pinned
    host buffer and device buffers (grown on demand), so no allocation happens in steady state."""

    def __init__(self, cases: Iterable[Dict], device="cuda", depth: int = 2):
        self.cases, self.device, self.depth = cases, torch.device(device), max(2, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned: List[Optional[torch.Tensor]] = [None] * self.depth
        self._dev32: List[Optional[torch.Tensor]] = [None] * self.depth
        self._uploaded: List[Optional[torch.cuda.Event]] = [None] * self.depth    # H2D out of the slot's pinned buffer done

    def _stage(self, slot: int, case: Dict) -> StagedCase:
        from . import ops
        feats = case["features"] if "features" in case else case["x"]
        feats = feats.reshape(-1, feats.shape[-1])
        L, C = feats.shape
        n = L * C
        if self._uploaded[slot] is not None:
            self._uploaded[slot].synchronize()        # the host is about to overwrite this slot's pinned buffer
        with torch.cuda.stream(self.copy_stream):
            if feats.dtype == torch.float16:
                pin = self._pinned[slot]
                if pin is None or pin.numel() < n or pin.dtype != torch.float16:
                    pin = self._pinned[slot] = torch.empty(n, dtype=torch.float16).pin_memory()
                _host_copy(pin[:n], feats.reshape(-1))
                x = torch.empty(L, C, dtype=torch.float16, device=self.device)
                x.view(-1).copy_(pin[:n], non_blocking=True)
            else:
                pin = self._pinned[slot]
                if pin is None or pin.numel() < n or pin.dtype != torch.float32:
                    pin = self._pinned[slot] = torch.empty(n, dtype=torch.float32).pin_memory()
                _host_copy(pin[:n], feats.reshape(-1))
                d32 = self._dev32[slot]
                if d32 is None or d32.numel() < n:
                    d32 = self._dev32[slot] = torch.empty(n, dtype=torch.float32, device=self.device)
                d32[:n].copy_(pin[:n], non_blocking=True)
                x = torch.empty(L, C, dtype=torch.float16, device=self.device)
                ops.cast_f32_to_f16(d32[:n], x)                      # on the copy stream (ops use the current stream)
            self._uploaded[slot] = torch.cuda.Event()
            self._uploaded[slot].record(self.copy_stream)
            genes = flatten_genes(case["genes"]).to(self.device, non_blocking=True)
            text = torch.as_tensor(case["text"], dtype=torch.float32).to(self.device, non_blocking=True)
            clin = case.get("clinical")
            if clin is not None and len(clin):
                clin = torch.as_tensor(clin, dtype=torch.float32).to(self.device, non_blocking=True)
            else:
                clin = None
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return StagedCase(x, case["coords"], genes, text, clin, case.get("case_id"), ready)

    def __iter__(self) -> Iterator[StagedCase]:
        it = iter(self.cases)
        slot = 0
        try:
            nxt = self._stage(slot, next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            slot = (slot + 1) % self.depth
            try:
                # device tensors are fresh allocations per case; the reused staging buffers are ordered by the copy
                # stream itself (device side) and by the slot's upload event (host side): the copy stream never waits
                # for the compute stream, so the upload overlaps the running step
                nxt = self._stage(slot, next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(cur.ready)
            for t in (cur.x, cur.genes, cur.text, cur.clinical):
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield cur
