"""smoke(): one tiny train step of the HIP hot path on cuda:0, checked against the CPU oracle."""
import os
import sys

import numpy as np
import torch


def smoke(L: int = 96, depth: int = 3, seed: int = 5):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import modaltune_oracle as O          # checker only (test infrastructure)
    from . import synth
    from .config import ModelConfig, segment_lengths
    from .engine import Engine
    from .trainer import TrainStep

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    inter = tuple((i, i) for i in range(depth))
    cfg = ModelConfig(depth=depth, interaction_indexes=inter, slide_ngrids=64)
    sizes = synth.toy_group_sizes()
    sd = synth.synth_state_dict(cfg, sizes, seed)
    inp = synth.synth_inputs(L, sizes, seed, grid=64)
    psd = synth.projector_state(seed)
    eng = Engine(cfg, sizes, "cuda:0")
    eng.load_state_dict(sd)
    ts = TrainStep(eng)
    ts.set_projector(psd)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    loss = ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=True)
    torch.cuda.synchronize()
    logits = ts.last_logits.double().cpu()
    # oracle (fp32 CPU)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    psdt = {k: torch.from_numpy(v) for k, v in psd.items()}
    with torch.no_grad():
        ref = O.multitask_logits(sdt, cfg, torch.from_numpy(inp["x"]), torch.from_numpy(inp["coords"]),
                                 [torch.from_numpy(a) for a in inp["genes"]], segment_lengths())
        ref_loss = O.distill_loss(ref, O.projector_forward(torch.from_numpy(inp["text"]), psdt))
    err = float((logits - ref.double()).abs().max() / ref.double().abs().max())
    lerr = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    print(f"smoke: L={L} depth={depth} logits rel err {err:.2e} loss {float(loss):.6f} (oracle {float(ref_loss):.6f}, rel {lerr:.1e})")
    if not (err < 1e-3 and lerr < 1e-3 and int(ts.step_dev) == 1):
        raise AssertionError(f"smoke parity failed: logits {err:.3e} loss {lerr:.3e}")
