"""Gradient error of the HIP train step against every reference golden: worst norm error, worst full-tensor relative L2 error
(the numbers behind the tolerances of tests/test_model_gpu.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
gd = os.path.join(ROOT, "tests", "golden")
SCALES = [float(v) for v in os.environ.get("MT_SCALES", "32768").split(",")]
for name, sc in [(n, s_) for n in ["L37_d3", "L1500_d3", "L512_d12", "L37_d3_clin", "L37_d3_cat", "L37_d3_pan", "L129_d3_pan", "L37_d3_single"] for s_ in SCALES]:
    g, cfg, eng, ts, inp = T._build(os.path.join(gd, f"model_{name}.npz"))
    ts.scale.fill_(sc)
    name = f"{name}@{sc:g}"
    x = torch.from_numpy(inp["x"]).cuda(); genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    clin = torch.from_numpy(inp["clinical"]).cuda() if cfg.clinical else None
    ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=False, clinical=clin)
    torch.cuda.synchronize()
    grads = ts.unscaled_grads()
    names = [str(n) for n in g["f64_grad_names"]]
    ours = np.array([float(grads[n].double().norm()) for n in names]); ref = g["f64_grad_norms"]
    rel = np.abs(ours - ref) / (ref + 1e-6 * ref.max())
    order = np.argsort(-rel)[:4]
    full = sorted(((float(np.linalg.norm(grads[k[9:]].double().cpu().numpy() - g[k]) / (np.linalg.norm(g[k]) + 1e-300)), k[9:]) for k in g.files if k.startswith("f64_grad/")), reverse=True)
    print(name, "norms worst:", [(names[i], f"{rel[i]:.2e}") for i in order], "| >1%:", int((rel > 1e-2).sum()), "of", len(rel))
    print("   full worst:", [(k, f"{e:.2e}") for e, k in full[:4]])
