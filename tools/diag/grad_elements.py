"""Which ELEMENTS of a gradient tensor carry its deviation from the fp64 golden: python tools/diag/grad_elements.py
Runs the golden step of fixture L37_d3_clin_cat and prints, for the Extractor FFN's linear1 tensors, the relative L2 error, the six largest
element deviations (index, |diff|, reference, ours) and the share of the squared error in the largest one.  (A single ReLU gate of the
Extractor FFN decided the other way shows as ~100 % of the error in ONE hidden unit: tools/experiments/README.md.)"""
import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
name = "L37_d3_clin_cat"
gd = os.path.join(ROOT, "tests", "golden")
g, cfg, eng, ts, inp = T._build(os.path.join(gd, f"model_{name}.npz"))
x = torch.from_numpy(inp["x"]).cuda(); genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
clin = torch.from_numpy(inp["clinical"]).cuda() if cfg.clinical else None
ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=False, clinical=clin)
grads = ts.unscaled_grads()
for key in ("interactions.2.extractor.ffn.linear1.bias", "interactions.2.extractor.ffn.linear1.weight", "interactions.1.extractor.ffn.linear1.bias"):
    k = "f64_grad/" + key
    if k not in g.files: print(key, "not in golden full tensors"); continue
    ours = grads[key].double().cpu().numpy().reshape(-1); ref = g[k].reshape(-1)
    d = np.abs(ours - ref); o = np.argsort(-d)[:6]
    print(key, "rel L2", np.linalg.norm(ours - ref) / np.linalg.norm(ref), "norm", np.linalg.norm(ref), "n", ref.size)
    print("  top |diff| elements:", [(int(i), float(d[i]), float(ref[i]), float(ours[i])) for i in o])
    print("  share of squared error in the top element: %.3f" % (d[o[0]] ** 2 / (d ** 2).sum()))
