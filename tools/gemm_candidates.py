"""Every kernel / tiling the gemm_nt dispatch could pick, timed at the backbone shapes for a list of row counts (one process per candidate:
MT_GEMM_FORCE is read once): python tools/gemm_candidates.py 4097 8194 12291 ...  Prints, per (M, shape), the time of each candidate, the
dispatch's own choice ("auto") and the best."""
import os, subprocess, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Ms = [int(a) for a in sys.argv[1:]] or [4097, 8194, 12291]
cands = ["auto", "ps", "pp_big", "pp_small", "pp_mixed", "k128"]
res = collections.defaultdict(dict)
for c in cands:
    env = dict(os.environ)
    env.pop("MT_GEMM_FORCE", None)
    if c != "auto":
        env["MT_GEMM_FORCE"] = c
    for M in Ms:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_microbench.py"), "ours", str(M)], env=env, capture_output=True, text=True).stdout
        for m in re.finditer(r"N=(\d+) K=(\d+): ([\d.]+) us", out):
            res[(M, int(m.group(1)), int(m.group(2)))][c] = float(m.group(3))
tot_auto = tot_best = 0.0
for key in sorted(res):
    r = res[key]
    best = min(r, key=r.get)
    tot_auto += r["auto"]; tot_best += r[best]
    print("M=%6d N=%4d K=%4d  " % key + "  ".join(f"{c} {r.get(c, float('nan')):6.1f}" for c in cands) + f"   best {best} ({100 * (1 - r[best] / r['auto']):.0f} % under auto)")
print(f"sum auto {tot_auto:.0f} us, sum best {tot_best:.0f} us ({100 * (1 - tot_best / tot_auto):.1f} %)")
