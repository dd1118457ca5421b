"""Register use and spills of every kernel that spills (and of the attention / GEMM kernels), from the code-object metadata of the
library's sources compiled to ISA with the build's own flags: python tools/diag/spill_table.py > profiles/rNN_registers_and_spills.txt"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

tmp = tempfile.mkdtemp()


def isa(src):
    out = os.path.join(tmp, src.replace(".hip", ".s"))
    flags = [f for f in ge._flags_for(src) if f != "-fPIC"]
    subprocess.run([ge.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(ge.CSRC, src)], check=True, stderr=subprocess.DEVNULL)
    return out


def short_name(n):
    """kernel name + its integer / bool template arguments from the Itanium mangling (binutils' c++filt does not know DF16_)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n) or re.match(r"_Z(\d+)", n)
    if not m:
        return n
    k = int(m.group(1))
    name, rest = n[m.end():m.end() + k], n[m.end() + k:]
    targs = []
    if rest.startswith("I"):
        for t in re.finditer(r"L([bi])(\d+)E|(DF16_|f)(?=[ILDfE])", rest[1:rest.find("EEv") + 1 if "EEv" in rest else len(rest)]):
            targs.append(("true" if t.group(2) == "1" else "false") if t.group(1) == "b" else t.group(2) if t.group(1) else
                         ("f16" if t.group(3) == "DF16_" else "float"))
    return name + ("<" + ", ".join(targs) + ">" if targs else "")


with ThreadPoolExecutor(max_workers=4) as ex:
    paths = list(ex.map(isa, [s for s in ge.SOURCES if s != "layer.hip"]))
print("Register use and spills, from the code-object metadata of every csrc/*.hip compiled to ISA with the flags of __graft_entry__.build()")
print(f"(tree build id {ge._bid().tree_build_id()[:16]}).  Listed: every kernel with a spill, and the attention / GEMM kernels for reference.")
print("SGPR spills go to VGPR lanes (v_writelane / v_readlane: no memory traffic); VGPR spills go to scratch memory -- the last column counts")
print("the scratch loads / stores that sit INSIDE a loop (between a backward branch and its target).\n")
print(f"{'file':14s} {'kernel':58s} vgpr agpr vgpr_spill sgpr_spill scratch_ops_in_loops")
HOT = ("dilated_attn_", "dense_attn_bwd", "dense_attn_fwd", "gemm_nt_ps", "gemm_nt_pp")
for f in sorted(paths):
    t = open(f).read()
    lines = t.split("\n")
    for chunk in t.split("- .agpr_count:")[1:]:
        n = re.search(r"\.name:\s+(\S+)", chunk).group(1)
        v = int(re.search(r"\.vgpr_count:\s+(\d+)", chunk).group(1))
        s = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", chunk).group(1))
        ss = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", chunk).group(1))
        a = int(re.match(r"\s*(\d+)", chunk).group(1))
        dem = short_name(n)
        if not (s or ss or any(k in dem for k in HOT)):
            continue
        i0 = next(i for i, l in enumerate(lines) if l.startswith(n + ":"))
        i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))      # (a kernel may hold several s_endpgm)
        body = lines[i0:i1]
        labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        inloop = sum(1 for i, l in enumerate(body) if "scratch_" in l and any(a_ <= i <= b_ for a_, b_ in loops))
        print(f"{os.path.basename(f)[:-2]:14s} {dem[:58]:58s} {v:4d} {a:4d} {s:10d} {ss:10d} {inloop:6d}")
