"""Build the C-ABI library of another git revision next to the working-tree one, for same-box A/B timing:
    python tools/build_ref_lib.py HEAD build_variants/base/libmodaltune_hip.so
The A/B itself: tools/ab_lib.sh <libA.so> <libB.so> <script> [args]  (MODALTUNE_HIP_LIB selects the library)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

ref, out = sys.argv[1], os.path.abspath(sys.argv[2])
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "modaltune_amd", "csrc"))
os.makedirs(os.path.join(tmp, "include"))
files = subprocess.run(["git", "ls-tree", "-r", "--name-only", ref, "modaltune_amd/csrc", "include"], cwd=ROOT, capture_output=True,
                       text=True, check=True).stdout.split()
for f in files:
    with open(os.path.join(tmp, f), "wb") as fh:
        fh.write(subprocess.run(["git", "show", f"{ref}:{f}"], cwd=ROOT, capture_output=True, check=True).stdout)
objs = []
for src in G.SOURCES:
    o = os.path.join(tmp, src.replace(".hip", ".o"))
    if not os.path.exists(os.path.join(tmp, "modaltune_amd", "csrc", src)):
        continue                      # (a source the other revision does not have yet)
    subprocess.run([G.HIPCC] + G._flags_for(src) + ["-c", os.path.join(tmp, "modaltune_amd", "csrc", src), "-o", o], check=True)
    objs.append(o)
# the id of such a build names the revision (it never matches the working tree: bench.py says build_id_matches_tree = false)
with open(os.path.join(tmp, "build_id.cpp"), "w") as fh:
    fh.write('extern "C" const char* mt_build_id(void) { return "ref:%s"; }\n' % ref)
subprocess.run([G.HIPCC, "-O1", "-fPIC", "-c", os.path.join(tmp, "build_id.cpp"), "-o", os.path.join(tmp, "build_id.o")], check=True)
objs.append(os.path.join(tmp, "build_id.o"))
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print("built", out, "from", ref)
