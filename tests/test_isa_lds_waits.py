"""CPU: no kernel of the library retires an LDS read with a counted `s_waitcnt lgkmcnt(N > 0)` while an LDS operation of the OTHER banking
class (read, write or atomic) stays in flight.

Round 6 (modaltune_amd/csrc/common.h, `lds_f32`): beside another kernel's LDS traffic on the same CU -- the two pass groups of the train
step on two HIP streams -- such a count was met while an older 16-byte read had not delivered lanes 48-63; mt_token_mha_fwd
then summed stale value rows once in ~10 launches beside mt_gemm_tn_f16.  The kernels that mixed the classes (prompt self-attention,
pathway networks) now keep every LDS read of such a loop in ONE class; this test compiles every source to ISA with the build's own flags
and scans it (tools/diag/lds_wait_scan.py), so a later edit -- or a compiler that starts merging reads differently -- cannot bring the
pattern back unnoticed."""
import importlib.util
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_counted_lds_wait_spans_both_banking_classes(tmp_path):
    import __graft_entry__ as ge
    spec = importlib.util.spec_from_file_location("lds_wait_scan", os.path.join(ROOT, "tools", "diag", "lds_wait_scan.py"))
    scanner = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scanner)

    def isa(src):
        out = str(tmp_path / src.replace(".hip", ".s"))
        flags = [f for f in ge._flags_for(src) if f != "-fPIC"]
        subprocess.run([ge.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(ge.CSRC, src)], check=True,
                       stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(max_workers=4) as ex:
        paths = list(ex.map(isa, [s for s in ge.SOURCES if s != "layer.hip"]))      # (layer.hip is host code only)
    flagged = {}
    nk = 0
    for p in paths:
        text = open(p).read()
        nk += text.count(".amdhsa_kernel ")
        flagged.update(scanner.scan(p))
    assert nk >= 100, nk                      # the scan saw the kernels (template instantiations included)
    assert not flagged, flagged
