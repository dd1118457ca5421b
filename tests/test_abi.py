"""CPU: the C-ABI shared library loads and exports every symbol include/modaltune_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "modaltune_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from modaltune_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names        # ctypes table covers exactly the header
    assert lib.mt_version() >= 100
    assert lib.mt_status_string(-1).decode().startswith("bad argument")


def test_build_id_ties_the_binary_to_the_sources(tmp_path):
    """the library carries the sha256 of (csrc/*, include/*, flags) it was built from; build() compares THAT with the
    tree (not modification times) and a change to any source changes the id."""
    import __graft_entry__ as ge
    from modaltune_amd import _build_id, _lib
    ge.build()
    info = _lib.build_info()
    assert len(info["build_id"]) == 64 and info["build_id_matches_tree"] is True
    assert info["build_id"] == _build_id.tree_build_id()
    assert not ge.stale_objects()                             # nothing to rebuild right after a build
    flags = _build_id.flags_string(ge.FLAGS + ["-DX"], ge.FLAGS_PER_FILE, ge.NO_VGPR_FORM, ge.SOURCES)
    assert _build_id.tree_build_id(flags) != info["build_id"]      # another flag set: another id


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from modaltune_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


_NULL_PROBE = r'''
import sys
sys.path.insert(0, %r)
from modaltune_amd import _lib
lib = _lib.load()
for name, sig in _lib.SIGNATURES.items():
    if name in ("mt_version", "mt_status_string", "mt_build_id", "mt_pool_attn_workspace_floats", "mt_alibi_dist_halves"):      # (plain value functions)
        continue
    args = [0 if t in (_lib.I, _lib.L) else 0.0 if t in (_lib.F, _lib.D) else None for t in sig]
    print(name, getattr(lib, name)(*args), flush=True)
'''


def test_every_launcher_rejects_null_arguments_without_touching_the_device():
    """Error convention: launchers return a negative MtStatus, nothing crosses the C boundary as an exception or a
    fault.  Every entry point called with null pointers and zero sizes -- in a child process, so a fault would fail this test and
    not the runner -- answers MT_ERR_BAD_ARG / MT_ERR_UNSUPPORTED before any HIP call (there is no GPU here)."""
    import subprocess
    import sys
    import __graft_entry__ as ge
    ge.build()
    p = subprocess.run([sys.executable, "-c", _NULL_PROBE % ROOT], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-400:] + p.stderr[-400:]
    rows = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("mt_")]
    from modaltune_amd import _lib
    assert len(rows) == len(_lib.SIGNATURES) - 5
    wrong = [(n, rc) for n, rc in rows if int(rc) >= 0]
    assert not wrong, wrong
