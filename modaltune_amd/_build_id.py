"""Build provenance of the C-ABI library: ONE hash over everything the binary is made from.

`tree_build_id()` = sha256 over the kernel sources (csrc/*.hip, csrc/*.h), the public header (include/*.h) and the compiler flags
(`__graft_entry__.py`: FLAGS, FLAGS_PER_FILE, NO_VGPR_FORM, the source list).  `build()` compiles that value into the library
(a generated translation unit exporting `mt_build_id()`), rebuilds whenever the library's id differs from the tree's (not by
modification times: a prebuilt `.so` travels to the GPU box and mtimes say nothing there), and `bench.py` / `smoke()` print both
the id the loaded binary reports and whether it matches the sources next to it (`build_id_matches_tree`)."""
from __future__ import annotations

import hashlib
import os
from typing import Dict, Iterable, List, Sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "modaltune_amd", "csrc")
INCLUDE = os.path.join(ROOT, "include")


def _files(dirs_and_suffixes: Iterable[tuple]) -> List[str]:
    out = []
    for d, suf in dirs_and_suffixes:
        if os.path.isdir(d):
            out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(suf)]
    return out


def header_files() -> List[str]:
    return _files([(CSRC, ".h"), (INCLUDE, ".h")])


def source_files() -> List[str]:
    return _files([(CSRC, ".hip")])


def _digest(paths: Sequence[str], extra: str) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.relpath(p, ROOT).encode())
        h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(extra.encode())
    return h.hexdigest()


def flags_string(flags: Sequence[str], per_file: Dict[str, Sequence[str]], no_vgpr_form: Iterable[str], sources: Sequence[str]) -> str:
    return repr((list(flags), sorted((k, list(v)) for k, v in per_file.items()), sorted(no_vgpr_form), list(sources)))


def object_id(src_path: str, flags_for_file: Sequence[str]) -> str:
    """What one object file is made from: its source, every header, its own flags."""
    return _digest([src_path] + header_files(), repr(list(flags_for_file)))


def tree_build_id(flags: str = None) -> str:
    """The id of the sources in this tree.  `flags` defaults to what `__graft_entry__.py` compiles with."""
    if flags is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_mt_graft_entry_flags", os.path.join(ROOT, "__graft_entry__.py"))
        ge = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ge)
        flags = flags_string(ge.FLAGS, ge.FLAGS_PER_FILE, ge.NO_VGPR_FORM, ge.SOURCES)
    return _digest(source_files() + header_files(), flags)
