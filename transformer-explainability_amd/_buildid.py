"""Content hash of the sources libte_relprop.so is built from.

``build.py`` bakes ``source_hash() + "-" + flags_hash(...)`` into the library (exported as ``te_build_id()``);
``_lib.load()`` recomputes ``source_hash()`` from the tree it was imported from and refuses a library built from
other sources -- the prebuilt in-tree ``.so`` is what travels to the GPU box, so "the tests ran the shipped
sources" is checked there, not assumed from mtimes.  No torch / ctypes imports: shared by the build script and the
loader."""
from __future__ import annotations

import hashlib
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO, "include")


def source_files():
    """Every file a translation unit of the library reads: csrc/*.hip, csrc/*.h, include/*.h (sorted, repo-relative)."""
    out = []
    for d in (CSRC, INCLUDE):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                out.append(os.path.join(d, name))
    return out


def source_hash() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.relpath(path, REPO).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def flags_hash(flags) -> str:
    return hashlib.sha256("\0".join(flags).encode()).hexdigest()[:8]


def build_id(flags) -> str:
    return source_hash() + "-" + flags_hash(flags)
