"""Build libte_relprop.so (hand-written HIP for gfx950) in-tree with hipcc.

    python transformer-explainability_amd/build.py [--force] [--verbose]

hipcc cross-compiles gfx950 code objects without a GPU.  The library is linked against the HIP
runtime that PyTorch-ROCm bundles (torch/lib/libamdhip64.so) -- NOT /opt/rocm's -- so that a
hipStream_t / device pointer handed over by torch belongs to the runtime the kernels are launched
through (one HIP runtime per process; SURVEY.md section 7 "hard parts").  /opt/rocm/lib is kept as
an rpath fallback for hosts without torch.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libte_relprop.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")

SOURCES = ["te_api.hip", "te_elementwise.hip", "te_linear.hip", "te_linear_x6.hip", "te_attn.hip", "te_attn_mfma.hip", "te_attn_rules.hip", "te_attn_kb.hip", "te_attn_rc.hip", "te_attn_fwd6.hip", "te_attn_fwd6l.hip", "te_attn_bwd6l.hip", "te_attn_long.hip", "te_norm_act.hip",
           "te_rollout.hip", "te_heatmap.hip", "te_conv.hip", "te_perturb.hip"]

CXXFLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",                              # the reference rounds every op separately
    "-fhip-fp32-correctly-rounded-divide-sqrt",       # IEEE division inside safe_divide
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build libte_relprop.so)")


def torch_lib_dir() -> str | None:
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = os.path.join(os.path.dirname(spec.origin), "lib")
            if os.path.exists(os.path.join(d, "libamdhip64.so")):
                return d
    except Exception:
        pass
    return None


def _buildid_module():
    spec = importlib.util.spec_from_file_location("_te_buildid", os.path.join(PKG_DIR, "_buildid.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "te_common.h"), os.path.join(CSRC, "te_attn_l6.h"), os.path.join(INCLUDE, "te_relprop.h"), __file__]
    # measurement builds only (benchmarks/, scripts/): TE_BUILD_DEFINES="TE_X6_STUDY" adds -D flags; they are built NEXT TO the
    # shipped library (lib/libte_relprop_study.so, objects in build_study/) and selected with TE_RELPROP_LIB, so that a
    # measurement never replaces the library the tests load
    extra = ["-D" + d for d in os.environ.get("TE_BUILD_DEFINES", "").split() if d]
    obj_dir, lib_path = (OBJ_DIR + "_study", LIB_PATH.replace(".so", "_study.so")) if extra else (OBJ_DIR, LIB_PATH)
    os.makedirs(obj_dir, exist_ok=True)
    # provenance: a content hash of every source + the flags, baked into te_api.o (te_build_id()) and
    # checked by _lib.load() against the tree.  A changed id recompiles te_api.hip even when its own mtime says "fresh".
    bid = _buildid_module().build_id([*CXXFLAGS, *extra])
    stamp = os.path.join(obj_dir, "build_id.txt")
    old_bid = open(stamp).read().strip() if os.path.exists(stamp) else None
    if old_bid == bid and os.path.exists(lib_path) and not force:
        return lib_path          # same sources, same flags: up to date whatever the mtimes of a copied tree say
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        stale_id = src == "te_api.hip" and old_bid != bid
        if force or stale_id or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc, *CXXFLAGS, *extra, "-I", INCLUDE, "-I", CSRC, "-c", s, "-o", o]
            if src == "te_api.hip":
                cmd.insert(-4, f'-DTE_BUILD_ID="{bid}"')
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if rebuilt or force or not os.path.exists(lib_path):
        tl = torch_lib_dir()
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path]
        if tl:
            link += ["-L", tl, "-Wl,-rpath," + tl]
        link += ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
        if verbose:
            print(" ".join(link), flush=True)
        r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(bid + "\n")
    return lib_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
