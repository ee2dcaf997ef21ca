#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels of one csrc/*.hip file (no GPU needed: hipcc -S --cuda-device-only).

Per kernel: VGPRs / AGPRs, spills, LDS bytes, and instruction counts by class for the whole kernel and for its hottest
loop (the innermost backward-branch region with the most MFMAs): MFMA, other VALU, LDS, global / buffer memory, SALU,
waitcnt, barriers -- and VALU per MFMA, the figure the PMC passes report dynamically (profiles/*_pmc_summary.csv).

    python scripts/isa_stats.py transformer-explainability_amd/csrc/te_attn_rules.hip [--kernel av_rule] [--defines TE_STUDY]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math"]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse_kernels(asm):
    """-> {mangled name: [lines]} for every function body that contains an s_endpgm (a kernel may have several: an early
    return of idle waves precedes the main loop of the kb attention kernels), up to its .Lfunc_end label."""
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                if any("s_endpgm" in b for b in body):
                    out[name] = body
                name = None
                continue
            body.append(line)
    return out


def instructions(body):
    """[(index, label or None, opcode, operands)] of the real instructions and labels of a kernel body."""
    items = []
    for line in body:
        s = line.split(";")[0].strip()
        if not s or s.startswith((".", "//")) and not s.startswith(".LBB"):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            items.append(("label", m.group(1), ""))
            continue
        parts = s.split(None, 1)
        items.append(("inst", parts[0], parts[1] if len(parts) > 1 else ""))
    return items


def count(items):
    c = {}
    for kind, op, _ in items:
        if kind == "inst":
            k = classify(op)
            c[k] = c.get(k, 0) + 1
    return c


def hottest_loop(items):
    """Innermost region [label .. backward branch to it] with the most MFMAs (ties: most instructions)."""
    pos = {op: i for i, (kind, op, _) in enumerate(items) if kind == "label"}
    loops = []
    for i, (kind, op, args) in enumerate(items):
        if kind == "inst" and classify(op) == "branch":
            tgt = args.strip().split()[-1] if args.strip() else ""
            if tgt in pos and pos[tgt] < i:
                loops.append((pos[tgt], i))
    best = None
    for a, b in loops:
        inner = not any((a2 > a or b2 < b) and a2 >= a and b2 <= b for a2, b2 in loops if (a2, b2) != (a, b))
        c = count(items[a:b + 1])
        key = (c.get("mfma", 0), inner, b - a)
        if best is None or key > best[0]:
            best = (key, c, items[a][1])
    return (best[1], best[2]) if best else ({}, None)


def metadata(asm):
    """per-kernel .amdhsa metadata: name -> dict(vgpr, agpr, sgpr_spill, vgpr_spill, lds)."""
    md, cur = {}, {}
    for line in asm.splitlines():
        line = line.strip()
        for key, tag in ((".name:", "name"), (".vgpr_count:", "vgpr"), (".agpr_count:", "agpr"),
                         (".sgpr_spill_count:", "sgpr_spill"), (".vgpr_spill_count:", "vgpr_spill"),
                         (".group_segment_fixed_size:", "lds")):
            if line.startswith(key) or line.startswith("- " + key):
                v = line.split(":", 1)[1].strip()
                if tag == "name":
                    cur = md.setdefault(v, {})
                else:
                    cur[tag] = v
    return md


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, stdout=subprocess.PIPE, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default="", help="substring filter on the demangled kernel name")
    ap.add_argument("--defines", default="", help="space-separated -D macros")
    a = ap.parse_args()
    src = a.source if os.path.isabs(a.source) else os.path.join(ROOT, a.source)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", *FLAGS, *("-D" + d for d in a.defines.split() if d), "-I", os.path.join(ROOT, "include"),
               "-I", os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.exit(r.stdout)
        asm = open(out).read()
    kernels, md = parse_kernels(asm), metadata(asm)
    pretty = demangle(list(kernels))
    cols = ("mfma", "valu", "lds", "vmem", "salu", "waitcnt", "barrier", "branch")
    for name, body in kernels.items():
        if name not in md or (a.kernel and a.kernel not in pretty[name]):
            continue
        items = instructions(body)
        whole, (loop, label) = count(items), hottest_loop(items)
        m = md[name]
        short = re.sub(r"\(anonymous namespace\)::", "", pretty[name]).split("(")[0]
        print(f"{short}")
        print(f"    vgpr {m.get('vgpr')} agpr {m.get('agpr')} spills sgpr {m.get('sgpr_spill')} vgpr {m.get('vgpr_spill')} static lds {m.get('lds')} B")
        for tag, c in (("kernel", whole), (f"hottest loop {label}", loop)):
            if not c:
                continue
            per = f"{c.get('valu', 0) / c['mfma']:.2f} valu/mfma" if c.get("mfma") else "no mfma"
            print("    " + f"{tag:28s}" + " ".join(f"{k} {c.get(k, 0):5d}" for k in cols) + f"   {per}")


if __name__ == "__main__":
    main()
