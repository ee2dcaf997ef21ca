#!/usr/bin/env python3
"""Static check of the hand-counted s_waitcnt of the kb attention kernels (te_attn_kb.hip) on the compiled ISA.

The tile loops issue their global loads as inline asm (`ld128_hidden`): hipcc does not know the destination registers are
still in flight, so nothing stops it from COPYING one (register coalescing at a loop back-edge, a phi with the prologue's
register) or from scheduling a move above the hand-written `s_waitcnt vmcnt(n)`.  Such a move reads whatever the register
held before the load lands -- almost always the load has landed long ago, which makes the failure rare and timing dependent
(the QK x6 study kernel: 2 of 10 graph replays differed in one sample's last bits, round 5).

This script walks the ISA of every kernel whose name matches, linearly: the code after the last vmcnt(0) before the tile loop,
then the loop body twice (so that loads carried over the back-edge meet their waits), then the code laid out after the loop up to
its first vmcnt(0) (round 6), and reports every instruction that mentions a register with a load in flight.  In flight = not yet covered by an `s_waitcnt vmcnt(n)`: a load is complete after
a wait iff at least n LOADS were issued after it (stores are ignored, which only makes the check stricter).

    hipcc --offload-arch=gfx950 -O3 ... -S -o te_attn_kb.s te_attn_kb.hip       (device pass of --save-temps)
    python scripts/check_hidden_loads.py te_attn_kb.s av6_kb_kernel

Exit status 1 if anything is reported.
"""
import re
import sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def kernels(lines, pattern):
    start = None
    name = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            if start is not None and pattern in name:
                yield name, start, i
            start, name = i, m.group(1)
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            if start is not None and pattern in name:
                yield name, start, i
            start = None


def check(lines, lo, hi, name):
    body = lines[lo:hi]
    # the tile loops: every depth-1 loop that contains a buffer_load_dwordx4 and is closed by a branch to its header
    headers = []
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if not m:
            continue
        note, k = ln, i + 1                               # the annotation may continue on comment-only lines
        while k < len(body) and body[k].lstrip().startswith(";") and not body[k].startswith(";;"):
            note += body[k]
            k += 1
        if "This Loop Header: Depth=1" in note or "This Inner Loop Header: Depth=1" in note:
            headers.append((i, m.group(1)))
    problems = []
    for hidx, label in headers:
        # extent of the loop: every block hipcc annotates with this header (a latch block may be laid out BEFORE the header)
        tag = label[2:]                                   # "BB13_18"
        member = [i for i, ln in enumerate(body)
                  if re.search(r"(in Loop: Header=|Parent Loop )" + re.escape(tag) + r"\b", ln)]
        first = min([hidx] + member)
        end = max([hidx] + member)
        while first > 0 and not re.match(r"^\.LBB\d+_\d+:", body[first]) and first != hidx:
            first -= 1
        while end + 1 < len(body) and not re.match(r"^(\.LBB\d+_\d+:|; %bb\.)", body[end + 1]):
            end += 1
        loop = body[hidx:end + 1] + body[first:hidx]      # one trip, starting at the header
        numbered = [lo + hidx + k for k in range(end + 1 - hidx)] + [lo + first + k for k in range(hidx - first)]
        if not any("buffer_load" in x for x in loop):
            continue
        pro = first
        while pro > 0 and not re.search(r"s_waitcnt.*vmcnt\(0\)", body[pro]):
            pro -= 1
        trace = [(lo + pro + k, x) for k, x in enumerate(body[pro:first])]
        trace += list(zip(numbered, loop)) * 2
        # ... and the code that follows the loop in layout order, up to the first full drain (the compiler may
        # reuse a register whose prefetch beyond the last tile is still in flight, or hoist epilogue work above the wait)
        k = end + 1
        while k < len(body) and k < end + 600:
            trace.append((lo + k, body[k]))
            if re.search(r"s_waitcnt.*vmcnt\(0\)", body[k]) or body[k].lstrip().startswith(("s_endpgm", "s_cbranch", "s_branch")):
                break
            k += 1
        inflight = []                                     # [(regs, line)] oldest first
        for no, ln in trace:
            if not ln.startswith("\t") or ln.lstrip().startswith((";", ".")):
                continue
            ins = ln.strip()
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if ins.startswith("s_waitcnt") and m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
                if n == 0:
                    inflight = []
                continue
            if ins.startswith("s_waitcnt"):
                continue
            used = regs_of(ins)
            for regs, at in inflight:
                hit = used & regs
                if hit:
                    problems.append((no + 1, ins, at + 1, sorted(hit)))
            if ins.startswith("buffer_load"):
                dst = ins.split()[1].rstrip(",")
                inflight.append((regs_of(dst), no))
        print(f"{name}: loop {label}, {len(loop)} lines, "
              f"{sum('buffer_load' in x for x in loop)} loads / {sum('buffer_store' in x for x in loop)} stores per trip")
    seen = set()
    for no, ins, at, hit in problems:
        if (no, at) in seen:
            continue
        seen.add((no, at))
        print(f"  line {no}: `{ins}` touches v{hit} of the load issued at line {at} before a wait covers it")
    return len(seen)


def check_linear(lines, lo, hi, name):
    """Kernels whose hidden loads sit in STRAIGHT-LINE code (te_attn_rc.hip: phases unrolled over their K16 steps, the
    arithmetic of a step under a forward, wave-uniform branch): one linear walk over the whole function in layout order, i.e.
    the path on which every guarded block runs -- the path that touches the most registers between a request and its wait.
    Same rule as check(): an instruction may not mention a register whose buffer_load no `s_waitcnt vmcnt(n)` has covered
    (n counts the hidden loads issued after it; compiler-visible loads in flight only make the real wait longer)."""
    problems, inflight, loads = [], [], 0
    for no in range(lo, hi):
        ln = lines[no]
        if not ln.startswith("\t") or ln.lstrip().startswith((";", ".")):
            continue
        ins = ln.strip()
        m = re.search(r"vmcnt\((\d+)\)", ins)
        if ins.startswith("s_waitcnt"):
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if 0 < n < len(inflight) else ([] if n == 0 else inflight)
            continue
        used = regs_of(ins)
        for regs, at in inflight:
            hit = used & regs
            if hit:
                problems.append((no + 1, ins, at + 1, sorted(hit)))
        if ins.startswith("buffer_load"):
            inflight.append((regs_of(ins.split()[1].rstrip(",")), no))
            loads += 1
    print(f"{name}: linear walk, {hi - lo} lines, {loads} hidden loads")
    seen = set()
    for no, ins, at, hit in problems:
        if (no, at) not in seen:
            seen.add((no, at))
            print(f"  line {no}: `{ins}` touches v{hit} of the load issued at line {at} before a wait covers it")
    return len(seen)


def main():
    args = [a for a in sys.argv[1:] if a != "--linear"]
    linear = "--linear" in sys.argv[1:]
    path, pattern = args[0], args[1] if len(args) > 1 else "_kb_kernel"
    lines = open(path).read().split("\n")
    bad = 0
    for name, lo, hi in kernels(lines, pattern):
        bad += check_linear(lines, lo, hi, name) if linear else check(lines, lo, hi, name)
    print("in-flight register touched: %d place(s)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
