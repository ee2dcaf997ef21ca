"""CPU: a static check of the compiled gfx950 ISA of the attention kernels that issue their global loads as inline asm
(csrc/te_attn_kb.hip: `ld128_hidden` + hand-counted `s_waitcnt vmcnt(n)`).  hipcc does not know those destination registers
are in flight and may copy one before the wait covers it -- round 5's QK study kernel did exactly that on its loop back-edge
and was wrong in 2 of 10 graph replays, in the last bits of one sample (attention section).  The check
(scripts/check_hidden_loads.py) walks every tile loop twice and fails on any instruction that touches a register whose load
no wait has covered yet; it runs on the shipped build's code, i.e. the default AV kernels."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("defines", [[], ["-DTE_STUDY"]], ids=["shipped", "study"])
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path, defines):
    build = _load("_te_build_isa", os.path.join(ROOT, "transformer-explainability_amd", "build.py"))
    checker = _load("_te_check_hidden_loads", os.path.join(ROOT, "scripts", "check_hidden_loads.py"))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "te_attn_kb.s"
    cmd = [hipcc, *build.CXXFLAGS, *defines, "--cuda-device-only", "-S", "-I", build.INCLUDE, "-I", build.CSRC,
           os.path.join(build.CSRC, "te_attn_kb.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    lines = out.read_text().split("\n")
    found = list(checker.kernels(lines, "_kb_kernel"))
    assert len(found) >= 2, "the kb kernels were not found in the ISA listing"
    assert any("av6_kb_kernel" in name for name, _, _ in found)
    bad = sum(checker.check(lines, lo, hi, name) for name, lo, hi in found)
    assert bad == 0, f"{bad} instruction(s) touch a register with a hidden load in flight (see the captured output)"


@pytest.mark.parametrize("defines", [[], ["-DTE_STUDY"]], ids=["shipped", "study"])
def test_rc_kernels_straight_line_hidden_loads(tmp_path, defines):
    """csrc/te_attn_rc.hip (round 6: the QK rule / softmax backward with row-block and key-block owners) keeps its hidden loads
    in STRAIGHT-LINE code -- its first, rolled version had hipcc copy loop-carried registers with loads in flight on the
    back-edge, which this checker found before the kernel ever ran -- and is checked by one linear walk over each kernel:
    448 / 504 hidden loads, no instruction may touch a destination before its hand-counted wait."""
    build = _load("_te_build_isa_rc", os.path.join(ROOT, "transformer-explainability_amd", "build.py"))
    checker = _load("_te_check_hidden_loads_rc", os.path.join(ROOT, "scripts", "check_hidden_loads.py"))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "te_attn_rc.s"
    cmd = [hipcc, *build.CXXFLAGS, *defines, "--cuda-device-only", "-S", "-I", build.INCLUDE, "-I", build.CSRC,
           os.path.join(build.CSRC, "te_attn_rc.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    lines = out.read_text().split("\n")
    found = list(checker.kernels(lines, "qk_rc_kernel"))
    assert len(found) == 2, "qk_rc_kernel<RULE> / <BWD> were not found in the ISA listing"
    for name, lo, hi in found:
        assert sum("buffer_load_dword" in ln for ln in lines[lo:hi]) >= 400
        # the loop-based check reports on (and only on) loops that contain a hidden load: none may be left
        assert checker.check(lines, lo, hi, name) == 0
    bad = sum(checker.check_linear(lines, lo, hi, name) for name, lo, hi in found)
    assert bad == 0, f"{bad} instruction(s) touch a register with a hidden load in flight (see the captured output)"


def test_checker_sees_a_planted_copy(tmp_path):
    """The checker itself: a listing with a move of an in-flight register before its wait, and on the loop back-edge."""
    checker = _load("_te_check_hidden_loads2", os.path.join(ROOT, "scripts", "check_hidden_loads.py"))
    listing = """_ZN4test9kb_kernelEv:
\ts_waitcnt vmcnt(0)
\tbuffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen
.LBB0_1:                                ; =>This Loop Header: Depth=1
\tv_mov_b32_e32 v20, v8
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v9, v4, v5
\tbuffer_load_dwordx4 v[8:11], v1, s[0:3], 0 offen
\tv_mov_b32_e32 v4, v8
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
""".split("\n")
    found = list(checker.kernels(listing, "kb_kernel"))
    assert len(found) == 1
    name, lo, hi = found[0]
    assert checker.check(listing, lo, hi, name) == 2      # line 9 (in the body) and line 5 (second trip: carried over the back-edge)
    # the linear walk (straight-line kernels) sees the copy in the body too, and a wait that is one step short
    assert checker.check_linear(listing, lo, hi, name) >= 1
    short = """_ZN4test9rc_kernelEv:
\tbuffer_load_dword v4, v1, s[0:3], 0 offen
\tbuffer_load_dword v5, v1, s[0:3], 0 offen
\ts_waitcnt vmcnt(1)
\tv_add_f32_e32 v9, v4, v4
\tv_add_f32_e32 v9, v5, v5
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v9, v5, v5
\ts_endpgm
.Lfunc_end0:
""".split("\n")
    (name, lo, hi), = list(checker.kernels(short, "rc_kernel"))
    assert checker.check_linear(short, lo, hi, name) == 1  # v5 read under vmcnt(1): its load is the one still in flight
