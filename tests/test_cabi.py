"""C-ABI surface checks that need no GPU: the library is built, loads through ctypes, exports every
symbol include/te_relprop.h declares, and the host-side workspace queries / argument validation work."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "te_relprop.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from transformer_explainability_amd import _lib
    return _lib.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(te_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in te_relprop.h but not exported by libte_relprop.so"


def test_binding_covers_header():
    from transformer_explainability_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_status(lib):
    assert lib.te_version() >= 501
    # the library in the tree is the one that travels to the GPU box: it must be the shipped build, not a measurement build
    # (TE_BUILD_DEFINES=TE_STUDY / TE_X6_STUDY: getenv switches, study schedules)
    assert lib.te_x6_study_build() == 0, "a measurement build is in the tree: python transformer-explainability_amd/build.py --force"
    assert lib.te_status_string(0) == b"ok"
    assert b"workspace" in lib.te_status_string(-2)


def test_workspace_queries(lib):
    T, i, o = 12608, 768, 3072
    assert lib.te_linear_relprop_workspace_bytes(T, i, o, 0) >= T * o * 4
    assert lib.te_linear_relprop_workspace_bytes(T, i, o, 1) >= 2 * T * o * 4
    assert lib.te_matmul_relprop_av_workspace_bytes(64, 12, 197, 64) >= 64 * 12 * 197 * 64 * 4
    assert lib.te_matmul_relprop_qk_workspace_bytes(64, 12, 197, 64) >= 64 * 12 * 197 * 197 * 4
    assert lib.te_add_relprop_workspace_bytes(64, 197 * 768) > 0
    assert lib.te_add_bcast_relprop_workspace_bytes(32, 12, 512) > 0
    assert lib.te_rollout_workspace_bytes(12, 64, 197) >= 13 * 64 * 197 * 197 * 4
    assert lib.te_linear_relprop_workspace_bytes(0, 1, 1, 0) == 0


def test_argument_validation_without_device(lib):
    # null pointers / bad sizes are rejected on the host before any HIP call
    assert lib.te_clone_relprop_f32(None, None, None, None, None, 16, None) == -1
    assert lib.te_linear_relprop_f32(None, None, None, None, 1, 4, 4, 1.0, 0, None, 0, None) == -1
    assert lib.te_rollout_f32(None, 1, 0, 1, 4, 0, None, None, 0, None) == -1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from transformer_explainability_amd import ops, TeError
    x = torch.zeros(2, 8)
    with pytest.raises(TeError):
        ops.clone_relprop([x, x], x)


def test_build_id_matches_tree_and_loader_refuses_other_sources(monkeypatch):
    """te_build_id() = content hash of csrc/ + include/ (+ flags) baked in by build.py; the loader
    recomputes the source half from the tree and refuses a library built from anything else."""
    import pytest
    from transformer_explainability_amd import _buildid, _lib
    bid = _lib.build_id()
    src, flags = bid.split("-")
    assert len(src) == 16 and len(flags) == 8 and src == _buildid.source_hash()
    assert all(p.endswith((".hip", ".h")) for p in _buildid.source_files()) and len(_buildid.source_files()) >= 16
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_buildid, "source_hash", lambda: "0" * 16)
    monkeypatch.delenv("TE_RELPROP_LIB", raising=False)
    monkeypatch.delenv("TE_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(_lib.TeError, match="built from other sources"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().te_build_id().decode() == bid
