"""ctypes binding of libte_relprop.so (the C ABI declared in include/te_relprop.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device is
visible when an op is called, this module raises -- loudly.  PyTorch is imported first so that the
library binds to the HIP runtime torch already loaded (one libamdhip64 per process).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must precede dlopen of libte_relprop: loads torch's libamdhip64 first)

_PKG = os.path.dirname(os.path.abspath(__file__))
# (TE_RELPROP_LIB: measurement builds of the same sources under another name, e.g. an A/B of two -D variants in one process tree)
LIB_PATH = os.environ.get("TE_RELPROP_LIB") or os.path.join(_PKG, "lib", "libte_relprop.so")

TE_OK = 0
MIN_LIB_VERSION = 600      # te_version(): 0.6.0, the round-6 ABI (te_build_id)
TE_ERR_UNSUPPORTED = -3
TE_VARIANT_OURS = 0
TE_VARIANT_LRP = 1
TE_IMPL_SIMPLE = 0x100
TE_ROLLOUT_NORMALISE = 1
TE_ROLLOUT_CLS_FIXUP = 2
TE_ROLLOUT_ROW0 = 4

_P, _I64, _F, _I, _SZ = c_void_p, c_int64, c_float, c_int, c_size_t

# name -> (restype, argtypes); mirrors include/te_relprop.h one to one
SIGNATURES = {
    "te_version": (_I, []),
    "te_status_string": (c_char_p, [_I]),
    "te_device_check": (_I, []),
    "te_x6_study_build": (_I, []),
    "te_build_id": (c_char_p, []),
    "te_linear_relprop_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "te_linear_relprop_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_linear_zpass_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "te_linear_cpass_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "te_linear_zpass_fwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "te_linear_relprop_fwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _SZ, _P]),
    "te_linear_zpass_fwd_scaled_f32": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "te_linear_relprop_fwd_scaled_f32": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _SZ, _P]),
    "te_matmul_relprop_av_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64]),
    "te_matmul_relprop_av_f32": (_I, [_P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64,
                                      _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_matmul_relprop_qk_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64]),
    "te_matmul_relprop_av_fwd_f32": (_I, [_P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64,
                                          _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_matmul_relprop_av_fwdz_f32": (_I, [_P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _P,
                                           _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_attention_forward_supported": (_I, [_I64, _I64]),
    "te_attention_forward_f32": (_I, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P]),
    "te_attention_forward_planes_f32": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _I64, _I64, _I64, _I64, _F, _P]),
    "te_attention_backward_f32": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "te_attention_backward_out_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "te_attention_strided_supported": (_I, [_I64, _I64]),
    "te_attention_backward_strided_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "te_attention_forward_strided_f32": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P,
                                              _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _P]),
    "te_attention_backward_strided_f32": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64,
                                               _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64,
                                               _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_attention_backward_strided_out_f32": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64,
                                                   _P, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P,
                                                   _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_linear_relprop_x6_supported": (_I, [_I64, _I64, _I64]),
    "te_linear_relprop_x6_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "te_linear_x6_weight_planes_bytes": (_SZ, [_I64, _I64]),
    "te_linear_x6_planes_bytes": (_SZ, [_I64, _I64]),
    "te_linear_x6_prepare_weights_f32": (_I, [_P, _I64, _I64, _P, _SZ, _P]),
    "te_linear_x6_split_abs_f32": (_I, [_P, _I64, _I64, _P, _SZ, _P]),
    "te_linear_relprop_x6_f32": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P, _SZ,
                                      _P]),
    "te_linear_relprop_x6_check": (_I, [_P, _I64, _I64, _I64, _P]),
    "te_linear_relprop_x6_general_supported": (_I, [_I64, _I64, _I64, _I]),
    "te_linear_x6_weight_planes_lrp_bytes": (_SZ, [_I64, _I64]),
    "te_linear_x6_prepare_weights_lrp_f32": (_I, [_P, _I64, _I64, _P, _SZ, _P]),
    "te_linear_relprop_x6_general_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I]),
    "te_linear_relprop_x6_general_f32": (_I, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _I,
                                              _I, _P, _P, _SZ, _P]),
    "te_gemm_x6_supported": (_I, [_I64, _I64, _I64]),
    "te_gemm_x6_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "te_linear_x6_split_matrix_f32": (_I, [_P, _I64, _I64, _I, _P, _SZ, _P]),
    "te_linear_x6_split_dual_f32": (_I, [_P, _I64, _I64, _P, _P, _SZ, _P]),
    "te_gemm_x6_f32": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
    "te_layernorm_supported": (_I, [_I64]),
    "te_layernorm_forward_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "te_layernorm_backward_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "te_gelu_forward_f32": (_I, [_P, _P, _I64, _P]),
    "te_gelu_backward_f32": (_I, [_P, _P, _P, _I64, _P]),
    "te_gelu_backward_x6_planes_f32": (_I, [_P, _P, _I64, _I64, _P, _SZ, _P]),
    "te_gelu_forward_x6_planes_f32": (_I, [_P, _P, _I64, _I64, _P, _P, _SZ, _P]),
    "te_matmul_relprop_qk_fwd_f32": (_I, [_P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64,
                                          _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_matmul_relprop_qk_fwd_scaled_f32": (_I, [_P, _P, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _I64,
                                                 _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P,
                                                 _SZ, _P]),
    "te_matmul_relprop_qk_f32": (_I, [_P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64,
                                      _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P, _SZ, _P]),
    "te_add_relprop_workspace_bytes": (_SZ, [_I64, _I64]),
    "te_add_relprop_f32": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _SZ, _P]),
    "te_add_relprop_deferred_workspace_bytes": (_SZ, [_I64, _I64]),
    "te_add_relprop_deferred_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _SZ, _P]),
    "te_add_bcast_relprop_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "te_add_bcast_relprop_f32": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _SZ, _P]),
    "te_add_bcast_relprop_deferred_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _SZ, _P]),
    "te_clone_relprop_f32": (_I, [_P, _P, _P, _P, _P, _I64, _P]),
    "te_clone_relprop_scaled_f32": (_I, [_P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _P]),
    "te_index_select_relprop_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _P]),
    "te_gradcam_headmean_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "te_heatmap_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _P]),
    "te_conv2d_zb_relprop_workspace_bytes": (_SZ, [_I64] * 6),
    "te_conv2d_zb_relprop_f32": (_I, [_P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _P, _SZ,
                                      _P]),
    "te_perturb_workspace_bytes": (_SZ, [_I64, _I64]),
    "te_perturb_f32": (_I, [_P, _P, _P, _I64, _I64, _I64, _P, _I64, _P, _P, _P, _SZ, _P]),
    "te_rollout_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "te_rollout_row0_workspace_bytes": (_SZ, [_I64, _I64]),
    "te_rollout_f32": (_I, [_P, _I64, _I64, _I64, _I64, _I, _P, _P, _SZ, _P]),
}


class TeError(RuntimeError):
    pass


_lib = None


def hip_runtimes_loaded():
    """Distinct libamdhip64 images mapped into this process (must be exactly one on a GPU box)."""
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    seen.add(line.split()[-1])
    except OSError:
        pass
    return sorted(seen)


def load():
    """dlopen libte_relprop.so and attach argtypes.  Raises TeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            f"(python transformer-explainability_amd/build.py or __graft_entry__.build()); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.te_version() < MIN_LIB_VERSION:      # a stale in-tree build: argument lists changed (x6 flags / status words)
        raise TeError(f"{LIB_PATH} is version {lib.te_version()}, this package needs >= {MIN_LIB_VERSION}: rebuild it "
                      f"(python transformer-explainability_amd/build.py --force)")
    # provenance: the library must have been built from the sources of THIS tree (the prebuilt in-tree .so is what reaches the
    # GPU box; mtimes prove nothing there).  TE_RELPROP_LIB / TE_ALLOW_STALE_LIB=1: measurement builds of other sources.
    from ._buildid import source_hash
    bid = (lib.te_build_id() or b"").decode()
    want = source_hash()
    if bid.split("-")[0] != want and not (os.environ.get("TE_RELPROP_LIB") or os.environ.get("TE_ALLOW_STALE_LIB") == "1"):
        raise TeError(f"{LIB_PATH} was built from other sources (te_build_id() = {bid!r}, this tree hashes to {want!r}): "
                      f"rebuild it (python transformer-explainability_amd/build.py)")
    _lib = lib
    return lib


def build_id() -> str:
    """te_build_id() of the loaded library: '<source hash>-<flags hash>' (see _buildid.py)."""
    return (load().te_build_id() or b"").decode()


def check(status: int, what: str):
    if status != TE_OK:
        msg = load().te_status_string(status)
        raise TeError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")


def require_device():
    """Raise unless a gfx950 device is usable through the same HIP runtime torch uses."""
    if not torch.cuda.is_available():
        raise TeError("no HIP device visible to PyTorch: the relprop hot path runs only on MI355X (gfx950); "
                      "there is no CPU fallback")
    lib = load()
    rts = hip_runtimes_loaded()
    if len(rts) > 1:
        raise TeError(f"two HIP runtimes are mapped into this process: {rts}; import torch before loading "
                      f"libte_relprop so that both share torch's libamdhip64")
    check(lib.te_device_check(), "te_device_check")
