// te_attn_mfma.hip -- LDS-tiled fp32-MFMA kernels for the attention einsum/MatMul relprop rules.
// (placeholder: the tiled kernels are not wired yet; the API falls back to the simple kernels of
// te_attn.hip while *_supported() returns false.)
#include "te_common.h"

namespace te_attn_mfma {

bool av_supported(int64_t, int64_t) { return false; }
bool qk_supported(int64_t, int64_t) { return false; }

int av_launch(const float*, int64_t, int64_t, int64_t, const float*, const float*, int64_t, int64_t, int64_t,
              float*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float, float*,
              hipStream_t) {
  return TE_ERR_UNSUPPORTED;
}
int qk_launch(const float*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, int64_t, int64_t,
              float*, int64_t, int64_t, int64_t, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
              int64_t, float, float*, hipStream_t) {
  return TE_ERR_UNSUPPORTED;
}

}  // namespace te_attn_mfma
