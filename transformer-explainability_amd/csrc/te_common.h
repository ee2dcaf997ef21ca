// te_common.h -- shared device helpers for libte_relprop (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "te_relprop.h"

#define TE_WAVE 64

// Launch-check: kernels are enqueued asynchronously; only launch-configuration errors surface here.
#define TE_RETURN_IF_LAUNCH_FAILED()            \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// nn.GELU, exact erf form (modules/layers_ours.py:70, ViT_LRP.py:57) and its derivative times an incoming gradient: ONE
// definition for the stand-alone producers (te_norm_act.hip) and for the producers that emit operand planes instead of
// fp32 (te_linear_x6.hip), so that both give the same bits.
constexpr float kTeInvSqrt2 = 0.70710678118654752440f;
constexpr float kTeInvSqrt2Pi = 0.39894228040143267794f;      // 1 / sqrt(2 pi)
__device__ __forceinline__ float te_gelu(float v) { return (v * 0.5f) * (1.0f + erff(v * kTeInvSqrt2)); }
__device__ __forceinline__ float te_gelu_grad(float g, float v) {
  const float cdf = 0.5f * (1.0f + erff(v * kTeInvSqrt2));
  const float pdf = expf(-0.5f * (v * v)) * kTeInvSqrt2Pi;
  return g * (cdf + v * pdf);
}

// safe_divide of the reference (modules/layers_ours.py:10-13), evaluated exactly as the reference
// does in fp32: den = b + 1e-9 (one rounding), an exact-zero den is replaced by 1e-9, IEEE
// division, then a multiplication by the 0/1 mask (b != 0).  Compiled with -ffp-contract=off.
__device__ __forceinline__ float te_sd(float a, float b) {
  float den = b + 1e-9f;
  den = (den == 0.0f) ? 1e-9f : den;
  float q = a / den;
  return q * ((b != 0.0f) ? 1.0f : 0.0f);
}

__device__ __forceinline__ double te_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, TE_WAVE);
  return v;
}

// Block-wide sum of up to 3 doubles; result valid in thread 0.  `smem` holds 3*(blockDim/64) doubles.
__device__ __forceinline__ void te_block_sum3(double& a, double& b, double& c, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  a = te_wave_sum(a);
  b = te_wave_sum(b);
  c = te_wave_sum(c);
  if (lane == 0) {
    smem[wave * 3 + 0] = a;
    smem[wave * 3 + 1] = b;
    smem[wave * 3 + 2] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0, sc = 0;
    for (int w = 0; w < nw; ++w) {
      sa += smem[w * 3 + 0];
      sb += smem[w * 3 + 1];
      sc += smem[w * 3 + 2];
    }
    a = sa;
    b = sb;
    c = sc;
  }
}

// Patch geometry of a stride == kernel convolution (ViT patch embedding): row t = (b, py, px) of the im2col matrix,
// column k = (c, dy, dx); te_zb_index maps (t, k) to the element's offset in the NCHW image.
struct TeZbGeom {
  const float* lohi;    // [B][2] per-sample pixel min / max
  int64_t P;            // patches per sample = Hp * Wp
  int C, H, W, p, Wp;
};
__host__ __device__ __forceinline__ int64_t te_zb_index(const TeZbGeom& g, int64_t t, int64_t k) {
  const int64_t b = t / g.P;
  const int tl = (int)(t - b * g.P), py = tl / g.Wp, px = tl - py * g.Wp;
  const int pp = g.p * g.p, c = (int)(k / pp), rem = (int)(k - (int64_t)c * pp), dy = rem / g.p, dx = rem - dy * g.p;
  return ((b * g.C + c) * g.H + (int64_t)py * g.p + dy) * g.W + (int64_t)px * g.p + dx;
}
bool te_internal_zb_cpass_tiled(const float* S, const float* W, const float* X, float* out, int64_t T, int64_t in_f,
                                int64_t out_f, const TeZbGeom& zb, hipStream_t stream);

static inline bool te_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t te_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t te_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
