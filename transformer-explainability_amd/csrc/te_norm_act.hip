// te_norm_act.hip -- producers of the relprop path's inputs, second slice: the LayerNorm and GELU
// layers of the LRP-instrumented models (modules/layers_ours.py:70-77; used at baselines/ViT/ViT_LRP.py:57,184,187,266
// and BERT_explainability/modules/BERT/BERT.py:18,52,416,463), forward and input-gradient backward, as streaming
// kernels for gfx950.  Their relprop rules are the identity; what the path needs from them is the forward value (the
// X / Y every Linear / Add / Clone rule caches) and the gradient that flows on to the attention maps.
//
//   LayerNorm forward   y = (x - mean) * rstd * w + b,  mean / rstd [T] kept for the backward
//   LayerNorm backward  dx = rstd * (a - mean(a) - xhat * mean(a * xhat)) [+ add],  a = dy * w, xhat = (x - mean) * rstd
//                       `add` (optional) is the gradient of the residual branch that bypasses the LayerNorm
//                       (x1, x2 = clone(x); ... add([x1, f(norm(x2))]), ViT_LRP.py:203-205): autograd would run a
//                       separate [T,C] addition kernel for it
//   GELU forward        y = 0.5 x (1 + erf(x / sqrt 2))                       (nn.GELU default, exact erf form)
//   GELU backward       dx = dy (0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi))
//
// All HBM-bound: one wave per LayerNorm row (C <= 2048, a multiple of 4: the row lives in <= 8 float4 registers per
// lane; mean by a fixed-order butterfly, variance as the mean of squared deviations from it -- two passes over
// registers, not E[x^2] - mean^2), one float4 per thread for GELU.  No atomics: a row / an element is owned by one wave /
// thread, so a batch equals its samples run one by one, bit for bit.
#include "te_common.h"

namespace {

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kThreads = 256;
constexpr int kRowsPerBlock = kThreads / 64;
constexpr int kMaxChunks = 8;      // float4 per lane: C <= 64 * 4 * 8 = 2048

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);      // butterfly: every lane ends with the sum
  return v;
}

template <int NV>
__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int64_t T,
                                                          int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nc = C >> 2;
  const float* xr = x + row * C;
  f32x4 v[NV];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nc) v[i] = *reinterpret_cast<const f32x4*>(xr + (c << 2));
    s = s + ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3]));
  }
  const float m = wave_sum_all(s) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nc) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - m;
        q = q + d * d;
      }
    }
  }
  const float r = 1.0f / sqrtf(wave_sum_all(q) / (float)C + eps);
  float* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nc) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (c << 2));
      const f32x4 bv = b ? *reinterpret_cast<const f32x4*>(b + (c << 2)) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ((v[i][e] - m) * r) * wv[e] + bv[e];
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yr + (c << 2)));
    }
  }
  if (lane == 0) {
    mean[row] = m;
    rstd[row] = r;
  }
}

template <int NV>
__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ add,
                                                          float* __restrict__ dx, int64_t T, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nc = C >> 2;
  const float m = mean[row], r = rstd[row];
  const float* xr = x + row * C;
  const float* gr = dy + row * C;
  f32x4 a[NV], xh[NV];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nc) {
      const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gr + (c << 2)));
      const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (c << 2)));
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (c << 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[i][e] = g[e] * wv[e];
        xh[i][e] = (xv[e] - m) * r;
        s1 = s1 + a[i][e];
        s2 = s2 + a[i][e] * xh[i][e];
      }
    }
  }
  const float c1 = wave_sum_all(s1) / (float)C, c2 = wave_sum_all(s2) / (float)C;
  float* dr = dx + row * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nc) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = r * ((a[i][e] - c1) - xh[i][e] * c2);
      if (add) {
        const f32x4 av = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(add + row * C + (c << 2)));
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = av[e] + o[e];
      }
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dr + (c << 2)));
    }
  }
}

__global__ __launch_bounds__(kThreads) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = te_gelu(v[e]);
  __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y) + i);
}

__global__ __launch_bounds__(kThreads) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ dx, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy) + i);
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = te_gelu_grad(g[e], v[e]);
  __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dx) + i);
}

template <typename F>
inline int for_chunks(int64_t C, F&& launch) {
  const int nv = (int)((C / 4 + 63) / 64);
  switch (nv) {
    case 1: launch(std::integral_constant<int, 1>{}); break;
    case 2: launch(std::integral_constant<int, 2>{}); break;
    case 3: launch(std::integral_constant<int, 3>{}); break;
    case 4: launch(std::integral_constant<int, 4>{}); break;
    case 5: case 6: launch(std::integral_constant<int, 6>{}); break;
    case 7: case 8: launch(std::integral_constant<int, 8>{}); break;
    default: return TE_ERR_UNSUPPORTED;
  }
  return TE_OK;
}

}  // namespace

extern "C" int te_layernorm_supported(int64_t C) { return (C >= 4 && C % 4 == 0 && C <= 64 * 4 * kMaxChunks) ? 1 : 0; }

extern "C" int te_layernorm_forward_f32(const float* x, const float* weight, const float* bias, float* y, float* mean,
                                        float* rstd, int64_t T, int64_t C, float eps, te_stream_t stream_) {
  if (!x || !weight || !y || !mean || !rstd || T <= 0) return TE_ERR_INVALID_ARG;
  if (!te_layernorm_supported(C) || (T + kRowsPerBlock - 1) / kRowsPerBlock > 0x7fffffff) return TE_ERR_UNSUPPORTED;
  if (!te_aligned16(x) || !te_aligned16(weight) || !te_aligned16(y) || (bias && !te_aligned16(bias))) return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((unsigned)((T + kRowsPerBlock - 1) / kRowsPerBlock)), blk(kThreads);
  const int rc = for_chunks(C, [&](auto nv) {
    ln_fwd_kernel<decltype(nv)::value><<<grid, blk, 0, stream>>>(x, weight, bias, y, mean, rstd, T, (int)C, eps);
  });
  if (rc != TE_OK) return rc;
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_layernorm_backward_f32(const float* dy, const float* x, const float* weight, const float* mean,
                                         const float* rstd, const float* add, float* dx, int64_t T, int64_t C,
                                         te_stream_t stream_) {
  if (!dy || !x || !weight || !mean || !rstd || !dx || T <= 0) return TE_ERR_INVALID_ARG;
  if (!te_layernorm_supported(C) || (T + kRowsPerBlock - 1) / kRowsPerBlock > 0x7fffffff) return TE_ERR_UNSUPPORTED;
  if (!te_aligned16(dy) || !te_aligned16(x) || !te_aligned16(weight) || !te_aligned16(dx) || (add && !te_aligned16(add)))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((unsigned)((T + kRowsPerBlock - 1) / kRowsPerBlock)), blk(kThreads);
  const int rc = for_chunks(C, [&](auto nv) {
    ln_bwd_kernel<decltype(nv)::value><<<grid, blk, 0, stream>>>(dy, x, weight, mean, rstd, add, dx, T, (int)C);
  });
  if (rc != TE_OK) return rc;
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_gelu_forward_f32(const float* x, float* y, int64_t n, te_stream_t stream_) {
  if (!x || !y || n <= 0) return TE_ERR_INVALID_ARG;
  if (n % 4 || !te_aligned16(x) || !te_aligned16(y) || n / 4 / kThreads > 0x7ffffffe) return TE_ERR_UNSUPPORTED;
  const int64_t n4 = n / 4;
  gelu_fwd_kernel<<<dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream_>>>(x, y, n4);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_gelu_backward_f32(const float* dy, const float* x, float* dx, int64_t n, te_stream_t stream_) {
  if (!dy || !x || !dx || n <= 0) return TE_ERR_INVALID_ARG;
  if (n % 4 || !te_aligned16(dy) || !te_aligned16(x) || !te_aligned16(dx) || n / 4 / kThreads > 0x7ffffffe) return TE_ERR_UNSUPPORTED;
  const int64_t n4 = n / 4;
  gelu_bwd_kernel<<<dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream_>>>(dy, x, dx, n4);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
