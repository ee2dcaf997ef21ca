// te_conv.hip -- Conv2d.relprop, z^B rule, for the ViT patch-embedding convolution (gfx950).
//
// Reference (modules/layers_ours.py:242-256, the `self.X.shape[1] == 3` branch; same text in modules/layers_lrp.py):
//     L = per-sample pixel minimum, H = per-sample pixel maximum (broadcast to X's shape)
//     Za = conv(X, W) - conv(L, W+) - conv(H, W-) + 1e-9 ;  S = R / Za                      (plain division)
//     out = X * convT(S, W) - L * convT(S, W+) - H * convT(S, W-)
// reached only by method="full" (ViT_LRP.py:337-343): SURVEY.md section 8(f), row 3.
//
// A patch embedding is a convolution with stride == kernel and no padding: patches do not overlap, so it is a
// Linear layer on the im2col matrix [T = B*P, K = C*p*p] and the rule has the shape of Linear.relprop:
//   Z-pass  conv(X, W) is the forward output minus the bias (cached by the rule module's forward hook), and the two
//           constant-image convolutions collapse to l_b * sum_k W+[e,k] and h_b * sum_k W-[e,k]: Za needs NO product,
//           one streaming kernel forms S (zb_s_kernel, with the NCHW -> token-major transpose done through LDS);
//   C-pass  convT(S, W) = S W+ + S W-: the two products of the Linear C-pass kernel (te_linear.hip, MODE 3) with the
//           epilogue x (P + N) - l_b P - h_b N, reading x from and writing out to the NCHW image through the patch
//           geometry -- no im2col / col2im copies.
// Rounding: Za differs from the reference's three convolutions by the order of fp32 additions only (all terms of
// sum_k (x - l) w+ + (x - h) w- are >= 0, so Za is well conditioned).
#include "te_common.h"

namespace {

constexpr int kMinMaxThreads = 1024;

// l_b, h_b: one block per sample
__global__ __launch_bounds__(kMinMaxThreads) void zb_minmax_kernel(const float* __restrict__ X, float* __restrict__ lohi,
                                                                 int64_t n) {
  __shared__ float s_lo[kMinMaxThreads / TE_WAVE], s_hi[kMinMaxThreads / TE_WAVE];
  const float* x = X + (int64_t)blockIdx.x * n;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += kMinMaxThreads) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, TE_WAVE));
    hi = fmaxf(hi, __shfl_down(hi, off, TE_WAVE));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kMinMaxThreads / TE_WAVE; ++w) {
      lo = fminf(lo, s_lo[w]);
      hi = fmaxf(hi, s_hi[w]);
    }
    lohi[2 * blockIdx.x] = lo;
    lohi[2 * blockIdx.x + 1] = hi;
  }
}

// cp[e] = sum_k max(W[e,k], 0), cn[e] = sum_k min(W[e,k], 0): one wave per output channel, fp64 accumulation
__global__ __launch_bounds__(256) void zb_wsum_kernel(const float* __restrict__ W, float* __restrict__ cpn, int64_t E,
                                                      int64_t K) {
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const int lane = threadIdx.x & 63;
  double sp = 0.0, sn = 0.0;
  for (int64_t k = lane; k < K; k += TE_WAVE) {
    const float w = W[e * K + k];
    sp += (double)fmaxf(w, 0.0f);
    sn += (double)fminf(w, 0.0f);
  }
  sp = te_wave_sum(sp);
  sn = te_wave_sum(sn);
  if (lane == 0) {
    cpn[2 * e] = (float)sp;
    cpn[2 * e + 1] = (float)sn;
  }
}

// S[t, e] = R[t, e] / ((((Y[b, e, tl] - bias[e]) - l_b cp[e]) - h_b cn[e]) + 1e-9).  Y is NCHW (tokens contiguous),
// R and S are token-major (channels contiguous): a 32x32 tile goes through LDS so both sides stay coalesced.
__global__ __launch_bounds__(256) void zb_s_kernel(const float* __restrict__ R, const float* __restrict__ Y,
                                                   const float* __restrict__ bias, const float* __restrict__ lohi,
                                                   const float* __restrict__ cpn, float* __restrict__ S, int64_t P,
                                                   int64_t E, int64_t r_bs) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int64_t t0 = (int64_t)blockIdx.x * 32, e0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float lo = lohi[2 * b], hi = lohi[2 * b + 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = e0 + ty + 8 * i, t = t0 + tx;
    if (e < E && t < P) {
      const float lin = Y[(b * E + e) * P + t] - (bias ? bias[e] : 0.0f);
      tile[ty + 8 * i][tx] = ((lin - lo * cpn[2 * e]) - hi * cpn[2 * e + 1]) + 1e-9f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t t = t0 + ty + 8 * i, e = e0 + tx;
    if (e < E && t < P) S[(b * P + t) * E + e] = R[b * r_bs + t * E + e] / tile[tx][ty + 8 * i];
  }
}

// any-shape C-pass: one thread per image element, e-ordered fmaf chains (cross-check of the tiled MODE 3 kernel)
__global__ __launch_bounds__(256) void zb_cpass_simple(const float* __restrict__ S, const float* __restrict__ W,
                                                       const float* __restrict__ X, float* __restrict__ out, int64_t T,
                                                       int64_t K, int64_t E, TeZbGeom zb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * K) return;
  const int64_t t = i / K, k = i - t * K;
  float pp = 0.0f, pn = 0.0f;
  for (int64_t e = 0; e < E; ++e) {
    const float s = S[t * E + e], w = W[e * K + k];
    pp = fmaf(s, fmaxf(w, 0.0f), pp);
    pn = fmaf(s, fminf(w, 0.0f), pn);
  }
  const int64_t b = t / zb.P, at = te_zb_index(zb, t, k);
  out[at] = (X[at] * (pp + pn) - zb.lohi[2 * b] * pp) - zb.lohi[2 * b + 1] * pn;
}

struct Layout {
  size_t s_off, lohi_off, cpn_off, total;
};
inline Layout ws_layout(int64_t B, int64_t P, int64_t E) {
  Layout l;
  l.s_off = 0;
  l.lohi_off = te_align_up((size_t)B * P * E * sizeof(float), 256);
  l.cpn_off = l.lohi_off + te_align_up((size_t)B * 2 * sizeof(float), 256);
  l.total = l.cpn_off + te_align_up((size_t)E * 2 * sizeof(float), 256);
  return l;
}

}  // namespace

extern "C" size_t te_conv2d_zb_relprop_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int64_t E,
                                                       int64_t p) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || E <= 0 || p <= 0 || H % p || W % p) return 0;
  return ws_layout(B, (H / p) * (W / p), E).total;
}

extern "C" int te_conv2d_zb_relprop_f32(const float* R, int64_t r_bs, const float* X, const float* Wt, const float* Y,
                                        const float* bias, float* out, int64_t B, int64_t C, int64_t H, int64_t W,
                                        int64_t E, int64_t p, int flags, void* ws, size_t ws_bytes,
                                        te_stream_t stream_) {
  if (!R || !X || !Wt || !Y || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || E <= 0 || p <= 0)
    return TE_ERR_INVALID_ARG;
  if (H % p || W % p) return TE_ERR_UNSUPPORTED;          // stride == kernel, no padding: whole patches only
  const int64_t Hp = H / p, Wp = W / p, P = Hp * Wp, K = C * p * p, T = B * P;
  if (r_bs < P * E) return TE_ERR_INVALID_ARG;
  if (H * W * C > INT32_MAX || K > INT32_MAX) return TE_ERR_UNSUPPORTED;
  const Layout l = ws_layout(B, P, E);
  if (!ws || ws_bytes < l.total || !te_aligned16(ws)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* S = (float*)((char*)ws + l.s_off);
  float* lohi = (float*)((char*)ws + l.lohi_off);
  float* cpn = (float*)((char*)ws + l.cpn_off);

  zb_minmax_kernel<<<dim3((unsigned)B), dim3(kMinMaxThreads), 0, stream>>>(X, lohi, C * H * W);
  zb_wsum_kernel<<<dim3((unsigned)te_ceil_div(E, 4)), dim3(256), 0, stream>>>(Wt, cpn, E, K);
  zb_s_kernel<<<dim3((unsigned)te_ceil_div(P, 32), (unsigned)te_ceil_div(E, 32), (unsigned)B), dim3(256), 0, stream>>>(
      R, Y, bias, lohi, cpn, S, P, E, r_bs);
  TeZbGeom zb;
  zb.lohi = lohi;
  zb.P = P;
  zb.C = (int)C;
  zb.H = (int)H;
  zb.W = (int)W;
  zb.p = (int)p;
  zb.Wp = (int)Wp;
  const bool simple = (flags & TE_IMPL_SIMPLE) != 0;
  if (simple || !te_internal_zb_cpass_tiled(S, Wt, X, out, T, K, E, zb, stream))
    zb_cpass_simple<<<dim3((unsigned)te_ceil_div(T * K, 256)), dim3(256), 0, stream>>>(S, Wt, X, out, T, K, E, zb);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
