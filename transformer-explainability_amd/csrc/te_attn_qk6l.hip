// te_attn_qk6l.hip -- the QK relprop rule (modules/layers_ours.py:48-60 via ViT_LRP.py:165-173, BERT.py:389-393) for LONG
// sequences on bf16 MFMAs (round 6; VERDICT r5 item 3 "and for the long-N path"), 64 < N <= 640, head dim 64, any strides:
//
//   S = sd(R_nn * f, Z_qk) [N,N]      cam_q = q .(S k) * scale      cam_k = k .(S^T q) * scale
//
// te_attn_rules.hip's qk_rule_kernel keeps a [32, keys] tile of S in LDS between barriers and sums cam_q over key groups in a
// second launch (fp32 MFMAs: 0.20 of the HBM roofline at N = 577).  Here the two products are the two halves of te_attn_bwd6l.hip
// with S in the place of the softmax backward's d_s:
//
//   rows   a wave owns 32 query rows and walks the keys a chunk at a time: the R and Z blocks come in through the wave's tile
//          (128-byte runs of eight rows per load instruction), S is formed in the accumulator layout, split into planes per K16
//          step and meets the chunk's k^T planes (LDS): cam_q^T accumulates in registers -- one k-ordered chain per output;
//   cols   a wave owns 32 keys and walks the query rows: a lane (key, h) loads rows 8 h + (0..7) of its column of R and Z (a
//          half-wave reads 128 contiguous bytes of a row), forms S, splits it, and meets the chunk's q^T planes (LDS).
//
// S = sd(R, Z) is evaluated once per side (te_attn_rc.hip does the same at N <= 224); the second read of R / Z comes from the
// Infinity Cache where it still holds them.  Every reduction has a fixed order that depends on N only.
#include <stdlib.h>
#include <string.h>

#include "te_attn_l6.h"

namespace te_attn_qk6l {

namespace {

using namespace te_attn_l6;

template <int W>
__global__ __launch_bounds__(64 * W, 2) void qk6l_rows_kernel(const float* __restrict__ Rnn, const float* __restrict__ Z,
                                                              const float* __restrict__ q, Strided qs, const float* __restrict__ k,
                                                              Strided ks, float* __restrict__ cam_q, Strided cqs, int H, int N, int BH,
                                                              int G, int RB, float scale, const float* __restrict__ r_scale,
                                                              int64_t r_scale_stride) {
  typedef Cfg<W> C;
  constexpr int kKC = C::kKC, NKB = C::kNKB, kBuf = C::kBuf, kPlane = C::kPlane;
  extern __shared__ __attribute__((aligned(16))) unsigned char Pl[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / G) * 8 + xcd, part = slot % G;
  if (bh >= BH) return;
  const int b = bh / H, h = bh - b * H;
  const int NBr = (N + 31) >> 5, NC = (N + kKC - 1) / kKC;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  const int blk = part * RB + wave;
  const bool owner = wave < RB && blk < NBr;
  const int i = blk * 32 + n, i0 = blk * 32;
  const float* k_bh = k + b * ks.sb + h * ks.sh;
  const unsigned nn_bytes = (unsigned)(N * N * 4);
  const Rsrc r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rnn) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const Rsrc z_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  float* const tile = reinterpret_cast<float*>(Pl + C::kTileOff) + wave * (32 * kTileLd);
  const bool has_f = r_scale != nullptr;
  const float f = has_f ? r_scale[(int64_t)b * r_scale_stride] : 1.0f;      // deferred per-sample factor of the mask Add (BERT.py:386-388)

  VReq kr;
  request_v<W>(kr, k_bh, ks.sn, N, 0);
  write_v<W>(Pl, kr, N, 0);
  __syncthreads();

  const unsigned char* const lane_frag = Pl + lane * 16;
  f32x16 o[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[mb][e] = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + (c & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    if (!last) request_v<W>(kr, k_bh, ks.sn, N, c + 1);
    if (owner) {
      f32x4 rv[NKB][4], zv[NKB][4];
#pragma unroll
      for (int u = 0; u < NKB; ++u) {
        block_in_request(rv[u], r_rs, i0, kKC * c + 32 * u, N);
        block_in_request(zv[u], z_rs, i0, kKC * c + 32 * u, N);
      }
      f32x16 S[NKB];
#pragma unroll
      for (int u = 0; u < NKB; ++u) {
        f32x16 r16, z16;
        block_in_land(tile, rv[u], r16);
        block_in_land(tile, zv[u], z16);
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          f32x2 r = {r16[2 * e2], r16[2 * e2 + 1]};
          if (has_f) r = r * f32x2{f, f};
          const f32x2 sv = sd2(r, f32x2{z16[2 * e2], z16[2 * e2 + 1]});
          S[u][2 * e2] = sv[0], S[u][2 * e2 + 1] = sv[1];
        }
        if (last) {
          // keys at or beyond N: the loads wrapped into the next row (finite values in general, but nothing bounds them): zero
#pragma unroll
          for (int e = 0; e < 16; ++e) S[u][e] = (kKC * c + 32 * u + 8 * (e >> 2) + 4 * kh + (e & 3) < N) ? S[u][e] : 0.0f;
        }
      }
#pragma unroll
      for (int s = 0; s < 2 * NKB; ++s) {
        const int u = s >> 1, g0 = 2 * (s & 1);
        const float x[8] = {S[u][4 * g0],     S[u][4 * g0 + 1], S[u][4 * g0 + 2], S[u][4 * g0 + 3],
                            S[u][4 * g0 + 4], S[u][4 * g0 + 5], S[u][4 * g0 + 6], S[u][4 * g0 + 7]};
        bf16x8 pb[3];
        planes_of8(x, pb);
        bf16x8 a[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(buf + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) o[mb] = TE_MFMA_BF16(a[mb][PA[p6]], pb[PB[p6]], o[mb]);
      }
    }
    if (!last) {
      write_v<W>(nbuf, kr, N, c + 1);
      __syncthreads();
    }
  }
  if (owner && i < N) {
    const float* qrow = q + b * qs.sb + h * qs.sh + (int64_t)i * qs.sn + 4 * kh;
    float* orow = cam_q + b * cqs.sb + h * cqs.sh + (int64_t)i * cqs.sn + 4 * kh;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 x4 = *reinterpret_cast<const f32x4_u*>(qrow + 32 * mb + 8 * g);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (x4[e] * o[mb][4 * g + e]) * scale;
        *reinterpret_cast<f32x4_u*>(orow + 32 * mb + 8 * g) = r;
      }
  }
}

template <int W>
__global__ __launch_bounds__(64 * W, 2) void qk6l_cols_kernel(const float* __restrict__ Rnn, const float* __restrict__ Z,
                                                              const float* __restrict__ q, Strided qs, const float* __restrict__ k,
                                                              Strided ks, float* __restrict__ cam_k, Strided cks, int H, int N, int BH,
                                                              int G, int RB, float scale, const float* __restrict__ r_scale,
                                                              int64_t r_scale_stride) {
  typedef Cfg<W> C;
  constexpr int kKC = C::kKC, NS = 2 * C::kNKB, kBuf = C::kBuf, kPlane = C::kPlane;
  extern __shared__ __attribute__((aligned(16))) unsigned char Pl[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / G) * 8 + xcd, part = slot % G;
  if (bh >= BH) return;
  const int b = bh / H, h = bh - b * H;
  const int NBr = (N + 31) >> 5, NC = (N + kKC - 1) / kKC;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  const int blk = part * RB + wave;
  const bool owner = wave < RB && blk < NBr;
  const int j = blk * 32 + n;                        // this lane's key
  const float* q_bh = q + b * qs.sb + h * qs.sh;
  const unsigned nn_bytes = (unsigned)(N * N * 4);
  const Rsrc r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rnn) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const Rsrc z_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const bool has_f = r_scale != nullptr;
  const float f = has_f ? r_scale[(int64_t)b * r_scale_stride] : 1.0f;

  VReq qr;
  request_v<W, false>(qr, q_bh, qs.sn, N, 0);
  // the column panels: set s = the K16 step s of the current chunk (rows kKC c + 16 s + 8 h + t of this lane's key column; rows at
  // or beyond N lie past the end of the views: 0)
  float ra[NS][8], za[NS][8];
  const unsigned row_bytes = (unsigned)N * 4u;
  const unsigned col0 = ((unsigned)(8 * kh) * (unsigned)N + (unsigned)j) * 4u;
  auto request_set = [&](int c, int s) __attribute__((always_inline)) {
    const unsigned off = col0 + (unsigned)(kKC * c + 16 * s) * row_bytes;
#pragma unroll
    for (int t = 0; t < 8; ++t) ra[s][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, off + (unsigned)t * row_bytes, 0, 0));
#pragma unroll
    for (int t = 0; t < 8; ++t) za[s][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rs, off + (unsigned)t * row_bytes, 0, 0));
  };
  if (owner) {
#pragma unroll
    for (int s = 0; s < NS; ++s) request_set(0, s);
  }
  write_v<W, false>(Pl, qr, N, 0);
  __syncthreads();

  const unsigned char* const lane_frag = Pl + lane * 16;
  f32x16 ak[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) ak[mb][e] = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + (c & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    if (!last) request_v<W, false>(qr, q_bh, qs.sn, N, c + 1);
    if (owner) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float x[8];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
          f32x2 r = {ra[s][2 * t2], ra[s][2 * t2 + 1]};
          if (has_f) r = r * f32x2{f, f};
          const f32x2 sv = sd2(r, f32x2{za[s][2 * t2], za[s][2 * t2 + 1]});
          x[2 * t2] = sv[0], x[2 * t2 + 1] = sv[1];
        }
        // (a lane whose key is at or beyond N reads finite or garbage values that reach its own, never stored, column only)
        bf16x8 sb[3];
        planes_of8(x, sb);
        request_set(c + 1, s);             // (behind the last chunk: rows beyond N, zeros nobody reads)
        bf16x8 a[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(buf + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) ak[mb] = TE_MFMA_BF16(a[mb][PA[p6]], sb[PB[p6]], ak[mb]);
      }
    }
    if (!last) {
      write_v<W, false>(nbuf, qr, N, c + 1);
      __syncthreads();
    }
  }
  if (owner && j < N) {
    const float* krow = k + b * ks.sb + h * ks.sh + (int64_t)j * ks.sn + 4 * kh;
    float* orow = cam_k + b * cks.sb + h * cks.sh + (int64_t)j * cks.sn + 4 * kh;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 x4 = *reinterpret_cast<const f32x4_u*>(krow + 32 * mb + 8 * g);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (x4[e] * ak[mb][4 * g + e]) * scale;
        *reinterpret_cast<f32x4_u*>(orow + 32 * mb + 8 * g) = r;
      }
  }
}

template <int W>
int launch_w(const float* Rnn, const float* Z, const float* q, Strided qs, const float* k, Strided ks, float* cam_q, Strided cqs, float* cam_k,
             Strided cks, int64_t B, int64_t H, int64_t N, float scale, const float* r_scale, int64_t r_scale_stride, hipStream_t stream) {
  const int NBr = (int)((N + 31) >> 5);
  const int G = (NBr + W - 1) / W, RB = (NBr + G - 1) / G;
  const int64_t BH = B * H, slots = ((BH + 7) / 8) * G;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qk6l_rows_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<W>::kLds);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(qk6l_cols_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<W>::kLds);
  if (e != hipSuccess) return (int)e;
  qk6l_rows_kernel<W><<<dim3((unsigned)(slots * 8)), dim3(64 * W), Cfg<W>::kLds, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, (int)H, (int)N, (int)BH, G, RB,
                                                                                          scale, r_scale, r_scale_stride);
  qk6l_cols_kernel<W><<<dim3((unsigned)(slots * 8)), dim3(64 * W), Cfg<W>::kLds, stream>>>(Rnn, Z, q, qs, k, ks, cam_k, cks, (int)H, (int)N, (int)BH, G, RB,
                                                                                          scale, r_scale, r_scale_stride);
  return TE_OK;
}

inline bool ok4(int64_t sb, int64_t sh, int64_t sn) { return sb >= 0 && sh >= 0 && sn >= 64 && (sn % 4) == 0 && (sh % 4) == 0 && (sb % 4) == 0; }

}  // namespace

// (strides in multiples of four floats: q / k / cam rows move as 16-byte pieces)
bool supported(int64_t B, int64_t H, int64_t N, int64_t D, int64_t q_sb, int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
               int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn) {
  return D == 64 && N > 64 && N <= kMaxN && B >= 1 && H >= 1 && B * H <= (1 << 24) && ok4(q_sb, q_sh, q_sn) && ok4(k_sb, k_sh, k_sn) &&
         ok4(cq_sb, cq_sh, cq_sn) && ok4(ck_sb, ck_sh, ck_sn);
}

int launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
           const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
           int64_t B, int64_t H, int64_t N, float scale, const float* r_scale, int64_t r_scale_stride, hipStream_t stream) {
  const te_attn_l6::Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
  const int NBr = (int)((N + 31) >> 5), G8 = (NBr + 7) / 8, G4 = (NBr + 3) / 4;
  bool w8 = 5 * G8 * 8 <= 6 * G4 * 4;            // (te_attn_fwd6l.hip: launch)
#ifdef TE_STUDY
  static const int wenv = [] { const char* e = getenv("TE_QK6L_WAVES"); return e ? atoi(e) : 0; }();
  if (wenv == 8) w8 = true;
  if (wenv == 4) w8 = false;
#endif
  return w8 ? launch_w<8>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks, B, H, N, scale, r_scale, r_scale_stride, stream)
            : launch_w<4>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks, B, H, N, scale, r_scale, r_scale_stride, stream);
}

}  // namespace te_attn_qk6l
