// te_attn_bwd6l.hip -- the ROW side of the attention-gradient backward pass (the gradient hook of
// baselines/ViT/ViT_LRP.py:144-145, BERT.py:349-350) for LONG sequences on bf16 MFMAs with row-block owners (round 6; VERDICT r5
// item 3 "and for the long-N path"), 64 < N <= 640, head dim 64, any [B,H,N,64] strides:
//
//   d_attn [BH,N,N] = d_out v^T                      (the tensor save_attn_gradients receives)
//   rowdot [BH,N]   = sum_j d_attn . attn = d_out . out      (out = attn v, d_attn = d_out v^T: 64 products per row from the block's
//                                                             own forward output instead of a pass over two N x N rows)
//   d_s             = ((d_attn - rowdot) . attn) * scale      (softmax backward, then '* scale'; never stored)
//   d_q             = d_s k
//
// te_attn_long.hip's attn_bwd_rows_kernel (round 3) keeps the [32, N] panel of d_attn in LDS between eleven barriers per 32 rows
// (fp32 MFMAs: 0.17 of the HBM roofline at N = 577).  Here -- the structure of te_attn_fwd6l.hip -- a wave owns 32 query rows for the
// whole kernel and walks the keys ONCE, a chunk of 8 W keys at a time: d_attn^T of the chunk from the v planes (A, key-major, LDS)
// and the wave's d_out rows (B, registers); the block leaves through the wave's tile; the attn block comes in the same way
// (requested before the products); d_s is formed in the accumulator layout, split into planes per K16 step and meets the chunk's
// k^T planes (A, LDS): d_q^T accumulates in registers.  One barrier per chunk, the next chunk's v / k rows in flight meanwhile.
// The column side (d_v, d_k: bwd6l_cols_kernel below) reads rowdot from the workspace this kernel fills.
//
// Every reduction has a fixed order that depends on N only: a batch equals its samples run one by one, bit for bit.
#include <stdlib.h>

#include "te_attn_l6.h"

namespace te_attn_bwd6l {

namespace {

using namespace te_attn_l6;

template <int W>
__global__ __launch_bounds__(64 * W, 2) void bwd6l_rows_kernel(const float* __restrict__ dout, Strided dos, const float* __restrict__ outp,
                                                               Strided os, const float* __restrict__ k, Strided ks,
                                                               const float* __restrict__ v, Strided vs, const float* __restrict__ attn,
                                                               float* __restrict__ dattn, float* __restrict__ rowdot,
                                                               float* __restrict__ dq, Strided dqs, int H, int N, int BH, int G, int RB,
                                                               float scale, int need_qk) {
  typedef Cfg<W> C;
  constexpr int kKC = C::kKC, NKB = C::kNKB, kBuf = C::kBuf, kOperand = C::kOperand, kPlane = C::kPlane;
  extern __shared__ __attribute__((aligned(16))) unsigned char Pl[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / G) * 8 + xcd, part = slot % G;
  if (bh >= BH) return;
  const int b = bh / H, h = bh - b * H;
  const int NBr = (N + 31) >> 5, NC = (N + kKC - 1) / kKC;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  const int blk = part * RB + wave;
  const bool owner = wave < RB && blk < NBr;
  const int i = blk * 32 + n;                        // this lane's query row
  const bool row_ok = owner && i < N;
  const float* g_bh = dout + b * dos.sb + h * dos.sh;
  const float* k_bh = k + b * ks.sb + h * ks.sh;
  const float* v_bh = v + b * vs.sb + h * vs.sh;
  const unsigned nn_bytes = (unsigned)(N * N * 4);
  const Rsrc d_rs = __builtin_amdgcn_make_buffer_rsrc(dattn + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const Rsrc a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attn) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  float* const tile = reinterpret_cast<float*>(Pl + C::kTileOff) + wave * (32 * kTileLd);

  // ---- chunk 0 of v (key-major planes) and of k (head-dimension-major planes); the wave's d_out rows as B planes; rowdot ----
  KReq vr;           // (the v rows travel as te_attn_fwd6l.hip's k rows do, and the other way round)
  VReq kr;
  request_k<W>(vr, v_bh, vs.sn, N, 0);
  if (need_qk) request_v<W>(kr, k_bh, ks.sn, N, 0);
  bf16x8 gb[4][3];
  float rd = 0.0f;
  {
    f32x4 gv[4][2], ov[4][2];
    const int ic = min(i, N - 1);
    const float* gr = g_bh + (int64_t)ic * dos.sn + 8 * kh;
    const float* orow = outp + b * os.sb + h * os.sh + (int64_t)ic * os.sn + 8 * kh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      gv[s][0] = *reinterpret_cast<const f32x4_u*>(gr + 16 * s);
      gv[s][1] = *reinterpret_cast<const f32x4_u*>(gr + 16 * s + 4);
      if (need_qk) {
        ov[s][0] = *reinterpret_cast<const f32x4_u*>(orow + 16 * s);
        ov[s][1] = *reinterpret_cast<const f32x4_u*>(orow + 16 * s + 4);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = row_ok ? gv[s][0][e] : 0.0f, x[4 + e] = row_ok ? gv[s][1][e] : 0.0f;
      planes_of8(x, gb[s]);
      if (need_qk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rd = rd + x[e] * ov[s][0][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) rd = rd + x[4 + e] * ov[s][1][e];
      }
    }
    if (need_qk) {
      rd = rd + __shfl_xor(rd, 32, 64);                  // (a + b = b + a: both lanes of a row hold the same bits)
      if (row_ok && kh == 0) rowdot[(int64_t)bh * N + i] = rd;
    }
  }
  write_k<W>(Pl, vr, N, 0);
  if (need_qk) write_v<W>(Pl + kOperand, kr, N, 0);
  __syncthreads();

  const RowOff ro = make_rowoff(blk * 32, N);
  const unsigned char* const lane_frag = Pl + lane * 16;
  f32x16 acc[NKB];
  f32x16 o[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[mb][e] = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + (c & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    if (!last) {
      request_k<W>(vr, v_bh, vs.sn, N, c + 1);
      if (need_qk) request_v<W>(kr, k_bh, ks.sn, N, c + 1);
    }
    if (owner) {
      f32x4 av[NKB][4];
      if (need_qk) {
#pragma unroll
        for (int u = 0; u < NKB; ++u) block_in_request(av[u], a_rs, ro, kKC * c + 32 * u);
      }
      scores<W>(acc, buf, gb);                            // d_attn^T of the chunk: acc[u][4 g + e] <-> key kKC c + 32 u + 8 g + 4 h + e of row i
      if (last) {
#pragma unroll
        for (int u = 0; u < NKB; ++u) block_out<true>(tile, acc[u], d_rs, ro, kKC * c + 32 * u, N);
      } else {
#pragma unroll
        for (int u = 0; u < NKB; ++u) block_out<false>(tile, acc[u], d_rs, ro, kKC * c + 32 * u, N);
      }
      if (need_qk) {
#pragma unroll
        for (int u = 0; u < NKB; ++u) {
          f32x16 p;
          block_in_land(tile, av[u], p);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[u][e] = ((acc[u][e] - rd) * p[e]) * scale;      // softmax backward, then '* scale'
        }
        const unsigned char* const kfrag = buf + kOperand;
#pragma unroll
        for (int s = 0; s < 2 * NKB; ++s) {
          const int u = s >> 1, g0 = 2 * (s & 1);
          const float x[8] = {acc[u][4 * g0],     acc[u][4 * g0 + 1], acc[u][4 * g0 + 2], acc[u][4 * g0 + 3],
                              acc[u][4 * g0 + 4], acc[u][4 * g0 + 5], acc[u][4 * g0 + 6], acc[u][4 * g0 + 7]};
          bf16x8 pb[3];
          planes_of8(x, pb);
          bf16x8 a[2][3];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(kfrag + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
          for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) o[mb] = TE_MFMA_BF16(a[mb][PA[p6]], pb[PB[p6]], o[mb]);
        }
      }
    }
    if (!last) {
      write_k<W>(nbuf, vr, N, c + 1);
      if (need_qk) write_v<W>(nbuf + kOperand, kr, N, c + 1);
      __syncthreads();
    }
  }
  if (row_ok && need_qk) {
    float* o_row = dq + b * dqs.sb + h * dqs.sh + (int64_t)i * dqs.sn + 4 * kh;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(o_row + 32 * mb + 8 * g) = f32x4{o[mb][4 * g], o[mb][4 * g + 1], o[mb][4 * g + 2], o[mb][4 * g + 3]};
  }
}

template <int W>
int launch_w(const float* dout, Strided dos, const float* outp, Strided os, const float* k, Strided ks, const float* v, Strided vs,
             const float* attn, float* dattn, float* rowdot, float* dq, Strided dqs, int64_t B, int64_t H, int64_t N, float scale,
             int need_qk, hipStream_t stream) {
  const int NBr = (int)((N + 31) >> 5);
  const int G = (NBr + W - 1) / W, RB = (NBr + G - 1) / G;
  const int64_t BH = B * H, slots = ((BH + 7) / 8) * G;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd6l_rows_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<W>::kLds);
  if (e != hipSuccess) return (int)e;
  bwd6l_rows_kernel<W><<<dim3((unsigned)(slots * 8)), dim3(64 * W), Cfg<W>::kLds, stream>>>(dout, dos, outp, os, k, ks, v, vs, attn, dattn, rowdot, dq,
                                                                                           dqs, (int)H, (int)N, (int)BH, G, RB, scale, need_qk);
  return TE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The COLUMN side: d_v = attn^T d_out, d_k = d_s^T q with d_s recomputed from attn, d_attn and rowdot (what the row side left).
// te_attn_long.hip's attn_bwd_cols_kernel gives a wave a key block and walks the query rows two at a time on fp32 MFMAs with its
// operands straight from global memory (0.19 of the HBM roofline at N = 577).  Here a wave owns a key block too -- d_v^T and d_k^T
// [64, 32 keys] accumulate in its registers over ALL query rows, one k-ordered chain each, nothing shared or summed across waves --
// but the products run on bf16 MFMAs with split operands, sixteen rows per step: the B operands are the wave's own column panels
// of attn and d_s, a lane (key, h) loading rows 8 h + (0..7) of its key column (a half-wave reads 128 contiguous bytes of a row),
// a ring of one register set per K16 step of a chunk, re-requested for the next chunk as soon as its values are planes; the A
// operands are the chunk's d_out^T and q^T planes in LDS (head-dimension-major, plain row order), staged as on the row side: one
// barrier per chunk of 8 W rows.
template <int W>
__global__ __launch_bounds__(64 * W, 2) void bwd6l_cols_kernel(const float* __restrict__ attn, const float* __restrict__ dattn,
                                                               const float* __restrict__ rowdot, const float* __restrict__ dout,
                                                               Strided dos, const float* __restrict__ q, Strided qs,
                                                               float* __restrict__ dv, Strided dvs, float* __restrict__ dk, Strided dks,
                                                               int H, int N, int BH, int G, int RB, float scale, int need_qk) {
  typedef Cfg<W> C;
  constexpr int kKC = C::kKC, NS = 2 * C::kNKB, kBuf = C::kBuf, kOperand = C::kOperand, kPlane = C::kPlane;
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
  const float* g_bh = dout + b * dos.sb + h * dos.sh;
  const float* q_bh = q + b * qs.sb + h * qs.sh;
  const unsigned nn_bytes = (unsigned)(N * N * 4);
  const Rsrc a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attn) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const Rsrc d_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dattn) + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  const Rsrc r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowdot) + (int64_t)bh * N, 0, (unsigned)(N * 4), 0x00020000);

  VReq gr, qr;
  request_v<W, false>(gr, g_bh, dos.sn, N, 0);
  if (need_qk) request_v<W, false>(qr, q_bh, qs.sn, N, 0);
  // the chunk's rowdot values travel with its planes: rdl[buffer 2][kKC] (rows beyond N: 0, the descriptor's range check)
  float* const rdl = reinterpret_cast<float*>(Pl + C::kBiasOff);
  const int rt = min((int)threadIdx.x, kKC - 1);
  auto request_rd = [&](int c) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, (unsigned)(kKC * c + rt) * 4u, 0, 0));
  };
  float rdv = need_qk ? request_rd(0) : 0.0f;
  // the column panels: set s = the K16 step s of the current chunk (rows kKC c + 16 s + 8 h + t of this lane's key column)
  float pa[NS][8], ga[NS][8];
  const unsigned row_bytes = (unsigned)N * 4u;
  const unsigned col0 = ((unsigned)(8 * kh) * (unsigned)N + (unsigned)j) * 4u;
  auto request_set = [&](int c, int s) __attribute__((always_inline)) {
    const unsigned off = col0 + (unsigned)(kKC * c + 16 * s) * row_bytes;
#pragma unroll
    for (int t = 0; t < 8; ++t) pa[s][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, off + (unsigned)t * row_bytes, 0, 0));
    if (need_qk) {
#pragma unroll
      for (int t = 0; t < 8; ++t) ga[s][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d_rs, off + (unsigned)t * row_bytes, 0, 0));
    }
  };
  if (owner) {
#pragma unroll
    for (int s = 0; s < NS; ++s) request_set(0, s);
  }
  write_v<W, false>(Pl, gr, N, 0);
  if (need_qk) {
    write_v<W, false>(Pl + kOperand, qr, N, 0);
    if ((int)threadIdx.x < kKC) rdl[rt] = rdv;
  }
  __syncthreads();

  const unsigned char* const lane_frag = Pl + lane * 16;
  f32x16 av[2], ak[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) av[mb][e] = 0.0f, ak[mb][e] = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + (c & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    if (!last) {
      request_v<W, false>(gr, g_bh, dos.sn, N, c + 1);
      if (need_qk) {
        request_v<W, false>(qr, q_bh, qs.sn, N, c + 1);
        rdv = request_rd(c + 1);
      }
    }
    if (owner) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8 pb[3], sb[3];
        {
          float x[8];
          f32x4 ra[2];
          if (need_qk) {
            ra[0] = *reinterpret_cast<const f32x4*>(rdl + (c & 1) * kKC + 16 * s + 8 * kh);
            ra[1] = *reinterpret_cast<const f32x4*>(rdl + (c & 1) * kKC + 16 * s + 8 * kh + 4);
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) x[t] = pa[s][t];
          planes_of8(x, pb);
          if (need_qk) {
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = ((ga[s][t] - ra[t >> 2][t & 3]) * pa[s][t]) * scale;      // softmax backward, then '* scale'
            planes_of8(x, sb);
          }
        }
        request_set(c + 1, s);             // (behind the last chunk: rows beyond N, zeros nobody reads)
        bf16x8 a[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(buf + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) av[mb] = TE_MFMA_BF16(a[mb][PA[p6]], pb[PB[p6]], av[mb]);
        if (need_qk) {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(buf + kOperand + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
          for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) ak[mb] = TE_MFMA_BF16(a[mb][PA[p6]], sb[PB[p6]], ak[mb]);
        }
      }
    }
    if (!last) {
      write_v<W, false>(nbuf, gr, N, c + 1);
      if (need_qk) {
        write_v<W, false>(nbuf + kOperand, qr, N, c + 1);
        if ((int)threadIdx.x < kKC) rdl[((c + 1) & 1) * kKC + rt] = rdv;
      }
      __syncthreads();
    }
  }
  if (owner && j < N) {
    // lane (key j, h) holds d_v[j][32 mb + 8 g + 4 h + (0..3)]: 16-byte pieces of the key's row
    float* v_row = dv + b * dvs.sb + h * dvs.sh + (int64_t)j * dvs.sn + 4 * kh;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(v_row + 32 * mb + 8 * g) = f32x4{av[mb][4 * g], av[mb][4 * g + 1], av[mb][4 * g + 2], av[mb][4 * g + 3]};
    if (need_qk) {
      float* k_row = dk + b * dks.sb + h * dks.sh + (int64_t)j * dks.sn + 4 * kh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(k_row + 32 * mb + 8 * g) = f32x4{ak[mb][4 * g], ak[mb][4 * g + 1], ak[mb][4 * g + 2], ak[mb][4 * g + 3]};
    }
  }
}

template <int W>
int launch_cols_w(const float* attn, const float* dattn, const float* rowdot, const float* dout, Strided dos, const float* q, Strided qs,
                  float* dv, Strided dvs, float* dk, Strided dks, int64_t B, int64_t H, int64_t N, float scale, int need_qk,
                  hipStream_t stream) {
  const int NBr = (int)((N + 31) >> 5);
  const int G = (NBr + W - 1) / W, RB = (NBr + G - 1) / G;
  const int64_t BH = B * H, slots = ((BH + 7) / 8) * G;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd6l_cols_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<W>::kLds);
  if (e != hipSuccess) return (int)e;
  bwd6l_cols_kernel<W><<<dim3((unsigned)(slots * 8)), dim3(64 * W), Cfg<W>::kLds, stream>>>(attn, dattn, rowdot, dout, dos, q, qs, dv, dvs, dk, dks,
                                                                                           (int)H, (int)N, (int)BH, G, RB, scale, need_qk);
  return TE_OK;
}

}  // namespace

bool supported(int64_t B, int64_t H, int64_t N, int64_t D) {
  return D == 64 && N > 64 && N <= kMaxN && B >= 1 && H >= 1 && B * H <= (1 << 24);
}

// out may be null with need_qk = 0 only (nothing but d_attn is formed then)
int launch_rows(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn, const float* out, int64_t o_sb, int64_t o_sh,
                int64_t o_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb, int64_t v_sh,
                int64_t v_sn, const float* attn, float* d_attn, float* rowdot, float* d_q, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                int64_t B, int64_t H, int64_t N, float scale, int need_qk, hipStream_t stream) {
  const te_attn_l6::Strided dos{do_sb, do_sh, do_sn}, os{o_sb, o_sh, o_sn}, ks{k_sb, k_sh, k_sn}, vs{v_sb, v_sh, v_sn}, dqs{dq_sb, dq_sh, dq_sn};
  const int NBr = (int)((N + 31) >> 5), G8 = (NBr + 7) / 8, G4 = (NBr + 3) / 4;
  bool w8 = 5 * G8 * 8 <= 6 * G4 * 4;            // (te_attn_fwd6l.hip: launch)
#ifdef TE_STUDY
  static const int wenv = [] { const char* e = getenv("TE_BWD6L_WAVES"); return e ? atoi(e) : 0; }();
  if (wenv == 8) w8 = true;
  if (wenv == 4) w8 = false;
#endif
  return w8 ? launch_w<8>(d_out, dos, out, os, k, ks, v, vs, attn, d_attn, rowdot, d_q, dqs, B, H, N, scale, need_qk, stream)
            : launch_w<4>(d_out, dos, out, os, k, ks, v, vs, attn, d_attn, rowdot, d_q, dqs, B, H, N, scale, need_qk, stream);
}


int launch_cols(const float* attn, const float* d_attn, const float* rowdot, const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn, float* d_k,
                int64_t dk_sb, int64_t dk_sh, int64_t dk_sn, int64_t B, int64_t H, int64_t N, float scale, int need_qk, hipStream_t stream) {
  const te_attn_l6::Strided dos{do_sb, do_sh, do_sn}, qs{q_sb, q_sh, q_sn}, dvs{dv_sb, dv_sh, dv_sn}, dks{dk_sb, dk_sh, dk_sn};
  const int NBr = (int)((N + 31) >> 5), G8 = (NBr + 7) / 8, G4 = (NBr + 3) / 4;
  bool w8 = 5 * G8 * 8 <= 6 * G4 * 4;
#ifdef TE_STUDY
  static const int wenv = [] { const char* e = getenv("TE_BWD6L_WAVES"); return e ? atoi(e) : 0; }();
  if (wenv == 8) w8 = true;
  if (wenv == 4) w8 = false;
#endif
  return w8 ? launch_cols_w<8>(attn, d_attn, rowdot, d_out, dos, q, qs, d_v, dvs, d_k, dks, B, H, N, scale, need_qk, stream)
            : launch_cols_w<4>(attn, d_attn, rowdot, d_out, dos, q, qs, d_v, dvs, d_k, dks, B, H, N, scale, need_qk, stream);
}

}  // namespace te_attn_bwd6l
