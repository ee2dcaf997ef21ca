// te_attn_l6.h -- what the long-sequence attention producers (te_attn_fwd6l.hip, te_attn_bwd6l.hip; round 6) share: the chunk
// geometry, the three-way bf16 split, the staging of a 64-wide operand's chunk as MFMA A-operand planes in LDS (key-major for a
// product that contracts the head dimension, head-dimension-major for one that contracts the keys), the product of a chunk with
// the wave's own 32 rows, and the way a 32 x 32 block of an [N, N] tensor travels between the accumulator layout and memory.
#pragma once
#include "te_common.h"

namespace te_attn_l6 {

constexpr int kMaxN = 640;
constexpr int kFrag = 1024;             // one plane fragment: [kh 2][r 32][8 bf16]
constexpr int kTileLd = 36;             // floats per row of the wave-private [32][36] tile (conflict-free 16-byte writes)
// W waves per workgroup; wave w owns row block part * RB + w.  A chunk = 8 W keys, so that staging it is one item per thread:
//   W = 8: 64-key chunks (2 key blocks, 4 K16 steps of the second product), 135 KB of LDS, one workgroup per CU
//   W = 4: 32-key chunks, 69 KB: TWO workgroups per CU that drift against each other (a barrier aligns the phases of the waves
//          it joins: every wave of a workgroup starts a chunk's MFMAs, then its exponentials, at the same time), and 19 row blocks
//          (N = 577) are 5 parts of 4 + 4 + 4 + 4 + 3 waves instead of 3 parts of 7 + 6 + 6 of 8
template <int W>
struct Cfg {
  static constexpr int kT = 64 * W;
  static constexpr int kKC = 8 * W;                  // keys per chunk
  static constexpr int kNKB = kKC / 32;              // key blocks per chunk
  static constexpr int kPlane = 4 * kNKB * kFrag;    // one plane of a chunk operand: k [step 4][jb kNKB], v^T [step 2 kNKB][mb 2]
  static constexpr int kOperand = 3 * kPlane;        // k planes, then v^T planes
  static constexpr int kBuf = 2 * kOperand;
  static constexpr int kBiasOff = 2 * kBuf;
  static constexpr int kTileOff = kBiasOff + kMaxN * 4;
  static constexpr int kLds = kTileOff + W * 32 * kTileLd * 4;
};

struct Strided {  // [B,H,N,64] view, 64 contiguous
  int64_t sb, sh, sn;
};

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;

#define TE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// x0, x1 -> three packed bf16 pairs (x0 low half): x = p[0] + p[1] + p[2] exactly (te_linear_x6.hip: split3_pk)
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned (&p)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    p[q] = u;
    x0 = x0 - __uint_as_float(u << 16);
    x1 = x1 - __uint_as_float(u & 0xffff0000u);
  }
}
__device__ __forceinline__ void planes_of8(const float (&x)[8], bf16x8 (&b)[3]) {
  unsigned pk[4][3];
#pragma unroll
  for (int t2 = 0; t2 < 4; ++t2) split3_pk(x[2 * t2], x[2 * t2 + 1], pk[t2]);
#pragma unroll
  for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]});
}

// ---- chunk c of k [N][64] as A planes with M = key, K = d:  Pk[plane 3][step 4][jb kNKB][kh 2][r 32][8], element = plane q of
// k[kKC c + 32 jb + r][16 step + 8 kh + t].  One item per thread: 8 consecutive d of one key (8 threads cover a key's 256 bytes).
struct KReq {
  f32x4 v0, v1;
};
template <int W>
__device__ __forceinline__ void request_k(KReq& r, const float* __restrict__ k, int64_t sn, int N, int c) {
  const int j = threadIdx.x >> 3, c8 = threadIdx.x & 7;
  const float* src = k + (int64_t)min(Cfg<W>::kKC * c + j, N - 1) * sn + 8 * c8;
  r.v0 = *reinterpret_cast<const f32x4_u*>(src), r.v1 = *reinterpret_cast<const f32x4_u*>(src + 4);
}
template <int W>
__device__ __forceinline__ void write_k(unsigned char* __restrict__ Pk, const KReq& r, int N, int c) {
  const int j = threadIdx.x >> 3, c8 = threadIdx.x & 7, step = c8 >> 1, kh = c8 & 1;
  const bool ok = Cfg<W>::kKC * c + j < N;
  float x[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = ok ? r.v0[e] : 0.0f, x[4 + e] = ok ? r.v1[e] : 0.0f;
  bf16x8 b[3];
  planes_of8(x, b);
  unsigned char* dst = Pk + (step * Cfg<W>::kNKB + (j >> 5)) * kFrag + (kh * 32 + (j & 31)) * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + q * Cfg<W>::kPlane) = b[q];
}
// ---- chunk c of v [N][64] as A planes with M = d, K = key, in the K order of a B operand that came out of an MFMA accumulator:
// Pv[plane 3][step 2 kNKB][mb 2][kh 2][r 32][8], element t = plane q of v[kKC c + 16 step + 8 (t >> 2) + 4 kh + (t & 3)][32 mb + r].
// One item per thread: the eight keys of one (step, kh) at one d (a wave reads 256 contiguous bytes of a row per instruction).
struct VReq {
  float x[8];
};
// PERM = false: the plain K order, element t = row kKC c + 16 step + 8 kh + t (a B operand loaded straight from memory: the column
// side of the backward pass, whose K index is the query row)
template <int W, bool PERM = true>
__device__ __forceinline__ void request_v(VReq& r, const float* __restrict__ v, int64_t sn, int N, int c) {
  const int d = threadIdx.x & 63, g8 = threadIdx.x >> 6, step = g8 >> 1, kh = g8 & 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = Cfg<W>::kKC * c + 16 * step + (PERM ? 8 * (t >> 2) + 4 * kh + (t & 3) : 8 * kh + t);
    r.x[t] = v[(int64_t)min(row, N - 1) * sn + d];
  }
}
template <int W, bool PERM = true>
__device__ __forceinline__ void write_v(unsigned char* __restrict__ Pv, const VReq& r, int N, int c) {
  const int d = threadIdx.x & 63, g8 = threadIdx.x >> 6, step = g8 >> 1, kh = g8 & 1;
  float x[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = Cfg<W>::kKC * c + 16 * step + (PERM ? 8 * (t >> 2) + 4 * kh + (t & 3) : 8 * kh + t);
    x[t] = (row < N) ? r.x[t] : 0.0f;
  }
  bf16x8 b[3];
  planes_of8(x, b);
  unsigned char* dst = Pv + (step * 2 + (d >> 5)) * kFrag + (kh * 32 + (d & 31)) * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + q * Cfg<W>::kPlane) = b[q];
}

// e / s, correctly rounded wherever no intermediate leaves the normal range (te_attn_fwd6.hip: div2)
__device__ __forceinline__ f32x2 div2(f32x2 e, float s, float rcs) {
  f32x2 q = e * f32x2{rcs, rcs};
  const f32x2 r = __builtin_elementwise_fma(f32x2{-s, -s}, q, e);
  return __builtin_elementwise_fma(r, f32x2{rcs, rcs}, q);
}
// exp(x) for two x <= 0 (a score minus its row's maximum; -inf for the keys beyond N) on packed fp32 instructions: 2^t on
// v_exp_f32 with t = x log2(e) carried as a rounded product plus its exact residual (fma) plus the low part of log2(e):
// e^x = 2^t_hi (1 + ln2 t_lo) to ~1 ulp.  x is clamped at -150 first (2^-216 = 0 on v_exp_f32; -inf - (-inf) never forms).
__device__ __forceinline__ f32x2 exp2_le0(f32x2 x) {
  constexpr float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-08f, kLn2 = 0.693147182464599609375f;
  x[0] = fmaxf(x[0], -150.0f), x[1] = fmaxf(x[1], -150.0f);
  const f32x2 t = x * f32x2{kL2eHi, kL2eHi};
  const f32x2 lo = __builtin_elementwise_fma(x, f32x2{kL2eLo, kL2eLo}, __builtin_elementwise_fma(x, f32x2{kL2eHi, kL2eHi}, -t));
  const f32x2 r = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  return __builtin_elementwise_fma(r * f32x2{kLn2, kLn2}, lo, r);
}
__device__ __forceinline__ float exp_le0(float x) { return exp2_le0(f32x2{x, x})[0]; }
__device__ __forceinline__ f32x2 add2(f32x2 a, f32x2 b) { return f32x2{a[0] + b[0], a[1] + b[1]}; }
__device__ __forceinline__ f32x2 sub2(float a, float b, float m) { return f32x2{a - m, b - m}; }

// One 32 x 32 block of an [N, N] tensor from the accumulator layout -- lane (row n, h) holds the keys 8 g + 4 h + (0..3) -- through
// the wave's LDS tile to memory: a store instruction covers 128 contiguous bytes of each of 8 rows.  Rows at or beyond N fall
// outside the descriptor (the (b, h)'s N x N matrix) and are dropped by the hardware; a piece at or beyond column N is sent there
// on purpose (offset past the end); the piece that straddles N (N % 4 != 0, last chunk only: TAIL) goes out element by element.
constexpr unsigned kDrop = 0xfffffff0u;
// the byte offsets of a lane's four tile rows inside the (b, h)'s matrix at column 4 (lane & 7), formed once per kernel; a block's
// column offset rides in the scalar offset of the buffer instruction: no vector address arithmetic per block
struct RowOff {
  unsigned v[4];
};
__device__ __forceinline__ RowOff make_rowoff(int i0, int N) {
  const int lane = threadIdx.x & 63, r8 = lane >> 3, c = lane & 7;
  RowOff ro;
#pragma unroll
  for (int m = 0; m < 4; ++m) ro.v[m] = (unsigned)(((i0 + r8 + 8 * m) * N + 4 * c) * 4);
  return ro;
}
template <bool TAIL>
__device__ __forceinline__ void block_out(float* __restrict__ tile, const f32x16& a, Rsrc rs, const RowOff& ro, int j0, int N) {
  const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(tile + n * kTileLd + 8 * g + 4 * kh) = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
  const int r8 = lane >> 3, c = lane & 7;
  const int col = j0 + 4 * c;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int r = r8 + 8 * m;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * kTileLd + 4 * c);
    if constexpr (TAIL) {
      const unsigned off = ro.v[m] + (unsigned)(j0 * 4);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (col + 3 < N) ? off : kDrop, 0, 0);
      if (col < N && col + 3 >= N) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (col + e < N) {
            const float x = v[e];      // (a copy: __builtin_bit_cast of the vector-element lvalue v[e] read element 0 for every e)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, off + 4u * e, 0, 0);
          }
      }
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ro.v[m], j0 * 4, 0);
    }
  }
}

constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};      // planes (1,1) (0,2) (2,0) (0,1) (1,0) (0,0): smallest first

// scores of one chunk, transposed: acc[u][4 g + c] = sum_d k[kKC c + 32 u + 8 g + 4 h + c][d] q[i][d] for lane (i, h); the key
// blocks' chains interleaved.  Called by both passes: the same instruction sequence on the same operands, the same bits.
template <int W>
__device__ __forceinline__ void scores(f32x16 (&acc)[Cfg<W>::kNKB], const unsigned char* __restrict__ frag, const bf16x8 (&qb)[4][3]) {
  constexpr int NKB = Cfg<W>::kNKB;
#pragma unroll
  for (int u = 0; u < NKB; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[u][e] = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 a[NKB][3];
#pragma unroll
    for (int u = 0; u < NKB; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[u][q] = *reinterpret_cast<const bf16x8*>(frag + q * Cfg<W>::kPlane + (s * NKB + u) * kFrag);
#pragma unroll
    for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
      for (int u = 0; u < NKB; ++u) acc[u] = TE_MFMA_BF16(a[u][PA[p6]], qb[s][PB[p6]], acc[u]);
  }
}

// The reverse of block_out: a 32 x 32 block of an [N, N] tensor from memory (128-byte runs of eight rows per load instruction) through
// the wave's tile into the accumulator layout.  request: four 16-byte buffer loads (rows at or beyond N read as zero; a piece at or
// beyond column N reads the next row's values or zero -- finite, and the caller multiplies them with zero planes); land: tile
// round trip.
__device__ __forceinline__ void block_in_request(f32x4 (&v)[4], Rsrc rs, const RowOff& ro, int j0) {
#pragma unroll
  for (int m = 0; m < 4; ++m) v[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ro.v[m], j0 * 4, 0));
}
__device__ __forceinline__ void block_in_land(float* __restrict__ tile, const f32x4 (&v)[4], f32x16& a) {
  const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5, r8 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4*>(tile + (r8 + 8 * m) * kTileLd + 4 * c) = v[m];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(tile + n * kTileLd + 8 * g + 4 * kh);
    a[4 * g] = t[0], a[4 * g + 1] = t[1], a[4 * g + 2] = t[2], a[4 * g + 3] = t[3];
  }
}

}  // namespace te_attn_l6
