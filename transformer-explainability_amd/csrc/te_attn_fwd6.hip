// te_attn_fwd6.hip -- the attention forward producer (ViT_LRP.py:132-152) with ROW-BLOCK OWNERS on bf16 MFMAs
// (round 6):
//
//   z_qk [BH,N,N] = q k^T (unscaled, cached for the QK rule);  attn [BH,N,N] = softmax(z_qk * scale);  out [B,N,C] = attn v
//
// on the fused qkv activation [B,N,3C] ('b n (qkv h d)'), one workgroup per (b, h), N <= 224, head dim 64.
//
// te_attn_rules.hip's attn_fwd_kernel walks the query rows in tiles of 32 with the score tile in LDS between three barriers per
// tile (fp32 MFMAs, every wave in the same phase at the same time: 0.26-0.27 of the HBM roofline since round 2).  Here wave w owns
// query-row block w for the WHOLE pipeline and the N x N tensors never touch LDS:
//
//   * scores, transposed:  D[j][i] = sum_d k[j][d] q[i][d]  -- A = a 32-key block of k (bf16 planes in MFMA-fragment order in LDS,
//     staged once per workgroup), B = the wave's own 32 rows of q (planes in registers), so a lane (i, h) ends up with ITS query
//     row's scores for the keys 32 jb + 8 g + 4 h + (0..3): runs of four consecutive keys = 16-byte pieces of z_qk and attn rows,
//     and the row's softmax is a reduction over the lane's own 4 x 7 x 4 registers plus one exchange with lane i + 32;
//   * out, transposed:     O[d][i] = sum_j v[j][d] p[i][j]  -- A = v^T (planes in LDS, written over k's after one barrier from loads
//     requested before the first product, K order = the accumulator layout's: request_vt / write_vt), B = the probabilities straight from the accumulator registers, split into planes per
//     K16 step; a lane (i, h) ends up with out[i][32 mb + 8 g + 4 h + (0..3)]: 16-byte pieces of the 'b n (h d)' row;
//   * both products on v_mfma_f32_32x32x16_bf16 with three-way split operands (te_linear_x6.hip: an fp32 value is the exact sum of
//     three bf16 values; the six partial products above 2^-24, smallest first, fp32 accumulation): fp32-class accuracy
//     (tests/test_gpu_producers.py: against stock PyTorch and against fp64);
//   * two barriers per WORKGROUP (k planes staged / k planes free for v^T), none per tile: the seven row-block waves drift apart and
//     one's stores and exponentials run beside another's MFMAs.
//
// Every reduction has a fixed order that depends on N only: a batch equals its samples run one by one, bit for bit.
#include <type_traits>

#include "te_common.h"

namespace te_attn_fwd6 {

namespace {

constexpr int kT = 512;                 // 8 waves; wave w owns query-row block w (N <= 224: at most 7)
constexpr int kMaxN = 224;
constexpr int kMaxB = kMaxN / 32;       // row / key blocks
constexpr int kMaxS = kMaxN / 16;       // K16 steps over the keys (second product)
constexpr int kFrag = 1024;             // one plane fragment: [kh 2][r 32][8 bf16]

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
// eight fp32 values (K order t = 0..7) -> the three plane operands of one lane
__device__ __forceinline__ void planes_of8(const float (&x)[8], bf16x8 (&b)[3]) {
  unsigned pk[4][3];
#pragma unroll
  for (int t2 = 0; t2 < 4; ++t2) split3_pk(x[2 * t2], x[2 * t2 + 1], pk[t2]);
#pragma unroll
  for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]});
}

// k [rows < N][64] as A planes with M = key, K = d:  Pk[plane 3][step 4][jb NB][kh 2][r 32][8]: element = plane q of
// k[32 jb + r][16 step + 8 kh + t].  One item = 8 consecutive d of one key: 8 threads cover the 256 bytes of a key's row.
__device__ __forceinline__ void stage_k(unsigned char* __restrict__ Pk, const float* __restrict__ k, int64_t sn, int N, int NB) {
  constexpr int kItems = (kMaxB * 32 * 8 + kT - 1) / kT;             // items per thread at most (4): every request first, then the splits
  f32x4 v0[kItems], v1[kItems];
#pragma unroll
  for (int u = 0; u < kItems; ++u) {
    const int item = threadIdx.x + u * kT, j = item >> 3, c8 = item & 7;
    const float* src = k + (int64_t)min(j, N - 1) * sn + 8 * c8;
    v0[u] = *reinterpret_cast<const f32x4_u*>(src), v1[u] = *reinterpret_cast<const f32x4_u*>(src + 4);
  }
#pragma unroll
  for (int u = 0; u < kItems; ++u) {
    const int item = threadIdx.x + u * kT;
    if (item < NB * 32 * 8) {
      const int j = item >> 3, c8 = item & 7;             // key, (step, kh) = chunk of 8 d
      const int step = c8 >> 1, kh = c8 & 1;
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (j < N) ? v0[u][e] : 0.0f, x[4 + e] = (j < N) ? v1[u][e] : 0.0f;
      bf16x8 b[3];
      planes_of8(x, b);
      unsigned char* dst = Pk + (size_t)(step * NB + (j >> 5)) * kFrag + (kh * 32 + (j & 31)) * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + (size_t)q * 4 * NB * kFrag) = b[q];
    }
  }
}

// v [rows < N][64] as A planes with M = d, K = key, the K order of a B operand that came out of an MFMA accumulator (te_attn_rc.hip:
// stage_planes<true>):  Pv[plane 3][step NS][mb 2][kh 2][r 32][8]: element t = plane q of v[16 step + 8 (t >> 2) + 4 kh + (t & 3)][32 mb + r].
// One item = 8 keys x 4 consecutive d: eight 16-B loads (a half-wave covers 256 contiguous bytes of a row).  Keys >= N: 0.
// (request: at most one item per thread -- kMaxS * 2 * 16 = 448 <= 512 -- issued BEFORE the first product so that the loads fly beside it;
//  write: after the barrier that frees the k planes)
__device__ __forceinline__ void request_vt(f32x4 (&x)[8], const float* __restrict__ v, int64_t sn, int N, int NS) {
  const int item = min((int)threadIdx.x, NS * 2 * 16 - 1);
  const int c = item & 15, g8 = item >> 4;
  const int step = g8 >> 1, kh = g8 & 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = 16 * step + 8 * (t >> 2) + 4 * kh + (t & 3);
    x[t] = *reinterpret_cast<const f32x4_u*>(v + (int64_t)min(row, N - 1) * sn + 4 * c);
  }
}
__device__ __forceinline__ void write_vt(unsigned char* __restrict__ Pv, f32x4 (&x)[8], int N, int NS) {
  const int item = threadIdx.x;
  if (item >= NS * 2 * 16) return;
  const int c = item & 15, g8 = item >> 4;
  const int step = g8 >> 1, kh = g8 & 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = 16 * step + 8 * (t >> 2) + 4 * kh + (t & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[t][e] = (row < N) ? x[t][e] : 0.0f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int d = 4 * c + e;
    const float col[8] = {x[0][e], x[1][e], x[2][e], x[3][e], x[4][e], x[5][e], x[6][e], x[7][e]};
    bf16x8 b[3];
    planes_of8(col, b);
    unsigned char* dst = Pv + (size_t)(step * 2 + (d >> 5)) * kFrag + (kh * 32 + (d & 31)) * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + (size_t)q * NS * 2 * kFrag) = b[q];
  }
}

// e / s, correctly rounded wherever no intermediate leaves the normal range (the hardware's own expansion of an IEEE division
// without its range scaling; s = a row's sum of exponentials, in [1, N])
__device__ __forceinline__ f32x2 div2(f32x2 e, float s, float rcs) {
  f32x2 q = e * f32x2{rcs, rcs};
  const f32x2 r = __builtin_elementwise_fma(f32x2{-s, -s}, q, e);
  return __builtin_elementwise_fma(r, f32x2{rcs, rcs}, q);
}

// exp(x) for x <= 0 (a score minus its row's maximum): 2^t on v_exp_f32 with t = x log2(e) carried as a rounded product plus its
// exact residual (fma) plus the low part of log2(e): e^x = 2^t_hi (1 + ln2 t_lo) to ~1 ulp; results below the normal range are 0
__device__ __forceinline__ float exp_le0(float x) {
  constexpr float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-08f, kLn2 = 0.693147182464599609375f;
  const float t = x * kL2eHi;
  const float lo = fmaf(x, kL2eLo, fmaf(x, kL2eHi, -t));
  const float r = __builtin_amdgcn_exp2f(t);
  return (x < -87.0f) ? 0.0f : fmaf(r * kLn2, lo, r);
}

// a row's 16-byte piece (keys j0 .. j0 + 3) of an [N, N] tensor; the piece that straddles N goes out element by element
__device__ __forceinline__ void store_piece(float* __restrict__ row, int j0, int N, f32x4 v) {
  if (j0 + 3 < N) {
    *reinterpret_cast<f32x4_u*>(row + j0) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j0 + e < N) row[j0 + e] = v[e];
  }
}

// One 32 x 32 block of an [N, N] tensor from the accumulator layout -- lane (row n, h) holds the keys 8 g + 4 h + (0..3) -- to memory
// through a wave-private LDS tile (row stride 144 B: conflict-free 16-byte writes), so that a store instruction covers 128 contiguous
// bytes of each of 8 rows instead of 32 bytes of each of 32 (the texture addresser's rate: profiles/r06_attention_qk_rc_phases.log).
constexpr int kTileLd = 36;                                 // floats per tile row
// LAST = the key block that may straddle N: its pieces are guarded per lane.  Every other block leaves as buffer stores whose hardware
// range check (descriptor = the (b, h)'s N x N matrix) drops the rows at or beyond N: no branch, no exec masking, 32-bit offsets.
typedef __amdgpu_buffer_rsrc_t Rsrc;
#ifdef TE_FWD6_NO_BUFSTORE
constexpr bool kNoBufStore = true;      // measurement builds: the global-store path for every block
#else
constexpr bool kNoBufStore = false;
#endif
template <bool LAST>
__device__ __forceinline__ void block_out(float* __restrict__ tile, const f32x16& a, float* __restrict__ base, Rsrc rs, int i0, int j0, int N) {
  const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(tile + n * kTileLd + 8 * g + 4 * kh) = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
  const int r8 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int r = r8 + 8 * m;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * kTileLd + 4 * c);
    if constexpr (LAST || kNoBufStore) {
      if (i0 + r < N && j0 + 4 * c < N) store_piece(base + (int64_t)(i0 + r) * N, j0 + 4 * c, N, v);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)(((i0 + r) * N + j0 + 4 * c) * 4), 0, 0);
    }
  }
}

template <int I, int END, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < END) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, END>(f);
  }
}

constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};      // planes (1,1) (0,2) (2,0) (0,1) (1,0) (0,0): smallest first

// (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second; te_linear_x6.hip)
__device__ __forceinline__ void swap_halves(unsigned& lo_keep, unsigned& hi_keep) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 r = __builtin_amdgcn_permlane32_swap(lo_keep, hi_keep, false, false);
  lo_keep = r[0];
  hi_keep = r[1];
}

// PLANES: `out` also leaves as the operand planes of the projection that consumes it -- the signed planes of out [B N, C] and the
// planes of |out| in te_linear_x6.hip's fragment-major order P3[row / 32][k / 16][plane][kh][r 32][8 bf16], exactly what that layer's
// split pass (te_linear_x6_split_dual_f32) would write from the fp32 tensor: the pass and its re-read of `out` disappear.
// NB (row / key blocks of 32) is a template parameter: the whole pipeline of a wave is then ONE basic block and hipcc interleaves the
// stores and exponentials of one key block with the MFMAs of the next (with run-time guards around every block the kernel was a chain of
// 350 small blocks); the second product runs 2 NB K16 steps (keys beyond N: zero probabilities against zero planes).
template <int NB, bool PLANES>
__global__ __launch_bounds__(kT) void fwd6_kernel(const float* __restrict__ qkv, float* __restrict__ zqk, float* __restrict__ attn,
                                                  float* __restrict__ out, int H, int N, float scale,
                                                  unsigned char* __restrict__ xs, unsigned char* __restrict__ xa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char Pl[];
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int C = H * 64;
  const int64_t sn = 3 * (int64_t)C;
  constexpr int NS = 2 * NB;
  constexpr size_t planes_bytes = (size_t)3 * kFrag * 4 * NB;      // k planes; the v^T planes take the same space
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  const float* q_bh = qkv + (int64_t)b * N * sn + h * 64;
  const float* k_bh = q_bh + C;
  const float* v_bh = q_bh + 2 * C;
  const int i = wave * 32 + n;                      // this lane's query row
  const bool owner = wave < NB, row_ok = owner && i < N;
  float* z_bh = zqk + (int64_t)bh * N * N;
  float* a_bh = attn + (int64_t)bh * N * N;
  const Rsrc z_rs = __builtin_amdgcn_make_buffer_rsrc(z_bh, 0, N * N * 4, 0x00020000);
  const Rsrc a_rs = __builtin_amdgcn_make_buffer_rsrc(a_bh, 0, N * N * 4, 0x00020000);

  // ---- the wave's q rows as B planes in registers (requested before the k planes are staged: in flight beside the staging) ----
  bf16x8 qb[4][3];
  {
    f32x4 qv[4][2];
    const float* qr = q_bh + (int64_t)min(i, N - 1) * sn + 8 * kh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qv[s][0] = *reinterpret_cast<const f32x4_u*>(qr + 16 * s);
      qv[s][1] = *reinterpret_cast<const f32x4_u*>(qr + 16 * s + 4);
    }
    stage_k(Pl, k_bh, sn, N, NB);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = row_ok ? qv[s][0][e] : 0.0f, x[4 + e] = row_ok ? qv[s][1][e] : 0.0f;
      planes_of8(x, qb[s]);
    }
  }
  __syncthreads();
  f32x4 vreq[8];
  request_vt(vreq, v_bh, sn, N, NS);

  float* const tile = reinterpret_cast<float*>(Pl + planes_bytes) + wave * (32 * kTileLd);      // wave-private [32][36] tile
  f32x16 acc[NB];         // scores, then probabilities: acc[jb][4 g + c] <-> key 32 jb + 8 g + 4 h + c of row i
  if (owner) {
    const unsigned char* const frag = Pl + lane * 16;
    constexpr size_t plane_k = (size_t)4 * NB * kFrag;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jb][e] = 0.0f;
    // four key blocks at a time: the same partial product of four independent accumulators between dependent MFMAs
#pragma unroll
    for (int jg = 0; jg < NB; jg += 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (jg + u < NB) a[u][q] = *reinterpret_cast<const bf16x8*>(frag + q * plane_k + (size_t)(s * NB + jg + u) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (jg + u < NB) acc[jg + u] = TE_MFMA_BF16(a[u][PA[p6]], qb[s][PB[p6]], acc[jg + u]);
      }
    }
    // ---- z_qk leaves, the row's softmax in registers ('dots = einsum(...) * self.scale', ViT_LRP.py:139-141) ----
    float mx = -INFINITY;
    static_for<0, NB>([&](auto jbi) __attribute__((always_inline)) {
      constexpr int jb = decltype(jbi)::value;
      {
        block_out<(jb == NB - 1)>(tile, acc[jb], z_bh, z_rs, wave * 32, 32 * jb, N);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * jb + 8 * g + 4 * kh;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float x = (j0 + c < N) ? acc[jb][4 * g + c] * scale : -INFINITY;
            acc[jb][4 * g + c] = x;
            mx = fmaxf(mx, x);
          }
        }
      }
    });
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[jb][e] = exp_le0(acc[jb][e] - mx);           // exp(-inf) = 0 for the keys beyond N
          sum = sum + acc[jb][e];
        }
      }
    }
    sum = sum + __shfl_xor(sum, 32, 64);                   // (a + b = b + a: both lanes of a row hold the same bits)
    float rcs = __builtin_amdgcn_rcpf(sum);
    rcs = fmaf(fmaf(-sum, rcs, 1.0f), rcs, rcs);
    static_for<0, NB>([&](auto jbi) __attribute__((always_inline)) {
      constexpr int jb = decltype(jbi)::value;
      {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x2 p0 = div2(f32x2{acc[jb][4 * g], acc[jb][4 * g + 1]}, sum, rcs);
          const f32x2 p1 = div2(f32x2{acc[jb][4 * g + 2], acc[jb][4 * g + 3]}, sum, rcs);
          acc[jb][4 * g] = p0[0], acc[jb][4 * g + 1] = p0[1], acc[jb][4 * g + 2] = p1[0], acc[jb][4 * g + 3] = p1[1];
        }
        block_out<(jb == NB - 1)>(tile, acc[jb], a_bh, a_rs, wave * 32, 32 * jb, N);
      }
    });
  }
  __syncthreads();                                         // every wave is done with the k planes
  write_vt(Pl, vreq, N, NS);
  __syncthreads();
  if (owner) {
    const unsigned char* const frag = Pl + lane * 16;
    constexpr size_t plane_v = (size_t)NS * 2 * kFrag;
    f32x16 o[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[mb][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      {
        // K16 step s = keys 16 s .. 16 s + 15 = (jb = s / 2, g = 2 (s & 1), 2 (s & 1) + 1): B element t = 4 gg + c of lane (i, h)
        const int jb = s >> 1, g0 = 2 * (s & 1);
        const float x[8] = {acc[jb][4 * g0],     acc[jb][4 * g0 + 1], acc[jb][4 * g0 + 2], acc[jb][4 * g0 + 3],
                            acc[jb][4 * g0 + 4], acc[jb][4 * g0 + 5], acc[jb][4 * g0 + 6], acc[jb][4 * g0 + 7]};
        bf16x8 pb[3];
        planes_of8(x, pb);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          bf16x8 a[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const bf16x8*>(frag + q * plane_v + (size_t)(s * 2 + mb) * kFrag);
#pragma unroll
          for (int p6 = 0; p6 < 6; ++p6) o[mb] = TE_MFMA_BF16(a[PA[p6]], pb[PB[p6]], o[mb]);
        }
      }
    }
    if (row_ok) {
      float* o_row = out + ((int64_t)b * N + i) * C + h * 64 + 4 * kh;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(o_row + 32 * mb + 8 * g) = f32x4{o[mb][4 * g], o[mb][4 * g + 1], o[mb][4 * g + 2], o[mb][4 * g + 3]};
    }
    if constexpr (PLANES) {
      // lane (i, h) holds out[i][32 mb + 8 g + 4 h + (0..3)]: K16 step 4 head + 2 mb + g / 2, k-half g % 2, positions 4 h .. 4 h + 3 of the
      // eight.  The halves are exchanged as in the x6 Z epilogue: lanes h = 0 end up with the 16-byte pieces g = 0, 1, lanes h = 1
      // with g = 2, 3.  |x| planes = x planes with plane 0's sign cleared and the low planes' signs flipped where x < 0 (split_kernel).
      const int64_t t = (int64_t)b * N + i;
      const int64_t nks = C >> 4;
      unsigned char* const dst0 = (unsigned char*)nullptr + ((t >> 5) * nks * 3) * kFrag + (t & 31) * 16;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        unsigned w[4][3][2], wa[4][3][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned lo[3], hi[3];
          split3_pk(o[mb][4 * g], o[mb][4 * g + 1], lo);
          split3_pk(o[mb][4 * g + 2], o[mb][4 * g + 3], hi);
          const unsigned m0 = lo[0] & 0x80008000u, m1 = hi[0] & 0x80008000u;
          // (a zero residual is +0 whatever the sign of x: only non-zero halves of the low planes change sign -- split_kernel)
          auto flip = [](unsigned p, unsigned m) { return p ^ (m & (((p & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u)); };
#pragma unroll
          for (int q = 0; q < 3; ++q) w[g][q][0] = lo[q], w[g][q][1] = hi[q];
          wa[g][0][0] = lo[0] & 0x7fff7fffu, wa[g][0][1] = hi[0] & 0x7fff7fffu;
#pragma unroll
          for (int q = 1; q < 3; ++q) wa[g][q][0] = flip(lo[q], m0), wa[g][q][1] = flip(hi[q], m1);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              swap_halves(w[g][q][d], w[g + 2][q][d]);
              swap_halves(wa[g][q][d], wa[g + 2][q][d]);
            }
        if (row_ok) {
          const size_t off = (size_t)(dst0 - (unsigned char*)nullptr) + (size_t)(4 * h + 2 * mb + kh) * 3 * kFrag;
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              *reinterpret_cast<u32x4*>(xs + off + q * kFrag + c * 512) = u32x4{w[c][q][0], w[c][q][1], w[c + 2][q][0], w[c + 2][q][1]};
              if (xa)
                *reinterpret_cast<u32x4*>(xa + off + q * kFrag + c * 512) = u32x4{wa[c][q][0], wa[c][q][1], wa[c + 2][q][0], wa[c + 2][q][1]};
            }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ a, unsigned char* __restrict__ b) {
  const size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  *reinterpret_cast<u32x4*>(a + off) = u32x4{0u, 0u, 0u, 0u};
  if (b) *reinterpret_cast<u32x4*>(b + off) = u32x4{0u, 0u, 0u, 0u};
}

}  // namespace

bool supported(int64_t B, int64_t H, int64_t N, int64_t D) {
  return D == 64 && N >= 1 && N <= kMaxN && B >= 1 && H >= 1 && B * H <= 0x7fffffff;
}

int launch(const float* qkv, float* z_qk, float* attn, float* out, int64_t B, int64_t H, int64_t N, float scale, hipStream_t stream,
           void* out_planes, void* out_abs_planes) {
  const int NB = (int)((N + 31) >> 5);
  // k planes, then v^T planes (12 KB per block of 32 keys: <= 84 KB) + one [32][36] fp32 tile per wave
  const size_t lds = (size_t)3 * kFrag * 4 * NB + (size_t)(kT / 64) * 32 * kTileLd * 4;
  if (out_planes && (B * N) % 32 != 0) {
    // rows beyond B N of the last 32-row block: zero, as the split pass leaves them (a kernel, not a memset node: the producers are
    // captured in HIP graphs, where memset nodes ran out of order on this stack -- DESIGN.md section 7)
    const size_t tail = (size_t)H * 64 * 6 * 32, total = (size_t)((B * N + 31) / 32) * tail;      // (one 32-row block of planes)
    zero_bytes_kernel<<<dim3((unsigned)(tail / 16 / 256)), dim3(256), 0, stream>>>((unsigned char*)out_planes + total - tail,
                                                                                   out_abs_planes ? (unsigned char*)out_abs_planes + total - tail : nullptr);
  }
  void (*kern)(const float*, float*, float*, float*, int, int, float, unsigned char*, unsigned char*) = nullptr;
  switch (NB * 2 + (out_planes ? 1 : 0)) {
#define TE_FWD6_CASE(nb) \
    case nb * 2: kern = fwd6_kernel<nb, false>; break; \
    case nb * 2 + 1: kern = fwd6_kernel<nb, true>; break;
    TE_FWD6_CASE(1) TE_FWD6_CASE(2) TE_FWD6_CASE(3) TE_FWD6_CASE(4) TE_FWD6_CASE(5) TE_FWD6_CASE(6) TE_FWD6_CASE(7)
#undef TE_FWD6_CASE
    default: return TE_ERR_UNSUPPORTED;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  kern<<<dim3((unsigned)(B * H)), dim3(kT), lds, stream>>>(qkv, z_qk, attn, out, (int)H, (int)N, scale, (unsigned char*)out_planes,
                                                          (unsigned char*)out_abs_planes);
  return TE_OK;
}

}  // namespace te_attn_fwd6
