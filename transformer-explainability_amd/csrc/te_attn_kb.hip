// te_attn_kb.hip -- attention relprop rules (modules/layers_ours.py:48-60,122-127 via ViT_LRP.py:154-177, BERT.py:367-393) and
// the attention-gradient producers on the same machinery, with WAVE-OWNED KEY BLOCKS (round 5).
//
// The one-pass kernels of te_attn_rules.hip move every [32, keys] tile of an N x N operand global -> registers -> LDS ->
// registers, all eight waves of the workgroup in lock-step through load / form-S / barrier / product / barrier phases:
// 15-23 vector instructions per MFMA, 45 % of a tile's cycles in MFMAs, 0.26-0.37 of the roofline for three rounds.
// Here the N x N operand NEVER touches LDS:
//
//   * wave w owns key block jb (32 keys) of its (b, h [, key group]) for the whole kernel.  A [32 rows x 32 keys] block of the
//     N x N operand is loaded straight into the MFMA ACCUMULATOR layout -- lane (lr, kh) holds column lr, rows
//     crow(e, kh) = (e & 3) + 8 (e >> 2) + 4 kh, e < 16: sixteen dword loads whose half-waves read 128 contiguous bytes of
//     one row.  Because the contraction index of an MFMA may be visited in any order, that very register set IS
//       - the A operand of the column-side product (K = query rows: step e multiplies rows crow(e, 0) | crow(e, 1)),
//       - the element-wise factor of the row-side result (cam_attn = attn . G lands in the same layout), and
//       - the layout the N x N result is stored from (sixteen dword stores, 128-byte segments).
//   * the key-side operand of the row product (v or k of the wave's 32 keys) lives in 32 registers per lane for the whole
//     kernel; only the [32, 64] row-side tile (S = sd(R, Z), d_out, q) goes through LDS, double-buffered, ONE barrier per tile.
//   * every load of tile it + 1 is requested before the MFMAs of tile it; stores drain behind them (vmcnt is never waited to
//     zero inside the loop).
//
// Per wave and tile: 64 MFMAs (32 row-side, 32 column-side) against 32 global memory instructions, 8 ds_read_b128, 32
// ds_read_b32 and ~40 vector-ALU instructions: < 2 other instructions per 64-cycle fp32 MFMA.
//
// Reductions run in an order that depends on N only: a batch equals its samples run one by one, bit for bit.
#include <type_traits>

#include "te_common.h"

namespace te_attn_kb {

namespace {

constexpr int TI = 32;         // query rows per tile
constexpr int kT = 512;        // threads per workgroup
constexpr int kWaves = kT / 64;
constexpr int SLD = 68;        // row stride (floats) of the [32][64] row-side tile in LDS: conflict-free 16-B fragment reads

struct Strided {  // [B,H,N,64] view, 64 contiguous
  int64_t sb, sh, sn;
};

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

#define TE_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// row (inside a 32-row block) of accumulator element e of lane half kh (v_mfma_f32_32x32x2_f32: D[i][j], j = lane & 31)
__device__ __forceinline__ int crow(int e, int kh) { return (e & 3) + 8 * (e >> 2) + 4 * kh; }

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.0f;
}

enum { RULE = 0, BWD = 1 };

// Buffer addressing (buffer_load / buffer_store ... s[rsrc], s_off offen): a 128-bit descriptor built from wave-uniform
// values, a 32-bit per-lane byte offset and a scalar byte offset; accesses past `bytes` return 0 / are dropped.
typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Rsrc make_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
// bytes of a strided [N, 64] view (row stride sn floats) from its first element
__device__ __forceinline__ unsigned view_bytes(int N, int64_t sn) { return ((unsigned)(N - 1) * (unsigned)sn + 64u) * 4u; }
__device__ __forceinline__ float ld32(Rsrc r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ f32x4 ld128(Rsrc r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void st32(float x, Rsrc r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, voff, 0, 0);
}
// A store the compiler's s_waitcnt insertion does not see.  With loads AND stores in flight hipcc assumes they may retire
// out of order and drains vmcnt to zero before every use of a loaded value; hidden, the loads alone are counted exactly.
// Safe: vmcnt counts these stores too, so a wait hipcc computes for its loads can only wait longer than it thinks, never
// shorter (loads retire in order among themselves); the store data is read at issue (no expcnt for VMEM stores on gfx9+).
__device__ __forceinline__ void st32_hidden(float x, Rsrc r, unsigned voff) {
  asm volatile("buffer_store_dword %0, %1, %2, 0 offen" : : "v"(x), "v"(voff), "s"(r));
}

// ------------------------------------------------------------------------------------------------
// AV rule (MODE RULE):  S = sd(R, Z) [N,64];  cam_attn = attn . (S v^T) * scale;  cam_v = v . (attn^T S) * scale
// attention backward, first half (MODE BWD):  d_attn = d_out v^T;  d_v = attn^T d_out          (R = d_out, Z unused)
// R, Z strided [B,H,N,64]; attn, cam_attn contiguous [B*H,N,N]; v, cam_v strided.
// grid = BH * ngroups (bh fastest); workgroup g of a (b, h) owns key blocks [g KBG, (g + 1) KBG), wave w block g KBG + w.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kT) void av_kb_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, Strided zs, const float* __restrict__ attn,
    const float* __restrict__ v, Strided vs, float* __restrict__ cam_attn, float* __restrict__ cam_v, Strided cs, int H,
    int N, int BH, int KBG, float scale) {
  __shared__ __attribute__((aligned(16))) float St[2][TI * SLD];
  const int bh = blockIdx.x % BH, g = blockIdx.x / BH;
  const int b = bh / H, h = bh - b * H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int nkb = (N + 31) >> 5, kb = g * KBG + wave;
  const bool has_blk = wave < KBG && kb < nkb;            // wave-uniform
  const int j = kb * 32 + lr;                             // this lane's key
  const int ntiles = (N + TI - 1) / TI;
  const int srow = threadIdx.x >> 4, sc = threadIdx.x & 15;      // this thread's float4 of the [32][64] row-side tile

  // Buffer descriptors of this (b, h)'s views: uniform base + ONE 32-bit per-lane byte offset per access (no 64-bit vector
  // address arithmetic).  The hardware range check on that offset does the edge handling: rows at or beyond N lie past the
  // end of a view, so their loads return 0 without touching memory (S = sd(0, 0) = 0: they contribute nothing) and their
  // stores are dropped -- which also makes every prefetch unconditional (a tile beyond the last one reads zeros).  Lanes
  // whose KEY is beyond N wrap into the next row of the N x N operands: finite values that only reach those lanes' own,
  // never stored, results.  (The scalar offset field is left at 0: it is not part of the range check.)
  const unsigned nn_bytes = (unsigned)N * (unsigned)N * 4u;
  const Rsrc a_rs = make_rsrc(attn + (int64_t)bh * N * N, nn_bytes);
  const Rsrc ca_rs = make_rsrc(cam_attn + (int64_t)bh * N * N, nn_bytes);
  const Rsrc r_rs = make_rsrc(R + (int64_t)b * rs.sb + (int64_t)h * rs.sh, view_bytes(N, rs.sn));
  const Rsrc z_rs = make_rsrc(MODE == RULE ? Z + (int64_t)b * zs.sb + (int64_t)h * zs.sh : R, MODE == RULE ? view_bytes(N, zs.sn) : 0u);
  const Rsrc v_rs = make_rsrc(v + (int64_t)b * vs.sb + (int64_t)h * vs.sh, view_bytes(N, vs.sn));
  const Rsrc cv_rs = make_rsrc(cam_v + (int64_t)b * cs.sb + (int64_t)h * cs.sh, view_bytes(N, cs.sn));
  const unsigned row_bytes = (unsigned)N * 4u;

  f32x4 rr = {0.f, 0.f, 0.f, 0.f}, zz = {0.f, 0.f, 0.f, 0.f};
  const unsigned r_off = ((unsigned)srow * (unsigned)rs.sn + 4u * sc) * 4u, z_off = ((unsigned)srow * (unsigned)zs.sn + 4u * sc) * 4u;
  const unsigned r_tile = (unsigned)TI * (unsigned)rs.sn * 4u, z_tile = (unsigned)TI * (unsigned)zs.sn * 4u;
  auto fetch_rz = [&](int it) __attribute__((always_inline)) {          // (any it: tiles beyond the last read zeros)
    rr = ld128(r_rs, r_off + (unsigned)it * r_tile);
    if constexpr (MODE == RULE) zz = ld128(z_rs, z_off + (unsigned)it * z_tile);
  };
  auto put_s = [&](int it) __attribute__((always_inline)) {
    f32x4 s = rr;                                                   // BWD: the tile of d_out itself
    if constexpr (MODE == RULE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = te_sd(rr[e], zz[e]);       // rows beyond N: sd(0, 0) = 0
    }
    *reinterpret_cast<f32x4*>(&St[it & 1][srow * SLD + (sc << 2)]) = s;
  };

  if (!has_blk) {
    // a wave without a key block (N = 197: wave 7) only helps to form the row-side tiles
    fetch_rz(0);
    put_s(0);
    fetch_rz(1);
    __syncthreads();
    for (int it = 0; it < ntiles; ++it) {
      put_s(it + 1);
      fetch_rz(it + 2);
      __syncthreads();
    }
    return;
  }

  // key-side operand of the row product, resident: v[j][8 kg + 4 kh + 0..3] (keys beyond N: zeros)
  f32x4 vf[8];
  {
    const unsigned off = ((unsigned)j * (unsigned)vs.sn + 4u * kh) * 4u;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) vf[kg] = ld128(v_rs, off + 32u * kg);
  }
  // the wave's [32 rows x 32 keys] block of an N x N operand in accumulator layout: element e = row i0 + crow(e, kh), key j
  const unsigned lane_nn = ((unsigned)(4 * kh) * (unsigned)N + (unsigned)j) * 4u;
  auto fetch_attn = [&](int it, float (&dst)[16]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(it * TI) * row_bytes;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[e] = ld32(a_rs, lane_nn + (base + (unsigned)((e & 3) + 8 * (e >> 2)) * row_bytes));
  };

  float ac[16], an[16];
  fetch_rz(0);
  fetch_attn(0, ac);
  put_s(0);
  fetch_rz(1);
  __syncthreads();

  f32x16 accv[2];
  zero16(accv[0]);
  zero16(accv[1]);
  // Memory order of a tile.  hipcc drains vmcnt to ZERO wherever loads and stores are in flight together (it assumes they
  // may retire out of order), and it merges its counts pessimistically wherever a branch encloses a memory instruction.  So:
  // every memory instruction of a tile is issued UNCONDITIONALLY in one burst at the tile's top, right after the wait for the
  // row-side operand -- the N x N result of the PREVIOUS tile (kept in registers for one tile; stores hipcc does not see:
  // st32_hidden), then the requests of the next tiles -- and nothing memory-related follows until the next tile's top, by
  // when all of it is a whole tile (~4000 MFMA-pipe cycles) old.
  f32x16 gp;                                       // cam_attn / d_attn block of the previous tile, not yet stored
  zero16(gp);
  const bool key_ok = j < N;
  auto store_g = [&](int it_prev) __attribute__((always_inline)) {
    if (key_ok) {                                  // (rows beyond N: dropped by the range check)
      const unsigned base = (unsigned)(it_prev * TI) * row_bytes;
#pragma unroll
      for (int e = 0; e < 16; ++e) st32_hidden(gp[e], ca_rs, lane_nn + (base + (unsigned)((e & 3) + 8 * (e >> 2)) * row_bytes));
    }
  };
  // FULL = all 32 rows exist (every tile but, for N % 32 != 0, the last)
  auto tile = [&](int it, auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int i0 = it * TI;
    const float* Sc = St[it & 1];
    put_s(it + 1);                                 // (its buffer's last readers finished before the previous barrier)
    if (it > 0) store_g(it - 1);
    fetch_rz(it + 2);
    fetch_attn(it + 1, an);
    __builtin_amdgcn_sched_barrier(0);             // (hipcc otherwise sinks the requests to the END of the tile's MFMAs)
    // ---- row side: G = S v^T for this wave's key block; cam_attn = attn . G straight from the accumulators ----
    f32x16 gacc;
    zero16(gacc);
    {
      const float* Ap = Sc + lr * SLD + 4 * kh;
      f32x4 a[8];
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) a[kg] = *reinterpret_cast<const f32x4*>(Ap + 8 * kg);
#pragma unroll
      for (int kg = 0; kg < 8; ++kg)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) gacc = TE_MFMA32(a[kg][jj], vf[kg][jj], gacc);
    }
    // ---- column side: cam_v += attn^T S (keys x 64), K = the 32 query rows in accumulator order ----
    const int kgmax = FULL ? 4 : (N - i0 + 7) >> 3;          // rows of the last tile beyond N are zero in S: skipped
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const float* Yp = Sc + kh * 4 * SLD + db * 32 + lr;
      float bq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bq[e] = Yp[((e & 3) + 8 * (e >> 2)) * SLD];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (FULL || g4 < kgmax) {
#pragma unroll
          for (int e = 4 * g4; e < 4 * g4 + 4; ++e) accv[db] = TE_MFMA32(ac[e], bq[e], accv[db]);
        }
      }
    }
    // ---- the N x N result of this tile: stored at the top of the next one ----
#pragma unroll
    for (int e = 0; e < 16; ++e) gp[e] = (MODE == RULE) ? (ac[e] * gacc[e]) * scale : gacc[e];
    __syncthreads();                               // S(it + 1) is published; S(it)'s buffer is free
#pragma unroll
    for (int e = 0; e < 16; ++e) ac[e] = an[e];
  };
  const int nfull = N / TI;
  for (int it = 0; it < nfull; ++it) tile(it, std::true_type{});
  if (nfull < ntiles) tile(nfull, std::false_type{});
  store_g(ntiles - 1);

  // ---- column epilogue: accv[db][e] = (attn^T S)[key = 32 kb + crow(e, kh)][d = 32 db + lr]; keys beyond N: dropped ----
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const unsigned d4 = (unsigned)(db * 32 + lr) * 4u;
    float x[16];
    if constexpr (MODE == RULE) {
      const unsigned xoff = (unsigned)(kb * 32 + 4 * kh) * (unsigned)vs.sn * 4u + d4;
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = ld32(v_rs, xoff + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)vs.sn * 4u);
    }
    const unsigned ooff = (unsigned)(kb * 32 + 4 * kh) * (unsigned)cs.sn * 4u + d4;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float val = accv[db][e];
      if constexpr (MODE == RULE) val = (x[e] * val) * scale;
      st32(val, cv_rs, ooff + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)cs.sn * 4u);
    }
  }
}

// key blocks per workgroup: at most eight (one per wave), the blocks of a (b, h) spread evenly over ceil(nkb / 8) workgroups
inline void groups_for(int64_t N, int& ng, int& kbg) {
  const int nkb = (int)((N + 31) / 32);
  ng = (nkb + kWaves - 1) / kWaves;
  kbg = (nkb + ng - 1) / ng;
}

}  // namespace

bool supported(int64_t B, int64_t H, int64_t N, int64_t D) {
  int ng, kbg;
  groups_for(N, ng, kbg);
  // (32-bit offsets inside a (b, h) view: N <= 4096 and, checked by the launchers, a row stride <= 2^16 floats)
  return D == 64 && N >= 1 && N <= 4096 && B * H * ng <= 0x7fffffff;
}

// mode 0: the AV rule; mode 1: d_attn / d_v of the attention backward (R = d_out, Z ignored, scale ignored)
int av_launch(int mode, const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn, const float* attn, const float* v,
              int64_t v_sb, int64_t v_sh, int64_t v_sn, const float* Z, int64_t z_sb, int64_t z_sh, int64_t z_sn,
              float* cam_attn, float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H, int64_t N,
              float scale, hipStream_t stream) {
  int ng, kbg;
  groups_for(N, ng, kbg);
  const int BH = (int)(B * H);
  if (r_sn > 65536 || z_sn > 65536) return TE_ERR_UNSUPPORTED;      // 32-bit row offsets inside a (b, h) view
  const dim3 grid((unsigned)(BH * ng)), blk(kT);
  const Strided rs{r_sb, r_sh, r_sn}, zs{z_sb, z_sh, z_sn}, vs{v_sb, v_sh, v_sn}, cs{cv_sb, cv_sh, cv_sn};
  if (mode == 0)
    av_kb_kernel<RULE><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale);
  else
    av_kb_kernel<BWD><<<grid, blk, 0, stream>>>(R, rs, nullptr, Strided{0, 0, 0}, attn, v, vs, cam_attn, cam_v, cs, (int)H,
                                                (int)N, BH, kbg, 1.0f);
  return TE_OK;
}

}  // namespace te_attn_kb
