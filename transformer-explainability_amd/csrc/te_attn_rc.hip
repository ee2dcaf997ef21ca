// te_attn_rc.hip -- the QK relprop rule (modules/layers_ours.py:48-60 via ViT_LRP.py:165-173, BERT.py:389-393) and the softmax
// half of the attention-gradient backward pass with ROW-BLOCK AND KEY-BLOCK OWNERS (round 6).
//
//   QK rule:   S = sd(R_nn * f, Z_qk) [N,N];          cam_q = q .(S k) * scale;   cam_k = k .(S^T q) * scale
//   backward:  d_s = attn .(d_attn - rowsum(d_attn . attn)) * scale [N,N];   d_q = d_s k;   d_k = d_s^T q
//
// Both products contract the N x N operand S along ONE of its axes, so whichever way a workgroup is cut one of them is a sum over
// the waves' pieces: te_attn_rules.hip keeps a [32, keys] tile of S in LDS between barriers (fp32 MFMAs, 0.26-0.30 of the HBM
// roofline for four rounds), the kb study of round 5 summed per-wave partials of cam_q in LDS under three counters per tile
// (5-8 % faster).  Here NOTHING of the N x N operand is shared between waves:
//
//   * wave w of the (b, h) workgroup owns query-row block w (32 rows) for the row-side product and key block w (32 keys) for
//     the column-side product.  It evaluates S for its whole row panel [32, N] and, in a second phase, for its whole column
//     panel [N, 32], straight from global memory into the MFMA B-operand layout -- the second read of R_nn / Z_qk is served by
//     the L2 / Infinity Cache (the (b, h)'s two N x N tensors are 310 KB at N = 197), and S = sd(R, Z) is simply evaluated
//     twice (15 + 15 vector instructions per element against the 40-60 of a trip through LDS and two barriers per tile).
//   * every output is then ONE k-ordered chain inside one wave (K = all keys / all query rows, sixteen at a time): no partial
//     sums, no LDS reduction, no arrival counters, no barrier inside a phase; a batch equals its samples bit for bit.
//   * the products run on bf16 MFMAs with split operands (te_linear_x6.hip: an fp32 value is the exact sum of three bf16
//     values; six partial products, smallest first, fp32 accumulation): the evaluated TRANSPOSED, D[d][i], so that a lane owns
//     ONE query row (key) and runs of four consecutive d -- q / cam_q (k / cam_k) move as 16-byte pieces.
//   * the only shared operand is the 64-wide one: k^T (row phase) and then q^T (column phase) as bf16 planes in MFMA-fragment
//     order in LDS (84 KB at N <= 224), staged once per phase -- one barrier per phase, not per tile.
//   * the global loads of a phase run three K16 steps ahead in a ring of register sets; they are inline asm hipcc's waitcnt
//     insertion does not see (te_attn_kb.hip explains why), waited for with hand-counted vmcnt; a set is re-requested only
//     after its values were consumed, and scripts/check_hidden_loads.py walks the compiled loops (tests/test_isa_hazards.py).
//
// N <= 224 (one workgroup per (b, h), the ViT-B/16 224^2 and DeiT shapes); longer sequences stay on te_attn_rules.hip.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "te_common.h"

namespace te_attn_rc {

namespace {

constexpr int kT = 512;                 // threads per workgroup: 8 waves, wave w owns row block w and key block w
constexpr int kMaxN = 224;              // 7 blocks of 32
constexpr int kMaxSteps = kMaxN / 16;   // K16 steps of a phase
constexpr int kFrag = 1024;             // one plane fragment: [kh 2][r 32][8 bf16]

struct Strided {  // [B,H,N,64] view, 64 contiguous
  int64_t sb, sh, sn;
};

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;

enum { RULE = 0, BWD = 1 };
enum { ROWS = 0, COLS = 1 };

#define TE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define TE_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TE_PIN(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ Rsrc make_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
// loads hipcc's s_waitcnt insertion does not see (te_attn_kb.hip): offsets past the descriptor's size return 0 per dword
__device__ __forceinline__ void ld128_hidden(f32x4& v, Rsrc r, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(r));
}
__device__ __forceinline__ void ld32_hidden(float& v, Rsrc r, unsigned voff) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(r));
}

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

// The 64-wide operand M [rows < N][64] (k or q of this (b, h)) as bf16 planes in MFMA A-fragment order in LDS:
//   Pl[plane 3][step NS][mb 2][kh 2][r 32][8]:  element = plane q of M[16 step + 8 kh + t][32 mb + r], t = 0..7
// i.e. the A operand (M index = d, K index = row of M) of K16 step `step` for d block `mb` is one contiguous 1-KiB fragment
// that a wave reads with one conflict-free ds_read_b128 per lane.  One item = 8 consecutive rows x 4 consecutive d: eight 16-B
// global loads (a half-wave covers 256 contiguous bytes of a row), sixteen pair splits, twelve 16-B LDS stores.  Rows >= N: 0.
template <bool PERM>
__device__ __forceinline__ void stage_planes(unsigned char* __restrict__ Pl, const float* __restrict__ M, int64_t sn, int N,
                                             int NS) {
  for (int item = threadIdx.x; item < NS * 2 * 16; item += kT) {
    const int c = item & 15, g8 = item >> 4;               // d chunk (4 floats), (step, kh)
    const int step = g8 >> 1, kh = g8 & 1;
    // rows of M behind element t of the fragment: 16 step + 8 kh + t, or -- PERM, the row phase's k^T planes: the K order of a
    // B operand that came out of an MFMA accumulator (phase_rows) -- 16 step + 8 (t >> 2) + 4 kh + (t & 3)
    f32x4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {    // all eight requests first (rows beyond N re-read the last row and are zeroed below)
      const int row = PERM ? 16 * step + 8 * (t >> 2) + 4 * kh + (t & 3) : 16 * step + 8 * kh + t;
      v[t] = *reinterpret_cast<const f32x4_u*>(M + (int64_t)min(row, N - 1) * sn + 4 * c);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = PERM ? 16 * step + 8 * (t >> 2) + 4 * kh + (t & 3) : 16 * step + 8 * kh + t;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][e] = (row < N) ? v[t][e] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                          // d = 4 c + e: mb = d >> 5, r = d & 31
      const int d = 4 * c + e;
      unsigned pl[4][3];
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2) split3_pk(v[2 * t2][e], v[2 * t2 + 1][e], pl[t2]);
      unsigned char* dst = Pl + ((step * 2 + (d >> 5)) * kFrag) + (kh * 32 + (d & 31)) * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<u32x4*>(dst + (size_t)q * NS * 2 * kFrag) = u32x4{pl[0][q], pl[1][q], pl[2][q], pl[3][q]};
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
template <int N_>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// safe_divide (modules/layers_ours.py:10-13) of two element pairs on packed fp32 instructions: den = b + 1e-9 (one rounding),
// an exact-zero den replaced by 1e-9, a / den, zero where b == 0.  The quotient is formed as in the hardware's own expansion of
// an IEEE division without its range scaling (v_rcp_f32, one Newton step on the reciprocal, q = a rc, the exact residual
// r = a - den q by fma, q + r rc): correctly rounded wherever no intermediate leaves the normal range -- |den| >= 1e-16 by
// construction, relevance values and attention scores are far inside it -- at 8 instead of 17 vector instructions per element.
// The kernel is bound by vector-instruction issue (phase stamps: profiles/r06_attention_qk_rc_*.log), S is evaluated twice per
// element, and the six-product sums that consume it are re-associated against the reference anyway.
__device__ __forceinline__ f32x2 sd2(f32x2 a, f32x2 b) {
  f32x2 den = b + f32x2{1e-9f, 1e-9f};
  den[0] = (den[0] == 0.0f) ? 1e-9f : den[0];
  den[1] = (den[1] == 0.0f) ? 1e-9f : den[1];
  f32x2 rc = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 e = __builtin_elementwise_fma(-den, rc, f32x2{1.0f, 1.0f});
  rc = __builtin_elementwise_fma(e, rc, rc);
  f32x2 q = a * rc;
  const f32x2 r = __builtin_elementwise_fma(-den, q, a);
  q = __builtin_elementwise_fma(r, rc, q);
  q[0] = (b[0] != 0.0f) ? q[0] : 0.0f;
  q[1] = (b[1] != 0.0f) ? q[1] : 0.0f;
  return q;
}

// One phase of one wave.  SIDE = ROWS: the wave's 32 query rows against every key (A = k^T planes; a lane's S values are 8
// consecutive keys of its row: two 16-B loads per operand and step).  SIDE = COLS: the wave's 32 keys against every query row
// (A = q^T planes; a lane's S values are 8 consecutive rows of its key column: eight 4-B loads per operand and step, a
// half-wave covering 128 contiguous bytes of a row).  D[m = d][n = own row / key]: lane (n = lane & 31, h = lane >> 5) ends up
// with acc[mb][4 g + c] = the sum for d = 32 mb + 8 g + 4 h + c.
//
// Schedule.  The global loads run RING steps ahead (ROWS: 6 -- first touch, HBM; COLS: 3 -- the second read comes from the L2 /
// Infinity Cache, and 3 x 16 four-byte requests are what the 6-bit vmcnt can count).  The S planes of step s + 1 are formed in
// the same basic block as the MFMAs of step s (independent work: hipcc interleaves the ~130 vector instructions of the one
// with the twelve MFMAs of the other; an in-order wave cannot start the next step's arithmetic behind a dependent MFMA).
// STRAIGHT-LINE over the (at most kMaxSteps) steps: a register with a hidden load in flight must never be copied, and a loop
// back-edge is where hipcc copies (the phi of a loop-carried value: found by scripts/check_hidden_loads.py in the first,
// rolled, version of this phase).  Every request is issued UNCONDITIONALLY (steps beyond NS read past the views: zeros, or
// harmless in-range values that are never used); only the arithmetic of a step is under the wave-uniform guard, so the
// values that meet at its merge point are landed ones.
template <int MODE, int SIDE>
__device__ __forceinline__ void phase(f32x16 (&acc)[2], const unsigned char* __restrict__ Pl, size_t plane, Rsrc r_rs, Rsrc z_rs,
                                      int N, int NS, int blk, float f, bool has_f, float scale, float rd_own,
                                      const float* __restrict__ rdv) {
  const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  constexpr int LPS = (SIDE == ROWS) ? 4 : 16;                        // hidden loads per step
  constexpr int RING = (SIDE == ROWS) ? 6 : 3;                        // register sets: step s lives in set s % RING
  f32x4 rv[RING][2], zv[RING][2];                                     // ROWS: keys 8 kh + 0..3, 4..7 of the lane's row
  float rc[RING][8], zc[RING][8];                                     // COLS: rows 8 kh + 0..7 of the lane's key column
  const unsigned row_bytes = (unsigned)N * 4u;
  const unsigned base = (SIDE == ROWS) ? ((unsigned)(blk * 32 + n) * (unsigned)N + 8u * kh) * 4u
                                       : ((unsigned)(8 * kh) * (unsigned)N + (unsigned)(blk * 32 + n)) * 4u;
  auto issue = [&](int s, int set) __attribute__((always_inline)) {
    if constexpr (SIDE == ROWS) {
      const unsigned off = base + 64u * (unsigned)s;                  // 16 keys per step
      ld128_hidden(rv[set][0], r_rs, off);
      ld128_hidden(rv[set][1], r_rs, off + 16u);
      ld128_hidden(zv[set][0], z_rs, off);
      ld128_hidden(zv[set][1], z_rs, off + 16u);
    } else {
      const unsigned off = base + 16u * (unsigned)s * row_bytes;      // 16 rows per step
#pragma unroll
      for (int e = 0; e < 8; ++e) ld32_hidden(rc[set][e], r_rs, off + (unsigned)e * row_bytes);
#pragma unroll
      for (int e = 0; e < 8; ++e) ld32_hidden(zc[set][e], z_rs, off + (unsigned)e * row_bytes);
    }
  };
  auto pin = [&](int set) __attribute__((always_inline)) {
    if constexpr (SIDE == ROWS) {
      TE_PIN(rv[set][0]); TE_PIN(rv[set][1]); TE_PIN(zv[set][0]); TE_PIN(zv[set][1]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { TE_PIN(rc[set][e]); TE_PIN(zc[set][e]); }
    }
  };
  // the three bf16 planes of the lane's eight S values of step s, from the (landed, pinned) register set `set`
  auto make_planes = [&](int s, int set, bf16x8 (&b)[3]) __attribute__((always_inline)) {
    f32x2 sv[4];
    float rdr[8];
    if constexpr (MODE == BWD && SIDE == COLS) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(rdv + 16 * s + 8 * kh), a1 = *reinterpret_cast<const f32x4*>(rdv + 16 * s + 8 * kh + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) rdr[e] = a0[e], rdr[4 + e] = a1[e];
    }
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
      f32x2 r, z;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = 2 * t2 + u;
        if constexpr (SIDE == ROWS) r[u] = rv[set][e >> 2][e & 3], z[u] = zv[set][e >> 2][e & 3];
        else r[u] = rc[set][e], z[u] = zc[set][e];
      }
      if constexpr (MODE == RULE) {
        if (has_f) r = r * f32x2{f, f};                               // deferred per-sample factor of the mask Add (BERT.py:386-388)
        sv[t2] = sd2(r, z);
      } else {
        const f32x2 rd = (SIDE == ROWS) ? f32x2{rd_own, rd_own} : f32x2{rdr[2 * t2], rdr[2 * t2 + 1]};
        sv[t2] = (z * (r - rd)) * f32x2{scale, scale};                // softmax backward: attn . (d_attn - rowdot) * scale
      }
    }
    if constexpr (SIDE == ROWS) {
      // keys at or beyond N: the loads wrapped into the next row (finite values in general, but nothing bounds them): zero
      if (16 * s + 16 > N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e >> 1][e & 1] = (16 * s + 8 * kh + e < N) ? sv[e >> 1][e & 1] : 0.0f;
      }
    }
    // (COLS: a lane whose key is beyond N reads finite or garbage values that reach its own, never stored, column only;
    //  rows beyond N lie past the end of the views: 0)
    unsigned pk[4][3];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) split3_pk(sv[t2][0], sv[t2][1], pk[t2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]});
  };
  const unsigned char* const frag = Pl + lane * 16;
  static_for<0, RING>([&](auto i) __attribute__((always_inline)) { issue(decltype(i)::value, decltype(i)::value); });
  vm_wait<(RING - 1) * LPS>();                                        // step 0 has landed; RING - 1 younger steps in flight
  bf16x8 bc[3];
  pin(0);
  make_planes(0, 0, bc);
  issue(RING, 0);
  static_for<0, kMaxSteps>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    bf16x8 bn[3] = {bc[0], bc[1], bc[2]};
    if constexpr (s + 1 < kMaxSteps) {
      // requested so far: steps 0 .. min(s + RING, kMaxSteps - 1); younger than step s + 1:
      constexpr int last = (s + RING < kMaxSteps - 1) ? s + RING : kMaxSteps - 1;
      vm_wait<(last - (s + 1)) * LPS>();
    }
    if (s < NS) {
      bf16x8 a[2][3];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[mb][q] = *reinterpret_cast<const bf16x8*>(frag + q * plane + (size_t)(s * 2 + mb) * kFrag);
      if constexpr (s + 1 < kMaxSteps) {
        pin((s + 1) % RING);
        make_planes(s + 1, (s + 1) % RING, bn);                       // (of no use when s + 1 == NS: never multiplied)
      }
      // six partial products per block, smallest first (te_linear_x6.hip): planes (1,1) (0,2) (2,0) (0,1) (1,0) (0,0)
      constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[mb] = TE_MFMA_BF16(a[mb][PA[p6]], bc[PB[p6]], acc[mb]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) bc[q] = bn[q];
    if constexpr (s + 1 + RING < kMaxSteps) issue(s + 1 + RING, (s + 1) % RING);      // the set's values have been consumed
  });
  vm_wait<0>();                                                       // (already drained by the last step's wait)
  static_for<0, RING>([&](auto i) __attribute__((always_inline)) { pin(decltype(i)::value); });
}

// The row side on KEY-CONTIGUOUS loads (the first version gave every lane its own row -- two 16-B pieces per operand and step
// at a 4 N-byte stride between lanes: 64 different cache-line halves per load instruction, and the phase ran at the rate of the
// CU's texture addresser, 3 200 cycles per step against the column side's 1 700; profiles/r06_attention_qk_rc_phases.log).
// Here the wave walks its row panel [32 rows, N] one key block (32 keys = one ROUND of two steps) at a time with the column
// side's access pattern -- a lane owns a key, a half-wave reads 128 contiguous bytes of a row -- and forms S in that layout:
// lane (key, kh), eight rows.  The row product contracts over KEYS, so it needs S with a lane per ROW: the change of layout is
// two MFMAs per plane against a 0 / 1 selector,  T[key][row] = sum_k P[key][k] E[k][row]  (a bf16 plane value times 1.0,
// summed with zeros: exact), which leaves S^T in the accumulator layout -- lane (row, h), keys 8 g + 4 h + c -- i.e., read as
// two K16 steps (g = 0, 1 and g = 2, 3), a B operand whose K order is 8 (t >> 2) + 4 h + (t & 3): the k^T planes are staged in
// that order (stage_planes<true>).  Hidden loads: a ring of three steps (3 x 16 requests: what the 6-bit vmcnt can count), a
// set re-requested right after its values were consumed; straight-line code, see phase().
template <int MODE>
__device__ __forceinline__ void phase_rows(f32x16 (&acc)[2], const unsigned char* __restrict__ Pk, size_t plane, Rsrc r_rs,
                                           Rsrc z_rs, int N, int NB, int blk, float f, bool has_f, float scale,
                                           const float* __restrict__ rdv) {
  const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
  constexpr int LPS = 16, RING = 3;
  float rc[RING][8], zc[RING][8];
  const unsigned row_bytes = (unsigned)N * 4u;
  const unsigned base = ((unsigned)(blk * 32 + 8 * kh) * (unsigned)N + (unsigned)n) * 4u;
  auto issue = [&](int s, int set) __attribute__((always_inline)) {      // step s: key block s >> 1, rows 16 (s & 1) + 8 kh + e
    const unsigned off = base + ((unsigned)(16 * (s & 1)) * (unsigned)N + 32u * (unsigned)(s >> 1)) * 4u;
#pragma unroll
    for (int e = 0; e < 8; ++e) ld32_hidden(rc[set][e], r_rs, off + (unsigned)e * row_bytes);
#pragma unroll
    for (int e = 0; e < 8; ++e) ld32_hidden(zc[set][e], z_rs, off + (unsigned)e * row_bytes);
  };
  auto pin = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { TE_PIN(rc[set][e]); TE_PIN(zc[set][e]); }
  };
  auto make_planes = [&](int s, int set, bf16x8 (&b)[3]) __attribute__((always_inline)) {
    f32x2 sv[4];
    float rdr[8];
    if constexpr (MODE == BWD) {
      const float* rp = rdv + 32 * blk + 16 * (s & 1) + 8 * kh;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(rp), a1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) rdr[e] = a0[e], rdr[4 + e] = a1[e];
    }
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
      f32x2 r = {rc[set][2 * t2], rc[set][2 * t2 + 1]}, z = {zc[set][2 * t2], zc[set][2 * t2 + 1]};
      if constexpr (MODE == RULE) {
        if (has_f) r = r * f32x2{f, f};
        sv[t2] = sd2(r, z);
      } else {
        sv[t2] = (z * (r - f32x2{rdr[2 * t2], rdr[2 * t2 + 1]})) * f32x2{scale, scale};
      }
    }
    // a lane whose key is at or beyond N read the next row's values (finite in general, but nothing bounds them) and its S
    // values meet zeros of the k^T planes in the row product: zero them (rows beyond N lie past the end of the views: 0)
    if (32 * (s >> 1) + 32 > N) {
      const bool ok = 32 * (s >> 1) + n < N;
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2) sv[t2][0] = ok ? sv[t2][0] : 0.0f, sv[t2][1] = ok ? sv[t2][1] : 0.0f;
    }
    unsigned pk[4][3];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) split3_pk(sv[t2][0], sv[t2][1], pk[t2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]});
  };
  // selectors of the change of layout: E[sp] as a B operand [K = the 16 rows of step sp][N = the 32 rows of the block]
  bf16x8 E[2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) {
    const int tt = n - 16 * sp - 8 * kh;                               // the element of this lane that is 1.0, if 0 <= tt < 8
    u32x4 w;
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = (tt == 2 * u) ? 0x00003f80u : (tt == 2 * u + 1) ? 0x3f800000u : 0u;
    E[sp] = __builtin_bit_cast(bf16x8, w);
  }
  const unsigned char* const frag = Pk + lane * 16;
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  static_for<0, kMaxSteps / 2>([&](auto ri) __attribute__((always_inline)) {
    constexpr int r = decltype(ri)::value, s0 = 2 * r, s1 = 2 * r + 1;
    bf16x8 P0[3], P1[3];
    // step s0: requested so far 0 .. min(s0 + 2, kMaxSteps - 1)
    vm_wait<(((s0 + 2 < kMaxSteps - 1) ? s0 + 2 : kMaxSteps - 1) - s0) * LPS>();
    if (r < NB) {
      pin(s0 % RING);
      make_planes(s0, s0 % RING, P0);
    }
    if constexpr (s0 + RING < kMaxSteps) issue(s0 + RING, s0 % RING);
    vm_wait<(((s1 + 2 < kMaxSteps - 1) ? s1 + 2 : kMaxSteps - 1) - s1) * LPS>();
    if (r < NB) {
      pin(s1 % RING);
      make_planes(s1, s1 % RING, P1);
    }
    if constexpr (s1 + RING < kMaxSteps) issue(s1 + RING, s1 % RING);
    if (r < NB) {
      bf16x8 Ba[3], Bb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        f32x16 T;
#pragma unroll
        for (int e = 0; e < 16; ++e) T[e] = 0.0f;
        T = TE_MFMA_BF16(P0[q], E[0], T);
        T = TE_MFMA_BF16(P1[q], E[1], T);
        u32x4 wa, wb;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          wa[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{T[2 * u], T[2 * u + 1]}, bf16x2));
          wb[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{T[8 + 2 * u], T[8 + 2 * u + 1]}, bf16x2));
        }
        Ba[q] = __builtin_bit_cast(bf16x8, wa);
        Bb[q] = __builtin_bit_cast(bf16x8, wb);
      }
      constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 a[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            a[mb][q] = *reinterpret_cast<const bf16x8*>(frag + q * plane + (size_t)((s0 + half) * 2 + mb) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[mb] = TE_MFMA_BF16(a[mb][PA[p6]], half ? Bb[PB[p6]] : Ba[PB[p6]], acc[mb]);
      }
    }
  });
  vm_wait<0>();
  pin(0);
  pin(1);
  pin(2);
}

// rowdots of the softmax backward for the wave's 32 rows, rd[i] = sum_j d_attn[i][j] attn[i][j], into rdv[32 blk ..]: eight lanes
// per row, each 16 contiguous bytes of every 128-byte key unit (a load instruction covers eight rows x 128 contiguous bytes);
// a lane adds its products in ascending key order, the eight lanes' partials meet in a fixed butterfly (xor 1, 2, 4) -- an order
// that depends on N only.
__device__ __forceinline__ void row_dots(float* __restrict__ rdv, Rsrc r_rs, Rsrc z_rs, int N, int NB, int blk) {
  const int lane = threadIdx.x & 63, lr = lane >> 3, lc = lane & 7;
  constexpr int RING = 3, LPS = 8;                                      // units in flight, loads per unit (4 row groups x 2 operands)
  f32x4 rv[RING][4], zv[RING][4];
  const unsigned row_bytes = (unsigned)N * 4u;
  const unsigned base = ((unsigned)(blk * 32 + lr) * (unsigned)N + 4u * lc) * 4u;
  auto issue = [&](int u, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned off = base + (unsigned)(8 * p) * row_bytes + 128u * (unsigned)u;
      ld128_hidden(rv[set][p], r_rs, off);
      ld128_hidden(zv[set][p], z_rs, off);
    }
  };
  auto pin = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) { TE_PIN(rv[set][p]); TE_PIN(zv[set][p]); }
  };
  float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  static_for<0, kMaxSteps / 2>([&](auto ui) __attribute__((always_inline)) {      // straight-line: see phase()
    constexpr int u = decltype(ui)::value, NU = kMaxSteps / 2;
    vm_wait<(((u + 2 < NU - 1) ? u + 2 : NU - 1) - u) * LPS>();
    if (u < NB) {
      pin(u % RING);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = rv[u % RING][p][e] * zv[u % RING][p][e];
          part[p] = part[p] + ((32 * u + 4 * lc + e < N) ? v : 0.0f);      // (keys beyond N: the next row's values)
        }
    }
    if constexpr (u + RING < NU) issue(u + RING, u % RING);
  });
  vm_wait<0>();
  pin(0);
  pin(1);
  pin(2);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float v = part[p];
    v = v + __shfl_xor(v, 1, 64);
    v = v + __shfl_xor(v, 2, 64);
    v = v + __shfl_xor(v, 4, 64);
    if (lc == 0) rdv[32 * blk + 8 * p + lr] = v;                        // (rows beyond N: 0 -- their loads were out of range)
  }
}

// measurement builds (-DTE_STUDY): phase stamps of one workgroup of the second round (scripts/attn_rc_prof.py)
#ifdef TE_STUDY
__device__ long long g_rc_prof[8 * 16];
#define RC_MARK(i)                                                                  \
  do {                                                                              \
    if (blockIdx.x == 300 && (threadIdx.x & 63) == 0) {                             \
      g_rc_prof[(threadIdx.x >> 6) * 16 + (i)] = clock64();                         \
      if ((i) == 0) g_rc_prof[(threadIdx.x >> 6) * 16 + 15] = wall_clock64();       \
      if ((i) == 7) g_rc_prof[(threadIdx.x >> 6) * 16 + 14] = wall_clock64();       \
    }                                                                               \
  } while (0)
#else
#define RC_MARK(i) do { } while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(kT) void qk_rc_kernel(const float* __restrict__ Rnn, const float* __restrict__ Z,
                                                   const float* __restrict__ q, Strided qs, const float* __restrict__ k,
                                                   Strided ks, float* __restrict__ cam_q, Strided cqs,
                                                   float* __restrict__ cam_k, Strided cks, int H, int N, float scale,
                                                   const float* __restrict__ r_scale, int64_t r_scale_stride, int both,
                                                   const float* __restrict__ dO, const float* __restrict__ O, Strided os) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NS = (N + 15) >> 4, NB = (N + 31) >> 5;
  const size_t plane = (size_t)NS * 2 * kFrag;                         // one plane of one operand
  // both != 0 (N <= 208: 2 x 78 KB): the k^T AND the q^T planes are resident from the start, and no barrier separates the
  // phases -- every wave goes from its rows to its keys on its own
  unsigned char* const Pk = smem;
  unsigned char* const Pq = both ? smem + 3 * plane : smem;
  float* const rdv = reinterpret_cast<float*>(smem + (both ? 6 : 3) * plane);      // BWD: rowdots, [NS * 16 + 16]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
  const unsigned nn_bytes = (unsigned)N * (unsigned)N * 4u;
  const Rsrc r_rs = make_rsrc(Rnn + (int64_t)bh * N * N, nn_bytes), z_rs = make_rsrc(Z + (int64_t)bh * N * N, nn_bytes);
  const float* q_bh = q + (int64_t)b * qs.sb + (int64_t)h * qs.sh;
  const float* k_bh = k + (int64_t)b * ks.sb + (int64_t)h * ks.sh;
  const bool has_f = (MODE == RULE) && r_scale != nullptr;
  const float f = has_f ? r_scale[(int64_t)b * r_scale_stride] : 1.0f;
  const bool owner = wave < NB;                                        // wave-uniform

  RC_MARK(0);
  stage_planes<true>(Pk, k_bh, ks.sn, N, NS);                                // k^T planes: the row phase's A operand
  if (both) stage_planes<false>(Pq, q_bh, qs.sn, N, NS);                      // q^T planes: the column phase's
  RC_MARK(1);
  float rd_own = 0.0f;
  if constexpr (MODE == BWD) {
    if (dO != nullptr) {
      // The row dots without a pass over the N x N tensors (round 6):  sum_j attn[i][j] d_attn[i][j] = sum_j attn[i][j] sum_d d_out[i][d]
      // v[j][d] = sum_d d_out[i][d] out[i][d]  -- 64 products per row from the block's own forward output.  Four lanes per row,
      // sixteen products each in ascending d, the partials meet in a fixed butterfly: an order that depends on nothing.
      const float* dO_bh = dO + (int64_t)b * os.sb + (int64_t)h * os.sh;
      const float* O_bh = O + (int64_t)b * os.sb + (int64_t)h * os.sh;
      for (int r0 = 0; r0 < 32 * NB; r0 += kT / 4) {
        const int i = r0 + (threadIdx.x >> 2), c = threadIdx.x & 3;
        float part = 0.0f;
        if (i < N) {
          const float* a = dO_bh + (int64_t)i * os.sn + 16 * c;
          const float* o = O_bh + (int64_t)i * os.sn + 16 * c;
          f32x4 av[4], ov[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const f32x4_u*>(a + 4 * u), ov[u] = *reinterpret_cast<const f32x4_u*>(o + 4 * u);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) part = part + av[u][e] * ov[u][e];
        }
        part = part + __shfl_xor(part, 1, 64);
        part = part + __shfl_xor(part, 2, 64);
        if (c == 0 && i < 32 * NB) rdv[i] = part;
      }
    } else if (owner) {
      row_dots(rdv, r_rs, z_rs, N, NB, wave);                          // (both phases read rows < 32 NB only)
    }
  }
  __syncthreads();
  RC_MARK(2);

  // ---- row side: cam_q[i][d] = q[i][d] * (sum_j S[i][j] k[j][d]) * scale   (BWD: d_q = the sum) ----
  if (owner) {
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][e] = 0.0f;
    phase_rows<MODE>(acc, Pk, plane, r_rs, z_rs, N, NB, wave, f, has_f, scale, rdv);
    RC_MARK(3);
    const int i = wave * 32 + n;
    if (i < N) {
      const float* qrow = q_bh + (int64_t)i * qs.sn + 4 * hh;
      float* orow = cam_q + (int64_t)b * cqs.sb + (int64_t)h * cqs.sh + (int64_t)i * cqs.sn + 4 * hh;
      f32x4 x4[2][4];
      if constexpr (MODE == RULE) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) x4[mb][g] = *reinterpret_cast<const f32x4_u*>(qrow + 32 * mb + 8 * g);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = acc[mb][4 * g + c];
            o[c] = (MODE == RULE) ? (x4[mb][g][c] * v) * scale : v;
          }
          *reinterpret_cast<f32x4_u*>(orow + 32 * mb + 8 * g) = o;
        }
    }
  }
  RC_MARK(4);
  if (!both) {
    __syncthreads();                                                   // every wave is done with the k^T planes
    stage_planes<false>(Pq, q_bh, qs.sn, N, NS);                       // q^T planes: the column phase's A operand
    __syncthreads();
  }
  RC_MARK(5);

  // ---- column side: cam_k[j][d] = k[j][d] * (sum_i S[i][j] q[i][d]) * scale   (BWD: d_k = the sum) ----
  if (owner) {
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][e] = 0.0f;
    phase<MODE, COLS>(acc, Pq, plane, r_rs, z_rs, N, NS, wave, f, has_f, scale, rd_own, rdv);
    RC_MARK(6);
    const int j = wave * 32 + n;
    if (j < N) {
      const float* krow = k_bh + (int64_t)j * ks.sn + 4 * hh;
      float* orow = cam_k + (int64_t)b * cks.sb + (int64_t)h * cks.sh + (int64_t)j * cks.sn + 4 * hh;
      f32x4 x4[2][4];
      if constexpr (MODE == RULE) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) x4[mb][g] = *reinterpret_cast<const f32x4_u*>(krow + 32 * mb + 8 * g);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = acc[mb][4 * g + c];
            o[c] = (MODE == RULE) ? (x4[mb][g][c] * v) * scale : v;
          }
          *reinterpret_cast<f32x4_u*>(orow + 32 * mb + 8 * g) = o;
        }
    }
  }
  RC_MARK(7);
}

inline size_t lds_bytes(int64_t N, bool both) {
  const int64_t NS = (N + 15) >> 4;
  return (size_t)((both ? 6 : 3) * NS * 2 * kFrag) + (size_t)(NS * 16 + 16) * sizeof(float);
}
constexpr size_t kLdsMax = 160 * 1024;

}  // namespace

bool supported(int64_t B, int64_t H, int64_t N, int64_t D) {
  return D == 64 && N >= 1 && N <= kMaxN && B * H >= 1 && B * H <= 0x7fffffff;
}

// mode 0: the QK rule (Rnn = relevance of the scores, Z = the cached unscaled q k^T); mode 1: softmax backward (Rnn = d_attn,
// Z = attn; cam_q / cam_k receive d_q / d_k).  Views [B,H,N,64] with element strides; Rnn, Z contiguous [B*H,N,N].
// d_out / out (mode 1, optional): the attention block's output gradient and forward output as [B,H,N,64] views with the strides
// o_s*: the row dots of the softmax backward then come from them instead of a pass over d_attn and attn.
int qk_launch(int mode, const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb,
              int64_t k_sh, int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, float* cam_k,
              int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N, float scale, const float* r_scale,
              int64_t r_scale_stride, hipStream_t stream, const float* d_out, const float* out, int64_t o_sb, int64_t o_sh,
              int64_t o_sn) {
  if (!supported(B, H, N, 64)) return TE_ERR_UNSUPPORTED;
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
  const int both = lds_bytes(N, true) <= kLdsMax ? 1 : 0;
  const size_t lds = lds_bytes(N, both != 0);
  const dim3 grid((unsigned)(B * H));
  if (mode == RULE) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qk_rc_kernel<RULE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    qk_rc_kernel<RULE><<<grid, dim3(kT), lds, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks, (int)H, (int)N, scale, r_scale,
                                                       r_scale_stride, both, nullptr, nullptr, Strided{0, 0, 0});
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qk_rc_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    const bool from_out = d_out != nullptr && out != nullptr;
    qk_rc_kernel<BWD><<<grid, dim3(kT), lds, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks, (int)H, (int)N, scale, nullptr, 0,
                                                      both, from_out ? d_out : nullptr, from_out ? out : nullptr,
                                                      Strided{o_sb, o_sh, o_sn});
  }
  return TE_OK;
}

}  // namespace te_attn_rc

#ifdef TE_STUDY
extern "C" int te_attn_rc_profile(long long* host_out) {      // 8 waves x 16 stamps (measurement builds only)
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(te_attn_rc::g_rc_prof), sizeof(long long) * 8 * 16);
}
#endif
