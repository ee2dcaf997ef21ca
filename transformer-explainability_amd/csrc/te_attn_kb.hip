// te_attn_kb.hip -- attention relprop rules (modules/layers_ours.py:48-60,122-127 via ViT_LRP.py:154-177, BERT.py:367-393) and
// the attention-gradient producers on the same machinery, with WAVE-OWNED KEY BLOCKS (round 5).
//
// The one-pass kernels of te_attn_rules.hip move every [32, keys] tile of an N x N operand global -> registers -> LDS ->
// registers, all eight waves of the workgroup in lock-step through load / form-S / barrier / product / barrier phases:
// 15-23 vector instructions per MFMA, 45 % of a tile's cycles in MFMAs, 0.26-0.37 of the roofline for three rounds.
// Here the N x N operand NEVER touches LDS:
//
//   * wave w owns key block jb (32 keys) of its (b, h [, key group]) for the whole kernel.  A [32 rows x 32 keys] block of the
//     N x N operand arrives as four 16-byte-per-lane buffer loads (eight rows x 128 contiguous bytes each), requested two
//     tiles ahead, and changes to the MFMA ACCUMULATOR layout -- lane (lr, kh) holds column lr, rows
//     crow(e, kh) = (e & 3) + 8 (e >> 2) + 4 kh, e < 16 -- through a wave-private 4.6 KB LDS block (no barrier: a wave's LDS
//     instructions execute in order).  Because the contraction index of an MFMA may be visited in any order, that very
//     register set IS
//       - the A operand of the column-side product (K = query rows in accumulator order),
//       - the element-wise factor of the row-side result (cam_attn = attn . G lands in the same layout), and
//       - (back through the staging block) the layout the N x N result is stored from, as 16-byte pieces.
//   * the key-side operand of the row product (v of the wave's 32 keys) lives in registers for the whole kernel; only the
//     [32, 64] row-side tile (S = sd(R, Z), or d_out) is shared: THREE LDS buffers and an LDS arrival counter instead of a
//     barrier per tile -- no wave ever waits for another wave's MFMAs.
//   * the products run on bf16 MFMAs with every fp32 operand split into three bf16 planes, six partial products, fp32
//     accumulation (av6_kb_kernel, the shipped kernel: fp32 MFMA runs at the fp32 vector rate and its time ADDS to the
//     vector work of the SIMD -- av_kb_kernel, the fp32-MFMA version of the same structure, is kept for measurement builds).
//   * global loads and stores of the tile loop are inline asm hipcc's s_waitcnt insertion does not see, waited for with
//     hand-counted vmcnt; a register with such a load in flight must never be copied -- register sets alternate over a
//     two-tile loop body, and scripts/check_hidden_loads.py (tests/test_isa_hazards.py) checks the compiled ISA for it.
//   DESIGN.md, "Attention rules with wave-owned key blocks", has the measurements behind each of these choices.
//
// Reductions run in an order that depends on N only: a batch equals its samples run one by one, bit for bit.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "te_common.h"

namespace te_attn_kb {

namespace {

constexpr int TI = 32;         // query rows per tile
constexpr int kT = 512;        // threads per workgroup
constexpr int kWaves = kT / 64;
constexpr int XLD = 36;        // row stride (floats) of a wave's [32][32] staging block
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
// Loads hipcc's s_waitcnt insertion does not see either: across a loop back-edge it loses the age order of in-flight loads
// and waits vmcnt(0) at the first use of ANY of them -- a full drain of the prefetch pipeline once per tile.  The tile loop
// therefore issues all its global loads and stores as inline asm and waits with hand-counted s_waitcnt vmcnt(n), n = the
// number of YOUNGER LOADS in flight (stores are never counted: loads retire in order among themselves, and a store that
// retires late only makes the wait longer).  After the wait, TE_PIN makes the value's first use follow it in program order.
__device__ __forceinline__ f32x4 ld128_hidden(Rsrc r, unsigned voff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(r));
  return v;
}
#define TE_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TE_PIN(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ void st128_hidden(f32x4 x, Rsrc r, unsigned voff) {
  // (s_nop: a store of more than 64 bits reads the upper half of its data one cycle late -- the VALU instruction that follows
  //  must not write those registers.  hipcc pads this hazard for its own stores, not for inline asm: without the wait state
  //  cam_q came back with the upper 8 bytes of some lanes' 16-byte pieces replaced by whatever was written next, sporadically)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(x), "v"(voff), "s"(r));
}
__device__ __forceinline__ void st32_hidden(float x, Rsrc r, unsigned voff) {
  asm volatile("buffer_store_dword %0, %1, %2, 0 offen" : : "v"(x), "v"(voff), "s"(r));
}

// ------------------------------------------------------------------------------------------------
// AV rule (MODE RULE):  S = sd(R, Z) [N,64];  cam_attn = attn . (S v^T) * scale;  cam_v = v . (attn^T S) * scale
// attention backward, first half (MODE BWD):  d_attn = d_out v^T;  d_v = attn^T d_out          (R = d_out, Z unused)
// R, Z strided [B,H,N,64]; attn, cam_attn contiguous [B*H,N,N]; v, cam_v strided.
// grid = BH * ngroups (bh fastest); workgroup g of a (b, h) owns key blocks [g KBG, (g + 1) KBG), wave w block g KBG + w.
// ------------------------------------------------------------------------------------------------
// PROF (measurement builds: -DTE_STUDY, TE_ATTN_KB_PROF=1): workgroup 0 accumulates shader-clock cycles per wave and phase
#ifdef TE_STUDY
__device__ long long g_kb_prof[kWaves * 8];
#endif
#define KB_MARK(slot)                                                     \
  do {                                                                    \
    if constexpr (PROF) {                                                 \
      if (blockIdx.x == 0 && lane == 0) {                                 \
        const long long now__ = clock64();                                \
        prof_acc[slot] += now__ - tprev;                                  \
        tprev = now__;                                                    \
      }                                                                   \
    }                                                                     \
  } while (0)

// STUDY (measurement builds, TE_ATTN_KB_STUDY=n; results are garbage): 1 no N x N stores, 2 no N x N loads, 4 no row product,
// 5 no column product, 8 no MFMA at all, 9 no N x N loads and no stores
template <int MODE, bool PROF = false, int STUDY = 0>
__global__ __launch_bounds__(kT) void av_kb_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, Strided zs, const float* __restrict__ attn,
    const float* __restrict__ v, Strided vs, float* __restrict__ cam_attn, float* __restrict__ cam_v, Strided cs, int H,
    int N, int BH, int KBG, float scale) {
  // Row-side tiles in LDS: THREE buffers and an arrival counter instead of a barrier per tile.  S(k) lives in buffer k % 3; a
  // wave that has written its part of S(k) adds 1 to `arrived` (LDS operations of a wave execute in order: the add follows
  // its write); S(k) is complete at 8 (k + 1).  In iteration it every wave first waits for S(it) -- complete since the other
  // waves' previous iteration, so the poll normally falls through -- then writes its part of S(it + 1) over S(it - 2), whose
  // last readers (the MFMAs of iteration it - 2) every wave finished before it contributed to S(it).  No wave ever waits for
  // another wave's MFMAs: measured with s_barrier per tile (scripts/attn_kb_prof.py), the older wave of a SIMD ran its 64
  // MFMAs at full rate, the younger one's only started when those had finished (issue arbitration prefers the older wave,
  // whose dependent MFMA is always ready), and then BOTH waited at the barrier and formed the next S with the matrix pipe
  // idle: 12 500 cycles per tile for 8 200 cycles of MFMAs.
  __shared__ __attribute__((aligned(16))) float St[3][TI * SLD];
  __shared__ unsigned arrived;
  // Wave-private staging of the wave's [32 rows x 32 keys] blocks: the N x N operand is read and the N x N result written as
  // 16-byte pieces per lane -- lane l moves keys 4 (l & 7) .. + 3 of row 8 p + (l >> 3), four instructions per block, each
  // covering eight rows x 128 contiguous bytes -- and changes to / from the accumulator layout through LDS (no barrier: a
  // wave's LDS instructions execute in order).  Measured (profiles/r05_attention_av_kb_study.log): with one dword per lane
  // and instruction (16 + 16 instructions per block) the stores alone cost 57 of the kernel's 130 us at N = 197.
  __shared__ __attribute__((aligned(16))) float Xw[kWaves][2][TI * XLD];
  if (threadIdx.x == 0) arrived = 0;
  __syncthreads();
  const int bh = blockIdx.x % BH, g = blockIdx.x / BH;
  const int b = bh / H, h = bh - b * H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = PROF ? clock64() : 0;
  (void)prof_acc;
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
    rr = ld128_hidden(r_rs, r_off + (unsigned)it * r_tile);
    if constexpr (MODE == RULE) zz = ld128_hidden(z_rs, z_off + (unsigned)it * z_tile);
  };
  constexpr int kRz = (MODE == RULE) ? 2 : 1;      // loads per fetch_rz
  auto pin_rz = [&]() __attribute__((always_inline)) {
    TE_PIN(rr);
    if constexpr (MODE == RULE) TE_PIN(zz);
  };
  auto put_s = [&](int it) __attribute__((always_inline)) {
    f32x4 s = rr;                                                   // BWD: the tile of d_out itself
    if constexpr (MODE == RULE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = te_sd(rr[e], zz[e]);       // rows beyond N: sd(0, 0) = 0
    }
    *reinterpret_cast<f32x4*>(&St[it % 3][srow * SLD + (sc << 2)]) = s;
    // RELAXED atomics + compiler barriers, NOT release / acquire: data and counter both live in LDS, whose instructions a
    // wave executes in order, so the hardware needs nothing more -- while hipcc turns a workgroup-scope acquire (and a
    // __syncthreads()) into s_waitcnt vmcnt(0), i.e. a full drain of the wave's global-memory pipeline at every tile: the
    // prefetches just issued, two tiles ahead, and the stores (found in the ISA after the phase profile showed every wave
    // waiting ~2 500 cycles where it forms S: with the drain the kernel's time was memory time PLUS MFMA time)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's LDS writes have been PERFORMED before it arrives: a
                                                            // later single-lane atomic can overtake the tail of a 64-lane write
    if (lane == 0) __hip_atomic_fetch_add(&arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_s = [&](int it) __attribute__((always_inline)) {            // S(it) complete
    const unsigned target = (unsigned)kWaves * (unsigned)(it + 1);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  if (!has_blk) {
    // a wave without a key block (N = 197: wave 7) only helps to form the row-side tiles
    fetch_rz(0);
    TE_VM_WAIT(0);
    pin_rz();
    put_s(0);
    fetch_rz(1);
    for (int it = 0; it < ntiles; ++it) {
      wait_s(it);                                  // (throttle: S(it + 1) overwrites S(it - 2))
      TE_VM_WAIT(0);                               // R / Z of tile it + 1: this wave's only requests
      pin_rz();
      put_s(it + 1);
      fetch_rz(it + 2);
    }
    TE_VM_WAIT(0);
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
  float* const Ax = Xw[wave][0];                  // attn block: rows in, accumulator layout out
  float* const Gx = Xw[wave][1];                  // result block: accumulator layout in, rows out
  const int xr = lane >> 3, xc = (lane & 7) << 2;                  // this lane's row (+ 8 p) and first key of a 16-byte piece
  const unsigned lane_x4 = ((unsigned)xr * (unsigned)N + (unsigned)(kb * 32 + xc)) * 4u;
  auto fetch_attn = [&](int it, f32x4 (&dst)[4]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(it * TI) * row_bytes;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (STUDY != 2 && STUDY != 9) dst[p] = ld128_hidden(a_rs, lane_x4 + (base + (unsigned)(8 * p) * row_bytes));
      else dst[p] = f32x4{0.25f, 0.5f, 0.75f, 1.0f};
    }
  };
  auto to_acc = [&](const f32x4 (&src)[4], float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Ax + (8 * p + xr) * XLD + xc) = src[p];
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[e] = Ax[crow(e, kh) * XLD + lr];
  };

  // attn blocks are requested TWO tiles ahead (an2), so that the change of layout at a tile's end (an -> ac) never waits for
  // memory: with one tile of lookahead every wave's non-MFMA chain contained an HBM round trip, and a SIMD's two waves can
  // only cover each other's chains while those are shorter than an MFMA block (4 400 cycles)
  float ac[16];
  f32x4 an[4], an2[4];
  fetch_rz(0);
  fetch_attn(0, an2);
  TE_VM_WAIT(0);                                   // (prologue: hipcc's own loads of vf above included)
  pin_rz();
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) TE_PIN(vf[kg]);   // (hipcc's own wait for vf lands HERE, not at vf's first use inside the loop,
                                                   //  where the merged loop-header state would repeat it every iteration)
#pragma unroll
  for (int p = 0; p < 4; ++p) TE_PIN(an2[p]);
  put_s(0);
  fetch_rz(1);
  fetch_attn(1, an);
  to_acc(an2, ac);
  // In flight when the loop starts, oldest first: R / Z(1), attn(1) -- the order every iteration keeps: it issues the stores
  // of tile it - 1, then R / Z(it + 2), then attn(it + 2).

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
  const bool blk_full = kb * 32 + 32 <= N;         // wave-uniform: every key of the block exists (else: dword stores, masked)
  auto stage_g = [&]() __attribute__((always_inline)) {               // the tile's result -> rows (full key blocks)
    if (blk_full) {
#pragma unroll
      for (int e = 0; e < 16; ++e) Gx[crow(e, kh) * XLD + lr] = gp[e];
    }
  };
  auto store_g = [&](int it_prev) __attribute__((always_inline)) {    // (rows beyond N: dropped by the range check)
    const unsigned base = (unsigned)(it_prev * TI) * row_bytes;
    if (blk_full) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(Gx + (8 * p + xr) * XLD + xc);
        if constexpr (STUDY != 1 && STUDY != 9) st128_hidden(t, ca_rs, lane_x4 + (base + (unsigned)(8 * p) * row_bytes));
      }
    } else if (key_ok) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if constexpr (STUDY != 1 && STUDY != 9) st32_hidden(gp[e], ca_rs, lane_nn + (base + (unsigned)((e & 3) + 8 * (e >> 2)) * row_bytes));
    }
  };
  // FULL = all 32 rows exist (every tile but, for N % 32 != 0, the last)
  // (ax = the register set the attn block of tile it + 2 is requested into, ay = the set that holds tile it + 1's, requested
  // by the previous iteration: the callers alternate the two sets -- a register with a load in flight must not be copied)
  auto tile = [&](int it, f32x4 (&ax)[4], f32x4 (&ay)[4]) __attribute__((always_inline)) {
    const float* Sc = St[it % 3];
    KB_MARK(5);                                    // to_acc of the previous tile's end + loop overhead
    wait_s(it);
    KB_MARK(4);                                    // poll for S(it)
    // R / Z(it + 1) were requested a tile ago (iteration 0: in the prologue); younger loads in flight: attn(it + 1)
    if constexpr (STUDY != 2 && STUDY != 9) TE_VM_WAIT(4);
    else TE_VM_WAIT(0);
    pin_rz();
    put_s(it + 1);
    KB_MARK(6);                                    // S(it + 1): wait for R / Z, sd, LDS write, arrive
    if (it > 0) store_g(it - 1);
    fetch_rz(it + 2);
    fetch_attn(it + 2, ax);
    __builtin_amdgcn_sched_barrier(0);             // (hipcc otherwise sinks the requests to the END of the tile's MFMAs)
    KB_MARK(0);                                    // tile top: waits, S(it + 1), memory burst
    if (it < ntiles) {                             // (the pair loop runs one tile beyond an odd tile count: no products there)
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
          for (int jj = 0; jj < 4; ++jj) {
            if constexpr (STUDY != 4 && STUDY != 8) gacc = TE_MFMA32(a[kg][jj], vf[kg][jj], gacc);
            else gacc[jj] += a[kg][jj] * vf[kg][jj];
          }
      }
      KB_MARK(1);                                  // row product issued
      // ---- column side: cam_v += attn^T S (keys x 64), K = the 32 query rows in accumulator order (rows beyond N are zero
      // in S and in attn: no special case, the last tile of N = 197 spends 24 of its 64 MFMAs on them) ----
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const float* Yp = Sc + kh * 4 * SLD + db * 32 + lr;
        float bq[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bq[e] = Yp[((e & 3) + 8 * (e >> 2)) * SLD];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if constexpr (STUDY != 5 && STUDY != 8) accv[db] = TE_MFMA32(ac[e], bq[e], accv[db]);
          else accv[db][e] += ac[e] * bq[e];
        }
      }
      KB_MARK(2);                                  // column product issued
      // ---- the N x N result of this tile: stored at the top of the next one ----
#pragma unroll
      for (int e = 0; e < 16; ++e) gp[e] = (MODE == RULE) ? (ac[e] * gacc[e]) * scale : gacc[e];
      stage_g();
      KB_MARK(3);                                  // N x N result formed (the row product's MFMAs have finished)
    }
    // attn(it + 1) was requested a tile ago; younger loads in flight: R / Z(it + 2) and attn(it + 2) of this tile
    if constexpr (STUDY != 2 && STUDY != 9) {
      if constexpr (kRz == 2) TE_VM_WAIT(6);
      else TE_VM_WAIT(5);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) TE_PIN(ay[p]);
    to_acc(ay, ac);                                // the next tile's attn block
  };
  KB_MARK(7);                                      // prologue
  // ONE loop body for every tile (two tiles per trip, the register sets swapping roles): hipcc may copy a loop-carried
  // register where control flow forks -- a separate code path for the last tile read such a copy of a register whose load was
  // still in flight (wrong rows 192..196 at N = 197 until the tail was folded into the loop)
#pragma unroll 1
  for (int it = 0; it < ntiles; it += 2) {
    tile(it, an2, an);
    tile(it + 1, an, an2);
  }
  TE_VM_WAIT(0);                                   // (requests beyond the last tile: zeros, but their registers are in flight)
  if ((ntiles & 1) == 0) store_g(ntiles - 1);      // (odd tile count: the trip beyond the last tile has stored it)

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
#ifdef TE_STUDY
  if constexpr (PROF) {
    KB_MARK(7);                                    // prologue + last store burst + column epilogue (issue only)
    if (blockIdx.x == 0 && lane == 0)
      for (int q = 0; q < 8; ++q) g_kb_prof[wave * 8 + q] = prof_acc[q];
  }
#endif
}

// ================================================================================================
// The same kernel on bf16 MFMAs at fp32 accuracy ("x6", as te_linear_x6.hip): every fp32 operand is the exact sum of three
// bf16 planes, the six partial products above 2^-24 are kept (a1 b1 + a0 b2 + a2 b0 + a0 b1 + a1 b0 + a0 b0, smallest first),
// fp32 accumulation.  Why here: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and its time ADDS to the vector-ALU time
// of the SIMD's waves (per-wave phase profile, profiles/r05_attention_av_kb_phase_profile.log: MFMA blocks at full rate,
// every other phase 2-3x longer while the partner wave multiplies; tile = 8 200 MFMA cycles + ~4 000 others), while the
// bf16 matrix pipe is a separate unit: 48 MFMAs of 32 cycles per tile and wave instead of 64 of 64, beside ~300 vector
// instructions (sd, the three-way splits, the result's factor).
//   row product   G = S v^T:  A = S planes [row][d] from LDS (16-byte fragments, K = d in four steps of 16), B = v planes of
//                 the wave's 32 keys, resident in 48 registers
//   column product  cam_v += attn^T S:  A = the attn block's planes, split in registers from the accumulator-layout values
//                 (K step s covers rows crow(8 s .. 8 s + 7, kh): the contraction order is free), B = S planes TRANSPOSED
//                 [d][row position] from LDS, the row positions permuted to that order (pos = row with bits 2 and 3 swapped)
// The S producer (all 512 threads, one float4 each) writes both plane images of its tile.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define TE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// x = p[0] + p[1] + p[2] exactly (te_linear_x6.hip: split3_pk); p[q] = the packed pair (x0 low half, x1 high half) of plane q
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned (&p)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    p[q] = u;
    x0 = x0 - __builtin_bit_cast(float, u << 16);
    x1 = x1 - __builtin_bit_cast(float, u & 0xffff0000u);
  }
}
// eight consecutive K values -> one MFMA operand fragment per plane
__device__ __forceinline__ void split3_x8(const float (&x)[8], bf16x8 (&pl)[3]) {
  unsigned w[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3_pk(x[2 * i], x[2 * i + 1], w[i]);
#pragma unroll
  for (int q = 0; q < 3; ++q) pl[q] = __builtin_bit_cast(bf16x8, u32x4{w[0][q], w[1][q], w[2][q], w[3][q]});
}
// acc += a b with the six partial products, smallest first
__device__ __forceinline__ void mfma_x6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  acc = TE_MFMA_BF16(a[1], b[1], acc);
  acc = TE_MFMA_BF16(a[0], b[2], acc);
  acc = TE_MFMA_BF16(a[2], b[0], acc);
  acc = TE_MFMA_BF16(a[0], b[1], acc);
  acc = TE_MFMA_BF16(a[1], b[0], acc);
  acc = TE_MFMA_BF16(a[0], b[0], acc);
}

constexpr int kRLD = 144;                  // bytes per row of a row-major S plane [32 rows][64 d] (128 + 16: conflict-free b128)
constexpr int kTLD = 80;                   // bytes per row of a transposed S plane [64 d][32 row positions] (64 + 16)
constexpr int kPlR = TI * kRLD;            // 4 608
constexpr int kPlT = 64 * kTLD;            // 5 120
constexpr int kSBuf = 3 * kPlR + 3 * kPlT; // 29 184 bytes per S tile: both images, three planes each

template <int MODE, bool PROF = false>
__global__ __launch_bounds__(kT) void av6_kb_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, Strided zs, const float* __restrict__ attn,
    const float* __restrict__ v, Strided vs, float* __restrict__ cam_attn, float* __restrict__ cam_v, Strided cs, int H,
    int N, int BH, int KBG, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char Sb[3][kSBuf];       // S(k) lives in buffer k % 3 (see av_kb_kernel)
  __shared__ __attribute__((aligned(16))) float Xw[kWaves][TI * XLD];      // wave-private: layout changes of the N x N blocks
  __shared__ unsigned arrived;
  if (threadIdx.x == 0) arrived = 0;
  __syncthreads();
  const int bh = blockIdx.x % BH, g = blockIdx.x / BH;
  const int b = bh / H, h = bh - b * H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = PROF ? clock64() : 0;
  (void)prof_acc;
  const int nkb = (N + 31) >> 5, kb = g * KBG + wave;
  const bool has_blk = wave < KBG && kb < nkb;            // wave-uniform
  const int j = kb * 32 + lr;                             // this lane's key
  const int ntiles = (N + TI - 1) / TI;
  const int srow = threadIdx.x >> 4, sc = threadIdx.x & 15;      // this thread's float4 of the [32][64] row-side tile
  const int spos = (srow & 0x13) | ((srow & 4) << 1) | ((srow & 8) >> 1);      // its row's position in the transposed image

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
    rr = ld128_hidden(r_rs, r_off + (unsigned)it * r_tile);
    if constexpr (MODE == RULE) zz = ld128_hidden(z_rs, z_off + (unsigned)it * z_tile);
  };
  constexpr int kRz = (MODE == RULE) ? 2 : 1;
  auto pin_rz = [&]() __attribute__((always_inline)) {
    TE_PIN(rr);
    if constexpr (MODE == RULE) TE_PIN(zz);
  };
  // S(it) of this thread's four elements -> both plane images of buffer it % 3, then arrive (see av_kb_kernel::put_s)
  auto put_s = [&](int it) __attribute__((always_inline)) {
    f32x4 s = rr;
    if constexpr (MODE == RULE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = te_sd(rr[e], zz[e]);       // rows beyond N: sd(0, 0) = 0
    }
    unsigned p01[3], p23[3];
    split3_pk(s[0], s[1], p01);
    split3_pk(s[2], s[3], p23);
    unsigned char* base = Sb[it % 3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      *reinterpret_cast<u32x2*>(base + q * kPlR + srow * kRLD + sc * 8) = u32x2{p01[q], p23[q]};
      unsigned char* t = base + 3 * kPlR + q * kPlT + (4 * sc) * kTLD + spos * 2;
      *reinterpret_cast<unsigned short*>(t) = (unsigned short)(p01[q] & 0xffffu);
      *reinterpret_cast<unsigned short*>(t + kTLD) = (unsigned short)(p01[q] >> 16);
      *reinterpret_cast<unsigned short*>(t + 2 * kTLD) = (unsigned short)(p23[q] & 0xffffu);
      *reinterpret_cast<unsigned short*>(t + 3 * kTLD) = (unsigned short)(p23[q] >> 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's LDS writes have been PERFORMED before it arrives: a
                                                            // later single-lane atomic can overtake the tail of a 64-lane write
    if (lane == 0) __hip_atomic_fetch_add(&arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_s = [&](int it) __attribute__((always_inline)) {            // S(it) complete
    const unsigned target = (unsigned)kWaves * (unsigned)(it + 1);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  if (!has_blk) {
    fetch_rz(0);
    TE_VM_WAIT(0);
    pin_rz();
    put_s(0);
    fetch_rz(1);
    for (int it = 0; it < ntiles; ++it) {
      wait_s(it);                                  // (throttle: S(it + 1) overwrites S(it - 2))
      TE_VM_WAIT(0);
      pin_rz();
      put_s(it + 1);
      fetch_rz(it + 2);
    }
    TE_VM_WAIT(0);
    return;
  }

  // key-side operand of the row product, resident: the planes of v[j][16 s + 8 kh + 0..7], s = 0..3 (keys beyond N: zeros)
  bf16x8 vpl[4][3];
  {
    const unsigned off = ((unsigned)j * (unsigned)vs.sn + 8u * kh) * 4u;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x4 lo = ld128(v_rs, off + 64u * s4), hi = ld128(v_rs, off + 64u * s4 + 16u);
      const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      split3_x8(x, vpl[s4]);
    }
  }
  float* const Xb = Xw[wave];
  const int xr = lane >> 3, xc = (lane & 7) << 2;                  // this lane's row (+ 8 p) and first key of a 16-byte piece
  const unsigned lane_x4 = ((unsigned)xr * (unsigned)N + (unsigned)(kb * 32 + xc)) * 4u;
  const unsigned lane_nn = ((unsigned)(4 * kh) * (unsigned)N + (unsigned)j) * 4u;
  const bool key_ok = j < N;
  const bool blk_full = kb * 32 + 32 <= N;         // wave-uniform: every key of the block exists (else: dword stores, masked)
  auto fetch_attn = [&](int it, f32x4 (&dst)[4]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(it * TI) * row_bytes;
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[p] = ld128_hidden(a_rs, lane_x4 + (base + (unsigned)(8 * p) * row_bytes));
  };
  auto to_acc = [&](const f32x4 (&src)[4], float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Xb + (8 * p + xr) * XLD + xc) = src[p];
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[e] = Xb[crow(e, kh) * XLD + lr];
  };

  float ac[16];
  f32x4 an[4], an2[4];
  fetch_rz(0);
  fetch_attn(0, an2);
  TE_VM_WAIT(0);                                   // (prologue: hipcc's own loads of v above included)
  pin_rz();
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
    for (int q = 0; q < 3; ++q) TE_PIN(vpl[s4][q]);
#pragma unroll
  for (int p = 0; p < 4; ++p) TE_PIN(an2[p]);
  put_s(0);
  fetch_rz(1);
  fetch_attn(1, an);
  to_acc(an2, ac);
  // In flight when the loop starts, oldest first: R / Z(1), attn(1).  Every iteration issues R / Z(it + 2), attn(it + 2) at
  // its top and the stores of its own result at its end.

  f32x16 accv[2];
  zero16(accv[0]);
  zero16(accv[1]);
  auto tile = [&](int it, f32x4 (&ax)[4], f32x4 (&ay)[4]) __attribute__((always_inline)) {
    const unsigned char* Sc = Sb[it % 3];
    KB_MARK(5);                                    // to_acc of the previous tile's end + loop overhead
    wait_s(it);
    KB_MARK(4);                                    // poll for S(it)
    TE_VM_WAIT(4);                                 // R / Z(it + 1): younger loads in flight = attn(it + 1)
    pin_rz();
    put_s(it + 1);
    KB_MARK(6);                                    // S(it + 1): wait for R / Z, sd, split, LDS writes, arrive
    fetch_rz(it + 2);
    fetch_attn(it + 2, ax);
    __builtin_amdgcn_sched_barrier(0);
    KB_MARK(0);                                    // requests
    if (it < ntiles) {                             // (the pair loop runs one tile beyond an odd tile count: no products there)
      // ---- row side: G = S v^T for this wave's key block ----
      f32x16 gacc;
      zero16(gacc);
      {
        const unsigned char* Ab = Sc + lr * kRLD + 16 * kh;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          bf16x8 a[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const bf16x8*>(Ab + q * kPlR + 32 * s4);
          mfma_x6(gacc, a, vpl[s4]);
        }
      }
      KB_MARK(1);                                  // row product issued
      // ---- the N x N result of this tile: cam_attn = attn . G straight from the accumulators, to rows through LDS, stored as
      // 16-byte pieces (full key blocks), else dword stores masked by key ----
      float gp[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) gp[e] = (MODE == RULE) ? (ac[e] * gacc[e]) * scale : gacc[e];
      const unsigned base = (unsigned)(it * TI) * row_bytes;
      if (blk_full) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Xb[crow(e, kh) * XLD + lr] = gp[e];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(Xb + (8 * p + xr) * XLD + xc);
          st128_hidden(t, ca_rs, lane_x4 + (base + (unsigned)(8 * p) * row_bytes));
        }
      } else if (key_ok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st32_hidden(gp[e], ca_rs, lane_nn + (base + (unsigned)((e & 3) + 8 * (e >> 2)) * row_bytes));
      }
      KB_MARK(3);                                  // result formed, staged, stored
      // ---- column side: cam_v += attn^T S (keys x 64); K step s = rows crow(8 s .. 8 s + 7, kh) ----
      bf16x8 apl[2][3];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float x[8] = {ac[8 * s2], ac[8 * s2 + 1], ac[8 * s2 + 2], ac[8 * s2 + 3],
                            ac[8 * s2 + 4], ac[8 * s2 + 5], ac[8 * s2 + 6], ac[8 * s2 + 7]};
        split3_x8(x, apl[s2]);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const unsigned char* Bb = Sc + 3 * kPlR + (db * 32 + lr) * kTLD + 16 * kh;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 bq[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const bf16x8*>(Bb + q * kPlT + 32 * s2);
          mfma_x6(accv[db], apl[s2], bq);
        }
      }
    }
    KB_MARK(2);                                    // attn split + column product issued
    // attn(it + 1) was requested a tile ago; younger loads in flight: R / Z(it + 2) and attn(it + 2) of this tile
    if constexpr (kRz == 2) TE_VM_WAIT(6);
    else TE_VM_WAIT(5);
#pragma unroll
    for (int p = 0; p < 4; ++p) TE_PIN(ay[p]);
    to_acc(ay, ac);                                // the next tile's attn block (the staging block is free: its rows were read)
  };
  KB_MARK(7);
#pragma unroll 1
  for (int it = 0; it < ntiles; it += 2) {
    tile(it, an2, an);
    tile(it + 1, an, an2);
  }
  TE_VM_WAIT(0);                                   // (requests beyond the last tile: zeros, but their registers are in flight)

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
#ifdef TE_STUDY
  if constexpr (PROF) {
    KB_MARK(7);
    if (blockIdx.x == 0 && lane == 0)
      for (int q = 0; q < 8; ++q) g_kb_prof[wave * 8 + q] = prof_acc[q];
  }
#endif
}

#ifdef TE_STUDY      // (a study, see te_attn_rules.hip: use_kb_qk)
// ================================================================================================
// QK rule on the same machinery (einsum 'bhid,bhjd->bhij', layers_ours.py:48-60,122-127; ViT_LRP.py:165-173, BERT.py:386-393):
//   S = sd(R f, Z) [N,N];  cam_q = q . (S k) * scale;  cam_k = k . (S^T q) * scale          (f: the deferred factor of BERT's mask Add)
// R, Z contiguous [B*H,N,N]; q, k, cam_q, cam_k strided [B,H,N,64].  Wave w owns key block kb: its [32 rows x 32 keys] blocks of
// R and Z arrive as 16-byte pieces, change to the accumulator layout through LDS, S is formed in registers and is
//   - split along the ROWS (K of the column product cam_k += S^T q: A operand; B = the q tile's planes, transposed image in LDS),
//   - staged as rows and split along the KEYS (K of the row product  S k  over the wave's 32 keys: A operand; B = the planes of
//     k^T of the wave's keys, resident in 48 registers).
// The row product of a wave is a PARTIAL sum over its 32 keys: the partials meet in LDS (one [32][64] fp32 slab per wave) and
// every thread folds its four outputs over the waves in wave order -- an order that depends on N only.  Three LDS counters
// order the exchange (q planes produced / partials written / partials read); no barrier.  More than eight key blocks
// (N > 256): each workgroup of a (b, h) writes its unscaled partial of cam_q to `qpart`, folded by qk_finish_kernel in group order.
// ================================================================================================
constexpr int PLD = 68;                    // row stride (floats) of a wave's partial slab [32 rows][64 d]

template <int MODE, bool PROF = false>
__global__ __launch_bounds__(kT) void qk6_kb_kernel(
    const float* __restrict__ Rnn, const float* __restrict__ Znn, const float* __restrict__ q, Strided qs,
    const float* __restrict__ k, Strided ks, float* __restrict__ cam_q, Strided cqs, float* __restrict__ cam_k, Strided cks,
    float* __restrict__ qpart, int H, int N, int BH, int KBG, int ngroups, float scale, const float* __restrict__ r_scale,
    int64_t r_scale_stride) {
  static_assert(MODE == RULE, "the softmax-backward mode runs on te_attn_rules.hip's kernel");
  __shared__ __attribute__((aligned(16))) unsigned char Qb[3][3 * kPlT];    // q tile k lives in buffer k % 3: transposed planes
  __shared__ __attribute__((aligned(16))) float Xw[kWaves][TI * XLD];      // wave-private: layout changes
  __shared__ __attribute__((aligned(16))) float Pred[kWaves][TI * PLD];    // per-wave partials of the row product
  __shared__ unsigned cnt[4];                                              // [0] q planes  [1] partials written  [2] read
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int bh = blockIdx.x % BH, g = blockIdx.x / BH;
  const int b = bh / H, h = bh - b * H;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = PROF ? clock64() : 0;
  (void)prof_acc;
  const int nkb = (N + 31) >> 5, kb = g * KBG + wave;
  const bool has_blk = wave < KBG && kb < nkb;            // wave-uniform
  const int nwb = min(KBG, nkb - g * KBG);                // waves of this workgroup that own a key block
  const int j = kb * 32 + lr;
  const int ntiles = (N + TI - 1) / TI;
  const int niter = (ntiles + 1) & ~1;                    // the pair loop's trip count (one tile beyond an odd count: zeros)
  const int srow = threadIdx.x >> 4, sc = threadIdx.x & 15;
  const int spos = (srow & 0x13) | ((srow & 4) << 1) | ((srow & 8) >> 1);
  const float f = r_scale ? r_scale[(int64_t)b * r_scale_stride] : 1.0f;

  const unsigned nn_bytes = (unsigned)N * (unsigned)N * 4u;
  const Rsrc r_rs = make_rsrc(Rnn + (int64_t)bh * N * N, nn_bytes);
  const Rsrc z_rs = make_rsrc(Znn + (int64_t)bh * N * N, nn_bytes);
  const Rsrc q_rs = make_rsrc(q + (int64_t)b * qs.sb + (int64_t)h * qs.sh, view_bytes(N, qs.sn));
  const Rsrc k_rs = make_rsrc(k + (int64_t)b * ks.sb + (int64_t)h * ks.sh, view_bytes(N, ks.sn));
  const Rsrc cq_rs = (ngroups == 1) ? make_rsrc(cam_q + (int64_t)b * cqs.sb + (int64_t)h * cqs.sh, view_bytes(N, cqs.sn))
                                    : make_rsrc(qpart + ((int64_t)g * BH + bh) * N * 64, (unsigned)N * 256u);
  const unsigned cq_row = (ngroups == 1) ? (unsigned)cqs.sn * 4u : 256u;
  const Rsrc ck_rs = make_rsrc(cam_k + (int64_t)b * cks.sb + (int64_t)h * cks.sh, view_bytes(N, cks.sn));
  const unsigned row_bytes = (unsigned)N * 4u;

  auto arrive = [&](int c) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's LDS writes have been PERFORMED before it arrives: a
                                                            // later single-lane atomic can overtake the tail of a 64-lane write
    if (lane == 0) __hip_atomic_fetch_add(&cnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_for = [&](int c, unsigned target) __attribute__((always_inline)) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&cnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  // ---- the q tile: every thread one float4; q(it) -> transposed planes of buffer it % 3 (B operand of the column product) ----
  // Two register sets (qa / qb) alternate between "requested" and "landed", exactly like the R / Z sets, and the tile that
  // cam_q needs at the END of a tile (qcur) is taken from a landed set by explicit v_mov: with one variable and plain copies
  // (round 5's first version) hipcc merged the copy with the in-flight value's phi and placed moves of v[184:187] BEFORE
  // the hand-written s_waitcnt and on the loop back-edge -- reads of a register whose load had usually, not always, landed
  // (2 of 10 graph replays differed in one sample's last bits; scripts/check_hidden_loads.py finds such moves in the ISA).
  f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f}, qcur = {0.f, 0.f, 0.f, 0.f};
  const unsigned q_off = ((unsigned)srow * (unsigned)qs.sn + 4u * sc) * 4u, q_tile = (unsigned)TI * (unsigned)qs.sn * 4u;
  auto fetch_q = [&](int it, f32x4& dst) __attribute__((always_inline)) { dst = ld128_hidden(q_rs, q_off + (unsigned)it * q_tile); };
  auto keep_q = [&](const f32x4& src) __attribute__((always_inline)) {      // qcur = src (landed), a copy hipcc cannot fold
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t;
      asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(src[e]));
      qcur[e] = t;
    }
  };
  auto put_q = [&](int it, const f32x4& qq) __attribute__((always_inline)) {      // (rows beyond N: zeros)
    unsigned p01[3], p23[3];
    split3_pk(qq[0], qq[1], p01);
    split3_pk(qq[2], qq[3], p23);
    unsigned char* base = Qb[it % 3];
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      unsigned char* t = base + pq * kPlT + (4 * sc) * kTLD + spos * 2;
      *reinterpret_cast<unsigned short*>(t) = (unsigned short)(p01[pq] & 0xffffu);
      *reinterpret_cast<unsigned short*>(t + kTLD) = (unsigned short)(p01[pq] >> 16);
      *reinterpret_cast<unsigned short*>(t + 2 * kTLD) = (unsigned short)(p23[pq] & 0xffffu);
      *reinterpret_cast<unsigned short*>(t + 3 * kTLD) = (unsigned short)(p23[pq] >> 16);
    }
    arrive(0);
  };
  // ---- fold the partials of tile `it` (this thread: row srow, features 4 sc .. + 3), in wave order ----
  auto reduce_tile = [&](int it) __attribute__((always_inline)) {
    wait_for(1, (unsigned)nwb * (unsigned)(it + 1));
    f32x4 sum = *reinterpret_cast<const f32x4*>(&Pred[0][srow * PLD + 4 * sc]);
    for (int w = 1; w < nwb; ++w) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&Pred[w][srow * PLD + 4 * sc]);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] = sum[e] + t[e];
    }
    arrive(2);
    if (ngroups == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] = (qcur[e] * sum[e]) * scale;
    }
    st128_hidden(sum, cq_rs, ((unsigned)(it * TI + srow) * cq_row) + 16u * sc);      // rows beyond N: dropped
  };

  if (!has_blk) {
    // a wave without a key block: forms its part of the q planes and folds its share of the partials
    fetch_q(0, qb);
    TE_VM_WAIT(0);
    TE_PIN(qb);
    put_q(0, qb);
    keep_q(qb);
    fetch_q(1, qa);
    // (qx: the set tile it + 2 is requested into, qy: the set holding tile it + 1)
    auto idle_tile = [&](int it, f32x4& qx, f32x4& qy) __attribute__((always_inline)) {
      wait_for(0, (unsigned)kWaves * (unsigned)(it + 1));      // (throttle: q(it + 1) overwrites q(it - 2))
      TE_VM_WAIT(0);
      TE_PIN(qy);
      put_q(it + 1, qy);
      fetch_q(it + 2, qx);
      reduce_tile(it);
      keep_q(qy);
    };
#pragma unroll 1
    for (int it = 0; it < niter; it += 2) {
      idle_tile(it, qb, qa);
      idle_tile(it + 1, qa, qb);
    }
    TE_VM_WAIT(0);
    return;
  }

  // B operand of the row product, resident: the planes of k[16 s + 8 kh + 0..7][32 db + lr] of the wave's 32 keys
  bf16x8 kpl[2][2][3];
  {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          x[i] = ld32(k_rs, ((unsigned)(kb * 32 + 16 * s2 + 8 * kh + i) * (unsigned)ks.sn + (unsigned)(db * 32 + lr)) * 4u);
        split3_x8(x, kpl[db][s2]);
      }
  }
  float* const Xb = Xw[wave];
  const int xr = lane >> 3, xc = (lane & 7) << 2;
  const unsigned lane_x4 = ((unsigned)xr * (unsigned)N + (unsigned)(kb * 32 + xc)) * 4u;
  auto fetch_nn = [&](Rsrc rs_, int it, f32x4 (&dst)[4]) __attribute__((always_inline)) {
    const unsigned base = (unsigned)(it * TI) * row_bytes;
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[p] = ld128_hidden(rs_, lane_x4 + (base + (unsigned)(8 * p) * row_bytes));
  };
  auto to_acc = [&](const f32x4 (&src)[4], float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Xb + (8 * p + xr) * XLD + xc) = src[p];
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[e] = Xb[crow(e, kh) * XLD + lr];
  };

  float rc[16], zc[16];                            // this tile's R and Z blocks, accumulator layout
  f32x4 ra[4], za[4], rb[4], zb[4];                // two register sets for the blocks in flight
  fetch_q(0, qb);
  fetch_nn(r_rs, 0, rb);
  fetch_nn(z_rs, 0, zb);
  TE_VM_WAIT(0);                                   // (prologue: hipcc's own loads of k above included)
  TE_PIN(qb);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int pq = 0; pq < 3; ++pq) TE_PIN(kpl[db][s2][pq]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    TE_PIN(rb[p]);
    TE_PIN(zb[p]);
  }
  put_q(0, qb);
  keep_q(qb);
  fetch_q(1, qa);
  fetch_nn(r_rs, 1, ra);
  fetch_nn(z_rs, 1, za);
  to_acc(rb, rc);
  to_acc(zb, zc);
  // In flight when the loop starts, oldest first: q(1), R(1), Z(1).  Every iteration issues q(it + 2), R(it + 2), Z(it + 2) at
  // its top and the store of its share of cam_q at its end.

  f32x16 acck[2];
  zero16(acck[0]);
  zero16(acck[1]);
  // (qx / rx / zx: the sets tile it + 2 is requested into, qy / ry / zy: the sets holding tile it + 1)
  auto tile = [&](int it, f32x4& qx, f32x4 (&rx)[4], f32x4 (&zx)[4], f32x4& qy, f32x4 (&ry)[4], f32x4 (&zy)[4])
                  __attribute__((always_inline)) {
    const unsigned char* Qc = Qb[it % 3];
    wait_for(0, (unsigned)kWaves * (unsigned)(it + 1));      // the q planes of tile it
    TE_VM_WAIT(8);                                 // q(it + 1): younger loads in flight = R(it + 1), Z(it + 1)
    TE_PIN(qy);
    put_q(it + 1, qy);
    fetch_q(it + 2, qx);
    fetch_nn(r_rs, it + 2, rx);
    fetch_nn(z_rs, it + 2, zx);
    __builtin_amdgcn_sched_barrier(0);
    KB_MARK(0);                                    // poll q, wait, q planes, requests
    // ---- S = sd(R f, Z) in the accumulator layout (rows / keys beyond N: sd(0, 0) = 0) ----
    float sv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sv[e] = te_sd(rc[e] * f, zc[e]);
    // rows of S for the row product (staged before the column product: its LDS round trip hides under those MFMAs)
#pragma unroll
    for (int e = 0; e < 16; ++e) Xb[crow(e, kh) * XLD + lr] = sv[e];
    KB_MARK(1);                                    // sd, S rows staged
    // ---- column side: cam_k += S^T q (keys x 64); K step s = rows crow(8 s .. 8 s + 7, kh) ----
    {
      bf16x8 spl[2][3];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float x[8] = {sv[8 * s2], sv[8 * s2 + 1], sv[8 * s2 + 2], sv[8 * s2 + 3],
                            sv[8 * s2 + 4], sv[8 * s2 + 5], sv[8 * s2 + 6], sv[8 * s2 + 7]};
        split3_x8(x, spl[s2]);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const unsigned char* Bb = Qc + (db * 32 + lr) * kTLD + 16 * kh;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 bq[3];
#pragma unroll
          for (int pq = 0; pq < 3; ++pq) bq[pq] = *reinterpret_cast<const bf16x8*>(Bb + pq * kPlT + 32 * s2);
          mfma_x6(acck[db], spl[s2], bq);
        }
      }
    }
    KB_MARK(2);                                    // split + column product
    // ---- row side, this wave's 32 keys: P = S k; K step s = keys 16 s + 8 kh + 0..7 ----
    f32x16 pacc[2];
    zero16(pacc[0]);
    zero16(pacc[1]);
    {
      bf16x8 rpl[2][3];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Xb + lr * XLD + 16 * s2 + 8 * kh);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Xb + lr * XLD + 16 * s2 + 8 * kh + 4);
        const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        split3_x8(x, rpl[s2]);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) mfma_x6(pacc[db], rpl[s2], kpl[db][s2]);
    }
    KB_MARK(3);                                    // rows, split, row product
    // ---- the partials meet: wait until the previous tile's have been read, publish, fold this thread's outputs ----
    wait_for(2, (unsigned)kWaves * (unsigned)it);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) Pred[wave][crow(e, kh) * PLD + db * 32 + lr] = pacc[db][e];
    arrive(1);
    KB_MARK(4);                                    // wait for the readers of the previous tile, publish
    reduce_tile(it);
    keep_q(qy);
    KB_MARK(5);                                    // wait for the partials, fold, store
    // R / Z(it + 1) were requested a tile ago; younger loads in flight: q, R, Z of tile it + 2
    TE_VM_WAIT(9);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      TE_PIN(ry[p]);
      TE_PIN(zy[p]);
    }
    to_acc(ry, rc);
    to_acc(zy, zc);
    KB_MARK(6);                                    // wait for R / Z of the next tile, change of layout
  };
  KB_MARK(7);
#pragma unroll 1
  for (int it = 0; it < niter; it += 2) {
    tile(it, qb, rb, zb, qa, ra, za);
    tile(it + 1, qa, ra, za, qb, rb, zb);
  }
  TE_VM_WAIT(0);

  // ---- column epilogue: acck[db][e] = (S^T q)[key = 32 kb + crow(e, kh)][d = 32 db + lr]; keys beyond N: dropped ----
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const unsigned d4 = (unsigned)(db * 32 + lr) * 4u;
    float x[16];
    const unsigned xoff = (unsigned)(kb * 32 + 4 * kh) * (unsigned)ks.sn * 4u + d4;
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = ld32(k_rs, xoff + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)ks.sn * 4u);
    const unsigned ooff = (unsigned)(kb * 32 + 4 * kh) * (unsigned)cks.sn * 4u + d4;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      st32((x[e] * acck[db][e]) * scale, ck_rs, ooff + (unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)cks.sn * 4u);
  }
#ifdef TE_STUDY
  if constexpr (PROF) {
    KB_MARK(7);
    if (blockIdx.x == 0 && lane == 0)
      for (int qi = 0; qi < 8; ++qi) g_kb_prof[wave * 8 + qi] = prof_acc[qi];
  }
#endif
}

#endif      // TE_STUDY

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
  // 32-bit row offsets inside a (b, h) view, and the hardware range check that drops the rows >= N of a partial key block needs
  // a row stride of at least the 64 floats a row holds: every strided view of the launch
  auto stride_ok = [](int64_t sn) { return sn >= 64 && sn <= 65536; };
  if (!stride_ok(r_sn) || !stride_ok(v_sn) || !stride_ok(cv_sn) || (mode == 0 && !stride_ok(z_sn))) return TE_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(BH * ng)), blk(kT);
  const Strided rs{r_sb, r_sh, r_sn}, zs{z_sb, z_sh, z_sn}, vs{v_sb, v_sh, v_sn}, cs{cv_sb, cv_sh, cv_sn};
#ifdef TE_STUDY
  if (mode == 0) {
    const char* se = getenv("TE_ATTN_KB_STUDY");
    const int st = se ? atoi(se) : 0;
#define TE_KB_ST(n) if (st == n) { av_kb_kernel<RULE, false, n><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale); return TE_OK; }
    TE_KB_ST(1) TE_KB_ST(2) TE_KB_ST(4) TE_KB_ST(5) TE_KB_ST(8) TE_KB_ST(9)
#undef TE_KB_ST
    const char* e = getenv("TE_ATTN_KB_PROF");
    const char* av = getenv("TE_ATTN_AV");
    if (e && atoi(e) == 1 && av && !strcmp(av, "fp32kb")) {
      av_kb_kernel<RULE, true><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale);
      return TE_OK;
    }
  }
#endif
#ifdef TE_STUDY      // the fp32-MFMA version of this structure (TE_ATTN_AV=fp32kb): measurement builds only
  if (const char* e = getenv("TE_ATTN_AV"); e && !strcmp(e, "fp32kb")) {
    if (mode == 0)
      av_kb_kernel<RULE><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale);
    else
      av_kb_kernel<BWD><<<grid, blk, 0, stream>>>(R, rs, nullptr, Strided{0, 0, 0}, attn, v, vs, cam_attn, cam_v, cs, (int)H,
                                                  (int)N, BH, kbg, 1.0f);
    return TE_OK;
  }
  if (mode == 0) {
    const char* e = getenv("TE_ATTN_KB_PROF");
    if (e && atoi(e) == 1) {
      av6_kb_kernel<RULE, true><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale);
      return TE_OK;
    }
  }
#endif
  if (mode == 0)
    av6_kb_kernel<RULE><<<grid, blk, 0, stream>>>(R, rs, Z, zs, attn, v, vs, cam_attn, cam_v, cs, (int)H, (int)N, BH, kbg, scale);
  else
    av6_kb_kernel<BWD><<<grid, blk, 0, stream>>>(R, rs, nullptr, Strided{0, 0, 0}, attn, v, vs, cam_attn, cam_v, cs, (int)H,
                                                 (int)N, BH, kbg, 1.0f);
  return TE_OK;
}

// the QK rule (study builds only); *ngroups_out = workgroups per (b, h): with more than one, cam_q's per-group partials are in qpart
// [ngroups][B*H][N][64] (unscaled) and the caller runs its finishing kernel
int qk_launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb,
              int64_t k_sh, int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
              float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N, float scale,
              float* qpart, const float* r_scale, int64_t r_scale_stride, int* ngroups_out, hipStream_t stream) {
#ifdef TE_STUDY
  int ng, kbg;
  groups_for(N, ng, kbg);
  *ngroups_out = ng;
  const int BH = (int)(B * H);
  if (q_sn > 65536 || k_sn > 65536 || cq_sn > 65536 || ck_sn > 65536) return TE_ERR_UNSUPPORTED;
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
#ifdef TE_STUDY
  {
    const char* e = getenv("TE_ATTN_KB_PROF");
    if (e && atoi(e) == 2) {
      qk6_kb_kernel<RULE, true><<<dim3((unsigned)(BH * ng)), dim3(kT), 0, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks,
                                                                                  qpart, (int)H, (int)N, BH, kbg, ng, scale,
                                                                                  r_scale, r_scale_stride);
      return TE_OK;
    }
  }
#endif
  qk6_kb_kernel<RULE><<<dim3((unsigned)(BH * ng)), dim3(kT), 0, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, cam_k, cks, qpart,
                                                                        (int)H, (int)N, BH, kbg, ng, scale, r_scale,
                                                                        r_scale_stride);
  return TE_OK;
#else
  (void)Rnn, (void)q, (void)k, (void)Z, (void)cam_q, (void)cam_k, (void)qpart, (void)r_scale, (void)stream;
  *ngroups_out = 1;
  return TE_ERR_UNSUPPORTED;
#endif
}

}  // namespace te_attn_kb

#ifdef TE_STUDY
// measurement builds: the phase counters of the last profiled launch (8 waves x 8 slots), synchronising
extern "C" int te_attn_kb_prof_read(long long* host_out) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(te_attn_kb::g_kb_prof), sizeof(long long) * 64);
}
#endif
