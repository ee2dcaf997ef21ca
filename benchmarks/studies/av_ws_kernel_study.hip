// STUDY SOURCE -- not compiled into libte_relprop.so (transformer-explainability_amd/build.py does not list it).
//
// Wave-specialised variant of the AV relprop rule kernel (round 2): four MFMA waves + four memory waves per workgroup,
// padded LDS images, one barrier per 32-row tile.  It is CORRECT (the 119 attention / producer GPU tests pass with it
// in place of av_rule_kernel) and exactly as fast as the lock-step kernel it was meant to beat (ViT-B/16 batch 64:
// 166.5 us vs 167.8 us per layer; slower at N = 577 / 512), so the library keeps the lock-step kernel.  What the study
// measured, and why the two designs tie, is in DESIGN.md section 3 and profiles/r02_attention_ws_study.log.
// To rebuild it: paste this block into csrc/te_attn_rules.hip in front of lds_av() and launch av_ws_kernel from
// av_launch with ws_groups_for() / lds_av_ws() (dynamic LDS 149 KB).
//
// One build of this kernel that took a debug flag as a RUNTIME kernel argument hung the GPU; the same flag as a
// compile-time constant (identical code for flag = 0) never did.  The cause was not found -- another reason it is not
// in the library.
// ================================================================================================
// WAVE-SPECIALISED rule kernels (the default; TE_ATTN_WS=0 selects the lock-step kernels above).
//
// The lock-step kernels put all eight waves of the CU's one workgroup through the same phase at the same time
// (wait for the tile, form S, write LDS, MFMA, store): the matrix pipe idles during every phase but one (DESIGN.md
// section 3).  Here the workgroup's waves have two roles:
//   waves 0-3 (one per SIMD)  MFMA only: per tile they read fragments from LDS and issue MFMAs; the column-side
//                             accumulators live in their registers across all row tiles.
//   waves 4-7                 memory only.  Tile it+2 is requested into registers, tile it+1 (requested one trip ago)
//                             goes registers -> S formation -> LDS stage (it+1) & 1, and (AV rule) the finished
//                             cam_attn tile it-1, which the MFMA waves left IN PLACE of the attn tile, goes LDS ->
//                             global as 16-B row stores.
// One barrier per 32-row tile; two LDS stages alternate.  LDS images are PADDED, not swizzled ([rows][68] for the
// 64-wide tiles, [TI][256] / [TI][260] for the wide tile): every fragment address of an MFMA wave is one of a few
// per-lane bases plus a compile-time offset, so the hundred-odd LDS reads of a tile need no address arithmetic and no
// address registers (with XOR swizzles hipcc hoists ~100 loop-invariant addresses and spills them).
// ================================================================================================
constexpr int kCW = 4;               // MFMA waves
constexpr int SLD = 68;              // row stride of the [.][64] images: 16-B fragments of 8 rows hit 8 bank groups
constexpr int ALD = 256;             // row stride of the AV rule's [TI][keys] tile (b32 along a row, 16-B row chunks)
#define TE_OPAQUE(x) asm volatile("" : "+v"(x))
#define TE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
constexpr int SGB_MFMA = 0x8, SGB_DSR = 0x100;

// float4 chunk c4 of a row with nj >= 4 valid columns, without a branch: chunks past the row's end read its last four
// columns (always in bounds) -- ws_load_raw only issues the load; ws_fix_chunk (run one trip later, when the data is
// needed) assembles the partial chunk and zeroes what lies outside the tile.  Keeping the two apart matters: any
// arithmetic on the loaded value next to the load makes hipcc wait for it there, one HBM round trip per chunk.
__device__ __forceinline__ f32x4 ws_load_raw(const float* __restrict__ base, unsigned row_off, int c4, int nj) {
  // (uniform base + 32-bit element offset: one address VGPR per row instead of a 64-bit pair per chunk)
  return *reinterpret_cast<const f32x4_u*>(base + (row_off + (unsigned)((c4 < (nj >> 2)) ? (c4 << 2) : nj - 4)));
}
__device__ __forceinline__ f32x4 ws_fix_chunk(f32x4 L, int c4, int nj, bool row_ok) {
  const int tailc = nj >> 2, rem = nj & 3;
  const bool full = c4 < tailc, tail = (c4 == tailc) && rem != 0;
  f32x4 t;
  t[0] = rem == 1 ? L[3] : (rem == 2 ? L[2] : L[1]);
  t[1] = rem == 2 ? L[3] : (rem == 3 ? L[2] : 0.0f);
  t[2] = rem == 3 ? L[3] : 0.0f;
  t[3] = 0.0f;
  f32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = (row_ok && full) ? L[e] : ((row_ok && tail) ? t[e] : 0.0f);
  return out;
}
__device__ __forceinline__ void ws_store_chunk(float* __restrict__ base, unsigned row_off, int c4, int nj, bool row_ok,
                                               f32x4 v) {
  const int tailc = nj >> 2, rem = nj & 3;
  float* p = base + (row_off + (unsigned)(c4 << 2));
  if (row_ok) {
    if (c4 < tailc) {
      *reinterpret_cast<f32x4_u*>(p) = v;
    } else if (c4 == tailc) {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < rem) p[e] = v[e];
    }
  }
}
// key-side operand of this group: rows [0, nj) of src [.,64] -> LDS [nj32][SLD], rows >= nj zero (all 512 threads)
__device__ __forceinline__ void ws_stage_keys(float* __restrict__ Kt, const float* __restrict__ src, int64_t sn, int nj,
                                              int nj32) {
  for (int idx = threadIdx.x; idx < nj32 * 16; idx += kT) {
    const int row = idx >> 4, c = idx & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nj) v = *reinterpret_cast<const f32x4_u*>(src + (int64_t)row * sn + (c << 2));
    *reinterpret_cast<f32x4*>(Kt + row * SLD + (c << 2)) = v;
  }
}

// out[j][d] = RAW ? acc : (x[j][d] * acc) * scale for the MFMA waves' column accumulators; accumulator (s, db) of wave
// w = key block w + 4 s, d columns 32 db ..  x comes from its LDS image XL [keys][SLD] (XLDS) or from global memory
// -- all sixteen values of a block requested before the first is used (a load inside the per-element guard
// costs one HBM round trip per element: 64 in a row per lane).
template <bool RAW, bool XLDS>
__device__ __forceinline__ void ws_col_epilogue(const f32x16 (&acc)[2][2], const float* __restrict__ XL,
                                                const float* __restrict__ XG, int64_t xsn, float* __restrict__ out,
                                                int64_t osn, int nj, int wave, int lr, int kh, int njb, float scale) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kb = wave + s * kCW;
    if (kb < njb) {
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int d = db * 32 + lr;
        float x[16];
        if constexpr (!RAW) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int j = kb * 32 + crow(e, kh);
            if constexpr (XLDS) x[e] = XL[j * SLD + d];
            else x[e] = XG[(int64_t)min(j, nj - 1) * xsn + d];
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = kb * 32 + crow(e, kh);
          float val = acc[s][db][e];
          if constexpr (!RAW) val = (x[e] * val) * scale;
          if (j < nj) out[(int64_t)j * osn + d] = val;
        }
      }
    }
  }
}

// Column-side product of one tile for an MFMA wave, hand-pipelined: the LDS fragments of group g + 1 are requested
// between the MFMAs of group g (left alone, hipcc emits ds_read / s_waitcnt lgkmcnt(0) / v_mfma triples -- one LDS
// round trip per MFMA).  Wp = this lane's base into the wide tile (row kh, column 32 wave + lr), WLD its row stride;
// Yp = this lane's base into the [TI][SLD] tile (row kh, column lr).  Leaves b-fragments dead, returns nothing.
//   acc[s][db] += W^T[key block wave + 4 s][32 rows] Y[32 rows][32 db ..]
template <int WLD, class Tail>
__device__ __forceinline__ void ws_col_product(f32x16 (&acc)[2][2], const float* __restrict__ Wp,
                                               const float* __restrict__ Yp, bool has1, Tail&& tail_loads) {
  // (every fragment array is defined unconditionally: a conditionally defined one becomes a loop-carried value and
  //  stays live across the whole tile loop; the loads of an absent second block read in-bounds LDS and go unused)
  float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) a0[m] = Wp[2 * m * WLD];
#pragma unroll
  for (int m = 0; m < 16; ++m) b0[m] = Yp[2 * m * SLD];
  TE_FENCE();
#pragma unroll
  for (int m = 0; m < 16; ++m) b1[m] = Yp[2 * m * SLD + 32];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[0][0] = TE_MFMA32(a0[m], b0[m], acc[0][0]);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    TE_SGB(SGB_MFMA, 1);
    TE_SGB(SGB_DSR, 1);
  }
  TE_FENCE();
#pragma unroll
  for (int m = 0; m < 16; ++m) a1[m] = Wp[2 * m * WLD + 32 * kCW];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[0][1] = TE_MFMA32(a0[m], b1[m], acc[0][1]);
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    TE_SGB(SGB_MFMA, 1);
    TE_SGB(SGB_DSR, 1);
  }
  TE_FENCE();
  if (has1) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[1][0] = TE_MFMA32(a1[m], b0[m], acc[1][0]);
  }
  TE_FENCE();
  tail_loads();                        // the caller's next fragments, requested under the last sixteen MFMAs
  if (has1) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[1][1] = TE_MFMA32(a1[m], b1[m], acc[1][1]);
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    TE_SGB(SGB_MFMA, 1);
    TE_SGB(SGB_DSR, 1);
  }
  TE_FENCE();
}

template <int MODE>
__global__ __launch_bounds__(kT) void av_ws_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, Strided zs, const float* __restrict__ attn,
    const float* __restrict__ v, Strided vs, float* __restrict__ cam_attn, float* __restrict__ cam_v, Strided cs, int H,
    int N, int BH, int JG, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kStage = TI * ALD + TI * SLD;            // [TI][ALD] attn tile (then cam_attn), [TI][SLD] S tile
  float* Vt = smem + 2 * kStage;                         // [256][SLD] v of this key group (row-side B operand)
  const int bh = blockIdx.x % BH, g = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int j0 = g * JG, nj = min(JG, N - j0), nj32 = (nj + 31) & ~31, njb = nj32 >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const float* r_bh = R + (int64_t)b * rs.sb + (int64_t)h * rs.sh;
  const float* z_bh = (MODE == RULE) ? Z + (int64_t)b * zs.sb + (int64_t)h * zs.sh : nullptr;
  const float* a_bh = attn + (int64_t)bh * N * N + j0;
  float* ca_bh = cam_attn + (int64_t)bh * N * N + j0;
  const float* v_bh = v + (int64_t)b * vs.sb + (int64_t)h * vs.sh + (int64_t)j0 * vs.sn;
  const int ntiles = (N + TI - 1) / TI;
  ws_stage_keys(Vt, v_bh, vs.sn, nj, nj32);

  if (wave < kCW) {
    // ---------------------------------------------------------------- MFMA waves
    f32x16 accv[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      zero16(accv[s][0]);
      zero16(accv[s][1]);
    }
    const bool has0 = wave < njb, has1 = wave + kCW < njb;       // this wave's key blocks kb = wave, wave + 4
    const int jA0 = wave * 32 + lr;
    const float* Vp = Vt + jA0 * SLD + 4 * kh;                   // v row of this lane's key (block 1: + 128 rows)
    __syncthreads();                                             // stage 0 holds tile 0, Vt is complete
    for (int it = 0; it < ntiles; ++it) {
      float* At = smem + (it & 1) * kStage;
      if (has0) {
        const float* Wp = At + kh * ALD + jA0;                     // column-side A fragments: rows 2 m + kh
        const float* Yp = At + TI * ALD + kh * SLD + lr;           // column-side B fragments
        const float* Sp = At + TI * ALD + lr * SLD + 4 * kh;       // row-side A fragments: S[lr][8 q + 4 kh ..]
        float* Ep = At + 4 * kh * ALD + jA0;                       // the block's own elements: rows crow(e, kh)
        f32x4 ra[8], vb0[8], vb1[8];
        ws_col_product<ALD>(accv, Wp, Yp, has1, [&]() __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) ra[q] = *reinterpret_cast<const f32x4*>(Sp + 8 * q);
#pragma unroll
          for (int q = 0; q < 8; ++q) vb0[q] = *reinterpret_cast<const f32x4*>(Vp + 8 * q);
        });
        f32x16 g0, g1;
        float av0[16], av1[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) vb1[q] = *reinterpret_cast<const f32x4*>(Vp + 32 * kCW * SLD + 8 * q);
#pragma unroll
        for (int e = 0; e < 16; ++e) av0[e] = Ep[((e & 3) + 8 * (e >> 2)) * ALD];
        zero16(g0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) g0 = TE_MFMA32(ra[q][e], vb0[q][e], g0);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          TE_SGB(SGB_MFMA, 1);
          TE_SGB(SGB_DSR, 1);
        }
        TE_FENCE();
#pragma unroll
        for (int e = 0; e < 16; ++e) av1[e] = Ep[((e & 3) + 8 * (e >> 2)) * ALD + 32 * kCW];
        zero16(g1);
        if (has1) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) g1 = TE_MFMA32(ra[q][e], vb1[q][e], g1);
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          TE_SGB(SGB_MFMA, 1);
          TE_SGB(SGB_DSR, 1);
        }
        TE_FENCE();
        // the blocks' attn values are replaced by the rule's output (the memory waves carry the tile out next trip)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Ep[((e & 3) + 8 * (e >> 2)) * ALD] = (MODE == RULE) ? (av0[e] * g0[e]) * scale : g0[e];
        if (has1) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Ep[((e & 3) + 8 * (e >> 2)) * ALD + 32 * kCW] = (MODE == RULE) ? (av1[e] * g1[e]) * scale : g1[e];
        }
      }
      __syncthreads();
    }
    float* o_bh = cam_v + (int64_t)b * cs.sb + (int64_t)h * cs.sh + (int64_t)j0 * cs.sn;
    ws_col_epilogue<MODE == BWD, true>(accv, Vt, v_bh, vs.sn, o_bh, cs.sn, nj, wave, lr, kh, njb, scale);
  } else {
    // ---------------------------------------------------------------- memory waves: 8 threads per tile row
    const int m = threadIdx.x - kCW * 64, row = m >> 3, l8_ = m & 7;
    const int nr = nj32 >> 5;                            // float4 slots per thread of the wide tile (<= 8)
    // Slot r of a thread is chunk l8 + 8 r of its row.  Slots r < rt hold whole chunks in every lane (plain 16-B
    // accesses at base + 32 r floats), slot rt -- if nj is no multiple of 32 -- is the one that meets the row's end
    // (per-lane: whole / partial / nothing), slots above it are empty.  rt is uniform: the split costs scalar branches.
    const int tailc = nj >> 2, rt = tailc >> 3;
    struct TileRegs {
      f32x4 ta[8], rr[2], zz[2];
    };
    TileRegs X;
    auto request = [&](TileRegs& t, int it) __attribute__((always_inline)) {      // loads only (see ws_load_raw)
      const int i0 = it * TI, rows_valid = min(TI, N - i0);
      const unsigned gi = (unsigned)(i0 + min(row, rows_valid - 1));      // rows past N re-read the last row (zeroed later)
      int l8 = l8_;
      TE_OPAQUE(l8);       // (offsets are recomputed here: hoisted out of the tile loop they pin dozens of registers)
      const unsigned arow = gi * (unsigned)N, aoff = arow + (unsigned)(l8 << 2);
      (void)aoff;
#pragma unroll
      for (int r = 0; r < 8; ++r) t.ta[r] = ws_load_raw(a_bh, arow, l8 + 8 * r, nj);      // branch-free: twelve loads back to back
      const unsigned roff = gi * (unsigned)rs.sn + (unsigned)(l8 << 2), zoff = gi * (unsigned)zs.sn + (unsigned)(l8 << 2);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        t.rr[r] = *reinterpret_cast<const f32x4_u*>(r_bh + (roff + 32u * r));
        if constexpr (MODE == RULE) t.zz[r] = *reinterpret_cast<const f32x4_u*>(z_bh + (zoff + 32u * r));
      }
    };
    auto deposit = [&](const TileRegs& t, int it) __attribute__((always_inline)) {      // tile `it` -> stage it & 1
      float* At = smem + (it & 1) * kStage;
      float* St = At + TI * ALD;
      const bool row_ok = row < min(TI, N - it * TI);
      int l8 = l8_;
      TE_OPAQUE(l8);
      float* ap = At + row * ALD + (l8 << 2);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r < nr) *reinterpret_cast<f32x4*>(ap + 32 * r) = ws_fix_chunk(t.ta[r], l8 + 8 * r, nj, row_ok);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        f32x4 s = t.rr[r];                                // BWD: the tile of d_out itself
        if constexpr (MODE == RULE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = te_sd(t.rr[r][e], t.zz[r][e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = row_ok ? s[e] : 0.0f;      // rows beyond N
        *reinterpret_cast<f32x4*>(St + row * SLD + ((l8 + 8 * r) << 2)) = s;
      }
    };
    auto pick_up = [&](f32x4 (&pt)[8], int it) __attribute__((always_inline)) {      // finished tile `it`: LDS -> registers
      const float* Pt = smem + (it & 1) * kStage;
      int l8 = l8_;
      TE_OPAQUE(l8);
      const float* pp = Pt + row * ALD + (l8 << 2);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r < nr) pt[r] = *reinterpret_cast<const f32x4*>(pp + 32 * r);
    };
    auto send = [&](const f32x4 (&pt)[8], int it) __attribute__((always_inline)) {   // ... -> cam_attn rows, 16-B stores
      int l8 = l8_;
      TE_OPAQUE(l8);
      const int i0 = it * TI;
      if (row < min(TI, N - i0)) {
        const unsigned crow_off = (unsigned)(i0 + row) * (unsigned)N, coff = crow_off + (unsigned)(l8 << 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (r < rt) *reinterpret_cast<f32x4_u*>(ca_bh + (coff + 32u * r)) = pt[r];
          else if (r == rt && r < nr) ws_store_chunk(ca_bh, crow_off, l8 + 8 * r, nj, true, pt[r]);
        }
      }
    };
    request(X, 0);
    deposit(X, 0);
    if (ntiles > 1) request(X, 1);
    __syncthreads();
    for (int it = 0; it < ntiles; ++it) {
      // trip `it`: the finished tile it - 1 leaves stage (it + 1) & 1, tile it + 1 (requested one trip ago, in flight
      // while this wave sat at the barrier) takes its place, tile it + 2 is requested.  Order matters for vmcnt,
      // which counts loads and stores in issue order: the deposit waits for the OLDEST outstanding operations.
      f32x4 pt[8];
      if (it >= 1) pick_up(pt, it - 1);
      if (it + 1 < ntiles) deposit(X, it + 1);
      TE_FENCE();
      if (it + 2 < ntiles) request(X, it + 2);
      TE_FENCE();          // (stores last: registers of a store in flight are not handed to a load, which would wait for it)
      if (it >= 1) send(pt, it - 1);
      __syncthreads();
    }
    {
      f32x4 pt[8];
      pick_up(pt, ntiles - 1);
      send(pt, ntiles - 1);
    }
  }
}

