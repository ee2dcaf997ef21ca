// te_attn_fwd6l.hip -- the attention forward producer for LONG sequences on bf16 MFMAs with row-block owners
// (round 6; VERDICT r5 item 3 "and for the long-N path"): ViT-L/16 at 384^2 (N = 577, baselines/ViT/ViT_LRP.py:132-152) and BERT
// (N = 512; three separate projections, scores / sqrt(D), additive mask: BERT_explainability/modules/BERT/BERT.py:307-365).
//
//   z_qk [BH,N,N] = q k^T (unscaled: the QK rule's Z; optional)       x [BH,N,N] = z_qk * scale (optional: BERT's Add.X[0])
//   attn [BH,N,N] = softmax(x + mask)                                  out = attn v          head dim 64, 64 < N <= 640, any strides
//
// te_attn_fwd6.hip keeps a wave's whole score panel [32, N] in accumulator registers between the two products, which ends at
// N = 224; te_attn_long.hip (round 3: fp32 MFMAs, the [32, N] panel in LDS, eleven barriers per 32 rows) ran at 0.27 of the HBM
// roofline.  Here a wave owns 32 query rows for the whole kernel and walks the keys TWICE, 64 at a time:
//
//   pass 1   scores of the chunk (bf16 MFMAs on three-way split operands, te_attn_fwd6.hip), z_qk / x leave, the row's running
//            maximum and sum of exponentials (one rescale per chunk, per lane; the two lanes of a row meet once at the end);
//   pass 2   the SAME scores again -- the same instruction sequence on the same planes: bit for bit what pass 1 stored, at the
//            price of 24 MFMAs per 32 x 32 block instead of a second trip of the panel through memory --, the probabilities
//            exp(x - max) / sum leave as attn and, split into planes per K16 step straight from the accumulator registers, meet
//            the chunk's v^T planes: out = attn v from the stored probabilities, as in fwd6.
//
// k (both passes) and v^T (pass 2) travel through LDS as bf16 planes in MFMA-fragment order, one 64-key chunk per buffer, two
// buffers: the next chunk's rows are requested before the chunk's products and written behind them -- ONE barrier per chunk, and
// the waves drift inside a chunk (stores and exponentials of one beside the MFMAs of another).  A workgroup serves up to eight row
// blocks of one (b, h); the parts of a (b, h) are consecutive slots of ONE XCD (blockIdx -> (xcd, slot)), so that k and v are
// fetched from HBM once and re-read from that XCD's L2.  N x N tensors leave through a wave-private LDS tile as 128-byte runs of
// eight rows per buffer-store instruction (the descriptor's range check drops the rows at or beyond N).
//
// Every reduction has a fixed order that depends on N only: a batch equals its samples run one by one, bit for bit.
#include <stdlib.h>

#include <type_traits>

#include "te_attn_l6.h"

namespace te_attn_fwd6l {

namespace {

using namespace te_attn_l6;

// OPT (measurement builds, TE_FWD6L_OPT): 1 no N x N stores, 2 no exponentials, 4 no second product, 8 no pass 1, 16 no recomputation
template <int W, int OPT>
__global__ __launch_bounds__(64 * W, 2) void fwd6l_kernel(const float* __restrict__ q, Strided qs, const float* __restrict__ k, Strided ks,
                                                          const float* __restrict__ v, Strided vs, const float* __restrict__ mask,
                                                          float* __restrict__ zqk, float* __restrict__ xsc, float* __restrict__ attn,
                                                          float* __restrict__ out, Strided os, int H, int N, int BH, int G, int RB,
                                                          float scale) {
  typedef Cfg<W> C;
  constexpr int kT = C::kT, kKC = C::kKC, NKB = C::kNKB, kBuf = C::kBuf, kOperand = C::kOperand, kPlane = C::kPlane;
  extern __shared__ __attribute__((aligned(16))) unsigned char Pl[];
  // blockIdx -> (xcd, slot): the G parts of a (b, h) are consecutive slots of one XCD
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
  const float* q_bh = q + b * qs.sb + h * qs.sh;
  const float* k_bh = k + b * ks.sb + h * ks.sh;
  const float* v_bh = v + b * vs.sb + h * vs.sh;
  const unsigned nn_bytes = (unsigned)(N * N * 4);
  const Rsrc z_rs = __builtin_amdgcn_make_buffer_rsrc(zqk ? zqk + (int64_t)bh * N * N : attn, 0, zqk ? nn_bytes : 0, 0x00020000);
  const Rsrc x_rs = __builtin_amdgcn_make_buffer_rsrc(xsc ? xsc + (int64_t)bh * N * N : attn, 0, xsc ? nn_bytes : 0, 0x00020000);
  const Rsrc a_rs = __builtin_amdgcn_make_buffer_rsrc(attn + (int64_t)bh * N * N, 0, nn_bytes, 0x00020000);
  float* const bias = reinterpret_cast<float*>(Pl + C::kBiasOff);
  float* const tile = reinterpret_cast<float*>(Pl + C::kTileOff) + wave * (32 * kTileLd);

  // ---- the wave's q rows as B planes in registers; chunk 0 of k; the additive term of every key (mask, -inf beyond N) ----
  KReq kr;
  VReq vr;
  request_k<W>(kr, k_bh, ks.sn, N, 0);
  bf16x8 qb[4][3];
  {
    f32x4 qv[4][2];
    const float* qr = q_bh + (int64_t)min(i, N - 1) * qs.sn + 8 * kh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qv[s][0] = *reinterpret_cast<const f32x4_u*>(qr + 16 * s);
      qv[s][1] = *reinterpret_cast<const f32x4_u*>(qr + 16 * s + 4);
    }
    for (int j = threadIdx.x; j < NC * kKC; j += kT) bias[j] = (j < N) ? (mask ? mask[(int64_t)b * N + j] : 0.0f) : -INFINITY;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = row_ok ? qv[s][0][e] : 0.0f, x[4 + e] = row_ok ? qv[s][1][e] : 0.0f;
      planes_of8(x, qb[s]);
    }
  }
  write_k<W>(Pl, kr, N, 0);
  __syncthreads();

  const RowOff ro = make_rowoff(blk * 32, N);
  const unsigned char* const lane_frag = Pl + lane * 16;
  f32x16 acc[NKB];
  // x = acc * scale + (mask; -inf beyond N), in place
  auto soft_in = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NKB; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + kKC * c + 32 * u + 8 * g + 4 * kh);
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          // 'dots * self.scale' (ViT_LRP.py:139-141) / 'scores / sqrt(D)' (BERT.py:339), then '+ attention_mask' (BERT.py:341-342)
          const f32x2 t = f32x2{acc[u][4 * g + 2 * e2], acc[u][4 * g + 2 * e2 + 1]} * f32x2{scale, scale} + f32x2{bv[2 * e2], bv[2 * e2 + 1]};
          acc[u][4 * g + 2 * e2] = t[0], acc[u][4 * g + 2 * e2 + 1] = t[1];
        }
      }
  };
  auto blocks_out = [&](const f32x16 (&a)[NKB], Rsrc rs, int c, bool tail) __attribute__((always_inline)) {
    if constexpr (OPT & 1) return;
    if (tail) {
#pragma unroll
      for (int u = 0; u < NKB; ++u) block_out<true>(tile, a[u], rs, ro, kKC * c + 32 * u, N);
    } else {
#pragma unroll
      for (int u = 0; u < NKB; ++u) block_out<false>(tile, a[u], rs, ro, kKC * c + 32 * u, N);
    }
  };

  // ================= pass 1: z_qk (and x) leave; running maximum m and sum l of exp(x - m) over this lane's keys =================
  float m = -INFINITY, l = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + (c & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    const int cn = last ? 0 : c + 1;                 // behind the last chunk: chunk 0 again, with its v rows, for pass 2
    request_k<W>(kr, k_bh, ks.sn, N, cn);
    if (last) request_v<W>(vr, v_bh, vs.sn, N, 0);
    if (owner && !(OPT & 8)) {
      scores<W>(acc, buf, qb);
      if (zqk) blocks_out(acc, z_rs, c, last);
      if (xsc) {
        f32x16 xs[NKB];
#pragma unroll
        for (int u = 0; u < NKB; ++u)
#pragma unroll
          for (int e = 0; e < 16; ++e) xs[u][e] = acc[u][e] * scale;
        blocks_out(xs, x_rs, c, last);
      }
      soft_in(c);
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < NKB; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) cm = fmaxf(cm, acc[u][e]);
      const float mn = fmaxf(m, cm);
      const float mr = (mn == -INFINITY) ? 0.0f : mn;          // (a lane whose keys so far all lie beyond N)
      f32x2 s2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
      for (int u = 0; u < NKB; ++u)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2)
          s2[e2 & 1] = add2(s2[e2 & 1], ((OPT & 2) ? sub2(acc[u][2 * e2], acc[u][2 * e2 + 1], mr) : exp2_le0(sub2(acc[u][2 * e2], acc[u][2 * e2 + 1], mr))));
      const f32x2 s1 = add2(s2[0], s2[1]);
      l = l * exp_le0(m - mr) + (s1[0] + s1[1]);
      m = mn;
    }
    write_k<W>(nbuf, kr, N, cn);
    if (last) write_v<W>(nbuf + kOperand, vr, N, 0);
    __syncthreads();
  }
  // the two lanes of a row: (a + b = b + a: both end up with the same bits)
  float mx, sum, rcs;
  {
    const float mo = __shfl_xor(m, 32, 64);
    const float mf = fmaxf(m, mo);
    mx = (mf == -INFINITY) ? 0.0f : mf;
    const float a = l * exp_le0(m - mx);
    sum = a + __shfl_xor(a, 32, 64);
    rcs = __builtin_amdgcn_rcpf(sum);
    rcs = fmaf(fmaf(-sum, rcs, 1.0f), rcs, rcs);
  }

  // ================= pass 2: the same scores, attn leaves, out += attn v =================
  f32x16 o[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[mb][e] = 0.0f;
  for (int c = 0; c < NC; ++c) {
    const unsigned char* const buf = lane_frag + ((NC + c) & 1) * kBuf;
    unsigned char* const nbuf = Pl + ((NC + c + 1) & 1) * kBuf;
    const bool last = c + 1 == NC;
    if (!last) {
      request_k<W>(kr, k_bh, ks.sn, N, c + 1);
      request_v<W>(vr, v_bh, vs.sn, N, c + 1);
    }
    if (owner) {
      if constexpr (OPT & 16) {
#pragma unroll
        for (int u = 0; u < NKB; ++u)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[u][e] = (float)(e + c);
      } else {
        scores<W>(acc, buf, qb);
      }
      soft_in(c);
#pragma unroll
      for (int u = 0; u < NKB; ++u)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const f32x2 xx = sub2(acc[u][2 * e2], acc[u][2 * e2 + 1], mx);
          const f32x2 p = div2((OPT & 2) ? xx : exp2_le0(xx), sum, rcs);
          acc[u][2 * e2] = p[0], acc[u][2 * e2 + 1] = p[1];
        }
      blocks_out(acc, a_rs, c, last);
      // K16 step s of the chunk = keys 16 s .. 16 s + 15 = (u = s / 2, g = 2 (s & 1), 2 (s & 1) + 1): B element t = 4 gg + c of lane (i, h)
      const unsigned char* const vfrag = buf + kOperand;
#pragma unroll
      for (int s = 0; s < ((OPT & 4) ? 0 : 2 * NKB); ++s) {
        const int u = s >> 1, g0 = 2 * (s & 1);
        const float x[8] = {acc[u][4 * g0],     acc[u][4 * g0 + 1], acc[u][4 * g0 + 2], acc[u][4 * g0 + 3],
                            acc[u][4 * g0 + 4], acc[u][4 * g0 + 5], acc[u][4 * g0 + 6], acc[u][4 * g0 + 7]};
        bf16x8 pb[3];
        planes_of8(x, pb);
        bf16x8 a[2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) a[mb][qq] = *reinterpret_cast<const bf16x8*>(vfrag + qq * kPlane + (s * 2 + mb) * kFrag);
#pragma unroll
        for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) o[mb] = TE_MFMA_BF16(a[mb][PA[p6]], pb[PB[p6]], o[mb]);
      }
    }
    if (!last) {
      write_k<W>(nbuf, kr, N, c + 1);
      write_v<W>(nbuf + kOperand, vr, N, c + 1);
      __syncthreads();
    }
  }
  if (row_ok) {
    float* o_row = out + b * os.sb + h * os.sh + (int64_t)i * os.sn + 4 * kh;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(o_row + 32 * mb + 8 * g) = f32x4{o[mb][4 * g], o[mb][4 * g + 1], o[mb][4 * g + 2], o[mb][4 * g + 3]};
  }
}

template <int W, int OPT = 0>
int launch_w(const float* q, Strided qs, const float* k, Strided ks, const float* v, Strided vs, const float* mask, float* z_qk,
             float* x_scaled, float* attn, float* out, Strided os, int64_t B, int64_t H, int64_t N, float scale, hipStream_t stream) {
  const int NBr = (int)((N + 31) >> 5);
  const int G = (NBr + W - 1) / W, RB = (NBr + G - 1) / G;      // parts of a (b, h), row blocks (waves) of a part
  const int64_t BH = B * H, slots = ((BH + 7) / 8) * G;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fwd6l_kernel<W, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<W>::kLds);
  if (e != hipSuccess) return (int)e;
  fwd6l_kernel<W, OPT><<<dim3((unsigned)(slots * 8)), dim3(64 * W), Cfg<W>::kLds, stream>>>(q, qs, k, ks, v, vs, mask, z_qk, x_scaled, attn, out, os,
                                                                                      (int)H, (int)N, (int)BH, G, RB, scale);
  return TE_OK;
}

}  // namespace

// N > 64 so that every lane's half of a row meets a valid key in chunk 0 (shorter sequences: te_attn_fwd6.hip / te_attn_long.hip);
// 16-byte pieces of q / k / out rows: strides in multiples of four floats (the caller checks them)
bool supported(int64_t B, int64_t H, int64_t N, int64_t D) {
  return D == 64 && N > 64 && N <= kMaxN && B >= 1 && H >= 1 && B * H <= (1 << 24);
}

int launch(const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
           const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, const float* mask, float* z_qk, float* x_scaled, float* attn,
           float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int64_t B, int64_t H, int64_t N, float scale, hipStream_t stream) {
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, vs{v_sb, v_sh, v_sn}, os{o_sb, o_sh, o_sn};
  // 8-wave workgroups are ~10 % faster at equal wave utilisation (measured: profiles/r06_attention_fwd_long_ab.log); the 4-wave
  // cut wins where it leaves fewer waves without a row block: owners / (parts x waves) at least 1.2 x the 8-wave cut's
  const int NBr = (int)((N + 31) >> 5), G8 = (NBr + 7) / 8, G4 = (NBr + 3) / 4;
  bool w8 = 5 * G8 * 8 <= 6 * G4 * 4;            // (NBr / (4 G4)) / (NBr / (8 G8)) < 1.2
#ifdef TE_STUDY      // TE_FWD6L_WAVES=4 / 8 forces the cut, TE_FWD6L_OPT the measurement switches (measurement builds)
  static const int wenv = [] { const char* e = getenv("TE_FWD6L_WAVES"); return e ? atoi(e) : 0; }();
  if (wenv == 8) w8 = true;
  if (wenv == 4) w8 = false;
  static const int opt = [] { const char* e = getenv("TE_FWD6L_OPT"); return e ? atoi(e) : 0; }();
#define TE_FWD6L_ARGS q, qs, k, ks, v, vs, mask, z_qk, x_scaled, attn, out, os, B, H, N, scale, stream
  if (w8) {
    switch (opt) {
      case 1: return launch_w<8, 1>(TE_FWD6L_ARGS);
      case 2: return launch_w<8, 2>(TE_FWD6L_ARGS);
      case 4: return launch_w<8, 4>(TE_FWD6L_ARGS);
      case 8: return launch_w<8, 8>(TE_FWD6L_ARGS);
      case 16: return launch_w<8, 16>(TE_FWD6L_ARGS);
      case 31: return launch_w<8, 31>(TE_FWD6L_ARGS);
      default: break;
    }
  } else {
    switch (opt) {
      case 1: return launch_w<4, 1>(TE_FWD6L_ARGS);
      case 2: return launch_w<4, 2>(TE_FWD6L_ARGS);
      case 4: return launch_w<4, 4>(TE_FWD6L_ARGS);
      case 8: return launch_w<4, 8>(TE_FWD6L_ARGS);
      case 16: return launch_w<4, 16>(TE_FWD6L_ARGS);
      case 31: return launch_w<4, 31>(TE_FWD6L_ARGS);
      default: break;
    }
  }
#undef TE_FWD6L_ARGS
#endif
  return w8 ? launch_w<8>(q, qs, k, ks, v, vs, mask, z_qk, x_scaled, attn, out, os, B, H, N, scale, stream)
            : launch_w<4>(q, qs, k, ks, v, vs, mask, z_qk, x_scaled, attn, out, os, B, H, N, scale, stream);
}

}  // namespace te_attn_fwd6l
