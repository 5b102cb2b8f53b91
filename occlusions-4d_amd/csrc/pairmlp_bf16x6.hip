// Training: the pair tensors of a merged-form attention layer (what csrc/crossattn16p.hip's pair_mlp_kernel produces for
// the recompute in backward; model/point_transformer_layer.py:168-176 before the softmax) on
// SPLIT-PRECISION matrix instructions -- opt-in, fp32-class (csrc/bf16x6.hpp: bf16 x 3 pieces, six partial products, fp32
// accumulate):
//     a[p]      = aq[p / K] - kt[idx[p]] + Wp r[p]        (P, 832)   written stage by stage, BEFORE the ReLU
//     logits[p] = W2 relu(a[p])                           (P, 416)   (attn_mlp[2].bias left out: it cancels in the softmax)
//     pe[p]     = P2 r[p] + c2                            (P, 416)
// The weight stream, the stage protocol (ring of three 45 KB stages, DMA one stage ahead, the two waves of a SIMD half a
// stage out of phase) and the GEMM1^T -> GEMM2 register hand-over are those of csrc/crossattn_bf16x6.hip; what differs:
// a workgroup owns 256 CONSECUTIVE pair rows (no query structure: there is no softmax here, every MFMA row is live), r
// is an input, and GEMM2 / GEMM3 run with the weight fragments on the A side, so that a lane holds four consecutive
// channels of one pair row and every store is 16 bytes.  The channel halves are separate workgroups (both compute GEMM1;
// half h stores hidden tile u = h of every stage).
#include <stdlib.h>

#include "bf16x6.hpp"

namespace {

constexpr int ZD = 416;                   // channels
constexpr int ZHID = 2 * ZD;              // hidden units of attn_mlp
constexpr int ZHALF = ZD / 2;             // channels per workgroup
constexpr int ZT = ZHALF / 16;            // 13 channel tiles
constexpr int ZS = ZHID / 32;             // 26 hidden stages of 32
constexpr int ZFW = 256;                  // u32 words per fragment image
constexpr int ZW2F = 3 * ZT;              // 39 W2 fragments of a stage
constexpr int ZSF = ZW2F + 6;             // + 2 x 3 Wp fragments = 45
constexpr int ZSTAGE = ZSF * ZFW;         // 46080 B
constexpr int ZNSTAGE = ZS + 1;           // + the P2 stage
constexpr int ZWAVES = 8;
constexpr int ZROWS = 32 * ZWAVES;        // 256 pair rows per workgroup

struct PairX6Args {
  const float* aq; int64_t ld_aq;
  const float* kt; int64_t ld_kt;
  const float* r;                         // (P, 32) contiguous
  const int32_t* idx;                     // (P) = (N, K) flat
  const float* c2;
  const unsigned* wstream;                // occ4d_pack_attn_bf16x6_stream_f32: [half][27 stages][45][64 lanes][4 words]
  float* a_out;                           // (P, 832) contiguous
  float* logits;                          // (P, 416) contiguous
  float* pe;                              // (P, 416) contiguous
  int P, K;
  int groups, per;                        // groups of 256 pair rows; groups per XCD slab
};

__global__ __launch_bounds__(512, 2) void pair_mlp_bf16x6_kernel(const PairX6Args a) {
  __shared__ __attribute__((aligned(16))) unsigned buf0[ZSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[ZSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf2[ZSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // workgroup b runs on XCD b % 8: XCDs 0-3 take channel half 0, 4-7 half 1; each XCD one contiguous slab of row groups
  const int x = blockIdx.x & 7, half = x >> 2, slab = x & 3, in_slab = blockIdx.x >> 3;
  const int group = slab * a.per + in_slab;
  if (in_slab >= a.per || group >= a.groups) return;
  const int ch0 = ZHALF * half;
  const unsigned* const wst = a.wstream + (int64_t)half * ZNSTAGE * ZSTAGE;
  const unsigned lane16 = lane * 16;
  const bool grp_b = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) != 0;   // (phase skew: crossattn_bf16x6.hip)

  auto dma_part = [&](int stage_no, const unsigned* dst, int i) {
    const int f = min(wave + ZWAVES * i, ZSF - 1);                                  // wave-uniform; the tail repeats 44
    dma_frag_x(wst + (int64_t)stage_no * ZSTAGE + f * ZFW, lds_addr_x(dst) + (unsigned)f * (ZFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_part(0, buf0, i);

  // ---- this lane's pair of each row tile (column c of the tile); rows past P are clamped for the loads, never stored
  Split rs[2];                            // r[p][8 g + j], three pieces
  unsigned aq_off[2], kt_off[2];
  int prow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int p = group * ZROWS + 32 * wave + 16 * rt + c;
    prow[rt] = p;
    const int pc = min(p, a.P - 1);
    const int q = pc / a.K;
    const int j = a.idx[pc];
    const float* rr = a.r + (int64_t)pc * 32 + 8 * g;
    rs[rt] = split8(*reinterpret_cast<const f32x4*>(rr), *reinterpret_cast<const f32x4*>(rr + 4));
    aq_off[rt] = (unsigned)(q * (int)a.ld_aq + 4 * g) * 4u;
    kt_off[rt] = (unsigned)(j * (int)a.ld_kt + 4 * g) * 4u;
  }
  auto slice = [](const float* base, unsigned off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  f32x4 ia[2][2], ik[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ia[rt][u] = slice(a.aq + 16 * u, aq_off[rt]);
      ik[rt][u] = slice(a.kt + 16 * u, kt_off[rt]);
    }
  f32x4 acc[2][ZT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < ZT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  dma_wait_x();
  __builtin_amdgcn_s_barrier();                       // barrier 0: stage 0 is complete
  if (grp_b) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_part(1, buf1, i);
  }

  auto stage = [&](const int s, const unsigned* __restrict__ cur, const unsigned* dA, const unsigned* dB) {
    const unsigned* f = cur + lane * 4;
    // ---- GEMM1: Hpre^T tiles (rt, u), init Aq - Kt
    f32x4 h[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        h[rt][u] = ia[rt][u] - ik[rt][u];
    const int sn = s + 1 < ZS ? s + 1 : s;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ia[rt][u] = slice(a.aq + 32 * sn + 16 * u, aq_off[rt]);
        ik[rt][u] = slice(a.kt + 32 * sn + 16 * u, kt_off[rt]);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const u32x4 wh = *reinterpret_cast<const u32x4*>(f + (ZW2F + 3 * u + 0) * ZFW);
      const u32x4 wm = *reinterpret_cast<const u32x4*>(f + (ZW2F + 3 * u + 1) * ZFW);
      const u32x4 wl = *reinterpret_cast<const u32x4*>(f + (ZW2F + 3 * u + 2) * ZFW);
      mm6x2_b(wh, wm, wl, rs[0], rs[1], h[0][u], h[1][u]);
    }
    // the hidden activation BEFORE the ReLU: lane (g, c) holds units 32 s + 16 u + 4 g .. + 3 of pair row c; this
    // workgroup's share is tile u = half
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      if (prow[rt] < a.P)
        *reinterpret_cast<f32x4*>(a.a_out + (int64_t)prow[rt] * ZHID + 32 * s + 16 * half + 4 * g) = half ? h[rt][1] : h[rt][0];
    // ---- ReLU + three-way split: GEMM2's per-row operand of both row tiles
    Split hs[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) hs[rt] = split8(relu4x(h[rt][0]), relu4x(h[rt][1]));
    // ---- GEMM2: 13 channel tiles x (3 fragment reads, 12 MFMAs), weights on the A side (TRANSPOSED tiles)
    u32x4 bh = *reinterpret_cast<const u32x4*>(f);
    u32x4 bm = *reinterpret_cast<const u32x4*>(f + ZFW);
    u32x4 bl = *reinterpret_cast<const u32x4*>(f + 2 * ZFW);
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
      const u32x4 ch = bh, cm = bm, cl = bl;
      if (t + 1 < ZT) {
        bh = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1)) * ZFW);
        bm = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1) + 1) * ZFW);
        bl = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1) + 2) * ZFW);
      }
      if (t < 6) {
        if (!grp_b && s + 1 < ZNSTAGE) dma_part(s + 1, dA, t);
      } else if (t > 6) {
        if (grp_b && s + 2 < ZNSTAGE) dma_part(s + 2, dB, t - 7);
      }
      mm6x2_b(ch, cm, cl, hs[0], hs[1], acc[0][t], acc[1][t]);
      if (t == 6 && grp_b) {
        dma_wait_x();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!grp_b) {
      dma_wait_x();
      __builtin_amdgcn_s_barrier();
    }
  };
#pragma clang loop unroll(disable)
  for (int s = 0; s < ZS - 2; s += 3) {
    stage(s, buf0, buf1, buf2);
    stage(s + 1, buf1, buf2, buf0);
    stage(s + 2, buf2, buf0, buf1);
  }
  stage(ZS - 2, buf0, buf1, buf2);
  stage(ZS - 1, buf1, buf2, buf0);

  // ---- epilogue: the logits, then pe = P2 r + c2 on the P2 stage (buf2 = 26 % 3; complete since barrier 26).  Lane
  // (g, c) holds channels ch0 + 16 t + 4 g .. + 3 of pair row c in every tile
  const unsigned* fp = buf2 + lane * 4;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const bool live = prow[rt] < a.P;       // (only the stores are predicated: no matrix instruction under divergence)
    float* const lrow = a.logits + (int64_t)prow[rt] * ZD + ch0 + 4 * g;
    float* const prw = a.pe + (int64_t)prow[rt] * ZD + ch0 + 4 * g;
#pragma unroll
    for (int t = 0; t < ZT; ++t)
      if (live) *reinterpret_cast<f32x4*>(lrow + 16 * t) = acc[rt][t];
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
      const u32x4 ph = *reinterpret_cast<const u32x4*>(fp + (3 * t) * ZFW);
      const u32x4 pm = *reinterpret_cast<const u32x4*>(fp + (3 * t + 1) * ZFW);
      const u32x4 pl = *reinterpret_cast<const u32x4*>(fp + (3 * t + 2) * ZFW);
      f32x4 e = *reinterpret_cast<const f32x4*>(a.c2 + ch0 + 16 * t + 4 * g);
      e = mm(pl, rs[rt].h, e);
      e = mm(ph, rs[rt].l, e);
      e = mm(pm, rs[rt].m, e);
      e = mm(pm, rs[rt].h, e);
      e = mm(ph, rs[rt].m, e);
      e = mm(ph, rs[rt].h, e);
      if (live) *reinterpret_cast<f32x4*>(prw + 16 * t) = e;
    }
  }
}

}  // namespace

extern "C" int occ4d_pt_pair_mlp_bf16x6_f32(const float* aq, int64_t ld_aq, const float* kt, int64_t ld_kt, const float* r,
                                            const int32_t* idx, const float* c2, const float* wstream, float* a_out,
                                            float* logits, float* pe, int n, int m, int k, int d, void* stream) {
  const char* who = "occ4d_pt_pair_mlp_bf16x6_f32";
  OCC4D_REQUIRE(d == ZD, "%s: built for d = %d, got %d", who, ZD, d);
  OCC4D_REQUIRE(k >= 1 && m >= 1 && n >= 0, "%s: bad n / m / k", who);
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(aq && kt && r && idx && c2 && wstream && a_out && logits && pe, "%s: null pointer", who);
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ld_aq >= ZHID && ld_kt >= ZHID && ((uintptr_t)aq % 16) == 0 &&
                    ((uintptr_t)kt % 16) == 0 && ((uintptr_t)r % 16) == 0 && ((uintptr_t)c2 % 16) == 0 &&
                    ((uintptr_t)wstream % 16) == 0 && ((uintptr_t)a_out % 16) == 0 && ((uintptr_t)logits % 16) == 0 &&
                    ((uintptr_t)pe % 16) == 0,
                "%s: aq / kt rows of at least 832 floats with ld %% 4 == 0; every pointer 16-byte aligned", who);
  const int64_t pairs = (int64_t)n * k;
  OCC4D_REQUIRE(pairs < ((int64_t)1 << 31) - ZROWS && (int64_t)n * ld_aq < ((int64_t)1 << 29) &&
                    (int64_t)m * ld_kt < ((int64_t)1 << 29),
                "%s: 32-bit row offsets: n * k below 2^31, n * ld_aq and m * ld_kt below 2^29 floats", who);
  PairX6Args a{aq, ld_aq, kt, ld_kt, r, idx, c2, reinterpret_cast<const unsigned*>(wstream), a_out, logits, pe, (int)pairs, k,
               0, 0};
  a.groups = (int)occ4d::cdiv(pairs, ZROWS);
  a.per = (int)occ4d::cdiv(a.groups, 4);
  pair_mlp_bf16x6_kernel<<<8 * a.per, 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch(who);
}
