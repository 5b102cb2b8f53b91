// Fused vector attention over K <= 14 neighbours, D = 416 (E3 of SURVEY.md §8(a); model/point_transformer_layer.py:
// 168-179) on SPLIT-PRECISION matrix instructions: every GEMM of the layer -- GEMM1 (32 -> 832 hidden), GEMM2 (832 ->
// 416 logits, 87 % of the layer's FLOP) and GEMM3 (32 -> 416 positional encoding) -- runs on v_mfma_f32_16x16x32_bf16
// with BOTH operands split three ways into bf16 pieces and SIX of the nine partial products accumulated in fp32:
//
//     x = x1 + x2 + x3 exactly   (x1 = x truncated to bf16, x2 = (x - x1) truncated, x3 = x - x1 - x2: 3 x 8 = 24 bits)
//     a b ~ a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3, a3 b2, a3 b3 <= 2^-23 |a b|)
//
// Every bf16 x bf16 product is exact in fp32, so the result differs from an fp32 GEMM by the dropped terms (2^-23
// relative to |a||b| per term: one fp32 rounding of the product) and by the accumulation order: fp32-CLASS arithmetic
// (every saturated-regime gate of the fp32 path holds), unlike the two-piece bf16x3 logit mode of round 1
// (2^-16 relative operand error).  Six 16-cycle bf16 MFMAs of K = 32 replace eight 32-cycle fp32 MFMAs of K = 4 per 16 x
// 16 x 32 block: 2.67 x less matrix time.
//
// Work decomposition (wave64, 8 waves, one workgroup per CU, 16 queries x HALF of the channels per workgroup):
//   wave w  = two row tiles of 16 pair rows: rows 0-13 = the 14 neighbours of query q0 + 2 w + rt; rows 14, 15 = two
//             neighbour slots of one of the workgroup's two EXTRA queries (q0 + 16 + tile / 7, slots 2 (tile % 7), + 1;
//             252 of 256 MFMA rows live; their per-tile partial softmaxes meet in LDS at the end, as the 9th query of
//             crossattn16p.hip) x 13 channel tiles of 16 (the workgroup's 208 channels): 104 accumulator registers; every weight fragment
//             read from LDS serves both row tiles (at one row tile per wave the three bf16 pieces of the weights would
//             need the whole LDS bandwidth of the CU).
//   channel halves are separate WORKGROUPS: vector attention normalises per channel, so the halves never meet; both
//             compute GEMM1 (7 % of the work, duplicated) and stream only their half of W2: a stage is 45 KB instead of
//             85, two of them fit the LDS.  Workgroups of XCDs 0-3 take half 0, XCDs 4-7 half 1: an L2 holds one half.
//   stage s = 32 hidden units: 13 tiles x 3 pieces of W2 + 2 x 3 of the merged Wp (GEMM1), 1 KB fragments (64 lanes x 8
//             bf16), double buffered, DMA (global_load_lds_dwordx4) one stage ahead, one barrier per stage; the last
//             stage of the stream holds P2 for the epilogue's GEMM3.
// Chain per stage: GEMM1 in transposed form, Hpre^T[hid][pair] = Wp[hid][:] r[pair][:] + (Aq[query] - Kt[neighbour])
// (exact fp32 init), so its C/D registers (lane (g, c): hidden 16 u + 4 g + i of pair c) become, after ReLU and the
// three-way split, GEMM2's A operand (lane (g, pair): k = 8 g + j <-> hidden 16 (j >> 2) + 4 g + (j & 3); the packed W2
// fragments use the same k order).  Softmax / aggregation as in crossattn16p.hip, over the 14 live rows of a tile.
//
// Measured alternatives (round 5, commit 9a25220 holds their code): (1) PAIRED 4-wave workgroups, two per
// CU, weights in 24 KB sub-stages (the crossattn16p.hip architecture: everything of one workgroup overlaps the other's
// MFMA stream): 1.41-1.46 ms per 32256-query launch against 1.40 here, at twice the L2 -> LDS stream; (2) one row tile
// per wave with all 416 channels (GEMM1 and the split once per row: -7 % MFMAs, half the VALU instructions): 1.47 ms.
// All three sit at 67 % of the matrix pipe: the remaining loss is the issue slots of the non-MFMA instructions of a
// wave (one per MFMA on average, but bunched: the ReLU + split block, fragment reads, DMA issue), not overlap.
#include <stdlib.h>

#include "bf16x6.hpp"

namespace {

constexpr int XD = 416;                   // channels
constexpr int XHID = 2 * XD;              // hidden units of attn_mlp
constexpr int XHALF = XD / 2;             // channels per workgroup
constexpr int XT = XHALF / 16;            // 13 channel tiles
constexpr int XS = XHID / 32;             // 26 hidden stages of 32
constexpr int XFW = 256;                  // u32 words per fragment image (64 lanes x 16 B)
constexpr int XNSTAGE = XS + 1;           // + the P2 stage
constexpr int XWAVES = 8;
constexpr int XTILES = 2 * XWAVES;        // 16 row tiles per workgroup
constexpr int XQPB = XTILES + 2;          // 18 queries per workgroup: one per row tile + two spread over the tiles' rows 14, 15
constexpr int XPART = XTILES * 3 * XHALF; // floats of the extra queries' partial softmaxes [tile 16][3][XHALF]
// stage geometry of a split scheme S (S::NP pieces per operand: csrc/bf16x6.hpp)
template <typename S> struct XG {
  static constexpr int W2F = S::NP * XT;          // W2 fragments of a stage: 39 (bf16 x 3) / 26 (fp16 x 2)
  static constexpr int SF = W2F + 2 * S::NP;      // + the merged Wp fragments of GEMM1: 45 / 30
  static constexpr int STAGE = SF * XFW;          // words: 46080 B / 30720 B
  static constexpr int PARTS = (SF + XWAVES - 1) / XWAVES;   // fragments a wave fetches per stage: 6 / 4
  static constexpr bool PART_IN_RING = STAGE >= XPART;       // the partial softmaxes fit the oldest stage buffer
  static_assert(PARTS <= 6, "the DMA slots sit behind channel tiles 0-5 and 7-12");
};

struct AttnX6Args {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;         // Wv f + c2
  const float* P1; const float* c1;
  const unsigned* wstream;                // [half][XNSTAGE][SF][64 lanes][4 words]
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
  int groups, per;                        // query groups of 16; groups per XCD slab
  int skew;                               // phase-skew grouping of the waves (see the kernel)
  int stamps;                             // debug: record phase time stamps
  float* logits;                          // STORE >= 1: (N * K, 416) pre-softmax logits W2 relu(a), unscaled
  float* a_out;                           // STORE == 2: (N * K, 832) hidden pre-activations a (before the ReLU), unscaled
  float* pe_out;                          // STORE == 2: (N * K, 416) pe = P2 r + c2
  const float* c2;                        // STORE == 2: pos_mlp[2].bias (the kernel's GEMM3 starts from 0)
};

// debug (OCC4D_X6_STAMPS=1): s_memtime at the phase boundaries of waves 0 and 4 of the first 1024 workgroups
__device__ unsigned long long g_x6_stamps[1024 * 2 * 6];

// PRESCALED (schemes with HSCALE != 1 only): aq and kt arrive multiplied by S::HSCALE (the path-level entry points scale
// the merged matrices that produce them, csrc/path.hip: exact, a power of two); otherwise the kernel multiplies the init
// term itself (8 packed multiplies per stage and wave)
// STORE (training forward): 1 = the logits also go to HBM, row q K + slot; 2 = a and pe as well (csrc/crossattn16p.hip has the
// fp32 twin of this; a by the workgroups of channel half 0 only: both halves compute the same hidden units)
template <typename S, bool PRESCALED, int STORE = 0>
__global__ __launch_bounds__(512, 2) void cross_attn_split_kernel(const AttnX6Args a) {
  using G = XG<S>;
  using Op = typename S::Op;
  constexpr int NP = S::NP, XW2F = G::W2F, XSF = G::SF, XSTAGE = G::STAGE, PARTS = G::PARTS;
  // a ring of three stage buffers: stage s lives in buffer s % 3
  __shared__ __attribute__((aligned(16))) unsigned buf0[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf2[XSTAGE];
  __shared__ __attribute__((aligned(16))) float s_part_own[G::PART_IN_RING ? 4 : XPART];
  __shared__ __attribute__((aligned(16))) float s_p1[32 * 4];   // (P1[m][0..2], c1[m])
  __shared__ int s_idx[XQPB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // workgroup b runs on XCD b % 8: XCDs 0-3 take channel half 0, 4-7 half 1; each XCD one contiguous slab of groups
  const int x = blockIdx.x & 7, half = x >> 2, slab = x & 3, in_slab = blockIdx.x >> 3;
  const int group = slab * a.per + in_slab;
  if (in_slab >= a.per || group >= a.groups) return;
  const int q0 = group * XQPB;
  const int ch0 = XHALF * half;
  unsigned long long ts[6];
  ts[0] = __builtin_amdgcn_s_memtime();
  const unsigned* const wst = a.wstream + (int64_t)half * XNSTAGE * XSTAGE;
  const unsigned lane16 = lane * 16;
  // PHASE SKEW inside the workgroup.  Waves 0-3 (group A) and 4-7 (group B) share the four SIMDs pairwise; they execute
  // the same number of barriers, but A's barrier sits at the END of a stage and B's after channel tile 6 of the same
  // stage, so B runs half a stage behind A for the whole kernel: B's VALU phase (ReLU + three-way split, GEMM1) and its
  // barrier / DMA waits fall under A's GEMM2 stream on the same SIMD and vice versa.  Barrier k (k = 1 .. 26) is "A
  // finished stage k - 1" = "B is half way through stage k - 1"; right after it every wave issues ITS fragments of stage
  // k + 1 (target buffer (k + 1) % 3 = (k - 2) % 3: A finished stage k - 2 at barrier k - 1, B before barrier k), waits
  // for them before it arrives at barrier k + 1, and the first reader (A) starts stage k + 1 after that barrier.
// (the two waves of this workgroup on one SIMD sit in its wave slots 0 and 1: HW_REG_HW_ID[3:0] = WAVE_ID; any
  // assignment of waves to the groups is correct, only the right one puts an A and a B wave on every SIMD.
  // a.skew: 0 = no skew, 1 = waves 4-7, 2 = odd wave slot; OCC4D_X6_SKEW, performance only)
  const bool grp_b = a.skew == 2 ? (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) != 0
                                 : (a.skew == 1 && wave >= 4);

  // this wave's i-th fragment of a stage (PARTS per wave and stage; the tail repeats the last fragment: same bytes, same place)
  auto dma_part = [&](int stage_no, const unsigned* dst, int i) {
    const int f = min(wave + XWAVES * i, XSF - 1);                                  // wave-uniform
    dma_frag_x(wst + (int64_t)stage_no * XSTAGE + f * XFW, lds_addr_x(dst) + (unsigned)f * (XFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < PARTS; ++i) dma_part(0, buf0, i);
  if (tid < XQPB * 16) {
    const int q = min(q0 + (tid >> 4), a.N - 1);
    const int s = min(tid & 15, a.K - 1);
    s_idx[tid] = a.idx[(int64_t)q * a.K + s];
  }
  if (tid < 32) {
    s_p1[4 * tid + 0] = a.P1[3 * tid + 0];
    s_p1[4 * tid + 1] = a.P1[3 * tid + 1];
    s_p1[4 * tid + 2] = a.P1[3 * tid + 2];
    s_p1[4 * tid + 3] = a.c1[tid];
  }
  __syncthreads();

  // ---- this lane's pair of each row tile: column c < 14 = neighbour slot c of the tile's own query; c = 14, 15 = slots
  // 2 (tile % 7) + c - 14 of extra query tile / 7 (tiles 14, 15: dead rows)
  Op rs[2];                               // r = relu(P1 d + c1), hidden-pos units 8 g + j, NP pieces
  unsigned aq_off[2], kt_off[2];
  int a_idx[2];                           // STORE == 2: float index of this lane's four hidden units of stage 0, per row tile
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int tile = 2 * wave + rt;
    const int ql = c < 14 ? tile : XTILES + tile / 7;
    const int slot = c < 14 ? c : 2 * (tile % 7) + c - 14;
    const bool my_valid = slot < a.K && (c < 14 || tile < 14);
    const int my_q = min(q0 + min(ql, XQPB - 1), a.N - 1);
    const int my_j = s_idx[min(ql, XQPB - 1) * 16 + slot];
    const float* qp = a.qpos + (int64_t)my_q * a.qs;
    const float* ap = a.apos + (int64_t)my_j * a.as;
    const float dx = qp[0] - ap[0], dy = qp[1] - ap[1], dz = qp[2] - ap[2];
    float rr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(s_p1 + 4 * (8 * g + j));
      const float v = fmaf(dz, w.z, fmaf(dy, w.y, dx * w.x)) + w.w;
      rr[j] = my_valid ? fmaxf(v, 0.f) : 0.f;
    }
    rs[rt] = S::split8(f32x4{rr[0], rr[1], rr[2], rr[3]}, f32x4{rr[4], rr[5], rr[6], rr[7]});
    if (STORE == 2)
      a_idx[rt] = (half == 0 && my_valid && q0 + ql < a.N) ? ((q0 + ql) * a.K + slot) * XHID + 4 * g : -1;
    aq_off[rt] = (unsigned)(my_q * (int)a.ld_aq + 4 * g) * 4u;
    kt_off[rt] = (unsigned)(my_j * (int)a.ld_kt + 4 * g) * 4u;
  }
  auto slice = [](const float* base, unsigned off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  // GEMM1 accumulator init (Aq - Kt) of stage 0; later stages are fetched one stage ahead
  f32x4 ia[2][2], ik[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ia[rt][u] = slice(a.aq + 16 * u, aq_off[rt]);
      ik[rt][u] = slice(a.kt + 16 * u, kt_off[rt]);
    }
  f32x4 acc[2][XT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < XT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  ts[1] = __builtin_amdgcn_s_memtime();
  dma_wait_x();
  __builtin_amdgcn_s_barrier();                       // barrier 0: stage 0 is complete
  ts[2] = __builtin_amdgcn_s_memtime();
  if (grp_b) {                                        // (A issues its part of stage 1 inside stage 0, tile by tile)
#pragma unroll
    for (int i = 0; i < PARTS; ++i) dma_part(1, buf1, i);
  }

  // stage s from `cur`; group A issues its fragments of stage s + 1 (-> dA) behind tiles 0-5, group B its fragments of
  // stage s + 2 (-> dB) behind tiles 7-12: both right after "their" barrier s
  auto stage = [&](const int s, const unsigned* __restrict__ cur, const unsigned* dA, const unsigned* dB) {
    const unsigned* f = cur + lane * 4;
    // ---- GEMM1: Hpre^T tiles (rt, u), init Aq - Kt
    f32x4 h[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        h[rt][u] = (S::HSCALE == 1.f || PRESCALED) ? ia[rt][u] - ik[rt][u] : (ia[rt][u] - ik[rt][u]) * S::HSCALE;   // (Wp is packed * HSCALE)
    // next stage's slices (consumed at its top)
    const int sn = s + 1 < XS ? s + 1 : s;
#ifndef OCC4D_XA_ABL_NOINIT                           // (timing-only ablations: profiles/time_attn_split.py)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ia[rt][u] = slice(a.aq + 32 * sn + 16 * u, aq_off[rt]);
        ik[rt][u] = slice(a.kt + 32 * sn + 16 * u, kt_off[rt]);
      }
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 wf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) wf[p] = *reinterpret_cast<const u32x4*>(f + (XW2F + NP * u + p) * XFW);
#ifndef OCC4D_XA_ABL_NOGEMM1
      S::mm_x2_b(wf, rs[0], rs[1], h[0][u], h[1][u]);
#endif
      if (STORE == 2) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          if (a_idx[rt] >= 0)
            *reinterpret_cast<f32x4*>(a.a_out + a_idx[rt] + 32 * s + 16 * u) = h[rt][u] * (1.f / S::HSCALE);
      }
    }
    // ---- ReLU + split: GEMM2's A operand of both row tiles
    Op hs[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#ifdef OCC4D_XA_ABL_NOSPLIT                           // (the accumulators reinterpreted as operand registers: no VALU)
#pragma unroll
      for (int p = 0; p < NP; ++p) hs[rt].p[p] = __builtin_bit_cast(u32x4, h[rt][p & 1]);
#else
      hs[rt] = S::split8(relu4x(h[rt][0]), relu4x(h[rt][1]));          // (HSCALE x the hidden activations: ReLU commutes)
#endif
    }
    // ---- GEMM2: 13 channel tiles x (NP fragment reads, 2 x (6 | 3) MFMAs)
    u32x4 bn[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) bn[p] = *reinterpret_cast<const u32x4*>(f + p * XFW);
#pragma unroll
    for (int t = 0; t < XT; ++t) {
      u32x4 bc[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bc[p] = bn[p];
#ifndef OCC4D_XA_ABL_NOLDS                            // (timing only: tile 0's fragments serve every tile)
      if (t + 1 < XT) {
#pragma unroll
        for (int p = 0; p < NP; ++p) bn[p] = *reinterpret_cast<const u32x4*>(f + (NP * (t + 1) + p) * XFW);
      }
#endif
#ifndef OCC4D_XA_ABL_NODMA
      if (t < PARTS) {
        if (!grp_b && s + 1 < XNSTAGE) dma_part(s + 1, dA, t);
      } else if (t > 6 && t - 7 < PARTS) {
        if (grp_b && s + 2 < XNSTAGE) dma_part(s + 2, dB, t - 7);
      }
#endif
#ifdef OCC4D_XA_ABL_HALFGEMM2
      if (t & 1) continue;
#endif
#ifdef OCC4D_XA_ABL_NODEP                             // (timing only: the tile's products spread over four accumulators)
      if (NP == 2) {
        acc[0][t] = mmh(hs[0].p[1], bc[0], acc[0][t]); acc[1][t] = mmh(hs[1].p[1], bc[0], acc[1][t]);
        acc[0][(t + 6) % XT] = mmh(hs[0].p[0], bc[1], acc[0][(t + 6) % XT]); acc[1][(t + 6) % XT] = mmh(hs[1].p[0], bc[1], acc[1][(t + 6) % XT]);
        acc[0][t] = mmh(hs[0].p[0], bc[0], acc[0][t]); acc[1][t] = mmh(hs[1].p[0], bc[0], acc[1][t]);
      } else
#endif
      S::mm_x2(hs[0], hs[1], bc, acc[0][t], acc[1][t]);
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
  for (int s = 0; s < XS - 2; s += 3) {
    stage(s, buf0, buf1, buf2);
    stage(s + 1, buf1, buf2, buf0);
    stage(s + 2, buf2, buf0, buf1);
  }
  stage(XS - 2, buf0, buf1, buf2);
  stage(XS - 1, buf1, buf2, buf0);
  ts[3] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: buf2 (26 % 3) holds the P2 stage.  Per PAIR of channel tiles: GEMM3 (pe = P2 r; c2 is folded into vt),
  // then the per-channel softmax over the 14 own rows of a tile (registers i = rows 4 g + i, lane groups g) with the
  // two-operand lane-swap trees of csrc/crossattn16p.hip (one exchange tree serves both tiles), and the aggregation.
  // Rows 14, 15 (lane group 3, registers 2, 3) belong to an extra query: their partial softmax of this tile (max in the
  // log2 domain, denominator, numerator) goes to LDS.  The partials live in buf0: its last content was stage 24, which
  // every wave has left before barrier 26 (group B is in the second half of stage 25 at that barrier, group A done).
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor * S::INV_WSCALE / S::HSCALE;        // (the logits left the matrix pipe * WSCALE * HSCALE)
  const float NINF = -__builtin_inff();
  const unsigned* fp = buf2 + lane * 4;
  float* const s_part = G::PART_IN_RING ? reinterpret_cast<float*>(buf0) : s_part_own;          // [tile 16][3][XHALF]
  const bool g3 = g == 3;
  const float own23 = g3 ? 0.f : 1.f;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int tile = 2 * wave + rt;
    const int qm = q0 + tile;
    int voff[4];
    int loff[4];                                           // STORE: float index of this lane's element of register i's row
    bool act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      const int ql = row < 14 ? tile : min(XTILES + tile / 7, XQPB - 1);
      const int slot = row < 14 ? row : 2 * (tile % 7) + row - 14;
      voff[i] = s_idx[ql * 16 + slot] * (int)a.ld_vt + ch0 + c;
      act[i] = slot < a.K && (row < 14 || tile < 14);
      if (STORE) loff[i] = (act[i] && q0 + ql < a.N) ? ((q0 + ql) * a.K + slot) * XD + ch0 + c : -1;
    }
    // output: lane groups 0 / 2 store the first / second channel tile of a pair for this tile's own query
    float* const orow = a.agg + (int64_t)min(qm, a.N - 1) * a.ld_agg + ch0 + 16 * (g >> 1) + c;
    const bool o_lane = (g & 1) == 0 && qm < a.N;
    float* const sp_lane = s_part + tile * 3 * XHALF + c;
    const bool extra_writer = g3 && tile < 14;
#pragma unroll
    for (int tp = 0; tp < (XT + 1) / 2; ++tp) {
      const int tA = 2 * tp, tB = min(2 * tp + 1, XT - 1);          // (the last pair repeats tile 12: its copy is dropped)
      const bool single = 2 * tp + 1 >= XT;
      float vq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vq[0][i] = a.vt[voff[i] + 16 * tA];
        vq[1][i] = a.vt[voff[i] + 16 * tB];
      }
      f32x4 pe[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int t = x ? tB : tA;
        u32x4 pf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) pf[p] = *reinterpret_cast<const u32x4*>(fp + (NP * t + p) * XFW);
        pe[x] = S::mm_1(rs[rt], pf, f32x4{0.f, 0.f, 0.f, 0.f});
      }
      float am[2][4], val[2][4], m23[2], lm[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const f32x4 av = acc[rt][x ? tB : tA];
        if (STORE && !(x == 1 && single)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (loff[i] >= 0) a.logits[loff[i] + 16 * (x ? tB : tA)] = av[i] * (S::INV_WSCALE / S::HSCALE);
          if (STORE == 2) {
            const float c2v = a.c2[ch0 + 16 * (x ? tB : tA) + c];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (loff[i] >= 0) a.pe_out[loff[i] + 16 * (x ? tB : tA)] = fmaf(pe[x][i], S::INV_WSCALE, c2v);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          am[x][i] = act[i] ? av[i] : NINF;
          val[x][i] = S::WSCALE == 1.f ? pe[x][i] + vq[x][i] : fmaf(pe[x][i], S::INV_WSCALE, vq[x][i]);
        }
        m23[x] = fmaxf(am[x][2], am[x][3]);
        lm[x] = fmaxf(fmaxf(am[x][0], am[x][1]), g3 ? NINF : m23[x]);   // rows 14, 15 are not this query's
      }
      float mx[2];
      {
        const PairX p1 = swap16x(lm[0], lm[1]);
        const float m1 = fmaxf(p1.lo, p1.hi);            // rows: (A01, B01, A23, B23)
        const PairX p2 = swap32x(m1, m1);
        const float m2 = fmaxf(p2.lo, p2.hi);            // rows: (A, B, A, B)
        const PairX p3 = swap16x(m2, m2);
        mx[0] = p3.lo;
        mx[1] = p3.hi;
      }
      float den[2], num[2], d23[2], n23[2], msr[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const float mxs = mx[x] * sc;                    // logits in the log2 domain: acc / sqrt(D) * log2(e)
        msr[x] = m23[x] * sc;                            // (extra query: the reference of this tile's two slots)
        const float live = msr[x] > NINF ? msr[x] : 0.f;
        const float ref23 = g3 ? live : mxs;
        const float e0 = __builtin_amdgcn_exp2f(fmaf(am[x][0], sc, -mxs));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(am[x][1], sc, -mxs));
        const float e2 = __builtin_amdgcn_exp2f(fmaf(am[x][2], sc, -ref23));
        const float e3 = __builtin_amdgcn_exp2f(fmaf(am[x][3], sc, -ref23));
        d23[x] = e2 + e3;
        n23[x] = fmaf(e3, val[x][3], e2 * val[x][2]);
        den[x] = fmaf(own23, d23[x], e0 + e1);
        num[x] = fmaf(own23, n23[x], fmaf(e1, val[x][1], e0 * val[x][0]));
      }
      {
        const PairX a1 = swap16x(den[0], num[0]);
        const float xa = a1.lo + a1.hi;                  // rows: (dA01, nA01, dA23, nA23)
        const PairX b1 = swap16x(den[1], num[1]);
        const float xb = b1.lo + b1.hi;
        const PairX z1 = swap32x(xa, xb);
        const float z = z1.lo + z1.hi;                   // rows: (dA, nA, dB, nB)
        const PairX z2 = swap16x(z, z);                  // lo = (dA, dA, dB, dB), hi = (nA, nA, nB, nB)
        const float o = z2.hi * __builtin_amdgcn_rcpf(z2.lo);
        if (o_lane && !(single && g >= 2)) orow[16 * tA] = o;
      }
      if (extra_writer) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (x == 1 && single) continue;
          float* sp = sp_lane + 16 * (x ? tB : tA);
          sp[0] = msr[x];
          sp[XHALF] = d23[x];
          sp[2 * XHALF] = n23[x];
        }
      }
    }
  }
  // ---- the two extra queries: combine the per-tile partial softmaxes of their seven tiles
  __syncthreads();
  for (int o = tid; o < 2 * XHALF; o += 512) {
    const int e = o / XHALF, ch = o % XHALF;
    const int qe = q0 + XTILES + e;
    if (qe < a.N) {
      const float* sp = s_part + (7 * e) * 3 * XHALF + ch;
      float m = NINF;
#pragma unroll
      for (int w = 0; w < 7; ++w) m = fmaxf(m, sp[w * 3 * XHALF]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int w = 0; w < 7; ++w) {
        const float wgt = __builtin_amdgcn_exp2f(sp[w * 3 * XHALF] - m);
        den = fmaf(wgt, sp[w * 3 * XHALF + XHALF], den);
        num = fmaf(wgt, sp[w * 3 * XHALF + 2 * XHALF], num);
      }
      a.agg[(int64_t)qe * a.ld_agg + ch0 + ch] = num / den;
    }
  }
  if (a.stamps && lane == 0 && (wave & 3) == 0 && blockIdx.x < 1024) {
    ts[4] = __builtin_amdgcn_s_memtime();
    unsigned long long* o = g_x6_stamps + ((int64_t)blockIdx.x * 2 + (wave >> 2)) * 6;
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = ts[i];
    o[5] = grp_b;
  }
}

// ---- packer: reference-layout matrices -> the kernel's fragment stream (three bf16 truncation pieces per weight)
template <typename S>
__global__ void pack_attn_split_kernel(const float* __restrict__ w2, const float* __restrict__ wp, const float* __restrict__ p2,
                                       unsigned* __restrict__ out) {
  constexpr int NP = S::NP, XW2F = XG<S>::W2F, XSF = XG<S>::SF, XSTAGE = XG<S>::STAGE;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)2 * XNSTAGE * XSTAGE;
  if (e >= total) return;
  const int word = (int)(e & 3), lane = (int)((e >> 2) & 63);
  const int frag = (int)((e / XFW) % XSF), stage = (int)((e / XSTAGE) % XNSTAGE), half = (int)(e / ((int64_t)XNSTAGE * XSTAGE));
  const int c = lane & 15, g = lane >> 4;
  unsigned res = 0u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = 2 * word + q;
    float v = 0.f, scale = S::WSCALE;
    int p = 0;
    if (frag < XW2F) {
      const int t = frag / NP;
      p = frag % NP;
      const int ch = XHALF * half + 16 * t + c;
      if (stage < XS) v = w2[(int64_t)ch * XHID + 32 * stage + 16 * (j >> 2) + 4 * g + (j & 3)];
      else v = p2[ch * 32 + 8 * g + j];
    } else if (stage < XS) {
      const int u = (frag - XW2F) / NP;
      p = (frag - XW2F) % NP;
      v = wp[(32 * stage + 16 * u + c) * 32 + 8 * g + j];      // A operand: row = lane & 15 = hidden unit
      scale = S::HSCALE;
    }
    res |= S::piece_scaled(v, p, scale) << (16 * q);
  }
  out[e] = res;
}

}  // namespace

namespace {
template <typename S> int64_t stream_floats() { return (int64_t)2 * XNSTAGE * XG<S>::STAGE; }

template <typename S>
int pack_stream(const char* who, const float* w2, const float* wp, const float* p2, float* wstream, void* stream) {
  OCC4D_REQUIRE(w2 && wp && p2 && wstream, "%s: null pointer", who);
  const int64_t total = stream_floats<S>();
  pack_attn_split_kernel<S><<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w2, wp, p2,
                                                                                      reinterpret_cast<unsigned*>(wstream));
  return occ4d::check_launch(who);
}

template <typename S, bool PRESCALED, int STORE = 0>
int launch_attn(const char* who, const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vtc, int64_t ld_vt,
                const float* pos0_w, const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg, int n, int m,
                int k, int d, float divisor, void* stream, float* logits = nullptr, float* a_out = nullptr,
                float* pe_out = nullptr, const float* c2 = nullptr) {
  OCC4D_REQUIRE(d == XD, "%s: built for d = %d, got %d", who, XD, d);
  OCC4D_REQUIRE(!STORE || (logits && ((uintptr_t)logits % 16) == 0 && (int64_t)n * k * XD < ((int64_t)1 << 31)),
                "%s: logits buffer null, misaligned or beyond 2^31 floats (chunk the queries)", who);
  OCC4D_REQUIRE(STORE != 2 || (a_out && pe_out && c2 && (((uintptr_t)a_out | (uintptr_t)pe_out) % 16) == 0 &&
                               (int64_t)n * k * XHID < ((int64_t)1 << 31)),
                "%s: a_out / pe_out / c2 null, misaligned or beyond 2^31 floats (chunk the queries)", who);
  OCC4D_REQUIRE(k >= 1 && k <= 14 && m >= 1 && n >= 0, "%s: k = %d (1 .. 14), m = %d, n = %d", who, k, m, n);
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vtc && pos0_w && pos0_b && wstream && agg, "%s: null pointer", who);
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ((uintptr_t)aq % 16) == 0 && ((uintptr_t)kt % 16) == 0 &&
                    ((uintptr_t)wstream % 16) == 0 && ld_aq >= XHID && ld_kt >= XHID && ld_vt >= XD && ld_agg >= XD &&
                    q_stride >= 3 && a_stride >= 3,
                "%s: misaligned or short rows", who);
  OCC4D_REQUIRE((int64_t)n * ld_aq < ((int64_t)1 << 29) && (int64_t)m * ld_kt < ((int64_t)1 << 29) &&
                    (int64_t)m * ld_vt < ((int64_t)1 << 31),
                "%s: 32-bit row offsets: n * ld_aq and m * ld_kt must stay below 2^29 floats", who);
  if (n == 0) return OCC4D_OK;
  AttnX6Args a{aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b,
               reinterpret_cast<const unsigned*>(wstream), agg, ld_agg, n, m, k, divisor, 0, 0, 0, 0, logits, a_out, pe_out, c2};
  static const int skew = [] { const char* e = getenv("OCC4D_X6_SKEW"); return e ? atoi(e) : 2; }();   // read once
  static const int stamps = [] { const char* e = getenv("OCC4D_X6_STAMPS"); return e ? atoi(e) : 0; }();
  a.skew = skew;
  a.stamps = stamps;
  a.groups = (int)occ4d::cdiv(n, XQPB);
  a.per = (int)occ4d::cdiv(a.groups, 4);
  cross_attn_split_kernel<S, PRESCALED, STORE><<<8 * a.per, 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch(who);
}
}  // namespace

extern "C" int64_t occ4d_pt_cross_attn_bf16x6_stream_floats(void) { return stream_floats<SplitBf16x6>(); }
extern "C" int64_t occ4d_pt_cross_attn_f16x3_stream_floats(void) { return stream_floats<SplitF16x3>(); }

extern "C" int occ4d_pack_attn_bf16x6_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream,
                                                 void* stream) {
  return pack_stream<SplitBf16x6>("occ4d_pack_attn_bf16x6_stream_f32", w2, wp, p2, wstream, stream);
}
extern "C" int occ4d_pack_attn_f16x3_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream,
                                                void* stream) {
  return pack_stream<SplitF16x3>("occ4d_pack_attn_f16x3_stream_f32", w2, wp, p2, wstream, stream);
}

extern "C" int occ4d_pt_cross_attn_bf16x6_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                              const float* apos, int64_t a_stride, const int32_t* idx, const float* kt,
                                              int64_t ld_kt, const float* vtc, int64_t ld_vt, const float* pos0_w,
                                              const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg, int n,
                                              int m, int k, int d, float divisor, void* stream) {
  return launch_attn<SplitBf16x6, false>("occ4d_pt_cross_attn_bf16x6_f32", aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt,
                                  vtc, ld_vt, pos0_w, pos0_b, wstream, agg, ld_agg, n, m, k, d, divisor, stream);
}
// ... that also leaves the pre-softmax logits in logits (n k, 416): the training forward of the split-precision step
extern "C" int occ4d_pt_cross_attn_bf16x6_logits_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                                     const float* apos, int64_t a_stride, const int32_t* idx, const float* kt,
                                                     int64_t ld_kt, const float* vtc, int64_t ld_vt, const float* pos0_w,
                                                     const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg,
                                                     float* logits, float* a_out, float* pe_out, const float* c2, int n, int m,
                                                     int k, int d, float divisor, void* stream) {
  const char* who = "occ4d_pt_cross_attn_bf16x6_logits_f32";
  OCC4D_REQUIRE((a_out != nullptr) == (pe_out != nullptr) && (a_out != nullptr) == (c2 != nullptr),
                "%s: a_out, pe_out and c2 come together (all three pair tensors) or not at all", who);
  if (a_out)
    return launch_attn<SplitBf16x6, false, 2>(who, aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w,
                                              pos0_b, wstream, agg, ld_agg, n, m, k, d, divisor, stream, logits, a_out, pe_out, c2);
  return launch_attn<SplitBf16x6, false, 1>(who, aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w,
                                            pos0_b, wstream, agg, ld_agg, n, m, k, d, divisor, stream, logits);
}
extern "C" int occ4d_pt_cross_attn_f16x3_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                             const float* apos, int64_t a_stride, const int32_t* idx, const float* kt,
                                             int64_t ld_kt, const float* vtc, int64_t ld_vt, const float* pos0_w,
                                             const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg, int n,
                                             int m, int k, int d, float divisor, void* stream) {
  return launch_attn<SplitF16x3, false>("occ4d_pt_cross_attn_f16x3_f32", aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt,
                                        ld_kt, vtc, ld_vt, pos0_w, pos0_b, wstream, agg, ld_agg, n, m, k, d, divisor, stream);
}
extern "C" float occ4d_pt_cross_attn_f16x3_hidden_scale(void) { return SplitF16x3::HSCALE; }
// The same with aq and kt already multiplied by occ4d_pt_cross_attn_f16x3_hidden_scale() (what the path-level entry points
// pass: they scale the merged matrices behind both once per weight update)
extern "C" int occ4d_pt_cross_attn_f16x3_prescaled_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                                       const float* apos, int64_t a_stride, const int32_t* idx,
                                                       const float* kt, int64_t ld_kt, const float* vtc, int64_t ld_vt,
                                                       const float* pos0_w, const float* pos0_b, const float* wstream,
                                                       float* agg, int64_t ld_agg, int n, int m, int k, int d, float divisor,
                                                       void* stream) {
  return launch_attn<SplitF16x3, true>("occ4d_pt_cross_attn_f16x3_prescaled_f32", aq, ld_aq, qpos, q_stride, apos, a_stride, idx,
                                       kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b, wstream, agg, ld_agg, n, m, k, d, divisor, stream);
}

// debug: the phase stamps of the last launch (OCC4D_X6_STAMPS=1): out[(workgroup * 2 + wave / 4) * 6 + i]
extern "C" int occ4d_debug_x6_stamps(unsigned long long* out, int n_words) {
  OCC4D_REQUIRE(out && n_words >= 0 && n_words <= 1024 * 2 * 6, "occ4d_debug_x6_stamps: bad arguments");
  const hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6_stamps), (size_t)n_words * 8);
  OCC4D_REQUIRE(e == hipSuccess, "occ4d_debug_x6_stamps: %s", hipGetErrorString(e));
  return OCC4D_OK;
}
