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
#include <stdlib.h>

#include <type_traits>

#include "bf16x6.hpp"

namespace {

constexpr int XD = 416;                   // channels
constexpr int XHID = 2 * XD;              // hidden units of attn_mlp
constexpr int XHALF = XD / 2;             // channels per workgroup
constexpr int XT = XHALF / 16;            // 13 channel tiles
constexpr int XS = XHID / 32;             // 26 hidden stages of 32
constexpr int XFW = 256;                  // u32 words per fragment image (64 lanes x 16 B)
constexpr int XW2F = 3 * XT;              // 39 W2 fragments of a stage
constexpr int XSF = XW2F + 6;             // + 2 x 3 Wp fragments = 45
constexpr int XSTAGE = XSF * XFW;         // 11520 words = 46080 B
constexpr int XNSTAGE = XS + 1;           // + the P2 stage
constexpr int XWAVES = 8;
constexpr int XTILES = 2 * XWAVES;        // 16 row tiles per workgroup
constexpr int XQPB = XTILES + 2;          // 18 queries per workgroup: one per row tile + two spread over the tiles' rows 14, 15

struct AttnX6Args {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;         // Wv f + c2
  const float* P1; const float* c1;
  const unsigned* wstream;                // [half][XNSTAGE][XSF][64 lanes][4 words]
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
  int groups, per;                        // query groups of 16; groups per XCD slab
  int skew;                               // phase-skew grouping of the waves (see the kernel)
  int stamps;                             // debug: record phase time stamps
};

// debug (OCC4D_X6_STAMPS=1): s_memtime at the phase boundaries of waves 0 and 4 of the first 1024 workgroups
__device__ unsigned long long g_x6_stamps[1024 * 2 * 6];

__global__ __launch_bounds__(512, 2) void cross_attn_bf16x6_kernel(const AttnX6Args a) {
  // a ring of three stage buffers: stage s lives in buffer s % 3
  __shared__ __attribute__((aligned(16))) unsigned buf0[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf2[XSTAGE];
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

  // this wave's i-th fragment of a stage (six per wave and stage; the tail repeats fragment 44: same bytes, same place)
  auto dma_part = [&](int stage_no, const unsigned* dst, int i) {
    const int f = min(wave + XWAVES * i, XSF - 1);                                  // wave-uniform
    dma_frag_x(wst + (int64_t)stage_no * XSTAGE + f * XFW, lds_addr_x(dst) + (unsigned)f * (XFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_part(0, buf0, i);
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
  Split rs[2];                            // r = relu(P1 d + c1), hidden-pos units 8 g + j, three pieces
  unsigned aq_off[2], kt_off[2];
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
    rs[rt] = split8(f32x4{rr[0], rr[1], rr[2], rr[3]}, f32x4{rr[4], rr[5], rr[6], rr[7]});
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
    for (int i = 0; i < 6; ++i) dma_part(1, buf1, i);
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
        h[rt][u] = ia[rt][u] - ik[rt][u];
    // next stage's slices (consumed at its top)
    const int sn = s + 1 < XS ? s + 1 : s;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ia[rt][u] = slice(a.aq + 32 * sn + 16 * u, aq_off[rt]);
        ik[rt][u] = slice(a.kt + 32 * sn + 16 * u, kt_off[rt]);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const u32x4 wh = *reinterpret_cast<const u32x4*>(f + (XW2F + 3 * u + 0) * XFW);
      const u32x4 wm = *reinterpret_cast<const u32x4*>(f + (XW2F + 3 * u + 1) * XFW);
      const u32x4 wl = *reinterpret_cast<const u32x4*>(f + (XW2F + 3 * u + 2) * XFW);
      mm6x2_b(wh, wm, wl, rs[0], rs[1], h[0][u], h[1][u]);
    }
    // ---- ReLU + three-way split: GEMM2's A operand of both row tiles
    Split hs[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) hs[rt] = split8(relu4x(h[rt][0]), relu4x(h[rt][1]));
    // ---- GEMM2: 13 channel tiles x (3 fragment reads, 12 MFMAs)
    u32x4 bh = *reinterpret_cast<const u32x4*>(f);
    u32x4 bm = *reinterpret_cast<const u32x4*>(f + XFW);
    u32x4 bl = *reinterpret_cast<const u32x4*>(f + 2 * XFW);
#pragma unroll
    for (int t = 0; t < XT; ++t) {
      const u32x4 ch = bh, cm = bm, cl = bl;
      if (t + 1 < XT) {
        bh = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1)) * XFW);
        bm = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1) + 1) * XFW);
        bl = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1) + 2) * XFW);
      }
      if (t < 6) {
        if (!grp_b && s + 1 < XNSTAGE) dma_part(s + 1, dA, t);
      } else if (t > 6) {
        if (grp_b && s + 2 < XNSTAGE) dma_part(s + 2, dB, t - 7);
      }
      mm6x2(hs[0], hs[1], ch, cm, cl, acc[0][t], acc[1][t]);
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
  const float sc = LOG2E / a.divisor;
  const float NINF = -__builtin_inff();
  const unsigned* fp = buf2 + lane * 4;
  float* const s_part = reinterpret_cast<float*>(buf0);          // [tile 16][3][XHALF]
  const bool g3 = g == 3;
  const float own23 = g3 ? 0.f : 1.f;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int tile = 2 * wave + rt;
    const int qm = q0 + tile;
    int voff[4];
    bool act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      const int ql = row < 14 ? tile : min(XTILES + tile / 7, XQPB - 1);
      const int slot = row < 14 ? row : 2 * (tile % 7) + row - 14;
      voff[i] = s_idx[ql * 16 + slot] * (int)a.ld_vt + ch0 + c;
      act[i] = slot < a.K && (row < 14 || tile < 14);
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
        const u32x4 ph = *reinterpret_cast<const u32x4*>(fp + (3 * t) * XFW);
        const u32x4 pm = *reinterpret_cast<const u32x4*>(fp + (3 * t + 1) * XFW);
        const u32x4 pl = *reinterpret_cast<const u32x4*>(fp + (3 * t + 2) * XFW);
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        e = mm(rs[rt].l, ph, e);
        e = mm(rs[rt].h, pl, e);
        e = mm(rs[rt].m, pm, e);
        e = mm(rs[rt].m, ph, e);
        e = mm(rs[rt].h, pm, e);
        e = mm(rs[rt].h, ph, e);
        pe[x] = e;
      }
      float am[2][4], val[2][4], m23[2], lm[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const f32x4 av = acc[rt][x ? tB : tA];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          am[x][i] = act[i] ? av[i] : NINF;
          val[x][i] = pe[x][i] + vq[x][i];
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

// =====================================================================================================================
// PAIRED workgroups (the architecture of csrc/crossattn16p.hip): the kernel above owns a CU (138 KB of LDS), so its
// prologue, its softmax epilogue and the tail of its phase skew have no partner -- measured 11 % of the workgroup's time
// (profiles/stamp_x6.py).  Here a workgroup has FOUR waves (one per SIMD; 8 row tiles: 8 queries + one extra query in the
// tiles' rows 14, 15) and streams the weights in SUB-stages -- (s, A): the GEMM1 fragments + channel tiles 0-5 of hidden
// stage s, (s, B): channel tiles 6-12; 24 fragments = 24 KB each, a ring of three -- so TWO workgroups share a CU (2 x 73
// KB).  They are independent (own barriers, own stream) and started out of phase, so one's barrier / DMA waits, VALU
// phases, prologue and epilogue sit under the other's MFMA stream.
constexpr int ZWAVES = 4;
constexpr int ZTILES = 2 * ZWAVES;        // 8 row tiles
constexpr int ZQPB = ZTILES + 1;          // 9 queries per workgroup
constexpr int ZSF = 24;                   // fragments per sub-stage
constexpr int ZSUB = ZSF * XFW;           // 6144 words = 24576 B
constexpr int ZNSUB = 2 * XS + 2;         // 52 hidden sub-stages + 2 of P2
constexpr int ZTA = 6;                    // channel tiles of an A sub-stage (B: 7)

struct AttnX6pArgs {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;         // Wv f + c2
  const float* P1; const float* c1;
  const unsigned* wstream;                // [half][ZNSUB][ZSF][64 lanes][4 words]
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
  int groups, per;                        // query groups of 9; groups per XCD slab
  int first_round, skew;                  // phase skew of the two workgroups of a CU (as in crossattn16p.hip)
};

__global__ __launch_bounds__(256, 2) void cross_attn_bf16x6p_kernel(const AttnX6pArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned slot0[ZSUB];
  __shared__ __attribute__((aligned(16))) unsigned slot1[ZSUB];
  __shared__ __attribute__((aligned(16))) unsigned slot2[ZSUB];
  __shared__ __attribute__((aligned(16))) float s_p1[32 * 4];
  __shared__ int s_idx[ZQPB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int x = blockIdx.x & 7, half = x >> 2, slab = x & 3, in_slab = blockIdx.x >> 3;
  const int group = slab * a.per + in_slab;
  if (in_slab >= a.per || group >= a.groups) return;
  const int q0 = group * ZQPB;
  const int ch0 = XHALF * half;
  const unsigned* const wst = a.wstream + (int64_t)half * ZNSUB * ZSUB;
  const unsigned lane16 = lane * 16;

  // this wave's i-th fragment (of six) of sub-stage `sub`
  auto dma_part = [&](int sub, const unsigned* dst, int i) {
    const int f = wave + ZWAVES * i;
    dma_frag_x(wst + (int64_t)sub * ZSUB + f * XFW, lds_addr_x(dst) + (unsigned)f * (XFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_part(0, slot0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_part(1, slot1, i);
  if (tid < ZQPB * 16) {
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
  // phase skew: the workgroup in the ODD wave slot of its SIMDs sleeps once in the first dispatch round; later
  // workgroups inherit slot and phase of the one they replace (performance only)
  if (a.skew > 0 && (int)blockIdx.x < a.first_round) {
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)
      for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(127);
  }
  __syncthreads();

  Split rs[2];
  unsigned aq_off[2], kt_off[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int tile = 2 * wave + rt;
    const int ql = c < 14 ? tile : ZTILES;
    const int slot = c < 14 ? c : 2 * tile + c - 14;          // (tile 7: slots 14, 15 of the extra query do not exist)
    const bool my_valid = slot < a.K && slot < 14;
    const int my_q = min(q0 + ql, a.N - 1);
    const int my_j = s_idx[ql * 16 + min(slot, 15)];
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
    rs[rt] = split8(f32x4{rr[0], rr[1], rr[2], rr[3]}, f32x4{rr[4], rr[5], rr[6], rr[7]});
    aq_off[rt] = (unsigned)(my_q * (int)a.ld_aq + 4 * g) * 4u;
    kt_off[rt] = (unsigned)(my_j * (int)a.ld_kt + 4 * g) * 4u;
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
  f32x4 acc[2][XT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < XT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  Split hs[2];
  dma_wait_x();
  __syncthreads();

  // Sub-stage sigma lives in slot sigma % 3.  At its top the wave issues its fragments of sub-stage sigma + 2 into the slot
  // sub-stage sigma - 1 has just left (everybody passed the barrier that ended it); before the barrier that ends sigma it
  // waits for all its DMA, so sub-stage sigma + 1 (issued a whole sub-stage earlier) is complete for every wave behind it.
  // (s, A): GEMM1 of hidden stage s (fragments 0-5: Wp tiles u = 0, 1 x 3 pieces), ReLU + split, channel tiles 0-5
  auto sub_a = [&](const int s, const unsigned* __restrict__ cur, const unsigned* dst) {
    const unsigned* f = cur + lane * 4;
    f32x4 h[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) h[rt][u] = ia[rt][u] - ik[rt][u];
    const int sn = s + 1 < XS ? s + 1 : s;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ia[rt][u] = slice(a.aq + 32 * sn + 16 * u, aq_off[rt]);
        ik[rt][u] = slice(a.kt + 32 * sn + 16 * u, kt_off[rt]);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const u32x4 wh = *reinterpret_cast<const u32x4*>(f + (3 * u + 0) * XFW);
      const u32x4 wm = *reinterpret_cast<const u32x4*>(f + (3 * u + 1) * XFW);
      const u32x4 wl = *reinterpret_cast<const u32x4*>(f + (3 * u + 2) * XFW);
      mm6x2_b(wh, wm, wl, rs[0], rs[1], h[0][u], h[1][u]);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) hs[rt] = split8(relu4x(h[rt][0]), relu4x(h[rt][1]));
    u32x4 bh = *reinterpret_cast<const u32x4*>(f + 6 * XFW);
    u32x4 bm = *reinterpret_cast<const u32x4*>(f + 7 * XFW);
    u32x4 bl = *reinterpret_cast<const u32x4*>(f + 8 * XFW);
#pragma unroll
    for (int t = 0; t < ZTA; ++t) {
      const u32x4 ch = bh, cm = bm, cl = bl;
      if (t + 1 < ZTA) {
        bh = *reinterpret_cast<const u32x4*>(f + (6 + 3 * (t + 1)) * XFW);
        bm = *reinterpret_cast<const u32x4*>(f + (6 + 3 * (t + 1) + 1) * XFW);
        bl = *reinterpret_cast<const u32x4*>(f + (6 + 3 * (t + 1) + 2) * XFW);
      }
      dma_part(2 * s + 2, dst, t);
      mm6x2(hs[0], hs[1], ch, cm, cl, acc[0][t], acc[1][t]);
    }
    dma_wait_x();
    __syncthreads();
  };
  // (s, B): channel tiles 6-12 (fragments 3 (t - 6) + piece)
  auto sub_b = [&](const int s, const unsigned* __restrict__ cur, const unsigned* dst) {
    const unsigned* f = cur + lane * 4;
    u32x4 bh = *reinterpret_cast<const u32x4*>(f);
    u32x4 bm = *reinterpret_cast<const u32x4*>(f + XFW);
    u32x4 bl = *reinterpret_cast<const u32x4*>(f + 2 * XFW);
#pragma unroll
    for (int t = ZTA; t < XT; ++t) {
      const u32x4 ch = bh, cm = bm, cl = bl;
      if (t + 1 < XT) {
        bh = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1 - ZTA)) * XFW);
        bm = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1 - ZTA) + 1) * XFW);
        bl = *reinterpret_cast<const u32x4*>(f + (3 * (t + 1 - ZTA) + 2) * XFW);
      }
      if (t - ZTA < 6) dma_part(2 * s + 3, dst, t - ZTA);
      mm6x2(hs[0], hs[1], ch, cm, cl, acc[0][t], acc[1][t]);
    }
    dma_wait_x();
    __syncthreads();
  };
  // three hidden stages = six sub-stages = two turns of the ring
#pragma clang loop unroll(disable)
  for (int s = 0; s < XS - 2; s += 3) {
    sub_a(s, slot0, slot2);
    sub_b(s, slot1, slot0);
    sub_a(s + 1, slot2, slot1);
    sub_b(s + 1, slot0, slot2);
    sub_a(s + 2, slot1, slot0);
    sub_b(s + 2, slot2, slot1);
  }
  sub_a(XS - 2, slot0, slot2);            // sub-stages 48 .. 51; the last two DMA the P2 sub-stages (52 -> slot1, 53 -> slot2)
  sub_b(XS - 2, slot1, slot0);
  sub_a(XS - 1, slot2, slot1);
  sub_b(XS - 1, slot0, slot2);

  // ---- epilogue (see the kernel above): P2 tiles 0-5 in slot1 (fragments 6 + 3 t + piece), tiles 6-12 in slot2; the extra
  // query's per-tile partials go to slot0 (sub-stage 51 has been left by everybody)
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor;
  const float NINF = -__builtin_inff();
  float* const s_part = reinterpret_cast<float*>(slot0);         // [tile 8][3][XHALF]
  const bool g3 = g == 3;
  const float own23 = g3 ? 0.f : 1.f;
  auto p2frag = [&](int t, int piece) {
    const unsigned* base = t < ZTA ? slot1 + (6 + 3 * t + piece) * XFW : slot2 + (3 * (t - ZTA) + piece) * XFW;
    return *reinterpret_cast<const u32x4*>(base + lane * 4);
  };
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int tile = 2 * wave + rt;
    const int qm = q0 + tile;
    int voff[4];
    bool act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      const int ql = row < 14 ? tile : ZTILES;
      const int slot = row < 14 ? row : 2 * tile + row - 14;
      voff[i] = s_idx[ql * 16 + min(slot, 15)] * (int)a.ld_vt + ch0 + c;
      act[i] = slot < a.K && slot < 14;
    }
    float* const orow = a.agg + (int64_t)min(qm, a.N - 1) * a.ld_agg + ch0 + 16 * (g >> 1) + c;
    const bool o_lane = (g & 1) == 0 && qm < a.N;
    float* const sp_lane = s_part + tile * 3 * XHALF + c;
    const bool extra_writer = g3 && tile < 7;
#pragma unroll
    for (int tp = 0; tp < (XT + 1) / 2; ++tp) {
      const int tA = 2 * tp, tB = min(2 * tp + 1, XT - 1);
      const bool single = 2 * tp + 1 >= XT;
      float vq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vq[0][i] = a.vt[voff[i] + 16 * tA];
        vq[1][i] = a.vt[voff[i] + 16 * tB];
      }
      f32x4 pe[2];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2) {
        const int t = x2 ? tB : tA;
        const u32x4 ph = p2frag(t, 0), pm = p2frag(t, 1), pl = p2frag(t, 2);
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        e = mm(rs[rt].l, ph, e);
        e = mm(rs[rt].h, pl, e);
        e = mm(rs[rt].m, pm, e);
        e = mm(rs[rt].m, ph, e);
        e = mm(rs[rt].h, pm, e);
        e = mm(rs[rt].h, ph, e);
        pe[x2] = e;
      }
      float am[2][4], val[2][4], m23[2], lm[2];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2) {
        const f32x4 av = acc[rt][x2 ? tB : tA];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          am[x2][i] = act[i] ? av[i] : NINF;
          val[x2][i] = pe[x2][i] + vq[x2][i];
        }
        m23[x2] = fmaxf(am[x2][2], am[x2][3]);
        lm[x2] = fmaxf(fmaxf(am[x2][0], am[x2][1]), g3 ? NINF : m23[x2]);
      }
      float mx[2];
      {
        const PairX p1 = swap16x(lm[0], lm[1]);
        const float m1 = fmaxf(p1.lo, p1.hi);
        const PairX p2 = swap32x(m1, m1);
        const float m2 = fmaxf(p2.lo, p2.hi);
        const PairX p3 = swap16x(m2, m2);
        mx[0] = p3.lo;
        mx[1] = p3.hi;
      }
      float den[2], num[2], d23[2], n23[2], msr[2];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2) {
        const float mxs = mx[x2] * sc;
        msr[x2] = m23[x2] * sc;
        const float live = msr[x2] > NINF ? msr[x2] : 0.f;
        const float ref23 = g3 ? live : mxs;
        const float e0 = __builtin_amdgcn_exp2f(fmaf(am[x2][0], sc, -mxs));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(am[x2][1], sc, -mxs));
        const float e2 = __builtin_amdgcn_exp2f(fmaf(am[x2][2], sc, -ref23));
        const float e3 = __builtin_amdgcn_exp2f(fmaf(am[x2][3], sc, -ref23));
        d23[x2] = e2 + e3;
        n23[x2] = fmaf(e3, val[x2][3], e2 * val[x2][2]);
        den[x2] = fmaf(own23, d23[x2], e0 + e1);
        num[x2] = fmaf(own23, n23[x2], fmaf(e1, val[x2][1], e0 * val[x2][0]));
      }
      {
        const PairX a1 = swap16x(den[0], num[0]);
        const float xa = a1.lo + a1.hi;
        const PairX b1 = swap16x(den[1], num[1]);
        const float xb = b1.lo + b1.hi;
        const PairX z1 = swap32x(xa, xb);
        const float z = z1.lo + z1.hi;
        const PairX z2 = swap16x(z, z);
        const float o = z2.hi * __builtin_amdgcn_rcpf(z2.lo);
        if (o_lane && !(single && g >= 2)) orow[16 * tA] = o;
      }
      if (extra_writer) {
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) {
          if (x2 == 1 && single) continue;
          float* sp = sp_lane + 16 * (x2 ? tB : tA);
          sp[0] = msr[x2];
          sp[XHALF] = d23[x2];
          sp[2 * XHALF] = n23[x2];
        }
      }
    }
  }
  __syncthreads();
  const int qe = q0 + ZTILES;
  if (qe < a.N && tid < XHALF) {
    const float* sp = s_part + tid;
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
    a.agg[(int64_t)qe * a.ld_agg + ch0 + tid] = num / den;
  }
}

__global__ void pack_attn_bf16x6p_kernel(const float* __restrict__ w2, const float* __restrict__ wp, const float* __restrict__ p2,
                                         unsigned* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)2 * ZNSUB * ZSUB;
  if (e >= total) return;
  const int word = (int)(e & 3), lane = (int)((e >> 2) & 63);
  const int frag = (int)((e / XFW) % ZSF), sub = (int)((e / ZSUB) % ZNSUB), half = (int)(e / ((int64_t)ZNSUB * ZSUB));
  const int c = lane & 15, g = lane >> 4;
  const int stage = sub >> 1;                  // hidden stage (26 = P2)
  const bool is_b = sub & 1;
  // which operand sits in this fragment: Wp tile u (A sub-stages, fragments 0-5) or channel tile t
  int t = -1, u = -1, p = 0;
  if (!is_b) {
    if (frag < 6) { u = frag / 3; p = frag % 3; }
    else { t = (frag - 6) / 3; p = (frag - 6) % 3; }
  } else if (frag < 21) {
    t = ZTA + frag / 3; p = frag % 3;
  }
  unsigned res = 0u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = 2 * word + q;
    float v = 0.f;
    if (t >= 0) {
      const int ch = XHALF * half + 16 * t + c;
      if (stage < XS) v = w2[(int64_t)ch * XHID + 32 * stage + 16 * (j >> 2) + 4 * g + (j & 3)];
      else v = p2[ch * 32 + 8 * g + j];
    } else if (u >= 0 && stage < XS) {
      v = wp[(32 * stage + 16 * u + c) * 32 + 8 * g + j];
    }
    res |= piece16(v, p) << (16 * q);
  }
  out[e] = res;
}

// =====================================================================================================================
// ONE ROW TILE PER WAVE, ALL 416 CHANNELS (round 5, third cut).  Measured on the two kernels above: the matrix pipe is
// busy 67 % of the time in both, whatever overlaps with what -- a v_mfma_f32_16x16x32_bf16 costs ~17 cycles and every VALU
// instruction of a wave its ~4 cycles of the same issue port, so what is left to gain is fewer instructions.  With the
// channel halves in separate workgroups every half recomputes GEMM1 and the ReLU + three-way split of the hidden
// activations (24 + 156 MFMAs and 104 VALU instructions per wave and stage).  Here a wave owns ONE row tile and all 26
// channel tiles (104 accumulators as before): GEMM1 and the split happen once per row tile (12 + 156 MFMAs, 52 VALU per
// hidden stage: -7 % matrix work, half the VALU work per row); the weights stream as before in stages of 13 channel
// tiles -- stage sigma = 2 s + half of the same packed stream -- so a weight fragment read from LDS now serves one row
// tile (110 of the CU's 256 B / clk), and the L2 -> LDS stream doubles to ~15 B / clk per CU.  Eight waves, one
// workgroup per CU, 8 queries + one extra in the tiles' rows 14, 15; ring of three stage buffers, phase skew of the two
// waves of a SIMD, barrier protocol: exactly the first kernel's, on 52 half-stages.
constexpr int CTILES = XWAVES;            // 8 row tiles
constexpr int CQPB = CTILES + 1;          // 9 queries per workgroup
constexpr int CT = 2 * XT;                // 26 channel tiles per wave
constexpr int CNS = 2 * XS;               // 52 compute stages (+ 2 of P2)

__global__ __launch_bounds__(512, 2) void cross_attn_bf16x6c_kernel(const AttnX6Args a) {
  __shared__ __attribute__((aligned(16))) unsigned buf0[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[XSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf2[XSTAGE];
  __shared__ __attribute__((aligned(16))) float s_p1[32 * 4];
  __shared__ int s_idx[CQPB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // every XCD one contiguous slab of query groups (its L2: the whole stream, 2.4 MB, and the Kt / Vt rows of that slab)
  const int xcd = blockIdx.x & 7, in_slab = blockIdx.x >> 3;
  const int group = xcd * a.per + in_slab;
  if (in_slab >= a.per || group >= a.groups) return;
  const int q0 = group * CQPB;
  const unsigned lane16 = lane * 16;
  const bool grp_b = a.skew == 2 ? (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) != 0
                                 : (a.skew == 1 && wave >= 4);
  // stage sigma of the sequence = stage sigma / 2 of channel half sigma % 2 in the packed stream
  auto dma_part = [&](int sigma, const unsigned* dst, int i) {
    const int f = min(wave + XWAVES * i, XSF - 1);
    dma_frag_x(a.wstream + ((int64_t)(sigma & 1) * XNSTAGE + (sigma >> 1)) * XSTAGE + f * XFW,
               lds_addr_x(dst) + (unsigned)f * (XFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_part(0, buf0, i);
  if (tid < CQPB * 16) {
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

  // this lane's pair: column c < 14 = neighbour slot c of query q0 + wave; c = 14, 15 = slots 2 wave + c - 14 of the extra query
  const int tile = wave;
  Split rs;
  unsigned aq_off, kt_off;
  {
    const int ql = c < 14 ? tile : CTILES;
    const int slot = c < 14 ? c : 2 * tile + c - 14;
    const bool my_valid = slot < a.K && slot < 14;
    const int my_q = min(q0 + ql, a.N - 1);
    const int my_j = s_idx[ql * 16 + min(slot, 15)];
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
    rs = split8(f32x4{rr[0], rr[1], rr[2], rr[3]}, f32x4{rr[4], rr[5], rr[6], rr[7]});
    aq_off = (unsigned)(my_q * (int)a.ld_aq + 4 * g) * 4u;
    kt_off = (unsigned)(my_j * (int)a.ld_kt + 4 * g) * 4u;
  }
  auto slice = [](const float* base, unsigned off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  f32x4 ia[2], ik[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    ia[u] = slice(a.aq + 16 * u, aq_off);
    ik[u] = slice(a.kt + 16 * u, kt_off);
  }
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  Split hs;
  dma_wait_x();
  __builtin_amdgcn_s_barrier();                       // barrier 0
  if (grp_b) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_part(1, buf1, i);
  }

  // six MFMAs of a tile, two tiles at a time on alternating accumulators (a wave has one row tile: the two chains are two
  // CHANNEL tiles)
  auto mm6pair = [&](const u32x4 (&b0)[3], const u32x4 (&b1)[3], f32x4& c0, f32x4& c1) {
    c0 = mm(hs.l, b0[0], c0); c1 = mm(hs.l, b1[0], c1);
    c0 = mm(hs.h, b0[2], c0); c1 = mm(hs.h, b1[2], c1);
    c0 = mm(hs.m, b0[1], c0); c1 = mm(hs.m, b1[1], c1);
    c0 = mm(hs.m, b0[0], c0); c1 = mm(hs.m, b1[0], c1);
    c0 = mm(hs.h, b0[1], c0); c1 = mm(hs.h, b1[1], c1);
    c0 = mm(hs.h, b0[0], c0); c1 = mm(hs.h, b1[0], c1);
  };
  // one stage of the sequence (HALF: its channel half, compile time); A issues stage sigma + 1 behind tile pairs 0-5 (-> dA),
  // B stage sigma + 2 behind the barrier that follows tile pair 2 (-> dB)
  auto stage = [&](auto halfc, const int sigma, const unsigned* __restrict__ cur, const unsigned* dA, const unsigned* dB) {
    constexpr int HALF = decltype(halfc)::value;
    const unsigned* f = cur + lane * 4;
    if (HALF == 0) {
      const int s = sigma >> 1;
      f32x4 h[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) h[u] = ia[u] - ik[u];
      const int sn = s + 1 < XS ? s + 1 : s;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ia[u] = slice(a.aq + 32 * sn + 16 * u, aq_off);
        ik[u] = slice(a.kt + 32 * sn + 16 * u, kt_off);
      }
      u32x4 w0[3], w1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        w0[p] = *reinterpret_cast<const u32x4*>(f + (XW2F + p) * XFW);
        w1[p] = *reinterpret_cast<const u32x4*>(f + (XW2F + 3 + p) * XFW);
      }
      // GEMM1, transposed: the Wp fragment is the A operand, r the B operand; the two hidden tiles alternate
      h[0] = mm(w0[2], rs.h, h[0]); h[1] = mm(w1[2], rs.h, h[1]);
      h[0] = mm(w0[0], rs.l, h[0]); h[1] = mm(w1[0], rs.l, h[1]);
      h[0] = mm(w0[1], rs.m, h[0]); h[1] = mm(w1[1], rs.m, h[1]);
      h[0] = mm(w0[1], rs.h, h[0]); h[1] = mm(w1[1], rs.h, h[1]);
      h[0] = mm(w0[0], rs.m, h[0]); h[1] = mm(w1[0], rs.m, h[1]);
      h[0] = mm(w0[0], rs.h, h[0]); h[1] = mm(w1[0], rs.h, h[1]);
      hs = split8(relu4x(h[0]), relu4x(h[1]));
    }
    // 13 channel tiles as 6 pairs + 1 (the last tile runs its six MFMAs alone)
#pragma unroll
    for (int tp = 0; tp < 7; ++tp) {
      const int t0 = 2 * tp, t1 = min(2 * tp + 1, XT - 1);
      u32x4 b0[3], b1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        b0[p] = *reinterpret_cast<const u32x4*>(f + (3 * t0 + p) * XFW);
        b1[p] = *reinterpret_cast<const u32x4*>(f + (3 * t1 + p) * XFW);
      }
      if (tp < 6) {
        if (tp < 3) {
          if (!grp_b && sigma + 1 < CNS + 2) { dma_part(sigma + 1, dA, 2 * tp); dma_part(sigma + 1, dA, 2 * tp + 1); }
        } else {
          if (grp_b && sigma + 2 < CNS + 2) { dma_part(sigma + 2, dB, 2 * (tp - 3)); dma_part(sigma + 2, dB, 2 * (tp - 3) + 1); }
        }
        mm6pair(b0, b1, acc[XT * HALF + t0], acc[XT * HALF + t1]);
      } else {
        f32x4& c0 = acc[XT * HALF + t0];
        c0 = mm(hs.l, b0[0], c0);
        c0 = mm(hs.h, b0[2], c0);
        c0 = mm(hs.m, b0[1], c0);
        c0 = mm(hs.m, b0[0], c0);
        c0 = mm(hs.h, b0[1], c0);
        c0 = mm(hs.h, b0[0], c0);
      }
      if (tp == 2 && grp_b) {
        dma_wait_x();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!grp_b) {
      dma_wait_x();
      __builtin_amdgcn_s_barrier();
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // six stages = two turns of the ring, three hidden stages
#pragma clang loop unroll(disable)
  for (int sg = 0; sg < CNS - 4; sg += 6) {
    stage(H0{}, sg, buf0, buf1, buf2);
    stage(H1{}, sg + 1, buf1, buf2, buf0);
    stage(H0{}, sg + 2, buf2, buf0, buf1);
    stage(H1{}, sg + 3, buf0, buf1, buf2);
    stage(H0{}, sg + 4, buf1, buf2, buf0);
    stage(H1{}, sg + 5, buf2, buf0, buf1);
  }
  stage(H0{}, CNS - 4, buf0, buf1, buf2);            // 48 .. 51; their DMAs bring the P2 stages 52 (-> buf1) and 53 (-> buf2)
  stage(H1{}, CNS - 3, buf1, buf2, buf0);
  stage(H0{}, CNS - 2, buf2, buf0, buf1);
  stage(H1{}, CNS - 1, buf0, buf1, buf2);
  // buf0 (stage 51) becomes the extra query's scratch: group A waits here until group B has left it as well
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: P2 of channel half 0 in buf1, of half 1 in buf2 (fragments 3 t + piece)
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor;
  const float NINF = -__builtin_inff();
  float* const s_part = reinterpret_cast<float*>(buf0);          // [tile 7][3][XD] = 34944 B
  const bool g3 = g == 3;
  const float own23 = g3 ? 0.f : 1.f;
  auto p2frag = [&](int t, int piece) {
    const unsigned* base = (t < XT ? buf1 + (3 * t + piece) * XFW : buf2 + (3 * (t - XT) + piece) * XFW);
    return *reinterpret_cast<const u32x4*>(base + lane * 4);
  };
  {
    const int qm = q0 + tile;
    int voff[4];
    bool act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      const int ql = row < 14 ? tile : CTILES;
      const int slot = row < 14 ? row : 2 * tile + row - 14;
      voff[i] = s_idx[ql * 16 + min(slot, 15)] * (int)a.ld_vt + c;
      act[i] = slot < a.K && slot < 14;
    }
    float* const orow = a.agg + (int64_t)min(qm, a.N - 1) * a.ld_agg + 16 * (g >> 1) + c;
    const bool o_lane = (g & 1) == 0 && qm < a.N;
    float* const sp_lane = s_part + tile * 3 * XD + c;
    const bool extra_writer = g3 && tile < 7;
#pragma unroll
    for (int tp = 0; tp < CT / 2; ++tp) {
      const int tA = 2 * tp, tB = 2 * tp + 1;
      float vq[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vq[0][i] = a.vt[voff[i] + 16 * tA];
        vq[1][i] = a.vt[voff[i] + 16 * tB];
      }
      f32x4 pe[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      {
        u32x4 pa[3], pb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { pa[p] = p2frag(tA, p); pb[p] = p2frag(tB, p); }
        pe[0] = mm(rs.l, pa[0], pe[0]); pe[1] = mm(rs.l, pb[0], pe[1]);
        pe[0] = mm(rs.h, pa[2], pe[0]); pe[1] = mm(rs.h, pb[2], pe[1]);
        pe[0] = mm(rs.m, pa[1], pe[0]); pe[1] = mm(rs.m, pb[1], pe[1]);
        pe[0] = mm(rs.m, pa[0], pe[0]); pe[1] = mm(rs.m, pb[0], pe[1]);
        pe[0] = mm(rs.h, pa[1], pe[0]); pe[1] = mm(rs.h, pb[1], pe[1]);
        pe[0] = mm(rs.h, pa[0], pe[0]); pe[1] = mm(rs.h, pb[0], pe[1]);
      }
      float am[2][4], val[2][4], m23[2], lm[2];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2) {
        const f32x4 av = acc[x2 ? tB : tA];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          am[x2][i] = act[i] ? av[i] : NINF;
          val[x2][i] = pe[x2][i] + vq[x2][i];
        }
        m23[x2] = fmaxf(am[x2][2], am[x2][3]);
        lm[x2] = fmaxf(fmaxf(am[x2][0], am[x2][1]), g3 ? NINF : m23[x2]);
      }
      float mx[2];
      {
        const PairX p1 = swap16x(lm[0], lm[1]);
        const float m1 = fmaxf(p1.lo, p1.hi);
        const PairX p2 = swap32x(m1, m1);
        const float m2 = fmaxf(p2.lo, p2.hi);
        const PairX p3 = swap16x(m2, m2);
        mx[0] = p3.lo;
        mx[1] = p3.hi;
      }
      float den[2], num[2], d23[2], n23[2], msr[2];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2) {
        const float mxs = mx[x2] * sc;
        msr[x2] = m23[x2] * sc;
        const float live = msr[x2] > NINF ? msr[x2] : 0.f;
        const float ref23 = g3 ? live : mxs;
        const float e0 = __builtin_amdgcn_exp2f(fmaf(am[x2][0], sc, -mxs));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(am[x2][1], sc, -mxs));
        const float e2 = __builtin_amdgcn_exp2f(fmaf(am[x2][2], sc, -ref23));
        const float e3 = __builtin_amdgcn_exp2f(fmaf(am[x2][3], sc, -ref23));
        d23[x2] = e2 + e3;
        n23[x2] = fmaf(e3, val[x2][3], e2 * val[x2][2]);
        den[x2] = fmaf(own23, d23[x2], e0 + e1);
        num[x2] = fmaf(own23, n23[x2], fmaf(e1, val[x2][1], e0 * val[x2][0]));
      }
      {
        const PairX a1 = swap16x(den[0], num[0]);
        const float xa = a1.lo + a1.hi;
        const PairX b1 = swap16x(den[1], num[1]);
        const float xb = b1.lo + b1.hi;
        const PairX z1 = swap32x(xa, xb);
        const float z = z1.lo + z1.hi;
        const PairX z2 = swap16x(z, z);
        const float o = z2.hi * __builtin_amdgcn_rcpf(z2.lo);
        if (o_lane) orow[16 * tA] = o;
      }
      if (extra_writer) {
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) {
          float* sp = sp_lane + 16 * (x2 ? tB : tA);
          sp[0] = msr[x2];
          sp[XD] = d23[x2];
          sp[2 * XD] = n23[x2];
        }
      }
    }
  }
  __syncthreads();
  const int qe = q0 + CTILES;
  if (qe < a.N && tid < XD) {
    const float* sp = s_part + tid;
    float m = NINF;
#pragma unroll
    for (int w = 0; w < 7; ++w) m = fmaxf(m, sp[w * 3 * XD]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int w = 0; w < 7; ++w) {
      const float wgt = __builtin_amdgcn_exp2f(sp[w * 3 * XD] - m);
      den = fmaf(wgt, sp[w * 3 * XD + XD], den);
      num = fmaf(wgt, sp[w * 3 * XD + 2 * XD], num);
    }
    a.agg[(int64_t)qe * a.ld_agg + tid] = num / den;
  }
}

// ---- packer: reference-layout matrices -> the kernel's fragment stream (three bf16 truncation pieces per weight)
__global__ void pack_attn_bf16x6_kernel(const float* __restrict__ w2, const float* __restrict__ wp, const float* __restrict__ p2,
                                        unsigned* __restrict__ out) {
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
    float v = 0.f;
    int p = 0;
    if (frag < XW2F) {
      const int t = frag / 3;
      p = frag % 3;
      const int ch = XHALF * half + 16 * t + c;
      if (stage < XS) v = w2[(int64_t)ch * XHID + 32 * stage + 16 * (j >> 2) + 4 * g + (j & 3)];
      else v = p2[ch * 32 + 8 * g + j];
    } else if (stage < XS) {
      const int u = (frag - XW2F) / 3;
      p = (frag - XW2F) % 3;
      v = wp[(32 * stage + 16 * u + c) * 32 + 8 * g + j];      // A operand: row = lane & 15 = hidden unit
    }
    res |= piece16(v, p) << (16 * q);
  }
  out[e] = res;
}

}  // namespace

// the stream holds BOTH layouts: the paired kernel's sub-stages first, then the single-workgroup kernel's stages
constexpr int64_t X6P_WORDS = (int64_t)2 * ZNSUB * ZSUB;
extern "C" int64_t occ4d_pt_cross_attn_bf16x6_stream_floats(void) { return X6P_WORDS + (int64_t)2 * XNSTAGE * XSTAGE; }

extern "C" int occ4d_pack_attn_bf16x6_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream,
                                                 void* stream) {
  OCC4D_REQUIRE(w2 && wp && p2 && wstream, "occ4d_pack_attn_bf16x6_stream_f32: null pointer");
  const int64_t total = (int64_t)2 * XNSTAGE * XSTAGE;
  pack_attn_bf16x6p_kernel<<<occ4d::cdiv(X6P_WORDS, 256), 256, 0, (hipStream_t)stream>>>(w2, wp, p2,
                                                                                         reinterpret_cast<unsigned*>(wstream));
  pack_attn_bf16x6_kernel<<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(
      w2, wp, p2, reinterpret_cast<unsigned*>(wstream) + X6P_WORDS);
  return occ4d::check_launch("occ4d_pack_attn_bf16x6_stream_f32");
}

extern "C" int occ4d_pt_cross_attn_bf16x6_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                              const float* apos, int64_t a_stride, const int32_t* idx, const float* kt,
                                              int64_t ld_kt, const float* vtc, int64_t ld_vt, const float* pos0_w,
                                              const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg, int n,
                                              int m, int k, int d, float divisor, void* stream) {
  const char* who = "occ4d_pt_cross_attn_bf16x6_f32";
  OCC4D_REQUIRE(d == XD, "%s: built for d = %d, got %d", who, XD, d);
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
  // OCC4D_X6_KERNEL: 2 (default) = one row tile per wave, all channels; 1 = paired 4-wave workgroups; 0 = the first cut
  // (two row tiles per wave, channel halves in separate workgroups).  Same results; A/B partners.
  static const int which = [] { const char* e = getenv("OCC4D_X6_KERNEL"); return e ? atoi(e) : 2; }();
  if (which == 2) {
    static const int cskew = [] { const char* e = getenv("OCC4D_X6_SKEW"); return e ? atoi(e) : 2; }();
    AttnX6Args c{aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b,
                 reinterpret_cast<const unsigned*>(wstream) + X6P_WORDS, agg, ld_agg, n, m, k, divisor, 0, 0, cskew, 0};
    c.groups = (int)occ4d::cdiv(n, CQPB);
    c.per = (int)occ4d::cdiv(c.groups, 8);
    cross_attn_bf16x6c_kernel<<<8 * c.per, 512, 0, (hipStream_t)stream>>>(c);
    return occ4d::check_launch(who);
  }
  if (which == 1) {
    static const int pskew = [] { const char* e = getenv("OCC4D_X6P_SKEW"); return e ? atoi(e) : 6; }();
    AttnX6pArgs b{aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b,
                  reinterpret_cast<const unsigned*>(wstream), agg, ld_agg, n, m, k, divisor, 0, 0, 2 * occ4d::cu_count(), pskew};
    b.groups = (int)occ4d::cdiv(n, ZQPB);
    b.per = (int)occ4d::cdiv(b.groups, 4);
    cross_attn_bf16x6p_kernel<<<8 * b.per, 256, 0, (hipStream_t)stream>>>(b);
    return occ4d::check_launch(who);
  }
  AttnX6Args a{aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b,
               reinterpret_cast<const unsigned*>(wstream) + X6P_WORDS, agg, ld_agg, n, m, k, divisor, 0, 0, 0, 0};
  static const int skew = [] { const char* e = getenv("OCC4D_X6_SKEW"); return e ? atoi(e) : 2; }();   // read once
  static const int stamps = [] { const char* e = getenv("OCC4D_X6_STAMPS"); return e ? atoi(e) : 0; }();
  a.skew = skew;
  a.stamps = stamps;
  a.groups = (int)occ4d::cdiv(n, XQPB);
  a.per = (int)occ4d::cdiv(a.groups, 4);
  cross_attn_bf16x6_kernel<<<8 * a.per, 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch(who);
}

// debug: the phase stamps of the last launch (OCC4D_X6_STAMPS=1): out[(workgroup * 2 + wave / 4) * 6 + i]
extern "C" int occ4d_debug_x6_stamps(unsigned long long* out, int n_words) {
  OCC4D_REQUIRE(out && n_words >= 0 && n_words <= 1024 * 2 * 6, "occ4d_debug_x6_stamps: bad arguments");
  const hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6_stamps), (size_t)n_words * 8);
  OCC4D_REQUIRE(e == hipSuccess, "occ4d_debug_x6_stamps: %s", hipGetErrorString(e));
  return OCC4D_OK;
}
