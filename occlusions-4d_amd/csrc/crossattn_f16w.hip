// Fused vector attention over K <= 14 neighbours, D = 416, in the fp16 two-piece split scheme (csrc/bf16x6.hpp: x = x1 + x2,
// three partial products, fp32 accumulate) on v_mfma_f32_32x32x16_f16 -- round 6, the re-lay VERDICT r5 item 1 names.
//
// Why another decomposition.  csrc/crossattn_bf16x6.hip (16 x 16 x 32 instructions) gives a wave two 16-row tiles x HALF of the
// channels; the two channel halves are separate workgroups, so GEMM1, the ReLU + split, the init gathers and the Aq read happen
// once per half, i.e. TWICE per pair row.  In the fp16 scheme, where GEMM2 has only three products left, those duplicated
// parts are a quarter of the kernel and its floor with everything but the matrix stream removed is 0.73 of the peak
// (profiles/r06_attn_split_ablations.txt).  Here a wave owns ONE tile of 32 pair rows x ALL 416 channels:
//   * 13 accumulator blocks of 32 x 32 (208 registers): one wave per SIMD, 4 waves per workgroup, 512-register budget;
//   * GEMM1 (transposed: Hpre^T[hidden 32][pair 32] = Wp r + (Aq - Kt)), ReLU, split and the init gathers once per pair row;
//   * a 1 KB weight fragment = 32 channels x 16 hidden units serves 32 rows: the LDS bytes per FLOP of the old kernel, half
//     its matrix-instruction count (each 32 x 32 x 16 instruction does the work of two 16 x 16 x 32);
//   * the Aq rows are read once (HBM-side traffic back to ~1.1 x algorithmic).
// Rows of a wave: 0-13 = the 14 neighbours of query 2 w, 14-27 = those of query 2 w + 1, 28-31 = slots 4 w .. 4 w + 3 of the
// workgroup's 9th query (its per-wave partial softmaxes meet in LDS): 9 queries per workgroup, 126 of 128 rows live.
// Register layouts (profiles/micro/mfma32_layout.hip checks them on the device): A: lane l -> row l % 32, k = 8 (l / 32) + e;
// B: lane l -> column l % 32, same k; C / D: lane l, register r -> column l % 32, row (r % 4) + 4 (l / 32) + 8 (r / 4).
// GEMM1's D registers of a lane are therefore 16 hidden units of ITS pair: registers 8 ks .. 8 ks + 7 are, after ReLU + split,
// GEMM2's A operand of k-step ks (the packed W2 fragments use the same hidden-unit order).
// Stage = 32 hidden units = 52 W2 fragments ([k-step 2][block 13][piece 2]) + 4 of the merged Wp ([k-step 2][piece 2]; those
// of the NEXT hidden stage: software pipeline), 56 KB, double buffered; every wave issues 14 of the next stage's fragments by DMA between its matrix instructions; one
// barrier per stage.  The last stage of the stream holds P2 for the epilogue's GEMM3.
// Scaling as in the 16 x 16 kernel's pre-scaled entry point: W2 and P2 packed * 2^8, Wp * 2^4, Aq / Kt arrive * 2^4.
//
// STATUS: parity-green (every f16x3 parity / regime test passes with OCC4D_F16W=1) and SLOWER than the 16 x 16 x 32 kernel:
// 1.06 ms per launch of the 32768-query decode chunk against 0.88-0.90 ms (profiles/r06_attn_f16w.txt), so it is an A/B
// switch, not the default.  The premise held on paper only: with ONE wave per SIMD (501 registers) nothing covers a wave's own
// waits, and the ablations put the loss there, not in the matrix stream -- no weight DMA 0.84 ms, no fragment reads 0.91,
// no GEMM1 / split 0.90, no init gathers 0.89, all four 0.65 (0.66 of the fp16 peak: no better than the 16 x 16 kernel's 0.62
// floor, whose two waves per SIMD hide each other's waits at 256 registers each).
#include <stdlib.h>

#include "bf16x6.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WD = 416;                   // channels
constexpr int WHID = 2 * WD;              // hidden units of attn_mlp
constexpr int WB = WD / 32;               // 13 channel blocks of 32
constexpr int WS = WHID / 32;             // 26 hidden stages of 32
constexpr int WFW = 256;                  // u32 words per fragment image (64 lanes x 16 B)
constexpr int WW2F = 2 * WB * 2;          // 52 W2 fragments of a stage: [k-step][block][piece]
constexpr int WSF = WW2F + 4;             // + Wp [k-step][piece] = 56
constexpr int WSTAGE = WSF * WFW;         // 14336 words = 57344 B
constexpr int WNSTAGE = WS + 1;           // + the P2 stage ([k-step][block][piece]: 52 fragments, 4 unused)
constexpr int WWAVES = 4;
constexpr int WQPB = 2 * WWAVES + 1;      // 9 queries per workgroup
constexpr int WPARTS = WSF / WWAVES;      // 14 fragments per wave and stage

struct AttnWArgs {
  const float* aq; int64_t ld_aq;         // (n, >= 832), * HSCALE
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;         // (m, >= 832), * HSCALE
  const float* vt; int64_t ld_vt;         // Wv f + c2
  const float* P1; const float* c1;
  const unsigned* wstream;                // [WNSTAGE][WSF][64 lanes][4 words]
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
  int groups;
};

__device__ __forceinline__ f32x16 mm32(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
struct Op2 { u32x4 hi, lo; };
__device__ __forceinline__ Op2 split8w(const float* v) {              // eight fp32 -> (high, low) fp16 piece registers
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2h(v[2 * i], v[2 * i + 1], h[i], l[i]);
  return Op2{u32x4{h[0], h[1], h[2], h[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}

__global__ __launch_bounds__(256, 1) void cross_attn_f16w_kernel(const AttnWArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned ring[2][WSTAGE];           // stage s lives in ring[s & 1]
  __shared__ __attribute__((aligned(16))) float s_p1[32 * 4];                 // (P1[m][0..2], c1[m])
  __shared__ int s_idx[WQPB * 16];
  __shared__ float s_part[WWAVES * 3 * WD];                                   // the 9th query: [wave][max, den, num][channel]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, g2 = lane >> 5;
  const int group = blockIdx.x;
  if (group >= a.groups) return;
  const int q0 = group * WQPB;
  const unsigned lane16 = lane * 16;
  constexpr float HS = SplitF16x3::HSCALE, INVW = SplitF16x3::INV_WSCALE;

  auto dma_part = [&](int stage_no, const unsigned* dst, int i) {               // this wave's i-th fragment of a stage
    const int f = wave + WWAVES * i;                                              // (wave-uniform; 4 x 14 = 56)
    dma_frag_x(a.wstream + (int64_t)stage_no * WSTAGE + f * WFW, lds_addr_x(dst) + (unsigned)f * (WFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < WPARTS; ++i) dma_part(0, ring[0], i);
  if (tid < WQPB * 16) {
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

  // ---- this lane's pair (tile row c32): which query, which neighbour slot
  auto row_query = [&](int row) { return row < 14 ? 2 * wave : row < 28 ? 2 * wave + 1 : 2 * WWAVES; };
  auto row_slot = [&](int row) { return row < 14 ? row : row < 28 ? row - 14 : 4 * wave + row - 28; };
  const int my_ql = row_query(c32), my_slot = row_slot(c32);
  const bool my_valid = my_slot < a.K && my_slot < 14;
  const int my_q = min(q0 + my_ql, a.N - 1);
  const int my_j = s_idx[my_ql * 16 + min(my_slot, 15)];
  Op2 rs[2];                                // r = relu(P1 d + c1): position-hidden units 16 ks + 8 g2 + e of this pair
  {
    const float* qp = a.qpos + (int64_t)my_q * a.qs;
    const float* ap = a.apos + (int64_t)my_j * a.as;
    const float dx = qp[0] - ap[0], dy = qp[1] - ap[1], dz = qp[2] - ap[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float rr[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(s_p1 + 4 * (16 * ks + 8 * g2 + e));
        const float v = fmaf(dz, w.z, fmaf(dy, w.y, dx * w.x)) + w.w;
        rr[e] = my_valid ? fmaxf(v, 0.f) : 0.f;
      }
      rs[ks] = split8w(rr);
    }
  }
  // GEMM1's init term: hidden units 8 b + 4 g2 + (0 .. 3) of the stage for this pair = four float4 per table
  const unsigned aq_off = (unsigned)(my_q * (int)a.ld_aq + 4 * g2) * 4u;
  const unsigned kt_off = (unsigned)(my_j * (int)a.ld_kt + 4 * g2) * 4u;
  auto slice = [](const float* base, unsigned off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  f32x4 ia[4], ik[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    ia[b] = slice(a.aq + 8 * b, aq_off);
    ik[b] = slice(a.kt + 8 * b, kt_off);
  }
  f32x16 acc[WB];
#pragma unroll
  for (int t = 0; t < WB; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // SOFTWARE PIPELINE.  One wave per SIMD: nothing else fills the matrix pipe while this wave waits, so GEMM1 + ReLU + split
  // of hidden stage s + 1 run BETWEEN the matrix instructions of stage s's second k-step (the Wp fragments of a stage
  // travel one stream slot early; the last slot carries stage 0's, read straight from the stream here), as two
  // independent 3-instruction chains (one per position k-step) that alternate with GEMM2's; the init slices of stage s + 1
  // are fetched at the top of stage s.  `hs` = GEMM2's A operand of the stage about to run.
  Op2 hs[2];
  auto finish_gemm1 = [&](const f32x16& h0, const f32x16& h1) {      // h0 + h1 -> ReLU -> split (HSCALE x the hidden activations)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(h0[8 * ks + e] + h1[8 * ks + e], 0.f);
      hs[ks] = split8w(v);
    }
  };
  auto init_term = [&]() {                                               // Aq - Kt of the slices in ia / ik (both * HSCALE)
    f32x16 h;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 d = ia[b] - ik[b];
      h[4 * b + 0] = d.x; h[4 * b + 1] = d.y; h[4 * b + 2] = d.z; h[4 * b + 3] = d.w;
    }
    return h;
  };
  {
    const unsigned* w0 = a.wstream + (int64_t)WS * WSTAGE + lane * 4;   // Wp of hidden stage 0: in the P2 stage's slot
    f32x16 h0 = init_term(), h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = 0.f;
    const u32x4 ah = *reinterpret_cast<const u32x4*>(w0 + (WW2F + 0) * WFW), al = *reinterpret_cast<const u32x4*>(w0 + (WW2F + 1) * WFW);
    const u32x4 bh = *reinterpret_cast<const u32x4*>(w0 + (WW2F + 2) * WFW), bl = *reinterpret_cast<const u32x4*>(w0 + (WW2F + 3) * WFW);
    h0 = mm32(al, rs[0].hi, h0); h1 = mm32(bl, rs[1].hi, h1);
    h0 = mm32(ah, rs[0].lo, h0); h1 = mm32(bh, rs[1].lo, h1);
    h0 = mm32(ah, rs[0].hi, h0); h1 = mm32(bh, rs[1].hi, h1);
    finish_gemm1(h0, h1);
  }
  dma_wait_x();
  __builtin_amdgcn_s_barrier();                       // stage 0 is complete

  // The init slices are fetched by inline asm and waited for by hand: the compiler does not see the DMA instructions (inline asm
  // too), so its own s_waitcnt for a visible load would be vmcnt(0) -- i.e. a wait for the whole next-stage weight stream in
  // the middle of the stage (measured: 0.2 ms of a 1.1 ms launch).  vmcnt counts in order: the slices are issued BEFORE the
  // stage's 14 DMA instructions, so "at most 14 outstanding" means they have arrived.
  auto aload = [](const float* base, unsigned off) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
  };
  auto stage = [&](const int s, const unsigned* __restrict__ cur, const unsigned* nxt) {
    const unsigned* f = cur + lane * 4;
    auto frag = [&](int i) { return *reinterpret_cast<const u32x4*>(f + i * WFW); };
    // the init slices of stage s + 1 (consumed from step 2 on; clamped at the end: results unused)
    const int sn = s + 1 < WS ? s + 1 : s;
#ifndef OCC4D_XW_ABL_NOINIT
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      ia[b] = aload(a.aq + 32 * sn + 8 * b, aq_off);
      ik[b] = aload(a.kt + 32 * sn + 8 * b, kt_off);
    }
#endif
    f32x16 h0, h1;
    // 6 steps = 2 k-steps x 3 block groups {0-3, 4-7, 8-12}: within a step the three partial products run OUTER and the
    // blocks inner, so a block's dependent matrix instructions are 4-5 instructions (128-160 cycles) apart.  The fragments of
    // step i + 1 are requested in step i.  The next stage's 14 DMA issues sit in steps 0 and 1; its GEMM1 (two independent
    // 3-instruction chains, one instruction of each per step) in steps 2-4, the ReLU + split in step 5: under this stage's
    // own matrix stream.  (Measured variants, profiles/r06_attn_f16w.txt: blocks in pairs with 14 steps 1.09-1.10 ms; this
    // one 1.06 ms; reads pinned in front of the previous step's matrix instructions and carried across the stage barrier,
    // 8 steps of 3-4 blocks, 1.12 ms.)
    constexpr int GB[4] = {0, 4, 8, 13};
    u32x4 fr[2][10];
    auto load_step = [&](int i, u32x4 (&dst)[10]) {
      const int ks = i / 3, g = i % 3;
#pragma unroll
      for (int b = GB[g]; b < GB[g + 1]; ++b) {
        dst[2 * (b - GB[g])] = frag((ks * WB + b) * 2);
        dst[2 * (b - GB[g]) + 1] = frag((ks * WB + b) * 2 + 1);
      }
    };
    load_step(0, fr[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ks = i / 3, g = i % 3, nb = GB[g + 1] - GB[g];
#ifndef OCC4D_XW_ABL_NOLDS                           // (timing-only ablations: profiles/time_attn_split.py f16w -D...)
      if (i + 1 < 6) load_step(i + 1, fr[(i + 1) & 1]);
#endif
#ifndef OCC4D_XW_ABL_NODMA
      if (i < 2) {
#pragma unroll
        for (int p = 0; p < 7; ++p) dma_part(s + 1, nxt, 7 * i + p);
      }
#endif
      const u32x4 (&w)[10] = fr[i & 1];
#ifndef OCC4D_XW_ABL_NOG1
      u32x4 ga, gb;                                    // the next stage's Wp fragments of this step's two GEMM1 instructions
      if (i >= 2 && i < 5) {
        ga = frag(WW2F + (i == 2 ? 1 : 0));            // chain h0 (position k-step 0): low, high, high
        gb = frag(WW2F + 2 + (i == 2 ? 1 : 0));        // chain h1 (position k-step 1)
        if (i == 2) {
#ifndef OCC4D_XW_ABL_NOINIT
          asm volatile("s_waitcnt vmcnt(14)"
                       : "+v"(ia[0]), "+v"(ia[1]), "+v"(ia[2]), "+v"(ia[3]), "+v"(ik[0]), "+v"(ik[1]), "+v"(ik[2]), "+v"(ik[3])
                       :: "memory");
#endif
          h0 = init_term();
#pragma unroll
          for (int r = 0; r < 16; ++r) h1[r] = 0.f;
        }
      }
#endif
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {                 // partial products: (low, high), (high, low), (high, high)
#ifndef OCC4D_XW_ABL_NOG1
        if (i >= 2 && i < 5 && pr == 1) {              // (between the products: its own accumulators, no dependence on GEMM2's)
          h0 = mm32(ga, i == 3 ? rs[0].lo : rs[0].hi, h0);
          h1 = mm32(gb, i == 3 ? rs[1].lo : rs[1].hi, h1);
        }
#endif
#pragma unroll
        for (int b = 0; b < nb; ++b)
          acc[GB[g] + b] = mm32(pr == 0 ? hs[ks].lo : hs[ks].hi, w[2 * b + (pr == 1 ? 1 : 0)], acc[GB[g] + b]);
      }
#ifndef OCC4D_XW_ABL_NOG1
      if (i == 5) finish_gemm1(h0, h1);                // (rewrites hs: every instruction that reads it has been issued)
#endif
      __builtin_amdgcn_sched_barrier(0);               // (a step's reads stay one step ahead: no hoisting of all 56 fragments)
    }
    dma_wait_x();
    __builtin_amdgcn_s_barrier();
  };
#pragma clang loop unroll(disable)
  for (int s = 0; s < WS; s += 2) {
    stage(s, ring[0], ring[1]);
    stage(s + 1, ring[1], ring[0]);
  }

  // ---- epilogue: ring[0] (26 & 1) holds the P2 stage.  Per block: GEMM3 (pe = P2 r; c2 is folded into vt), the per-channel
  // softmax over each query's 14 rows -- a lane holds 16 rows of its channel: 8 / 6 of query A, 6 / 8 of query B by lane half,
  // the upper half's last four = slots of the 9th query -- two v_permlane32_swap rounds, and the aggregation.
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor * INVW / HS;     // (the logits left the matrix pipe * WSCALE * HSCALE)
  const float NINF = -__builtin_inff();
  const unsigned* fp = ring[0] + lane * 4;
  int voff[16];
  bool act[16], isa[16], isb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 4 * g2 + 8 * (r >> 2);
    const int ql = row_query(row), slot = row_slot(row);
    voff[r] = s_idx[ql * 16 + min(slot, 15)] * (int)a.ld_vt + c32;
    act[r] = slot < a.K && slot < 14;
    isa[r] = row < 14;
    isb[r] = row >= 14 && row < 28;
  }
  const int q_out = q0 + 2 * wave + g2;               // lower lane half stores query A's channel, upper half query B's
  float* const orow = a.agg + (int64_t)min(q_out, a.N - 1) * a.ld_agg + c32;
  const bool o_ok = q_out < a.N;
  float* const sp = s_part + wave * 3 * WD + c32;
#pragma unroll
  for (int t = 0; t < WB; ++t) {
    float vq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vq[r] = a.vt[voff[r] + 32 * t];
    f32x16 pe;
#pragma unroll
    for (int r = 0; r < 16; ++r) pe[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 ph = *reinterpret_cast<const u32x4*>(fp + ((ks * WB + t) * 2) * WFW);
      const u32x4 pl = *reinterpret_cast<const u32x4*>(fp + ((ks * WB + t) * 2 + 1) * WFW);
      pe = mm32(rs[ks].lo, ph, pe);
      pe = mm32(rs[ks].hi, pl, pe);
      pe = mm32(rs[ks].hi, ph, pe);
    }
    float am[16], val[16];
    float ma = NINF, mb = NINF, mx = NINF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      am[r] = act[r] ? acc[t][r] : NINF;
      val[r] = fmaf(pe[r], INVW, vq[r]);
      ma = isa[r] ? fmaxf(ma, am[r]) : ma;
      mb = isb[r] ? fmaxf(mb, am[r]) : mb;
      mx = (!isa[r] && !isb[r]) ? fmaxf(mx, am[r]) : mx;
    }
    {                                                  // both queries' maxima in every lane
      const PairX p1 = swap32x(ma, mb);                // lo = (ma.lower, mb.lower), hi = (ma.upper, mb.upper)
      const float m1 = fmaxf(p1.lo, p1.hi);            // lower lanes: max A, upper lanes: max B
      const PairX p2 = swap32x(m1, m1);
      ma = p2.lo;                                      // (m1.lower everywhere)
      mb = p2.hi;
    }
    const float mas = ma * sc, mbs = mb * sc, mxs0 = mx * sc;
    const float mxs = mxs0 > NINF ? mxs0 : 0.f;        // (a wave without live slots of the 9th query)
    float da = 0.f, na = 0.f, db = 0.f, nb = 0.f, dx = 0.f, nx = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ref = isa[r] ? mas : isb[r] ? mbs : mxs;
      const float e = __builtin_amdgcn_exp2f(fmaf(am[r], sc, -ref));
      if (isa[r]) { da += e; na = fmaf(e, val[r], na); }
      else if (isb[r]) { db += e; nb = fmaf(e, val[r], nb); }
      else { dx += e; nx = fmaf(e, val[r], nx); }
    }
    {
      const PairX d1 = swap32x(da, db);
      const float den = d1.lo + d1.hi;                 // lower lanes: A, upper lanes: B
      const PairX n1 = swap32x(na, nb);
      const float num = n1.lo + n1.hi;
      if (o_ok) orow[32 * t] = num * __builtin_amdgcn_rcpf(den);
    }
    if (g2 == 1) {                                     // rows 28-31: this wave's slots of the 9th query
      sp[32 * t] = mxs0;
      sp[WD + 32 * t] = dx;
      sp[2 * WD + 32 * t] = nx;
    }
  }
  // ---- the 9th query: combine the four waves' partial softmaxes
  __syncthreads();
  const int qe = q0 + 2 * WWAVES;
  if (qe < a.N) {
    for (int ch = tid; ch < WD; ch += 64 * WWAVES) {
      float m = NINF;
#pragma unroll
      for (int w = 0; w < WWAVES; ++w) m = fmaxf(m, s_part[w * 3 * WD + ch]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int w = 0; w < WWAVES; ++w) {
        const float pm = s_part[w * 3 * WD + ch];
        const float wgt = pm > NINF ? __builtin_amdgcn_exp2f(pm - m) : 0.f;
        den = fmaf(wgt, s_part[w * 3 * WD + WD + ch], den);
        num = fmaf(wgt, s_part[w * 3 * WD + 2 * WD + ch], num);
      }
      a.agg[(int64_t)qe * a.ld_agg + ch] = num / den;
    }
  }
}

// ---- packer: reference-layout matrices -> the fragment stream (two round-to-nearest fp16 pieces per weight)
__global__ void pack_attn_f16w_kernel(const float* __restrict__ w2, const float* __restrict__ wp, const float* __restrict__ p2,
                                      unsigned* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)WNSTAGE * WSTAGE;
  if (e >= total) return;
  const int word = (int)(e & 3), lane = (int)((e >> 2) & 63);
  const int frag = (int)((e / WFW) % WSF), stage = (int)(e / WSTAGE);
  const int c32 = lane & 31, g2 = lane >> 5;
  unsigned res = 0u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = 2 * word + q;                        // element 0 .. 7 of the lane's operand register set
    float v = 0.f, scale = SplitF16x3::WSCALE;
    int p = 0;
    if (frag < WW2F) {
      const int ks = frag / (2 * WB), blk = (frag / 2) % WB;
      p = frag & 1;
      const int ch = 32 * blk + c32;                   // B operand: column = lane % 32
      if (stage < WS) v = w2[(int64_t)ch * WHID + 32 * stage + 8 * (2 * ks + (j >> 2)) + 4 * g2 + (j & 3)];
      else v = p2[ch * 32 + 16 * ks + 8 * g2 + j];
    } else {
      // the merged Wp of hidden stage s + 1 travels in slot s (software pipeline); the last slot carries stage 0's
      const int hst = stage + 1 < WNSTAGE ? stage + 1 : 0;
      const int ks = (frag - WW2F) >> 1;
      p = (frag - WW2F) & 1;
      if (hst < WS) v = wp[(32 * hst + c32) * 32 + 16 * ks + 8 * g2 + j];      // A operand: row = lane % 32 = hidden unit
      scale = SplitF16x3::HSCALE;
    }
    res |= SplitF16x3::piece_scaled(v, p, scale) << (16 * q);
  }
  out[e] = res;
}

}  // namespace

extern "C" int64_t occ4d_pt_cross_attn_f16w_stream_floats(void) { return (int64_t)WNSTAGE * WSTAGE; }

extern "C" int occ4d_pack_attn_f16w_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream, void* stream) {
  OCC4D_REQUIRE(w2 && wp && p2 && wstream, "occ4d_pack_attn_f16w_stream_f32: null pointer");
  const int64_t total = (int64_t)WNSTAGE * WSTAGE;
  pack_attn_f16w_kernel<<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w2, wp, p2, reinterpret_cast<unsigned*>(wstream));
  return occ4d::check_launch("occ4d_pack_attn_f16w_stream_f32");
}

extern "C" int occ4d_pt_cross_attn_f16w_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                            int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vtc,
                                            int64_t ld_vt, const float* pos0_w, const float* pos0_b, const float* wstream, float* agg,
                                            int64_t ld_agg, int n, int m, int k, int d, float divisor, void* stream) {
  const char* who = "occ4d_pt_cross_attn_f16w_f32";
  OCC4D_REQUIRE(d == WD, "%s: built for d = %d, got %d", who, WD, d);
  OCC4D_REQUIRE(k >= 1 && k <= 14 && m >= 1 && n >= 0, "%s: k = %d (1 .. 14), m = %d, n = %d", who, k, m, n);
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vtc && pos0_w && pos0_b && wstream && agg, "%s: null pointer", who);
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ((uintptr_t)aq % 16) == 0 && ((uintptr_t)kt % 16) == 0 &&
                    ((uintptr_t)wstream % 16) == 0 && ld_aq >= WHID && ld_kt >= WHID && ld_vt >= WD && ld_agg >= WD &&
                    q_stride >= 3 && a_stride >= 3,
                "%s: misaligned or short rows", who);
  OCC4D_REQUIRE((int64_t)n * ld_aq < ((int64_t)1 << 29) && (int64_t)m * ld_kt < ((int64_t)1 << 29) &&
                    (int64_t)m * ld_vt < ((int64_t)1 << 31),
                "%s: 32-bit row offsets: n * ld_aq and m * ld_kt must stay below 2^29 floats", who);
  if (n == 0) return OCC4D_OK;
  AttnWArgs a{aq, ld_aq, qpos, q_stride, apos, a_stride, idx, kt, ld_kt, vtc, ld_vt, pos0_w, pos0_b,
              reinterpret_cast<const unsigned*>(wstream), agg, ld_agg, n, m, k, divisor, 0};
  a.groups = (int)occ4d::cdiv(n, WQPB);
  cross_attn_f16w_kernel<<<a.groups, 64 * WWAVES, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch(who);
}
