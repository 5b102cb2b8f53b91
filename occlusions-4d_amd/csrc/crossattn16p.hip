// Fused vector attention over K <= 14 neighbours, D = 416 (E3 of SURVEY.md §8(a); K2-K4 of §2.1;
// model/point_transformer_layer.py:168-179) -- third generation: PAIRED workgroups.
//
// crossattn16.hip (second generation) runs one 8-wave workgroup per CU: both waves of a SIMD belong to the same
// workgroup, so they reach every barrier, the prologue and the VALU-bound softmax epilogue TOGETHER and the matrix pipe
// idles through all of them (measured: loop at 93.5 % of the pipe, epilogue 8.1 %, prologue 1.5 % of the kernel).
// Here a workgroup has FOUR waves (one per SIMD) and half the LDS footprint, so TWO workgroups share a CU; they are
// independent (own barriers, own weight stream) and started out of phase, so that one's epilogue / prologue / barrier
// waits sit under the other's MFMA stream -- the hardware's wave scheduler does the software pipelining.
//
// Work decomposition (wave64, 4 waves, 9 queries per workgroup in TWO passes of 64 pair rows):
//   pass ps, wave w = 16 pair rows: rows 0-13 = the 14 neighbours of query q0 + 4 ps + w; rows 14, 15 = neighbour slots
//                     8 ps + 2 w, + 1 of the workgroup's 9th query (8 + 6 slots over the two passes; 126 of 128 MFMA
//                     rows live, as before).  The 9th query's per-wave partial softmax stays in LDS between the passes
//                     (pass B merges into pass A's entry of the same wave: no barrier) and is combined at the end.
//   stage s       = 16 hidden units (52 of them): 26 fragments of W2 (416 x 16) + 2 of Wp (16 x 32), 1 KB each,
//                   28 KB, double buffered (56 KB per workgroup), one barrier per stage; then two stages of P2
//                   fragments (channel tiles 0-13 and 14-25) for the epilogue's GEMM3.
// MFMA chain per stage, as in the second generation: GEMM1 (transposed, K = 32, two alternating accumulators) gives
// Hpre^T whose C/D registers are GEMM2's A operand; 13 groups of 8 MFMAs advance the 26 channel tiles.
// Epilogue per P2 stage: GEMM3 (pe = P2 r + c2, accumulators initialised with c2), then the per-channel softmax over
// the 14 neighbours two channel tiles at a time: one v_permlane16/32_swap exchange serves both tiles (maxima), and
// den / num of both tiles share one exchange tree (two-operand swaps), so a tile pair costs 7 + 9 cross-lane
// instructions instead of 2 x 16.
// VALU diet: on gfx950 the fp32 MFMAs and the plain VALU share the SIMD's vector issue (profiles/micro/
// valu_beside_mfma.hip), every VALU instruction costs matrix time, so: (1) attn_mlp[2].bias is NOT added -- it is
// constant over the neighbour axis the softmax normalises over and cancels exactly (softmax(l + b) = softmax(l));
// (2) pos_mlp[2].bias comes folded into the value table (vt = Wv f + c2, built once per scene by the host), so
// GEMM3 starts from the inline constant 0; (3) the logit accumulators of a pass are zeroed by stage 0's MFMAs
// (srcC = 0), not by 104 moves; (4) DMA addresses are scalar base + one constant lane offset.
#include <type_traits>

#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PD = 416;                   // channels
constexpr int PTD = PD / 16;              // 26 channel tiles
constexpr int PHS = 2 * PD / 16;          // 52 hidden stages of 16 units
constexpr int PFRAG = 256;                // floats per fragment image (64 lanes x float4)
constexpr int PSF = PTD + 2;              // 28 fragments per stage: 26 W2 + 2 Wp
constexpr int PSTAGE = PSF * PFRAG;       // 7168 floats = 28672 B
constexpr int PTA = 14, PTB = 12;         // channel tiles of the two P2 stages
constexpr int PNSTAGE = PHS + 2;          // 54 stages in the packed stream
constexpr int PQPB = 9;                   // queries per workgroup
constexpr int PKMAX = 14;

struct Attn16pArgs {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;
  const float* P1; const float* c1;
  const float* wstream;                   // PNSTAGE stages (layout in include/occ4d.h)
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
  int first_round;                        // workgroups of the first dispatch round (2 per CU)
  int skew;                               // s_sleep(127) repeats of the later-placed workgroup of a CU in that round
  float* logits;                          // STORE >= 1: (N * K, 416) pre-softmax logits W2 relu(a) (no attn_mlp[2].bias)
  float* a_out;                           // STORE == 2: (N * K, 832) hidden pre-activations a (before the ReLU)
  float* pe_out;                          // STORE == 2: (N * K, 416) pe = P2 r + c2
  const float* c2;                        // STORE == 2: pos_mlp[2].bias (the kernel's own GEMM3 starts from 0)
};

__device__ __forceinline__ unsigned lds_addr_p(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// One fragment (1 KB), global (L2) -> LDS by DMA: scalar base (wave-uniform fragment address) + this lane's 16 bytes,
// LDS destination in M0.  No VALU instruction: on gfx950 the fp32 MFMAs and the plain VALU share the SIMD's vector
// issue (profiles/micro/valu_beside_mfma.hip: every VALU instruction beside a saturated v_mfma_f32_16x16x4_f32 stream
// costs its ~4 cycles in full), so address arithmetic in the stage loop is paid for in matrix throughput.  Inline asm
// on purpose (csrc/trunk.hip: with the builtin every fragment wait degrades to lgkmcnt(0)); ordering comes from
// dma_wait_p() + the stage barrier.
__device__ __forceinline__ void dma_frag_p(const float* __restrict__ src_frag, unsigned lds_dst, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(src_frag) : "memory");
}
// this wave's 7 fragments of a stage: fragments wave + 4 i
__device__ __forceinline__ void dma_stage_p(const float* __restrict__ src, const float* dst, int wave, unsigned lane16) {
#pragma unroll
  for (int i = 0; i < PSF / 4; ++i)
    dma_frag_p(src + (wave + 4 * i) * PFRAG, lds_addr_p(dst) + (unsigned)(wave + 4 * i) * (PFRAG * 4), lane16);
}
__device__ __forceinline__ void dma_wait_p() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 8 MFMAs on two accumulators, alternating
__device__ __forceinline__ void mm_ab_p(const f32x4 a, const f32x4 b0, const f32x4 b1, f32x4& c0, f32x4& c1) {
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c1, 0, 0, 0);
}
// the same with the operands swapped: the fragments are the A operand, so that the result is TRANSPOSED -- lane (g, c)
// holds channels 4 g .. 4 g + 3 of pair row c (one float4 per row: the training kernel's stores)
__device__ __forceinline__ void mm_ba_p(const f32x4 b, const f32x4 a0, const f32x4 a1, f32x4& c0, f32x4& c1) {
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, c1, 0, 0, 0);
}
// GEMM1 of a stage: K = 32 on ONE accumulator (a second accumulator would cost four VALU adds per stage; the dependent
// chain's 8 cycles per MFMA are the co-resident wave's)
__device__ __forceinline__ void mm_g1_p(const f32x4 a0, const f32x4 a1, const f32x4 b0, const f32x4 b1, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, c, 0, 0, 0);
}

// gfx950 lane-swap exchanges (16-lane rows r0..r3 of a wave):
//   swap16(x, y) -> lo = (x.r0, y.r0, x.r2, y.r2), hi = (x.r1, y.r1, x.r3, y.r3)
//   swap32(x, y) -> lo = (x.r0, x.r1, y.r0, y.r1), hi = (x.r2, x.r3, y.r2, y.r3)
struct Pair { float lo, hi; };
__device__ __forceinline__ Pair swap16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return Pair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ Pair swap32(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return Pair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}

// STORE (training forward): 1 = the logits also go to HBM, row q K + slot, so that backward does not run GEMM2 again
// (pair_hidden_kernel recomputes only the hidden pre-activations and pe); 2 = a and pe as well: the three pair tensors
// of pair_mlp_kernel, and backward recomputes nothing (6.4 GB per layer at BASELINE config 5).
template <bool K14, int STORE = 0>
__global__ __launch_bounds__(256, 2) void cross_attn16p_kernel(const Attn16pArgs a) {
  __shared__ __attribute__((aligned(16))) float buf0[PSTAGE];
  __shared__ __attribute__((aligned(16))) float buf1[PSTAGE];
  __shared__ float s_part[4 * 3 * PD];     // 9th query: per wave (max in the log2 domain, den, num) per channel
  __shared__ __attribute__((aligned(16))) float s_p1[32 * 4];   // (P1[m][0..2], c1[m])
  __shared__ int s_idx[PQPB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const bool g3 = g == 3;
  // XCD-aware group assignment (workgroup b runs on XCD b % 8): every XCD takes one contiguous range of query
  // groups, so its L2 holds the Kt / Vt rows near that slab of the grid.  Bijective for any grid size.
  const int nwg = gridDim.x, xcd = blockIdx.x & 7;
  const int per = nwg >> 3, rem = nwg & 7;
  const int group = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
  const int q0 = group * PQPB;

#ifdef OCC4D_CA16P_STAMP
  unsigned long long ts[8];
  ts[0] = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const unsigned lane16 = lane * 16;
  dma_stage_p(a.wstream, buf0, wave, lane16);
  if (tid < PQPB * 16) {
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
  // Phase skew.  The two workgroups of a CU are dispatched together in the first round and would stay in lock step
  // (same work, same rate: both in their epilogue at the same time, which is what this design exists to avoid).  The
  // one whose waves sit in the ODD wave slot of their SIMD (HW_ID.WAVE_ID: slot 0 = first placed, 1 = second) sleeps
  // a fraction of a pass once; later workgroups inherit the slot, and with it the phase, of the one they replace.
  // Performance only: a wrong guess about placement leaves results unchanged.
  if (a.skew > 0 && (int)blockIdx.x < a.first_round) {
    const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID[3:0] = WAVE_ID
    if (hw & 1)
      for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(127);
  }
  __syncthreads();

  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor;
  const float NINF = -__builtin_inff();
  f32x4 acc[PTD];

#pragma clang loop unroll(disable)
  for (int ps = 0; ps < 2; ++ps) {
    asm volatile("; OCC4D_MARK pass_prologue");
    // ---- this lane's pair in the operand layouts (pair row = lane & 15)
    const int my_ql = c < 14 ? 4 * ps + wave : 8;
    const int my_slot = c < 14 ? c : 8 * ps + 2 * wave + c - 14;
    const bool my_valid = my_slot < a.K;
    const int my_q = min(q0 + my_ql, a.N - 1);
    const int my_j = s_idx[my_ql * 16 + min(my_slot, 15)];
    // r = relu(P1 d + c1): this lane holds hidden units 4 s + g, s = 0..7 (MFMA step s consumes k = 4 s + g)
    float rr[8];
    {
      const float* qp = a.qpos + (int64_t)my_q * a.qs;
      const float* ap = a.apos + (int64_t)my_j * a.as;
      const float dx = qp[0] - ap[0], dy = qp[1] - ap[1], dz = qp[2] - ap[2];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(s_p1 + 4 * (4 * s + g));
        const float v = fmaf(dz, w.z, fmaf(dy, w.y, dx * w.x)) + w.w;
        rr[s] = my_valid ? fmaxf(v, 0.f) : 0.f;
      }
    }
    const f32x4 r_lo = {rr[0], rr[1], rr[2], rr[3]}, r_hi = {rr[4], rr[5], rr[6], rr[7]};
    // per-lane byte offsets of this pair's Aq / Kt rows (32-bit: the stage's slice is "uniform base + lane offset")
    const unsigned aq_off = (unsigned)(my_q * (int)a.ld_aq + 4 * g) * 4u;
    const unsigned kt_off = (unsigned)(my_j * (int)a.ld_kt + 4 * g) * 4u;
    auto slice = [](const float* base, unsigned off) {
      return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
    };
    // STORE == 2: this lane's four hidden units per stage of ITS pair row (operand layout: pair row = lane & 15)
    float* const my_a = (STORE == 2 && my_valid && q0 + my_ql < a.N)
                            ? a.a_out + ((int64_t)(q0 + my_ql) * a.K + my_slot) * (2 * PD) + 4 * g : nullptr;
    // GEMM1 accumulator init of stage 0 (later stages: fetched one stage ahead)
    f32x4 ia = slice(a.aq, aq_off);
    f32x4 ik = slice(a.kt, kt_off);
    // epilogue rows of this lane (C/D layout): 4 g + i; rows 0-13 = neighbours of query q0 + 4 ps + wave,
    // rows 14, 15 (g = 3, i = 2, 3) = slots 8 ps + 2 wave, + 1 of the 9th query
    int voff[4];
    bool act[4];
    float* lrow[4];                                      // STORE: this lane's element of the logits row of register i (or null)
    float* prow[4];                                      // STORE == 2: the same of the pe row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      const int ql = row < 14 ? 4 * ps + wave : 8;
      const int slot = row < 14 ? row : 8 * ps + 2 * wave + row - 14;
      voff[i] = s_idx[ql * 16 + min(slot, 15)] * (int)a.ld_vt + c;
      act[i] = slot < a.K;
      if (STORE) {
        const bool live = slot < a.K && q0 + ql < a.N;
        const int64_t e = ((int64_t)(q0 + ql) * a.K + slot) * PD + c;
        lrow[i] = live ? a.logits + e : nullptr;
        if (STORE == 2) prow[i] = live ? a.pe_out + e : nullptr;
      }
    }
    const int qm = q0 + 4 * ps + wave;
    const float own23 = g3 ? 0.f : 1.f;
    // output: lane groups 0 / 2 store the first / second channel tile of a pair for this wave's query
    float* const orow = a.agg + (int64_t)min(qm, a.N - 1) * a.ld_agg + 16 * (g >> 1) + c;
    const bool o_writer = (g & 1) == 0 && qm < a.N;
    float* const sp_lane = s_part + (wave * 3) * PD + c;
    const bool ninth_writer = g3 && (ps == 0 || wave < 3);
    dma_wait_p();
    __syncthreads();
#ifdef OCC4D_CA16P_STAMP
    ts[1 + 3 * ps] = __builtin_amdgcn_s_memtime();
#endif

    // ---- hidden-stage loop.  A stage is 14 groups of 8 MFMAs behind a fenced fragment pipeline: the two ds_read_b128
    // of group i + 1 are issued before the MFMAs of group i and nothing is scheduled across the fences.  Group 0:
    // GEMM1 (Wp fragments 26, 27 = K halves); groups 1 .. 13: GEMM2 on channel tiles 2 p, 2 p + 1 (W2 fragments).
    // Vector memory, one instruction per group (see crossattn16.hip): the next stage's Aq / Kt slices first (consumed
    // at the next stage's top, after the DMA has been waited for anyway), then this wave's 7 DMA fragments of the next
    // stage; the packed stream continues into the P2 stages, so "stage s + 1" is branch-free.
    auto stage = [&](auto firstc, const int s, const float* __restrict__ cur, const float* nxt) {
      constexpr bool FIRST = decltype(firstc)::value;
      f32x4 h = f32x4{ia.x - ik.x, ia.y - ik.y, ia.z - ik.z, ia.w - ik.w};
      const float* f = cur + lane * 4;
      f32x4 wa = *reinterpret_cast<const f32x4*>(f + 26 * PFRAG);
      f32x4 wb = *reinterpret_cast<const f32x4*>(f + 27 * PFRAG);
      __builtin_amdgcn_sched_barrier(0);
      const int sn = s + 1 < PHS ? s + 1 : s;          // (clamped: the last stage re-reads its own slices)
      const float* nsrc = a.wstream + (int64_t)(s + 1) * PSTAGE;
#pragma unroll
      for (int gq = 0; gq < 14; ++gq) {
        const f32x4 ca = wa, cb = wb;
        if (gq + 1 < 14) {
          wa = *reinterpret_cast<const f32x4*>(f + (2 * gq) * PFRAG);
          wb = *reinterpret_cast<const f32x4*>(f + (2 * gq + 1) * PFRAG);
        }
#ifndef OCC4D_CA16P_ABL_NOGATHER
        if (gq == 0) ia = slice(a.aq + 16 * sn, aq_off);
        if (gq == 1) ik = slice(a.kt + 16 * sn, kt_off);
#endif
#ifndef OCC4D_CA16P_ABL_NODMA
        if (gq >= 2 && gq <= 8)
          dma_frag_p(nsrc + (wave + 4 * (gq - 2)) * PFRAG, lds_addr_p(nxt) + (unsigned)(wave + 4 * (gq - 2)) * (PFRAG * 4), lane16);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (gq == 0) {
          mm_g1_p(ca, cb, r_lo, r_hi, h);
        } else {
          if (FIRST) acc[2 * (gq - 1)] = acc[2 * (gq - 1) + 1] = f32x4{0.f, 0.f, 0.f, 0.f};   // (folds into srcC = 0)
          mm_ab_p(h, ca, cb, acc[2 * (gq - 1)], acc[2 * (gq - 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gq == 0) {
          if (STORE == 2) {
            if (my_a) *reinterpret_cast<f32x4*>(my_a + 16 * s) = h;        // (before the ReLU)
            __builtin_amdgcn_sched_barrier(0);
          }
          h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        }
      }
    };
    asm volatile("; OCC4D_MARK loop");
    // two stages per loop trip so that the LDS buffers are compile-time objects; one barrier per stage
    stage(std::true_type{}, 0, buf0, buf1);
    dma_wait_p();
    __syncthreads();
    stage(std::false_type{}, 1, buf1, buf0);
    dma_wait_p();
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int s = 2; s < PHS; s += 2) {
      stage(std::false_type{}, s, buf0, buf1);
      dma_wait_p();
      __syncthreads();
      stage(std::false_type{}, s + 1, buf1, buf0);
      dma_wait_p();
      __syncthreads();
    }
#ifdef OCC4D_CA16P_STAMP
    ts[2 + 3 * ps] = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue, in chunks of NTH <= 4 channel tiles from T0 (P2 fragments from fragment 2 (T0 - TS) of the stage in
    // pbuf, TS = the stage's first tile): GEMM3 as a short fenced MFMA stage,
    // then the softmax / aggregation of those tiles two at a time: pure VALU + the gathered V rows (issued before
    // the chunk's MFMAs).  Small chunks keep pe + V at 32 registers; the MFMA latency a chunk waits for is the
    // co-resident workgroup's time on the matrix pipe.
    auto chunk = [&](auto T0c, auto NTc, auto TSc, const float* __restrict__ pbuf) {
      constexpr int T0 = decltype(T0c)::value, NTH = decltype(NTc)::value, TS = decltype(TSc)::value;
      f32x4 pe[NTH];
#pragma unroll
      for (int i = 0; i < NTH; ++i) pe[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float vq[NTH][4];
#pragma unroll
      for (int tl = 0; tl < NTH; ++tl) {
#ifndef OCC4D_CA16P_ABL_NOV
#pragma unroll
        for (int i = 0; i < 4; ++i) vq[tl][i] = a.vt[voff[i] + 16 * (T0 + tl)];
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) vq[tl][i] = (float)(voff[i] + tl);
#endif
      }
      {
        const float* fp = pbuf + lane * 4 + 2 * (T0 - TS) * PFRAG;
        f32x4 pa = *reinterpret_cast<const f32x4*>(fp);
        f32x4 pb = *reinterpret_cast<const f32x4*>(fp + 2 * PFRAG);
#pragma unroll
        for (int q = 0; q < NTH; ++q) {                 // group q: tile pair q >> 1, k half q & 1
          const int p = q >> 1, kh = q & 1;
          const f32x4 ca = pa, cb = pb;
          if (q + 1 < NTH) {
            const int pn = (q + 1) >> 1, kn = (q + 1) & 1;
            pa = *reinterpret_cast<const f32x4*>(fp + (4 * pn + kn) * PFRAG);
            pb = *reinterpret_cast<const f32x4*>(fp + (4 * pn + 2 + kn) * PFRAG);
          }
          __builtin_amdgcn_sched_barrier(0);
#ifndef OCC4D_CA16P_ABL_NOG3
          mm_ab_p(kh ? r_hi : r_lo, ca, cb, pe[2 * p], pe[2 * p + 1]);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int pr = 0; pr < NTH / 2; ++pr) {
        const int tA = T0 + 2 * pr;
        float am[2][4], val[2][4], m23[2], lm[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const f32x4 av = acc[tA + x];
          const f32x4 pv = pe[2 * pr + x];
          const float* vv = vq[2 * pr + x];
          if (STORE) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (lrow[i]) lrow[i][16 * (tA + x)] = av[i];
          }
          if (STORE == 2) {                              // pe = P2 r + c2 (GEMM3 started from 0: c2 sits in the value table)
            const float c2v = a.c2[16 * (tA + x) + c];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (prow[i]) prow[i][16 * (tA + x)] = pv[i] + c2v;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            am[x][i] = (K14 || act[i]) ? av[i] : NINF;     // (K = 14: every row a wave writes for is live)
            val[x][i] = pv[i] + vv[i];
          }
          m23[x] = fmaxf(am[x][2], am[x][3]);
          // rows 2, 3 of lane group 3 belong to the 9th query, not to this wave's own
          lm[x] = fmaxf(fmaxf(am[x][0], am[x][1]), g3 ? NINF : m23[x]);
        }
        // per-channel maximum over the 14 rows of both tiles with one exchange tree
        float mx[2];
        {
          const Pair p1 = swap16(lm[0], lm[1]);
          const float m1 = fmaxf(p1.lo, p1.hi);            // rows: (A01, B01, A23, B23)
          const Pair p2 = swap32(m1, m1);
          const float m2 = fmaxf(p2.lo, p2.hi);            // rows: (A, B, A, B)
          const Pair p3 = swap16(m2, m2);
          mx[0] = p3.lo;
          mx[1] = p3.hi;
        }
        float den[2], num[2], d23[2], n23[2], msr[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const float mxs = mx[x] * sc;                    // logits in the log2 domain: acc / sqrt(D) * log2(e)
          msr[x] = m23[x] * sc;                            // (9th query: the reference of this wave's two slots)
          const float live = K14 ? msr[x] : (msr[x] > NINF ? msr[x] : 0.f);
          const float ref23 = g3 ? live : mxs;
          const float e0 = __builtin_amdgcn_exp2f(fmaf(am[x][0], sc, -mxs));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(am[x][1], sc, -mxs));
          const float e2 = __builtin_amdgcn_exp2f(fmaf(am[x][2], sc, -ref23));
          const float e3 = __builtin_amdgcn_exp2f(fmaf(am[x][3], sc, -ref23));
          d23[x] = e2 + e3;
          n23[x] = fmaf(e3, val[x][3], e2 * val[x][2]);
          den[x] = fmaf(own23, d23[x], e0 + e1);           // (own23 = 0 in lane group 3: its rows 2, 3 are the 9th query's)
          num[x] = fmaf(own23, n23[x], fmaf(e1, val[x][1], e0 * val[x][0]));
        }
        // den / num of both tiles over the four lane groups with one exchange tree
        {
          const Pair a1 = swap16(den[0], num[0]);
          const float xa = a1.lo + a1.hi;                  // rows: (dA01, nA01, dA23, nA23)
          const Pair b1 = swap16(den[1], num[1]);
          const float xb = b1.lo + b1.hi;
          const Pair z1 = swap32(xa, xb);
          const float z = z1.lo + z1.hi;                   // rows: (dA, nA, dB, nB)
          const Pair z2 = swap16(z, z);                    // lo = (dA, dA, dB, dB), hi = (nA, nA, nB, nB)
          const float o = z2.hi * __builtin_amdgcn_rcpf(z2.lo);
          if (o_writer) orow[16 * tA] = o;
        }
        // 9th query: this wave's partial softmax over its two slots (registers 2, 3 of lane group 3)
        if (ninth_writer) {
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            float* sp = sp_lane + 16 * (tA + x);
            if (ps == 0) {
              sp[0] = msr[x];
              sp[PD] = d23[x];
              sp[2 * PD] = n23[x];
            } else {
              const float mA = sp[0], dA = sp[PD], nA = sp[2 * PD];
              const float m = fmaxf(mA, msr[x]);
              const float mm = K14 ? m : (m > NINF ? m : 0.f);
              const float wA = __builtin_amdgcn_exp2f(mA - mm), wB = __builtin_amdgcn_exp2f(msr[x] - mm);
              sp[0] = m;
              sp[PD] = fmaf(wA, dA, wB * d23[x]);
              sp[2 * PD] = fmaf(wA, nA, wB * n23[x]);
            }
          }
        }
      }
    };
    asm volatile("; OCC4D_MARK epilogue");
    // buf0 holds the first P2 stage (it followed the last hidden stage in the stream); the second lands in buf1 under it
    dma_stage_p(a.wstream + (int64_t)(PHS + 1) * PSTAGE, buf1, wave, lane16);
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using IA = std::integral_constant<int, PTA>;
    chunk(I0{}, I4{}, I0{}, buf0);
    chunk(I4{}, I4{}, I0{}, buf0);
    chunk(std::integral_constant<int, 8>{}, I4{}, I0{}, buf0);
    chunk(std::integral_constant<int, 12>{}, I2{}, I0{}, buf0);
    dma_wait_p();
    __syncthreads();
    // pass B's first hidden stage lands in buf0 under the second half of pass A's epilogue
    if (ps == 0) dma_stage_p(a.wstream, buf0, wave, lane16);
    chunk(IA{}, I4{}, IA{}, buf1);
    chunk(std::integral_constant<int, PTA + 4>{}, I4{}, IA{}, buf1);
    chunk(std::integral_constant<int, PTA + 8>{}, I4{}, IA{}, buf1);
#ifdef OCC4D_CA16P_STAMP
    ts[3 + 3 * ps] = __builtin_amdgcn_s_memtime();
#endif
  }
  asm volatile("; OCC4D_MARK tail");
  __syncthreads();
  // ---- 9th query: combine the four waves' partial softmaxes (maxima are in the log2 domain)
  const int q8 = q0 + 8;
  if (q8 < a.N) {
    for (int ch = tid; ch < PD; ch += 256) {
      float m = NINF;
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fmaxf(m, s_part[(w * 3) * PD + ch]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float wgt = __builtin_amdgcn_exp2f(s_part[(w * 3) * PD + ch] - m);
        den = fmaf(wgt, s_part[(w * 3 + 1) * PD + ch], den);
        num = fmaf(wgt, s_part[(w * 3 + 2) * PD + ch], num);
      }
      a.agg[(int64_t)q8 * a.ld_agg + ch] = num / den;
    }
  }
#ifdef OCC4D_CA16P_STAMP
  // debug build: absolute s_memtime stamps + HW_ID of every wave into the rows behind the N output rows
  if (lane == 0) {
    ts[7] = __builtin_amdgcn_s_memtime();
    unsigned long long* o = (unsigned long long*)(a.agg + (int64_t)(a.N + blockIdx.x) * a.ld_agg) + 10 * wave;
    for (int i = 0; i < 8; ++i) o[i] = ts[i];
    o[8] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    o[9] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) | ((__builtin_amdgcn_s_memrealtime() - rt0) << 8);   // XCC_ID, 100 MHz ticks
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Training: the pair tensors of the same layer in its merged form, for backward (the ops are
// model/point_transformer_layer.py:168-176 before the softmax).  One kernel instead of three generic GEMM launches:
//   a[p]      = aq[p / K] - kt[idx[p]] + Wp r[p]        (P, 832)   written stage by stage, BEFORE the ReLU
//   logits[p] = W2 relu(a[p])                           (P, 416)   (attn_mlp[2].bias is left out: constant over the
//                                                                   neighbour axis the softmax normalises over)
//   pe[p]     = P2 r[p] + c2                            (P, 416)
// Same weight stream, stage protocol and MFMA chain as the inference kernel above; a workgroup owns 128 consecutive
// pair rows (two passes of 64), every MFMA row is live, and there is no softmax: the epilogue is stores.  Rows past
// P are clamped to row P - 1 (they recompute and rewrite that row's values: no predicates in the loop).
struct PairMlpArgs {
  const float* aq; int64_t ld_aq;
  const float* kt; int64_t ld_kt;
  const float* r;                         // (P, 32) contiguous
  const int32_t* idx;                     // (P) = (N, K) flat
  const float* c2;
  const float* wstream;
  float* a_out;                           // (P, 832) contiguous
  float* logits;                          // (P, 416) contiguous
  float* pe;                              // (P, 416) contiguous
  int P, K;
  int first_round, skew;
};

__global__ __launch_bounds__(256, 2) void pair_mlp_kernel(const PairMlpArgs a) {
  __shared__ __attribute__((aligned(16))) float buf0[PSTAGE];
  __shared__ __attribute__((aligned(16))) float buf1[PSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const unsigned lane16 = lane * 16;
  dma_stage_p(a.wstream, buf0, wave, lane16);
  if (a.skew > 0 && (int)blockIdx.x < a.first_round) {      // phase skew of the two workgroups of a CU, as above
    const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    if (hw & 1)
      for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(127);
  }
  f32x4 acc[PTD];
  auto slice = [](const float* base, unsigned off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
  };
  auto put = [](float* base, unsigned off, const f32x4 v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + off) = v;
  };

#pragma clang loop unroll(disable)
  for (int ps = 0; ps < 2; ++ps) {
    asm volatile("; OCC4D_MARK pass_prologue");
    const int p = min((int)blockIdx.x * 128 + 64 * ps + 16 * wave + c, a.P - 1);
    const int q = p / a.K;
    const int j = a.idx[p];
    float rr[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) rr[s] = a.r[(int64_t)p * 32 + 4 * s + g];    // MFMA step s consumes k = 4 s + g
    const f32x4 r_lo = {rr[0], rr[1], rr[2], rr[3]}, r_hi = {rr[4], rr[5], rr[6], rr[7]};
    const unsigned aq_off = (unsigned)(q * (int)a.ld_aq + 4 * g) * 4u;
    const unsigned kt_off = (unsigned)(j * (int)a.ld_kt + 4 * g) * 4u;
    const unsigned a_off = ((unsigned)p * (2u * PD) + 4u * g) * 4u;
    const unsigned l_off = ((unsigned)p * (unsigned)PD + 4u * g) * 4u;
    f32x4 ia = slice(a.aq, aq_off);
    f32x4 ik = slice(a.kt, kt_off);
    dma_wait_p();
    __syncthreads();

    auto stage = [&](auto firstc, const int s, const float* __restrict__ cur, const float* nxt) {
      constexpr bool FIRST = decltype(firstc)::value;
      f32x4 h = f32x4{ia.x - ik.x, ia.y - ik.y, ia.z - ik.z, ia.w - ik.w};
      const float* f = cur + lane * 4;
      f32x4 wa = *reinterpret_cast<const f32x4*>(f + 26 * PFRAG);
      f32x4 wb = *reinterpret_cast<const f32x4*>(f + 27 * PFRAG);
      __builtin_amdgcn_sched_barrier(0);
      const int sn = s + 1 < PHS ? s + 1 : s;
      const float* nsrc = a.wstream + (int64_t)(s + 1) * PSTAGE;
#pragma unroll
      for (int gq = 0; gq < 14; ++gq) {
        const f32x4 ca = wa, cb = wb;
        if (gq + 1 < 14) {
          wa = *reinterpret_cast<const f32x4*>(f + (2 * gq) * PFRAG);
          wb = *reinterpret_cast<const f32x4*>(f + (2 * gq + 1) * PFRAG);
        }
        if (gq == 0) ia = slice(a.aq + 16 * sn, aq_off);
        if (gq == 1) ik = slice(a.kt + 16 * sn, kt_off);
        if (gq >= 2 && gq <= 8)
          dma_frag_p(nsrc + (wave + 4 * (gq - 2)) * PFRAG, lds_addr_p(nxt) + (unsigned)(wave + 4 * (gq - 2)) * (PFRAG * 4), lane16);
        __builtin_amdgcn_sched_barrier(0);
        if (gq == 0) {
          mm_g1_p(ca, cb, r_lo, r_hi, h);
        } else {
          if (FIRST) acc[2 * (gq - 1)] = acc[2 * (gq - 1) + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          mm_ba_p(h, ca, cb, acc[2 * (gq - 1)], acc[2 * (gq - 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gq == 0) {
#ifndef OCC4D_PM_ABL_NOA
          put(a.a_out + 16 * s, a_off, h);                 // units 16 s + 4 g .. + 3 of this lane's pair row
#endif
          __builtin_amdgcn_sched_barrier(0);
          h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
        }
      }
    };
    asm volatile("; OCC4D_MARK loop");
    stage(std::true_type{}, 0, buf0, buf1);
    dma_wait_p();
    __syncthreads();
    stage(std::false_type{}, 1, buf1, buf0);
    dma_wait_p();
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int s = 2; s < PHS; s += 2) {
      stage(std::false_type{}, s, buf0, buf1);
      dma_wait_p();
      __syncthreads();
      stage(std::false_type{}, s + 1, buf1, buf0);
      dma_wait_p();
      __syncthreads();
    }
    asm volatile("; OCC4D_MARK epilogue");
    // ---- epilogue: the logits, then pe = P2 r + c2 on the two P2 stages (buf0 holds the first, the second lands in
    // buf1 under it).  GEMM2 / GEMM3 ran with the fragments as the A operand (mm_ba_p): lane (g, c) holds channels
    // 16 t + 4 g .. + 3 of pair row c in every accumulator tile
    dma_stage_p(a.wstream + (int64_t)(PHS + 1) * PSTAGE, buf1, wave, lane16);
#pragma unroll
    for (int t = 0; t < PTD; ++t) {
#ifdef OCC4D_PM_ABL_NOLOGITS
      if (acc[t].x == 123.456f)                      // (ablation: keep the accumulators live, store nothing)
#endif
      put(a.logits + 16 * t, l_off, acc[t]);
    }
    auto pe_tiles = [&](auto T0c, auto NPc, const float* __restrict__ pbuf) {
      constexpr int T0 = decltype(T0c)::value, NP = decltype(NPc)::value;
      const float* fp = pbuf + lane * 4;
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const int t = T0 + 2 * pr;
        f32x4 e0 = *reinterpret_cast<const f32x4*>(a.c2 + 16 * t + 4 * g);
        f32x4 e1 = *reinterpret_cast<const f32x4*>(a.c2 + 16 * (t + 1) + 4 * g);
        mm_ba_p(r_lo, *reinterpret_cast<const f32x4*>(fp + (4 * pr) * PFRAG),
                *reinterpret_cast<const f32x4*>(fp + (4 * pr + 2) * PFRAG), e0, e1);
        mm_ba_p(r_hi, *reinterpret_cast<const f32x4*>(fp + (4 * pr + 1) * PFRAG),
                *reinterpret_cast<const f32x4*>(fp + (4 * pr + 3) * PFRAG), e0, e1);
#ifdef OCC4D_PM_ABL_NOPE
        if (e0.x == 123.456f)
#endif
        {
          put(a.pe + 16 * t, l_off, e0);
          put(a.pe + 16 * (t + 1), l_off, e1);
        }
      }
    };
    pe_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, PTA / 2>{}, buf0);
    dma_wait_p();
    __syncthreads();
    if (ps == 0) dma_stage_p(a.wstream, buf0, wave, lane16);   // pass B's first hidden stage
    pe_tiles(std::integral_constant<int, PTA>{}, std::integral_constant<int, PTB / 2>{}, buf1);
  }
}

// The same pair tensors WITHOUT the logits (the training forward kept them: cross_attn16p_kernel<., true>): a and pe only,
// 4992 B of stores per pair row against 42 KFLOP -- memory-bound, so nothing is staged and nothing is shared: a wave reads
// its Wp / P2 fragments (1 KB, coalesced, L2-resident: the packed stream of the kernels above) and its Aq / Kt slices
// straight into registers two hidden stages ahead, and never meets a barrier.  Same MFMA chain and operand order per
// stage as pair_mlp_kernel: a and pe are bit-identical to its.
__global__ __launch_bounds__(256) void pair_hidden_kernel(const PairMlpArgs a) {
  // a workgroup = 16 consecutive pair rows (53 KB of a, 26 KB of pe, contiguous); its four waves take a quarter of the hidden
  // stages / of the channel tile pairs each.  Measured (profiles/time_pair_hidden.py, 22976 x 14 pairs, 1.61 GB stored):
  // 0.78-0.89 ms against 2.0 ms for the full kernel on the same (random-neighbour) inputs and 0.23 ms for a fill of
  // the same bytes; one wave per 16 rows over all stages measured the same -- per 1 KB stored a wave reads 4 KB (two
  // fragments, the Aq and the Kt slice), so the L2 -> CU side, not the stores, is the bound.
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int p = min((int)blockIdx.x * 16 + c, a.P - 1);
  const int q = p / a.K;
  const int j = a.idx[p];
  float rr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) rr[s] = a.r[(int64_t)p * 32 + 4 * s + g];
  const f32x4 r_lo = {rr[0], rr[1], rr[2], rr[3]}, r_hi = {rr[4], rr[5], rr[6], rr[7]};
  const float* aq = a.aq + (int64_t)q * a.ld_aq + 4 * g;
  const float* kt = a.kt + (int64_t)j * a.ld_kt + 4 * g;
  float* ao = a.a_out + (int64_t)p * (2 * PD) + 4 * g;
  float* po = a.pe + (int64_t)p * PD + 4 * g;
  const float* ws = a.wstream + lane * 4;
  auto ld = [](const float* q4) { return *reinterpret_cast<const f32x4*>(q4); };
  constexpr int AHEAD = 2, PER = PHS / 4;                // stages in flight ahead of the one being computed; 13 per wave
  const int s_begin = PER * wave, s_end = s_begin + PER;
  f32x4 wa[AHEAD + 1], wb[AHEAD + 1], ia[AHEAD + 1], ik[AHEAD + 1];
#pragma unroll
  for (int u = 0; u < AHEAD; ++u) {
    const int s = s_begin + u;
    wa[u] = ld(ws + (int64_t)s * PSTAGE + 26 * PFRAG); wb[u] = ld(ws + (int64_t)s * PSTAGE + 27 * PFRAG);
    ia[u] = ld(aq + 16 * s); ik[u] = ld(kt + 16 * s);
  }
#pragma unroll 1
  for (int s0 = s_begin; s0 < s_end; s0 += AHEAD + 1) {  // (the slot index is a compile-time constant)
#pragma unroll
    for (int u = 0; u < AHEAD + 1; ++u) {
      const int s = s0 + u;
      if (s < s_end) {
        const int sn = min(s + AHEAD, s_end - 1);
        const int slot_n = (u + AHEAD) % (AHEAD + 1);
        wa[slot_n] = ld(ws + (int64_t)sn * PSTAGE + 26 * PFRAG); wb[slot_n] = ld(ws + (int64_t)sn * PSTAGE + 27 * PFRAG);
        ia[slot_n] = ld(aq + 16 * sn); ik[slot_n] = ld(kt + 16 * sn);
        f32x4 h = ia[u] - ik[u];
        mm_g1_p(wa[u], wb[u], r_lo, r_hi, h);
        *reinterpret_cast<f32x4*>(ao + 16 * s) = h;      // units 16 s + 4 g .. + 3 of this lane's pair row, BEFORE the ReLU
      }
    }
  }
  // pe = P2 r + c2: the two P2 stages of the stream (channel tiles 0-13, 14-25), a tile pair at a time
#pragma unroll 1
  for (int pr = wave; pr < PTD / 2; pr += 4) {
    const int t = 2 * pr;
    const bool second = t >= PTA;
    const float* fp = ws + (int64_t)(PHS + (second ? 1 : 0)) * PSTAGE + (int64_t)(4 * (second ? pr - PTA / 2 : pr)) * PFRAG;
    f32x4 e0 = ld(a.c2 + 16 * t + 4 * g), e1 = ld(a.c2 + 16 * (t + 1) + 4 * g);
    mm_ba_p(r_lo, ld(fp), ld(fp + 2 * PFRAG), e0, e1);
    mm_ba_p(r_hi, ld(fp + PFRAG), ld(fp + 3 * PFRAG), e0, e1);
    *reinterpret_cast<f32x4*>(po + 16 * t) = e0;
    *reinterpret_cast<f32x4*>(po + 16 * (t + 1)) = e1;
  }
}

using occ4d::cu_count;

}  // namespace

extern "C" int64_t occ4d_pt_cross_attn16p_stream_floats(void) { return (int64_t)PNSTAGE * PSTAGE; }

static int attn16p_launch(const char* who, const float* aq, int64_t ld_aq, const float* qpos, int64_t qs, const float* apos,
                          int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt, int64_t ld_vt,
                          const float* P1, const float* c1, const float* wstream, float* agg, int64_t ld_agg, int n, int m,
                          int k, int d, float divisor, int skew, float* logits, float* a_out, float* pe_out, const float* c2,
                          void* stream) {
  OCC4D_REQUIRE(d == PD, "%s: built for d = %d, got %d", who, PD, d);
  OCC4D_REQUIRE(k >= 1 && k <= PKMAX, "%s: k=%d outside [1,%d]", who, k, PKMAX);
  OCC4D_REQUIRE(m >= 1 && n >= 0, "%s: bad n/m", who);
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vt && P1 && c1 && wstream && agg, "%s: null pointer", who);
  OCC4D_REQUIRE(ld_aq >= 2 * d && ld_kt >= 2 * d && ld_vt >= d && ld_agg >= d && qs >= 3 && as >= 3,
                "%s: leading dimension too small", who);
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ((uintptr_t)aq % 16) == 0 && ((uintptr_t)kt % 16) == 0 &&
                    ((uintptr_t)wstream % 16) == 0,
                "%s: aq / kt / wstream must be 16-byte aligned with ld %% 4 == 0", who);
  OCC4D_REQUIRE((int64_t)m * ld_vt < (int64_t)1 << 31, "%s: vt too large for 32-bit row offsets", who);
  OCC4D_REQUIRE(divisor > 0.f, "%s: divisor must be > 0", who);
  OCC4D_REQUIRE(skew >= 0 && skew <= 64, "%s: skew=%d outside [0,64]", who, skew);
  Attn16pArgs a{aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wstream, agg, ld_agg, n, m, k, divisor,
                2 * cu_count(), skew, logits, a_out, pe_out, c2};
  const int grid = occ4d::cdiv(n, PQPB);
  hipStream_t st = (hipStream_t)stream;
  if (logits && a_out) {
    if (k == PKMAX) cross_attn16p_kernel<true, 2><<<grid, 256, 0, st>>>(a);
    else cross_attn16p_kernel<false, 2><<<grid, 256, 0, st>>>(a);
  } else if (logits) {
    if (k == PKMAX) cross_attn16p_kernel<true, 1><<<grid, 256, 0, st>>>(a);
    else cross_attn16p_kernel<false, 1><<<grid, 256, 0, st>>>(a);
  } else {
    if (k == PKMAX) cross_attn16p_kernel<true><<<grid, 256, 0, st>>>(a);
    else cross_attn16p_kernel<false><<<grid, 256, 0, st>>>(a);
  }
  return occ4d::check_launch(who);
}

extern "C" int occ4d_pt_cross_attn16p_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs, const float* apos,
                                          int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                                          int64_t ld_vt, const float* P1, const float* c1, const float* wstream,
                                          float* agg, int64_t ld_agg, int n, int m, int k, int d, float divisor,
                                          int skew, void* stream) {
  return attn16p_launch("occ4d_pt_cross_attn16p", aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wstream,
                        agg, ld_agg, n, m, k, d, divisor, skew, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int occ4d_pt_cross_attn16p_logits_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs,
                                                 const float* apos, int64_t as, const int32_t* idx, const float* kt,
                                                 int64_t ld_kt, const float* vt, int64_t ld_vt, const float* P1,
                                                 const float* c1, const float* wstream, float* agg, int64_t ld_agg,
                                                 float* logits, float* a_out, float* pe_out, const float* c2, int n, int m,
                                                 int k, int d, float divisor, int skew, void* stream) {
  const char* who = "occ4d_pt_cross_attn16p_logits";
  OCC4D_REQUIRE(logits && ((uintptr_t)logits % 16) == 0, "%s: null or misaligned logits buffer", who);
  OCC4D_REQUIRE((a_out != nullptr) == (pe_out != nullptr) && (a_out != nullptr) == (c2 != nullptr),
                "%s: a_out, pe_out and c2 come together (all three pair tensors) or not at all", who);
  OCC4D_REQUIRE(!a_out || (((uintptr_t)a_out | (uintptr_t)pe_out) % 16) == 0, "%s: misaligned a_out / pe_out", who);
  return attn16p_launch(who, aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wstream, agg, ld_agg, n, m, k, d,
                        divisor, skew, logits, a_out, pe_out, c2, stream);
}

extern "C" int occ4d_pt_pair_mlp_f32(const float* aq, int64_t ld_aq, const float* kt, int64_t ld_kt, const float* r,
                                     const int32_t* idx, const float* c2, const float* wstream, float* a_out,
                                     float* logits, float* pe, int n, int m, int k, int d, int skew, void* stream) {
  OCC4D_REQUIRE(d == PD, "occ4d_pt_pair_mlp: built for d = %d, got %d", PD, d);
  OCC4D_REQUIRE(k >= 1 && m >= 1 && n >= 0, "occ4d_pt_pair_mlp: bad n/m/k");
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(aq && kt && r && idx && c2 && wstream && a_out && pe, "occ4d_pt_pair_mlp: null pointer");
  OCC4D_REQUIRE(ld_aq >= 2 * d && ld_kt >= 2 * d && ld_aq % 4 == 0 && ld_kt % 4 == 0,
                "occ4d_pt_pair_mlp: aq / kt leading dimensions must be >= 832 and multiples of 4");
  auto al16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  OCC4D_REQUIRE(al16(aq) && al16(kt) && al16(c2) && al16(wstream) && al16(a_out) && (!logits || al16(logits)) && al16(pe),
                "occ4d_pt_pair_mlp: aq, kt, c2, wstream and the outputs must be 16-byte aligned");
  const int64_t pairs = (int64_t)n * k;
  OCC4D_REQUIRE(pairs * 2 * d * 4 < ((int64_t)1 << 32) && (int64_t)n * ld_aq * 4 < ((int64_t)1 << 32) &&
                    (int64_t)m * ld_kt * 4 < ((int64_t)1 << 32),
                "occ4d_pt_pair_mlp: 32-bit row offsets: n * k * 3328 B, n * ld_aq * 4 B and m * ld_kt * 4 B must stay "
                "below 4 GiB (split the queries into chunks)");
  OCC4D_REQUIRE(skew >= 0 && skew <= 64, "occ4d_pt_pair_mlp: skew=%d outside [0,64]", skew);
  PairMlpArgs a{aq, ld_aq, kt, ld_kt, r, idx, c2, wstream, a_out, logits, pe, (int)pairs, k, 2 * cu_count(), skew};
  // logits == nullptr: the caller holds them already (occ4d_pt_cross_attn16p_logits_f32): a and pe only, no GEMM2
  if (logits) pair_mlp_kernel<<<occ4d::cdiv(pairs, 128), 256, 0, (hipStream_t)stream>>>(a);
  else pair_hidden_kernel<<<occ4d::cdiv(pairs, 16), 256, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_pt_pair_mlp");
}
