// Fused vector attention over K <= 14 neighbours, D = 416 (E3 of SURVEY.md §8(a); K2-K4 of §2.1;
// model/point_transformer_layer.py:168-179) -- second generation, on v_mfma_f32_16x16x4_f32.
//
// Why a second kernel (crossattn.hip stays for D = 288 and the opt-in split-bf16 mode): with 32 x 32 MFMA tiles
// the 416-channel accumulators of a 32-pair row tile need 208 registers, so the first kernel splits the channels
// over two waves and both run GEMM1 for the same rows (+7 % MFMA work), stages the weights through 60 registers
// (10 of them spilled) and reads its gathers per 32-row tile.  With 16 x 16 tiles a wave owns 16 pair rows and ALL
// 416 channels in 104 accumulator registers: no duplicated GEMM1, no cross-wave dependence inside a hidden block,
// no spills, two waves per SIMD; the weight stream is DMA'd (global_load_lds) from a stage-packed copy.
//
// Work decomposition (wave64, 8 waves, 9 queries per workgroup):
//   wave w   = 16 pair rows: rows 0-13 = the 14 neighbours of query q0 + w, rows 14-15 = neighbour slots 2w, 2w + 1
//              of the workgroup's 9th query (7 waves x 2 = 14 slots; wave 7's two spare rows idle): 126 of 128 MFMA
//              rows carry live pairs.
//   stage hb = one 32-wide hidden block (26 of them): 52 fragments of W2 (416 x 32) + 4 of Wp (32 x 32), 1 KB each
//              (64 lanes x 16 B, lane-linear: conflict-free ds_read_b128), double buffered, one barrier per stage.
// Chained MFMAs, nothing between the two GEMMs of attn_mlp leaves the registers:
//   GEMM1 (transposed)  Hpre^T[hid][pair] = Wp[hid][:] . r[pair][:]  (K = 32), accumulator initialised with
//          Aq[query][hid] - Kt[neighbour][hid]; its C/D registers (lane = pair, 4 consecutive hidden units) ARE the
//   GEMM2  A operand:  logits[pair][ch] += relu(Hpre)[pair][hid] W2[ch][hid]   (C/D: lane = channel, 4 pair rows)
//   GEMM3  pe[pair][ch] = r[pair][:] . P2[ch][:]  (a 27th stage holds P2's fragments), same C/D layout.
// Per-channel softmax over a query's 14 neighbours: 4 rows in a lane's registers, the rest in lanes ^ 16, ^ 32, ^ 48:
// two exchanges per reduction.  The 9th query's seven 2-row partial softmaxes are merged through LDS at the end.
#include <type_traits>

#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AD = 416;                   // channels
constexpr int ATD = AD / 16;              // 26 channel tiles
constexpr int AHB = 2 * AD / 32;          // 26 hidden blocks of 32
constexpr int AFRAG = 256;                // floats per fragment image (64 lanes x float4)
constexpr int ASTAGE_FRAGS = 2 * ATD + 4; // 52 W2 + 4 Wp fragments
constexpr int ASTAGE = ASTAGE_FRAGS * AFRAG;   // 14336 floats = 57344 B
constexpr int AQPB = 9;                   // queries per workgroup
constexpr int AKMAX = 14;

struct Attn16Args {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;
  const float* P1; const float* c1;
  const float* wstream;                   // (AHB + 1) stages: [W2 | Wp] per hidden block, then [P2 | b2 | c2]
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
};

__device__ __forceinline__ unsigned lds_addr16(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// one stage = 56 fragments, 7 per wave, global (L2) -> LDS by DMA.  Inline asm on purpose (see csrc/trunk.hip: with
// the builtin every fragment wait degrades to lgkmcnt(0)); ordering comes from dma_wait16() + the stage barrier.
__device__ __forceinline__ void dma_stage16(const float* __restrict__ src, const float* dst, int wave, int lane) {
  const unsigned dst0 = lds_addr16(dst);
#pragma unroll
  for (int i = 0; i < ASTAGE_FRAGS / 8; ++i) {
    const int c = wave + 8 * i;
    const float* g = src + c * AFRAG + lane * 4;
    const unsigned d = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)c * (AFRAG * 4));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(d) : "memory");
  }
}
// the i-th of this wave's 7 fragments of a stage (fragment wave + 8 i)
__device__ __forceinline__ void dma_frag16(const float* __restrict__ src, const float* dst, int wave, int lane, int i) {
  const int c = wave + 8 * i;
  const float* g = src + c * AFRAG + lane * 4;
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_addr16(dst) + (unsigned)c * (AFRAG * 4));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(d) : "memory");
}
__device__ __forceinline__ void dma_wait16() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 8 MFMAs on two accumulators, alternating (A operands a0 / a1 share the B fragment element-wise or vice versa)
__device__ __forceinline__ void mm_ab(const f32x4 a, const f32x4 b0, const f32x4 b1, f32x4& c0, f32x4& c1) {
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c1, 0, 0, 0);
}
__device__ __forceinline__ void mm_ba(const f32x4 a0, const f32x4 a1, const f32x4 b, f32x4& c0, f32x4& c1) {
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, c1, 0, 0, 0);
}

// Cross-row exchanges on the gfx950 VALU lane-swap instructions (no LDS round trip): with both operands equal to v,
// v_permlane16_swap leaves (row0, row0, row2, row2) and (row1, row1, row3, row3) of the 16-lane rows,
// v_permlane32_swap (low half, low half) and (high half, high half): combining the two results reduces over
// lane ^ 16 and lane ^ 32 respectively.
__device__ __forceinline__ float max_x16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_x32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_x16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_x32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(512, 2) void cross_attn16_kernel(const Attn16Args a) {
  __shared__ __attribute__((aligned(16))) float buf0[ASTAGE];
  __shared__ __attribute__((aligned(16))) float buf1[ASTAGE];
  __shared__ int s_idx[AQPB * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // XCD-aware group assignment (workgroup b runs on XCD b % 8): every XCD takes one contiguous range of query
  // groups, so its L2 holds the Kt / Vt rows near that slab of the grid.  Bijective for any grid size.
  const int nwg = gridDim.x, xcd = blockIdx.x & 7;
  const int per = nwg >> 3, rem = nwg & 7;
  const int group = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
  const int q0 = group * AQPB;

#ifdef OCC4D_CA16_STAMP
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  dma_stage16(a.wstream, buf0, wave, lane);
  if (tid < AQPB * 16) {
    const int q = min(q0 + (tid >> 4), a.N - 1);
    const int s = min(tid & 15, a.K - 1);
    s_idx[tid] = a.idx[(int64_t)q * a.K + s];
  }
  // ---- this lane's pair in the operand layouts (pair row = lane & 15)
  const int my_ql = c < 14 ? wave : 8;
  const int my_slot = c < 14 ? c : 2 * wave + c - 14;
  const bool my_valid = my_slot < a.K && (c < 14 || wave < 7);
  const int my_q = min(q0 + my_ql, a.N - 1);
  const int my_j = a.idx[(int64_t)my_q * a.K + min(my_slot, a.K - 1)];
  // r = relu(P1 d + c1): this lane holds hidden units 4 s + g, s = 0..7 (MFMA step s consumes k = 4 s + g)
  float rr[8];
  {
    const float* qp = a.qpos + (int64_t)my_q * a.qs;
    const float* ap = a.apos + (int64_t)my_j * a.as;
    const float dx = qp[0] - ap[0], dy = qp[1] - ap[1], dz = qp[2] - ap[2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int m = 4 * s + g;
      const float* w = a.P1 + 3 * m;
      const float v = fmaf(dz, w[2], fmaf(dy, w[1], dx * w[0])) + a.c1[m];
      rr[s] = my_valid ? fmaxf(v, 0.f) : 0.f;
    }
  }
  const f32x4 r_lo = {rr[0], rr[1], rr[2], rr[3]}, r_hi = {rr[4], rr[5], rr[6], rr[7]};
  const float* aq_row = a.aq + (int64_t)my_q * a.ld_aq + 4 * g;
  const float* kt_row = a.kt + (int64_t)my_j * a.ld_kt + 4 * g;

  f32x4 acc[ATD];
#pragma unroll
  for (int t = 0; t < ATD; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // GEMM1 accumulator init of hidden block 0 (later blocks: fetched one block ahead)
  f32x4 ia0 = *reinterpret_cast<const f32x4*>(aq_row), ia1 = *reinterpret_cast<const f32x4*>(aq_row + 16);
  f32x4 ik0 = *reinterpret_cast<const f32x4*>(kt_row), ik1 = *reinterpret_cast<const f32x4*>(kt_row + 16);
  dma_wait16();
  __syncthreads();

  // ---- hidden-block loop.  A block is 28 groups of 8 MFMAs behind a fenced fragment pipeline: the two ds_read_b128 of
  // group i + 1 are issued before the MFMAs of group i and nothing is scheduled across the fences.  Groups 0, 1:
  // GEMM1 (Wp fragments 52 + 2 nt + kh); groups 2 .. 27: GEMM2, group (p, nt) = channel tiles 2 p, 2 p + 1, hidden
  // half nt (W2 fragments 2 t + nt).
  f32x4 h0, h1, wa, wb;
  auto frag_a = [](int gq) { return gq == 0 ? 52 : gq == 1 ? 53 : 4 * ((gq - 2) >> 1) + ((gq - 2) & 1); };
  auto frag_b = [](int gq) { return gq == 0 ? 54 : gq == 1 ? 55 : 4 * ((gq - 2) >> 1) + 2 + ((gq - 2) & 1); };
  // Order matters for the hardware's in-order vmcnt: the gathered Aq / Kt slices (compiler-tracked loads, issued one
  // block ago) are consumed FIRST, before any DMA of this block is issued.  The compiler does not see the DMA: had one
  // been issued before this use, the compiler's "all but my newest loads" wait would make the wave sit out the DMA's
  // full latency at the top of every block.
  auto begin_block = [&](const float* __restrict__ cur) {
    h0 = f32x4{ia0.x - ik0.x, ia0.y - ik0.y, ia0.z - ik0.z, ia0.w - ik0.w};
    h1 = f32x4{ia1.x - ik1.x, ia1.y - ik1.y, ia1.z - ik1.z, ia1.w - ik1.w};
    wa = *reinterpret_cast<const f32x4*>(cur + lane * 4 + 52 * AFRAG);
    wb = *reinterpret_cast<const f32x4*>(cur + lane * 4 + 54 * AFRAG);
    __builtin_amdgcn_sched_barrier(0);
  };
  // The block's 11 vector-memory instructions per wave (4 gathered Aq / Kt slices of the next block, 7 DMA fragments of
  // the next stage; the packed stream has AHB + 1 stages, so "stage hb + 1" is branch-free and the last one brings P2)
  // are SPREAD over the MFMA groups, one every other group.  Issued back to back at the top of the block, the eight
  // waves' 88 KB queued up in the CU's single vector-memory path (64 B / clk) and every wave's instruction stream --
  // MFMAs included -- sat behind its own stalled VMEM issue: measured 3.0 % (DMA) + 3.4 % (gathers) of the loop.
  // Gathers first (consumed at the next block's top, after the DMA has been waited for anyway), the last DMA
  // fragment 5 groups before the barrier.
  auto groups = [&](const int hb, const float* __restrict__ cur, const float* nxt) {
    const float* f = cur + lane * 4;
    const int hn = hb + 1 < AHB ? hb + 1 : hb;       // (clamped: the last block re-reads its own slices)
    const float* nsrc = a.wstream + (int64_t)(hb + 1) * ASTAGE;
#pragma unroll
    for (int gq = 0; gq < 28; ++gq) {
      const f32x4 ca = wa, cb = wb;
      if (gq + 1 < 28) {
        wa = *reinterpret_cast<const f32x4*>(f + frag_a(gq + 1) * AFRAG);
        wb = *reinterpret_cast<const f32x4*>(f + frag_b(gq + 1) * AFRAG);
      }
#ifndef OCC4D_CA16_ABL_NOGATHER
      if (gq == 1) ia0 = *reinterpret_cast<const f32x4*>(aq_row + 32 * hn);
      if (gq == 3) ia1 = *reinterpret_cast<const f32x4*>(aq_row + 32 * hn + 16);
      if (gq == 5) ik0 = *reinterpret_cast<const f32x4*>(kt_row + 32 * hn);
      if (gq == 7) ik1 = *reinterpret_cast<const f32x4*>(kt_row + 32 * hn + 16);
#endif
#ifndef OCC4D_CA16_ABL_NODMA
      if (gq >= 9 && gq <= 21 && (gq & 1)) dma_frag16(nsrc, nxt, wave, lane, (gq - 9) >> 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (gq == 0) mm_ba(ca, cb, r_lo, h0, h1);
      else if (gq == 1) mm_ba(ca, cb, r_hi, h0, h1);
      else mm_ab(((gq - 2) & 1) ? h1 : h0, ca, cb, acc[2 * ((gq - 2) >> 1)], acc[2 * ((gq - 2) >> 1) + 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (gq == 1) {
        h0.x = fmaxf(h0.x, 0.f); h0.y = fmaxf(h0.y, 0.f); h0.z = fmaxf(h0.z, 0.f); h0.w = fmaxf(h0.w, 0.f);
        h1.x = fmaxf(h1.x, 0.f); h1.y = fmaxf(h1.y, 0.f); h1.z = fmaxf(h1.z, 0.f); h1.w = fmaxf(h1.w, 0.f);
      }
    }
  };
#ifdef OCC4D_CA16_STAMP
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("; OCC4D_MARK loop");
  // two blocks per loop trip so that the LDS buffers are compile-time objects; one barrier per block
#pragma clang loop unroll(disable)
  for (int hb = 0; hb < AHB; hb += 2) {
    begin_block(buf0);
    groups(hb, buf0, buf1);
    dma_wait16();
    __syncthreads();
    begin_block(buf1);
    groups(hb + 1, buf1, buf0);
    dma_wait16();
    __syncthreads();
  }
#ifdef OCC4D_CA16_STAMP
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("; OCC4D_MARK epilogue");
  // buf0 now holds P2's fragments (stage AHB); buf1 is free: the 9th query's partial softmax (3 x 7 x D floats)

  // ---- epilogue.  C/D rows of this lane: 4 g + reg; rows 0-13 = neighbours of query q0 + wave, rows 14, 15 (g = 3,
  // reg 2, 3) = slots 2 wave, 2 wave + 1 of the 9th query.
  float* const s_part = buf1;
  int jrow[4];
  bool vmain[4], vninth[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i;
    const int ql = row < 14 ? wave : 8;
    const int slot = row < 14 ? row : 2 * wave + row - 14;
    jrow[i] = s_idx[ql * 16 + min(slot, 15)];
    vmain[i] = row < 14 && slot < a.K;
    vninth[i] = row >= 14 && wave < 7 && slot < a.K;
  }
  const int qm = q0 + wave;
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = LOG2E / a.divisor;
  const bool ninth_writer = g == 3 && wave < 7;
  const float* fp = buf0 + lane * 4;
  const float* s_b2 = buf0 + 52 * AFRAG;       // the P2 stage's spare fragments carry attn_mlp[2].bias ...
  const float* s_c2 = buf0 + 54 * AFRAG;       // ... and pos_mlp[2].bias
  // Two halves of 14 + 12 channel tiles.  Per half: (1) GEMM3 as one more fenced MFMA stage -- pe[pair][ch] =
  // r[pair][:] . P2[ch][:], two tiles advanced alternately, fragments (t, kh) = 2 t + kh of the P2 stage; (2) the
  // softmax / aggregation of those tiles: pure VALU + the gathered V rows (fetched two tiles ahead), no MFMA latency
  // on its critical path.  (Doing both per tile left every tile waiting for its own 8-MFMA chain and its loads:
  // the epilogue took 10 % of the kernel.)
  auto half = [&](auto T0c, auto NTc) {
    constexpr int T0 = decltype(T0c)::value, NTH = decltype(NTc)::value;
    f32x4 pe[NTH];
#pragma unroll
    for (int i = 0; i < NTH; ++i) pe[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float vq[3][4];
    auto loadV = [&](int t, float* V) {
#ifndef OCC4D_CA16_ABL_NOV
#pragma unroll
      for (int i = 0; i < 4; ++i) V[i] = a.vt[(int64_t)jrow[i] * a.ld_vt + 16 * t + c];
#else
#pragma unroll
      for (int i = 0; i < 4; ++i) V[i] = (float)(jrow[i] + t);
#endif
    };
    loadV(T0, vq[0]);
    loadV(T0 + 1, vq[1]);
    {
      f32x4 pa = *reinterpret_cast<const f32x4*>(fp + (2 * T0) * AFRAG);
      f32x4 pb = *reinterpret_cast<const f32x4*>(fp + (2 * T0 + 2) * AFRAG);
#pragma unroll
      for (int q = 0; q < NTH; ++q) {                 // group q: tile pair q >> 1, k half q & 1
        const int p = q >> 1, kh = q & 1;
        const f32x4 ca = pa, cb = pb;
        if (q + 1 < NTH) {
          const int pn = (q + 1) >> 1, kn = (q + 1) & 1;
          pa = *reinterpret_cast<const f32x4*>(fp + (2 * (T0 + 2 * pn) + kn) * AFRAG);
          pb = *reinterpret_cast<const f32x4*>(fp + (2 * (T0 + 2 * pn + 1) + kn) * AFRAG);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm_ab(kh ? r_hi : r_lo, ca, cb, pe[2 * p], pe[2 * p + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int tl = 0; tl < NTH; ++tl) {
      const int t = T0 + tl;
      const int ch = 16 * t + c;
      if (tl + 2 < NTH) loadV(t + 2, vq[(tl + 2) % 3]);
      const float* vv = vq[tl % 3];
      // logits in the log2 domain: lg = (acc + b2) / sqrt(D) * log2(e) = acc * sc + b2 * sc (b2 * sc comes pre-scaled
      // in the stream), so that exp(x - max) is one v_exp_f32 of a difference
      const float b2s = s_b2[ch], c2c = s_c2[ch];
      float lg[4], val[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lg[i] = fmaf(acc[t][i], sc, b2s);
        val[i] = (pe[tl][i] + c2c) + vv[i];
      }
      // main query: 14 rows over the four lane groups (masked rows: -inf -> weight 0)
      {
        float lm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lm[i] = vmain[i] ? lg[i] : -__builtin_inff();
        float mx = fmaxf(fmaxf(lm[0], lm[1]), fmaxf(lm[2], lm[3]));
        mx = max_x32(max_x16(mx));
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float e = __builtin_amdgcn_exp2f(lm[i] - mx);
          den += e;
          num = fmaf(e, val[i], num);
        }
        den = sum_x32(sum_x16(den));
        num = sum_x32(sum_x16(num));
        if (g == 0 && qm < a.N) a.agg[(int64_t)qm * a.ld_agg + ch] = num * __builtin_amdgcn_rcpf(den);
      }
      // 9th query: this wave's 2 slots live in registers 2, 3 of lane group 3.  Computed branch-free by every lane
      // (a dozen VALU ops), written by the owning lanes only.
      {
        const float l2 = vninth[2] ? lg[2] : -__builtin_inff(), l3 = vninth[3] ? lg[3] : -__builtin_inff();
        const float mx = fmaxf(l2, l3);
        const float ms = mx > -__builtin_inff() ? mx : 0.f;          // (no live slot: exp2(-inf - 0) = 0)
        const float e2 = __builtin_amdgcn_exp2f(l2 - ms), e3 = __builtin_amdgcn_exp2f(l3 - ms);
        if (ninth_writer) {
          s_part[(0 * 7 + wave) * AD + ch] = mx;
          s_part[(1 * 7 + wave) * AD + ch] = e2 + e3;
          s_part[(2 * 7 + wave) * AD + ch] = fmaf(e2, val[2], e3 * val[3]);
        }
      }
    }
  };
  half(std::integral_constant<int, 0>{}, std::integral_constant<int, 14>{});
  half(std::integral_constant<int, 14>{}, std::integral_constant<int, 12>{});
  __syncthreads();
  const int q8 = q0 + 8;
  if (q8 < a.N && tid < AD) {
    const int ch = tid;
    float m = -__builtin_inff();
#pragma unroll
    for (int w = 0; w < 7; ++w) m = fmaxf(m, s_part[(0 * 7 + w) * AD + ch]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int w = 0; w < 7; ++w) {
      const float wgt = __builtin_amdgcn_exp2f(s_part[(0 * 7 + w) * AD + ch] - m);    // maxima are in the log2 domain
      den += wgt * s_part[(1 * 7 + w) * AD + ch];
      num += wgt * s_part[(2 * 7 + w) * AD + ch];
    }
    a.agg[(int64_t)q8 * a.ld_agg + ch] = num / den;
  }
#ifdef OCC4D_CA16_STAMP
  // debug build: (prologue, loop, epilogue) cycles of every wave into the rows behind the N output rows
  if (lane == 0) {
    const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
    float* o = a.agg + (int64_t)(a.N + blockIdx.x) * a.ld_agg + 4 * wave;
    o[0] = (float)(ts1 - ts0); o[1] = (float)(ts2 - ts1); o[2] = (float)(ts3 - ts2);
  }
#endif
}

}  // namespace

extern "C" int64_t occ4d_pt_cross_attn16_stream_floats(void) { return (int64_t)(AHB + 1) * ASTAGE; }

extern "C" int occ4d_pt_cross_attn16_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs, const float* apos,
                                         int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                                         int64_t ld_vt, const float* P1, const float* c1, const float* wstream,
                                         float* agg, int64_t ld_agg, int n, int m, int k, int d, float divisor,
                                         void* stream) {
  OCC4D_REQUIRE(d == AD, "occ4d_pt_cross_attn16: built for d = %d, got %d", AD, d);
  OCC4D_REQUIRE(k >= 1 && k <= AKMAX, "occ4d_pt_cross_attn16: k=%d outside [1,%d]", k, AKMAX);
  OCC4D_REQUIRE(m >= 1 && n >= 0, "occ4d_pt_cross_attn16: bad n/m");
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vt && P1 && c1 && wstream && agg,
                "occ4d_pt_cross_attn16: null pointer");
  OCC4D_REQUIRE(ld_aq >= 2 * d && ld_kt >= 2 * d && ld_vt >= d && ld_agg >= d && qs >= 3 && as >= 3,
                "occ4d_pt_cross_attn16: leading dimension too small");
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ((uintptr_t)aq % 16) == 0 && ((uintptr_t)kt % 16) == 0 &&
                    ((uintptr_t)wstream % 16) == 0,
                "occ4d_pt_cross_attn16: aq / kt / wstream must be 16-byte aligned with ld %% 4 == 0");
  OCC4D_REQUIRE(divisor > 0.f, "occ4d_pt_cross_attn16: divisor must be > 0");
  Attn16Args a{aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wstream, agg, ld_agg, n, m, k, divisor};
  cross_attn16_kernel<<<occ4d::cdiv(n, AQPB), 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_pt_cross_attn16");
}
