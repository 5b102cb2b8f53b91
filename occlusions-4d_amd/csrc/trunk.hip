// Row-resident fused layers of the decoder trunk (D6 / K11 of SURVEY.md: ResnetBlockFC.forward,
// model/implicit.py:92-101, and the 416-wide Linear layers around the cross-attention blocks,
// model/modules.py:61-65): the activations of a row tile never leave the registers between the
// layers of a block; only the weights stream (L2 -> LDS, DMA) past them.
//
// Layout (wave64, v_mfma_f32_16x16x4_f32, exact fp32): a wave owns 16 rows; lane (g = lane >> 4,
// r = lane & 15) holds, for row r, the channels 16 t + 4 g + e (e = 0..3) of every 16-channel group
// t as one float4 -- 26 float4 = 104 registers for a 416-wide activation.  Every GEMM is run in
// TRANSPOSED form  out^T[n][row] = sum_k W[n][k] act^T[k][row]:
//   A operand = weight fragment W[n0 + r][16 t + 4 g + e]   (ds_read_b128 from LDS, element e per step)
//   B operand = act register (t, e)                          (lane (g, r) = row r, k = 16 t + 4 g + e)
//   C/D       = out^T[n0 + 4 g + reg][row r]                 -> again "row r, 4 consecutive channels":
// the output registers of one layer ARE the B operand of the next layer and, at the end, one float4
// global store per lane.  Two waves per SIMD (8 waves x 16 rows = 128 rows per workgroup) -- the
// 32x32x2 form would need 208 registers per activation and leave room for one wave per SIMD only.
//
// Residual block  y = x + W1 relu(W0 relu(x) + b0) + b1  as ONE rolled loop over 13 hidden chunks
// of 32 (the structure of the attention kernel): chunk j:  h_j = relu(W0[32 j.., :] relu(x) + b0)
// (2 accumulator tiles, 208 MFMAs), then  yacc += W1[:, 32 j..] h_j  (26 accumulator tiles, 208
// MFMAs); yacc is initialised with x + b1, so the residual costs nothing.  Weights are pre-packed on
// the host (ops.pack_trunk_weights) into the exact LDS image of each stage: 52 fragments of 1 KB
// (64 lanes x 16 B, lane-linear: conflict-free ds_read_b128, contiguous global_load_lds_dwordx4), two
// stage buffers of 52 KB: while a stage's MFMAs read one buffer the DMA fills the other; one
// s_waitcnt vmcnt(0) + barrier per stage ("my part of the next stage has landed" + "everybody's has").
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 416;                      // trunk width the kernels are built for
constexpr int TKG = TH / 16;                 // 26 channel groups of 16
constexpr int TNS = TH / 32;                 // 13 stages of 32 channels
constexpr int FRAG_FLOATS = 256;             // one (n-tile, k-group) fragment image: 64 lanes x float4
constexpr int STAGE_FRAGS = 2 * TKG;         // 52: (2 n-tiles x 26 k-groups) or (26 n-tiles x 2 k-groups)
constexpr int STAGE_FLOATS = STAGE_FRAGS * FRAG_FLOATS;   // 13312 floats = 53248 B
constexpr int TROWS = 128;                   // rows per workgroup

struct TrunkArgs {
  const float* x; int64_t ldx;               // input rows (n, 416)
  float* y; int64_t ldy;                     // output rows (n, N); may alias x (each workgroup reads its rows first)
  const float* w0p; const float* b0;         // stage-packed weights / bias of the first layer
  const float* w1p; const float* b1;         // ... of the second layer (residual block only)
  const float* res; int64_t ldr;             // rowlin: residual rows added to the output, or null
  // optional inverse-distance interpolation term added to the OUTPUT rows (the next block's lin_z):
  // y[i, :] += zconst[:] + sum_j zw[i, j] * ztab[zidx[i, j], :]
  const float* zconst; const float* ztab; int64_t ldz; const int32_t* zidx; const float* zw; int kz;
  int n;                                     // rows
  int n_stages;                              // rowlin: output channels / 32
  int relu_in;                               // rowlin: relu on the operand
  const float* mask; int64_t ldm;            // rowlin: output rows zeroed where mask <= 0 (ReLU mask of a data gradient), or null
};

// LDS byte address of a __shared__ object (wave-uniform, for M0)
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// One stage = 52 fragments of 1 KB, global (L2) -> LDS by DMA; wave w moves fragments w, w + 8, ...: one
// global_load_lds_dwordx4 per fragment (LDS destination = M0 base + lane * 16, source address per lane).
// Issued through inline asm ON PURPOSE: with the __builtin the compiler, knowing that an asynchronous LDS write is
// in flight, degrades every s_waitcnt of the fragment ds_reads to lgkmcnt(0) -- each group of MFMAs then waits
// for the reads issued just before it (a full LDS round trip per 8 MFMAs; measured: a wave running alone kept the
// matrix pipe 65 % busy).  The asm is invisible to that bookkeeping; the stage protocol supplies the ordering:
// dma_wait() (s_waitcnt vmcnt(0)) + barrier before anybody reads the buffer, barrier before it is overwritten.
// (round 3: scalar fragment address + one constant lane offset -- no VALU address arithmetic: on gfx950 the fp32 MFMAs
// and the plain VALU share the SIMD's vector issue, profiles/micro/valu_beside_mfma.hip)
__device__ __forceinline__ void dma_one(const float* __restrict__ src_frag, unsigned lds_dst, int lane) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(lds_dst), "s"(src_frag) : "memory");
}
__device__ __forceinline__ void dma_stage(const float* __restrict__ src, const float* dst, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int c = wave + 8 * i;                       // wave-uniform
    if (c < STAGE_FRAGS) dma_one(src + c * FRAG_FLOATS, lds_addr(dst) + (unsigned)c * (FRAG_FLOATS * 4), lane);
  }
}

// the i-th of this wave's (up to 7) fragments of a stage: fragment wave + 8 i (i = 6 exists for waves 0-3 only)
__device__ __forceinline__ void dma_frag(const float* __restrict__ src, const float* dst, int wave, int lane, int i) {
  const int c = wave + 8 * i;                         // wave-uniform
  if (c < STAGE_FRAGS) dma_one(src + c * FRAG_FLOATS, lds_addr(dst) + (unsigned)c * (FRAG_FLOATS * 4), lane);
}

__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma4(const f32x4 w, const f32x4 v, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, v.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, v.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, v.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, v.w, acc, 0, 0, 0);
  return acc;
}

// two accumulators advanced alternately: consecutive MFMAs never depend on each other (40-cycle latency vs 32-cycle issue)
__device__ __forceinline__ void mfma4x2(const f32x4 wa, const f32x4 wb, const f32x4 v, f32x4& acc_a, f32x4& acc_b) {
  acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, v.x, acc_a, 0, 0, 0);
  acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.x, v.x, acc_b, 0, 0, 0);
  acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, v.y, acc_a, 0, 0, 0);
  acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.y, v.y, acc_b, 0, 0, 0);
  acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, v.z, acc_a, 0, 0, 0);
  acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.z, v.z, acc_b, 0, 0, 0);
  acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, v.w, acc_a, 0, 0, 0);
  acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.w, v.w, acc_b, 0, 0, 0);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

// out[t] += zconst + sum_j zw[row, j] * ztab[zidx[row, j], 16 t + 4 g ..]   (one neighbour at a time: 26 gathers in flight)
__device__ __forceinline__ void interp_into(const TrunkArgs& a, int rowc, int g, f32x4* out) {
#pragma unroll
  for (int t = 0; t < TKG; ++t) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(a.zconst + 16 * t + 4 * g);
    out[t].x += c.x; out[t].y += c.y; out[t].z += c.z; out[t].w += c.w;
  }
  for (int j = 0; j < a.kz; ++j) {
    const float w = a.zw[(int64_t)rowc * a.kz + j];
    const float* zr = a.ztab + (int64_t)a.zidx[(int64_t)rowc * a.kz + j] * a.ldz + 4 * g;
#pragma unroll
    for (int t = 0; t < TKG; ++t) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(zr + 16 * t);
      out[t].x = fmaf(w, z.x, out[t].x); out[t].y = fmaf(w, z.y, out[t].y);
      out[t].z = fmaf(w, z.z, out[t].z); out[t].w = fmaf(w, z.w, out[t].w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// y = x + W1 relu(W0 relu(x) + b0) + b1  [+ interpolation term of the next block]
__global__ __launch_bounds__(512, 2) void resblock_kernel(const TrunkArgs a) {
  // two separate LDS objects: the DMA into one is provably disjoint from the fragment reads of the other (one
  // array split by an offset made the compiler wait for vmcnt(0) -- the DMA just issued -- before the first ds_read)
  __shared__ __attribute__((aligned(16))) float bufA[STAGE_FLOATS];   // W0 chunk of the current hidden chunk
  __shared__ __attribute__((aligned(16))) float bufB[STAGE_FLOATS];   // W1 chunk
  __shared__ __attribute__((aligned(16))) float s_b0[TH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * TROWS + wave * 16 + r;
  const int rowc = min(row, a.n - 1);

  dma_stage(a.w0p, bufA, wave, lane);
  if (tid < TH) s_b0[tid] = a.b0[tid];
  f32x4 xr[TKG], yacc[TKG];
  {
    const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * g;
#pragma unroll
    for (int t = 0; t < TKG; ++t) {
#ifdef OCC4D_TR_NOPRO
      const f32x4 v = {(float)t, (float)g, (float)r, 1.f};
#else
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 16 * t);
#endif
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + 16 * t + 4 * g);
      xr[t] = relu4(v);
      yacc[t].x = v.x + b.x; yacc[t].y = v.y + b.y; yacc[t].z = v.z + b.z; yacc[t].w = v.w + b.w;
    }
  }
  dma_wait();
  __syncthreads();
  const float* const fa = bufA + lane * 4;   // this lane's float4 inside a fragment image
  const float* const fb = bufB + lane * 4;

#ifdef OCC4D_TR_STAMP
  // per-wave cycle accounting (debug build): [0] stage A issue -> last MFMA, [1] wait at barrier 1, [2] stage B, [3] wait 2
  unsigned long long tacc[4] = {0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
#define STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define STAMP(i)
#endif
  asm volatile("; OCC4D_MARK loop");
  // the packed W0 stream carries TNS + 1 stages (the last repeats stage 0), so "prefetch stage j + 1" is branch-free
#pragma clang loop unroll(disable)
  for (int j = 0; j < TNS; ++j) {
    // ---- stage A: h = relu(W0[32 j .. 32 j + 32, :] relu(x) + b0); meanwhile W1's chunk j lands in bufB
    f32x4 h0 = *reinterpret_cast<const f32x4*>(s_b0 + 32 * j + 4 * g);
    f32x4 h1 = *reinterpret_cast<const f32x4*>(s_b0 + 32 * j + 16 + 4 * g);
    {
      // fragment pipeline: the ds_reads of group t + 1 are issued BEFORE the 8 MFMAs of group t and nothing may be
      // scheduled across the fences (left alone, the compiler sinks the reads to just before their use and every 16
      // MFMAs wait for a full LDS round trip: measured 65 % of the matrix pipe for a wave that runs alone)
      f32x4 wa = *reinterpret_cast<const f32x4*>(fa);
      f32x4 wb = *reinterpret_cast<const f32x4*>(fa + TKG * FRAG_FLOATS);
#pragma unroll
      for (int t = 0; t < TKG; ++t) {
#ifndef OCC4D_TR_NODMA
        // the other buffer's DMA, one fragment every third group, between MFMAs: issued back to back right after the
        // barrier, the eight waves' 52 KB queued up in the CU's single vector-memory path and every wave's
        // instruction stream -- MFMAs included -- sat behind its own stalled VMEM issue
        if (t >= 2 && t <= 20 && (t - 2) % 3 == 0) dma_frag(a.w1p + (int64_t)j * STAGE_FLOATS, bufB, wave, lane, (t - 2) / 3);
#endif
        const f32x4 ca = wa, cb = wb;
        if (t + 1 < TKG) {
          wa = *reinterpret_cast<const f32x4*>(fa + (t + 1) * FRAG_FLOATS);
          wb = *reinterpret_cast<const f32x4*>(fa + (TKG + t + 1) * FRAG_FLOATS);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma4x2(ca, cb, xr[t], h0, h1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    h0 = relu4(h0);
    h1 = relu4(h1);
    STAMP(0)
#ifndef OCC4D_TR_NOBAR
    dma_wait();
    __syncthreads();
#endif
    STAMP(1)
    // ---- stage B: yacc += W1[:, 32 j .. 32 j + 32] h; meanwhile W0's chunk j + 1 lands in bufA
    {
      // group (p, tt): output tiles 2 p and 2 p + 1, hidden half tt; same fenced pipeline
      f32x4 wa = *reinterpret_cast<const f32x4*>(fb);
      f32x4 wb = *reinterpret_cast<const f32x4*>(fb + 2 * FRAG_FLOATS);
#pragma unroll
      for (int q = 0; q < TKG; ++q) {
        const int p = q >> 1, tt = q & 1;
#ifndef OCC4D_TR_NODMA
        if (q >= 2 && q <= 20 && (q - 2) % 3 == 0) dma_frag(a.w0p + (int64_t)(j + 1) * STAGE_FLOATS, bufA, wave, lane, (q - 2) / 3);
#endif
        const f32x4 ca = wa, cb = wb;
        if (q + 1 < TKG) {
          const int pn = (q + 1) >> 1, tn = (q + 1) & 1;
          wa = *reinterpret_cast<const f32x4*>(fb + (4 * pn + tn) * FRAG_FLOATS);
          wb = *reinterpret_cast<const f32x4*>(fb + (4 * pn + 2 + tn) * FRAG_FLOATS);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma4x2(ca, cb, tt ? h1 : h0, yacc[2 * p], yacc[2 * p + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(2)
#ifndef OCC4D_TR_NOBAR
    dma_wait();
    __syncthreads();
#endif
    STAMP(3)
  }
  asm volatile("; OCC4D_MARK epilogue");
#ifdef OCC4D_TR_STAMP
  if (lane == 0 && a.zw) {     // debug build: a.zw doubles as the stamp buffer (5 x u64 per wave)
    unsigned long long* o = (unsigned long long*)a.zw + (size_t)(blockIdx.x * 8 + wave) * 6;
    o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3]; o[4] = tprev - tstart; o[5] = tstart;
  }
#endif
  if (a.ztab) interp_into(a, rowc, g, yacc);
#ifdef OCC4D_TR_NOEPI
  {
    float tsum = 0.f;
#pragma unroll
    for (int t = 0; t < TKG; ++t) tsum += yacc[t].x + yacc[t].y + yacc[t].z + yacc[t].w;
    if (tsum == 123.456f) a.y[0] = tsum;     // keep the accumulators live
    return;
  }
#endif
  if (row < a.n) {
    float* yp = a.y + (int64_t)row * a.ldy + 4 * g;
#pragma unroll
    for (int t = 0; t < TKG; ++t) *reinterpret_cast<f32x4*>(yp + 16 * t) = yacc[t];
  }
}

// ---------------------------------------------------------------------------------------------------
// y[:, 0 .. 32 S) = [res +] W [relu](x) + b  [+ interpolation term], K = 416, one stage per 32 output channels
__device__ __forceinline__ void rowlin_stage(const TrunkArgs& a, int s, const float* __restrict__ frag, const f32x4* xr,
                                             int row, int rowc, int g, const float* next_src, const float* next_dst,
                                             int wave, int lane) {
  const int c0 = 32 * s + 4 * g;
  // bias / residual: compiler-tracked global loads issued AFTER this stage's DMA and consumed at the END of the stage
  // (the hardware's vmcnt is in order: consuming them earlier would wait for the DMA as well)
  const f32x4 bz0 = *reinterpret_cast<const f32x4*>(a.b0 + c0);
  const f32x4 bz1 = *reinterpret_cast<const f32x4*>(a.b0 + c0 + 16);
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
  if (a.res) {
    r0 = *reinterpret_cast<const f32x4*>(a.res + (int64_t)rowc * a.ldr + c0);
    r1 = *reinterpret_cast<const f32x4*>(a.res + (int64_t)rowc * a.ldr + c0 + 16);
  }
  {
    f32x4 wa = *reinterpret_cast<const f32x4*>(frag);
    f32x4 wb = *reinterpret_cast<const f32x4*>(frag + TKG * FRAG_FLOATS);
#pragma unroll
    for (int t = 0; t < TKG; ++t) {
      if (t >= 2 && t <= 20 && (t - 2) % 3 == 0) dma_frag(next_src, next_dst, wave, lane, (t - 2) / 3);   // (see resblock_kernel)
      const f32x4 ca = wa, cb = wb;
      if (t + 1 < TKG) {
        wa = *reinterpret_cast<const f32x4*>(frag + (t + 1) * FRAG_FLOATS);
        wb = *reinterpret_cast<const f32x4*>(frag + (TKG + t + 1) * FRAG_FLOATS);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma4x2(ca, cb, xr[t], o0, o1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  o0.x += bz0.x + r0.x; o0.y += bz0.y + r0.y; o0.z += bz0.z + r0.z; o0.w += bz0.w + r0.w;
  o1.x += bz1.x + r1.x; o1.y += bz1.y + r1.y; o1.z += bz1.z + r1.z; o1.w += bz1.w + r1.w;
  if (a.ztab) {
    const f32x4 ca = *reinterpret_cast<const f32x4*>(a.zconst + c0);
    const f32x4 cb = *reinterpret_cast<const f32x4*>(a.zconst + c0 + 16);
    o0.x += ca.x; o0.y += ca.y; o0.z += ca.z; o0.w += ca.w;
    o1.x += cb.x; o1.y += cb.y; o1.z += cb.z; o1.w += cb.w;
    for (int j = 0; j < a.kz; ++j) {
      const float w = a.zw[(int64_t)rowc * a.kz + j];
      const float* zr = a.ztab + (int64_t)a.zidx[(int64_t)rowc * a.kz + j] * a.ldz + c0;
      const f32x4 za = *reinterpret_cast<const f32x4*>(zr);
      const f32x4 zb = *reinterpret_cast<const f32x4*>(zr + 16);
      o0.x = fmaf(w, za.x, o0.x); o0.y = fmaf(w, za.y, o0.y); o0.z = fmaf(w, za.z, o0.z); o0.w = fmaf(w, za.w, o0.w);
      o1.x = fmaf(w, zb.x, o1.x); o1.y = fmaf(w, zb.y, o1.y); o1.z = fmaf(w, zb.z, o1.z); o1.w = fmaf(w, zb.w, o1.w);
    }
  }
  if (a.mask) {      // dx = (x > 0) ? dx : 0: the ReLU of a relu_in layer, applied to its data gradient in the epilogue
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(a.mask + (int64_t)rowc * a.ldm + c0);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(a.mask + (int64_t)rowc * a.ldm + c0 + 16);
    o0.x = m0.x > 0.f ? o0.x : 0.f; o0.y = m0.y > 0.f ? o0.y : 0.f; o0.z = m0.z > 0.f ? o0.z : 0.f; o0.w = m0.w > 0.f ? o0.w : 0.f;
    o1.x = m1.x > 0.f ? o1.x : 0.f; o1.y = m1.y > 0.f ? o1.y : 0.f; o1.z = m1.z > 0.f ? o1.z : 0.f; o1.w = m1.w > 0.f ? o1.w : 0.f;
  }
  if (row < a.n) {
    float* yp = a.y + (int64_t)row * a.ldy + c0;
    *reinterpret_cast<f32x4*>(yp) = o0;
    *reinterpret_cast<f32x4*>(yp + 16) = o1;
  }
}

__global__ __launch_bounds__(512, 2) void rowlin_kernel(const TrunkArgs a) {
  __shared__ __attribute__((aligned(16))) float bufA[STAGE_FLOATS];
  __shared__ __attribute__((aligned(16))) float bufB[STAGE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int row = blockIdx.x * TROWS + wave * 16 + r;
  const int rowc = min(row, a.n - 1);
  dma_stage(a.w0p, bufA, wave, lane);
  f32x4 xr[TKG];
  {
    const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * g;
#pragma unroll
    for (int t = 0; t < TKG; ++t) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 16 * t);
      xr[t] = a.relu_in ? relu4(v) : v;
    }
  }
  dma_wait();
  __syncthreads();
  // the packed stream carries n_stages + 1 stages (the last repeats stage 0): prefetching is branch-free
#pragma clang loop unroll(disable)
  for (int s = 0; s < a.n_stages; s += 2) {
    rowlin_stage(a, s, bufA + lane * 4, xr, row, rowc, g, a.w0p + (int64_t)(s + 1) * STAGE_FLOATS, bufB, wave, lane);
    dma_wait();
    __syncthreads();
    if (s + 1 < a.n_stages)
      rowlin_stage(a, s + 1, bufB + lane * 4, xr, row, rowc, g, a.w0p + (int64_t)(s + 2) * STAGE_FLOATS, bufA, wave, lane);
    dma_wait();
    __syncthreads();
  }
}

int check_common(const TrunkArgs& a, const char* who) {
  OCC4D_REQUIRE(a.x && a.y && a.w0p && a.b0, "%s: null pointer", who);
  OCC4D_REQUIRE(a.n >= 0, "%s: n = %d", who, a.n);
  OCC4D_REQUIRE(a.ldx >= TH && a.ldx % 4 == 0 && a.ldy % 4 == 0 && ((uintptr_t)a.x % 16) == 0 &&
                    ((uintptr_t)a.y % 16) == 0 && ((uintptr_t)a.w0p % 16) == 0 && ((uintptr_t)a.b0 % 16) == 0,
                "%s: x / y / weights / bias must be 16-byte aligned with row strides %% 4 == 0 (ldx >= %d)", who, TH);
  if (a.ztab) {
    OCC4D_REQUIRE(a.zconst && a.zidx && a.zw && a.kz >= 1 && a.ldz % 4 == 0 && ((uintptr_t)a.ztab % 16) == 0 &&
                      ((uintptr_t)a.zconst % 16) == 0,
                  "%s: interpolation term needs zconst / zidx / zw, kz >= 1 and a 16-byte aligned table", who);
  }
  return OCC4D_OK;
}

}  // namespace

extern "C" int occ4d_trunk_width(void) { return TH; }
extern "C" int64_t occ4d_trunk_packed_floats(int n_out) { return (int64_t)(n_out / 32 + 1) * STAGE_FLOATS; }

extern "C" int occ4d_resblock_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w0_packed,
                                  const float* b0, const float* w1_packed, const float* b1, const float* zconst,
                                  const float* ztab, int64_t ldz, const int32_t* zidx, const float* zw, int kz, int n,
                                  void* stream) {
  TrunkArgs a{x, ldx, y, ldy, w0_packed, b0, w1_packed, b1, nullptr, 0, zconst, ztab, ldz, zidx, zw, kz, n, TNS, 1, nullptr, 0};
  if (n == 0) return OCC4D_OK;               // (an empty batch has no storage: nothing to check)
  if (int rc = check_common(a, "occ4d_resblock_f32")) return rc;
  OCC4D_REQUIRE(w1_packed && b1 && ((uintptr_t)w1_packed % 16) == 0 && ((uintptr_t)b1 % 16) == 0 && ldy >= TH,
                "occ4d_resblock_f32: second layer weights / bias missing or misaligned");
  if (n == 0) return OCC4D_OK;
  resblock_kernel<<<occ4d::cdiv(n, TROWS), 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_resblock_f32");
}

extern "C" int occ4d_rowlin_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                       const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                       const float* mask, int64_t ldm, int n, void* stream);

extern "C" int occ4d_rowlin_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                const float* zconst, const float* ztab, int64_t ldz, const int32_t* zidx,
                                const float* zw, int kz, int n, void* stream) {
  TrunkArgs a{x, ldx, y, ldy, w_packed, b, nullptr, nullptr, res, ldr, zconst, ztab, ldz, zidx, zw, kz, n,
              n_out / 32, relu_in, nullptr, 0};
  if (n == 0) return OCC4D_OK;
  if (int rc = check_common(a, "occ4d_rowlin_f32")) return rc;
  OCC4D_REQUIRE(n_out >= 32 && n_out % 32 == 0 && ldy >= n_out, "occ4d_rowlin_f32: n_out = %d must be a multiple of 32 <= ldy",
                n_out);
  OCC4D_REQUIRE(!res || (ldr % 4 == 0 && ((uintptr_t)res % 16) == 0 && ldr >= n_out),
                "occ4d_rowlin_f32: residual rows must be 16-byte aligned with ldr %% 4 == 0");
  if (n == 0) return OCC4D_OK;
  rowlin_kernel<<<occ4d::cdiv(n, TROWS), 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_rowlin_f32");
}

// y = mask > 0 ? ([res +] W [relu](x) + b) : 0 -- occ4d_rowlin_f32 with the ReLU mask of a data gradient applied in
// the epilogue (training: dx of a relu_in layer = (x > 0) . (g W); the separate masking pass read and wrote dx again)
extern "C" int occ4d_rowlin_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                       const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                       const float* mask, int64_t ldm, int n, void* stream) {
  TrunkArgs a{x, ldx, y, ldy, w_packed, b, nullptr, nullptr, res, ldr, nullptr, nullptr, 0, nullptr, nullptr, 0, n,
              n_out / 32, relu_in, mask, ldm};
  if (n == 0) return OCC4D_OK;
  if (int rc = check_common(a, "occ4d_rowlin_masked_f32")) return rc;
  OCC4D_REQUIRE(n_out >= 32 && n_out % 32 == 0 && ldy >= n_out, "occ4d_rowlin_masked_f32: n_out = %d must be a multiple of 32 <= ldy",
                n_out);
  OCC4D_REQUIRE(!res || (ldr % 4 == 0 && ((uintptr_t)res % 16) == 0 && ldr >= n_out),
                "occ4d_rowlin_masked_f32: residual rows must be 16-byte aligned with ldr %% 4 == 0");
  OCC4D_REQUIRE(mask && ldm % 4 == 0 && ((uintptr_t)mask % 16) == 0 && ldm >= n_out,
                "occ4d_rowlin_masked_f32: mask rows must be 16-byte aligned with ldm %% 4 == 0 and ldm >= n_out");
  rowlin_kernel<<<occ4d::cdiv(n, TROWS), 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_rowlin_masked_f32");
}
