// Row-resident fused layers of the decoder trunk, HALF-CU shape (D6 / K11 of SURVEY.md: ResnetBlockFC.forward,
// model/implicit.py:92-101, and the 416-wide Linear layers around the cross-attention blocks,
// model/modules.py:61-65).  Same arithmetic and register layout as csrc/trunk.hip (a wave owns 16 rows, every GEMM in
// transposed form on v_mfma_f32_16x16x4_f32, the activations of a row tile never leave the registers between the
// layers of a block), re-cut so that a workgroup takes HALF a CU:
//   * 4 waves (one per SIMD) x 16 rows = 64 rows per workgroup, stages of 16 channels (26 fragments = 26 KB), two
//     stage buffers = 52 KB of LDS: two workgroups share a CU -- of this kernel, or of whatever the other decode
//     stream is running (csrc/crossattn16p.hip has the same shape).  A workgroup's memory phases (the 104-register
//     row load, the row store, the gathered interpolation term) then sit under the co-resident workgroup's MFMA
//     stream instead of idling the CU: with one 8-wave workgroup per CU all waves were in those phases together
//     (measured: 16 of 180 us for the residual block).
//   * no VALU instruction in the stage loop beyond the activation function: on gfx950 the fp32 MFMAs and the plain
//     VALU share the SIMD's vector issue (profiles/micro/valu_beside_mfma.hip), so DMA addresses are scalar base +
//     one constant lane offset.
// Residual block  y = x + W1 relu(W0 relu(x) + b0) + b1  as a loop over 26 hidden chunks of 16: stage A:
// h = relu(W0[16 j .., :] relu(x) + b0) (104 MFMAs on two accumulators over the even / odd K groups), stage B:
// yacc += W1[:, 16 j ..] h (104 MFMAs into the 26 output tiles, initialised with x + b1).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int QH = 416;                      // trunk width
constexpr int QKG = QH / 16;                 // 26 channel groups of 16
constexpr int QNC = QH / 16;                 // 26 hidden chunks of 16
constexpr int QFRAG = 256;                   // floats per fragment image (64 lanes x float4)
constexpr int QSTAGE = QKG * QFRAG;          // 6656 floats = 26624 B
constexpr int QROWS = 64;                    // rows per workgroup

struct Trunk4Args {
  const float* x; int64_t ldx;
  float* y; int64_t ldy;
  const float* w0p; const float* b0;
  const float* w1p; const float* b1;
  const float* res; int64_t ldr;
  const float* zconst; const float* ztab; int64_t ldz; const int32_t* zidx; const float* zw; int kz;
  int n;
  int n_stages;                              // rowlin: output channels / 16
  int relu_in;
  const float* mask; int64_t ldm;            // rowlin: output zeroed where mask <= 0 (ReLU mask of a data gradient), or null
  // rowlin: the row tiles behind the last FULL dispatch round (full_tiles = a multiple of 2 x CUs) are each split over
  // `tail_parts` workgroups by output stage range, so that a nearly empty last round is a fraction of a round long
  int full_tiles, tail_parts;
  int res_post;                              // rowlin with mask: res is added AFTER the mask (skip gradient of a residual block)
};

__device__ __forceinline__ unsigned lds_addr_q(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
// one fragment (1 KB), global (L2) -> LDS by DMA: scalar fragment address + this lane's 16 bytes (no VALU)
__device__ __forceinline__ void dma_frag_q(const float* __restrict__ src_frag, unsigned lds_dst, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(src_frag) : "memory");
}
// the i-th of this wave's fragments of a 26-fragment stage: fragment wave + 4 i (i = 6 exists for waves 0, 1 only)
__device__ __forceinline__ void dma_part_q(const float* __restrict__ src, const float* dst, int wave, unsigned lane16, int i) {
  const int c = wave + 4 * i;
  if (c < QKG) dma_frag_q(src + c * QFRAG, lds_addr_q(dst) + (unsigned)c * (QFRAG * 4), lane16);
}
__device__ __forceinline__ void dma_stage_q(const float* __restrict__ src, const float* dst, int wave, unsigned lane16) {
#pragma unroll
  for (int i = 0; i < 7; ++i) dma_part_q(src, dst, wave, lane16, i);
}
__device__ __forceinline__ void dma_wait_q() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 8 MFMAs: two weight fragments against two activation registers into two accumulators, alternating
__device__ __forceinline__ void mm_kk(const f32x4 wa, const f32x4 wb, const f32x4 va, const f32x4 vb, f32x4& c0, f32x4& c1) {
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, va.x, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.x, vb.x, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, va.y, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.y, vb.y, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, va.z, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.z, vb.z, c1, 0, 0, 0);
  c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, va.w, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.w, vb.w, c1, 0, 0, 0);
}
// 8 MFMAs: two weight fragments against ONE activation register into two accumulators, alternating
__device__ __forceinline__ void mm_nn(const f32x4 wa, const f32x4 wb, const f32x4 v, f32x4& c0, f32x4& c1) {
  mm_kk(wa, wb, v, v, c0, c1);
}
__device__ __forceinline__ f32x4 relu4q(f32x4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

// out[t] += zconst + sum_j zw[row, j] * ztab[zidx[row, j], 16 t + 4 g ..]: one neighbour at a time, its 26 gathers in
// flight in the registers of the dead input activation.  A memory phase of the workgroup: the co-resident workgroup
// has the matrix pipe meanwhile.
__device__ __forceinline__ void interp_into_q(const Trunk4Args& a, int rowc, int g, f32x4* out) {
#pragma unroll
  for (int t = 0; t < QKG; ++t) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(a.zconst + 16 * t + 4 * g);
    out[t].x += c.x; out[t].y += c.y; out[t].z += c.z; out[t].w += c.w;
  }
  const float* wrow = a.zw + (int64_t)rowc * a.kz;
  const int32_t* irow = a.zidx + (int64_t)rowc * a.kz;
  for (int j = 0; j < a.kz; ++j) {
    const float w0 = wrow[j];
    const float* z0 = a.ztab + (int64_t)irow[j] * a.ldz + 4 * g;
    f32x4 za[QKG];
#pragma unroll
    for (int t = 0; t < QKG; ++t) za[t] = *reinterpret_cast<const f32x4*>(z0 + 16 * t);
#pragma unroll
    for (int t = 0; t < QKG; ++t) {
      out[t].x = fmaf(w0, za[t].x, out[t].x); out[t].y = fmaf(w0, za[t].y, out[t].y);
      out[t].z = fmaf(w0, za[t].z, out[t].z); out[t].w = fmaf(w0, za[t].w, out[t].w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// y = x + W1 relu(W0 relu(x) + b0) + b1  [+ interpolation term of the next block]
__global__ __launch_bounds__(256, 2) void resblock4_kernel(const Trunk4Args a) {
  // two separate LDS objects (csrc/trunk.hip: one array split by an offset made the compiler wait for the DMA just
  // issued before the first ds_read)
  __shared__ __attribute__((aligned(16))) float bufA[QSTAGE];   // W0 rows of the current hidden chunk
  __shared__ __attribute__((aligned(16))) float bufB[QSTAGE];   // W1 columns of the current hidden chunk
  __shared__ __attribute__((aligned(16))) float s_b0[QH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const unsigned lane16 = lane * 16;
  const int row = blockIdx.x * QROWS + wave * 16 + r;
  const int rowc = min(row, a.n - 1);

  dma_stage_q(a.w0p, bufA, wave, lane16);
  for (int i = tid; i < QH; i += 256) s_b0[i] = a.b0[i];
  f32x4 xr[QKG], yacc[QKG];
  {
    const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * g;
#pragma unroll
    for (int t = 0; t < QKG; ++t) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 16 * t);
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + 16 * t + 4 * g);
      xr[t] = relu4q(v);
      yacc[t].x = v.x + b.x; yacc[t].y = v.y + b.y; yacc[t].z = v.z + b.z; yacc[t].w = v.w + b.w;
    }
  }
  dma_wait_q();
  __syncthreads();
  const float* const fa = bufA + lane * 4;
  const float* const fb = bufB + lane * 4;

#ifdef OCC4D_TR4_STAMP
  // per-wave cycle accounting (debug build): [0] stage A, [1] wait at barrier 1, [2] stage B, [3] wait at barrier 2
  unsigned long long tacc[4] = {0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
  const unsigned long long rstart = __builtin_amdgcn_s_memrealtime();
#define STAMP4(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define STAMP4(i)
#endif
  // the packed W0 stream carries QNC + 1 stages (the last repeats stage 0), so "prefetch chunk j + 1" is branch-free
#pragma clang loop unroll(disable)
  for (int j = 0; j < QNC; ++j) {
    // ---- stage A: h = relu(W0[16 j .. 16 j + 16, :] relu(x) + b0): 13 groups of 8 MFMAs, group q = K groups 2 q and
    // 2 q + 1 on the accumulators h0 / h1; meanwhile W1's chunk j lands in bufB.  Fenced fragment pipeline: the
    // ds_reads of group q + 1 are issued before the MFMAs of group q (csrc/trunk.hip).
    f32x4 h0 = *reinterpret_cast<const f32x4*>(s_b0 + 16 * j + 4 * g);
    f32x4 h1 = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4 wa = *reinterpret_cast<const f32x4*>(fa);
      f32x4 wb = *reinterpret_cast<const f32x4*>(fa + QFRAG);
      const float* nsrc = a.w1p + (int64_t)j * QSTAGE;
#pragma unroll
      for (int q = 0; q < QKG / 2; ++q) {
#ifndef OCC4D_TR4_NODMA
        if (q >= 1 && q <= 7) dma_part_q(nsrc, bufB, wave, lane16, q - 1);
#endif
        const f32x4 ca = wa, cb = wb;
        if (q + 1 < QKG / 2) {
          wa = *reinterpret_cast<const f32x4*>(fa + (2 * q + 2) * QFRAG);
          wb = *reinterpret_cast<const f32x4*>(fa + (2 * q + 3) * QFRAG);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm_kk(ca, cb, xr[2 * q], xr[2 * q + 1], h0, h1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x4 h;
    h.x = fmaxf(h0.x + h1.x, 0.f); h.y = fmaxf(h0.y + h1.y, 0.f);
    h.z = fmaxf(h0.z + h1.z, 0.f); h.w = fmaxf(h0.w + h1.w, 0.f);
    STAMP4(0)
#ifndef OCC4D_TR4_NOBAR
    dma_wait_q();
    __syncthreads();
#endif
    STAMP4(1)
    // ---- stage B: yacc += W1[:, 16 j .. 16 j + 16] h: 13 groups, group p = output tiles 2 p, 2 p + 1; meanwhile W0's
    // chunk j + 1 lands in bufA
    {
      f32x4 wa = *reinterpret_cast<const f32x4*>(fb);
      f32x4 wb = *reinterpret_cast<const f32x4*>(fb + QFRAG);
      const float* nsrc = a.w0p + (int64_t)(j + 1) * QSTAGE;
#pragma unroll
      for (int p = 0; p < QKG / 2; ++p) {
#ifndef OCC4D_TR4_NODMA
        if (p >= 1 && p <= 7) dma_part_q(nsrc, bufA, wave, lane16, p - 1);
#endif
        const f32x4 ca = wa, cb = wb;
        if (p + 1 < QKG / 2) {
          wa = *reinterpret_cast<const f32x4*>(fb + (2 * p + 2) * QFRAG);
          wb = *reinterpret_cast<const f32x4*>(fb + (2 * p + 3) * QFRAG);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm_nn(ca, cb, h, yacc[2 * p], yacc[2 * p + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP4(2)
#ifndef OCC4D_TR4_NOBAR
    dma_wait_q();
    __syncthreads();
#endif
    STAMP4(3)
  }
#ifdef OCC4D_TR4_STAMP
  if (lane == 0 && a.zw) {     // debug build: a.zw doubles as the stamp buffer (6 x u64 per wave)
    unsigned long long* o = (unsigned long long*)a.zw + (size_t)(blockIdx.x * 4 + wave) * 6;
    o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3]; o[4] = tprev - tstart; o[5] = __builtin_amdgcn_s_memrealtime() - rstart;
  }
  if (a.ztab == (const float*)1) return;
#else
  if (a.ztab) interp_into_q(a, rowc, g, yacc);
#endif
  if (row < a.n) {
    float* yp = a.y + (int64_t)row * a.ldy + 4 * g;
#pragma unroll
    for (int t = 0; t < QKG; ++t) *reinterpret_cast<f32x4*>(yp + 16 * t) = yacc[t];
  }
}

// ---------------------------------------------------------------------------------------------------
// y[:, 0 .. 16 S) = [res +] W [relu](x) + b  [+ interpolation term], K = 416, one stage per 16 output channels
__device__ __forceinline__ void rowlin4_stage(const Trunk4Args& a, int s, const float* __restrict__ frag, const f32x4* xr,
                                              int row, int rowc, int g, const float* next_src, const float* next_dst,
                                              int wave, unsigned lane16) {
  const int c0 = 16 * s + 4 * g;
  // bias / residual: compiler-tracked global loads consumed at the END of the stage (the hardware's vmcnt is in order:
  // consuming them earlier would wait for the DMA as well)
  const f32x4 bz = *reinterpret_cast<const f32x4*>(a.b0 + c0);
  f32x4 rs = {0.f, 0.f, 0.f, 0.f};
  if (a.res) rs = *reinterpret_cast<const f32x4*>(a.res + (int64_t)rowc * a.ldr + c0);
  f32x4 mk = {1.f, 1.f, 1.f, 1.f};
  if (a.mask) mk = *reinterpret_cast<const f32x4*>(a.mask + (int64_t)rowc * a.ldm + c0);
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
  {
    f32x4 wa = *reinterpret_cast<const f32x4*>(frag);
    f32x4 wb = *reinterpret_cast<const f32x4*>(frag + QFRAG);
#pragma unroll
    for (int q = 0; q < QKG / 2; ++q) {
      if (q >= 1 && q <= 7) dma_part_q(next_src, next_dst, wave, lane16, q - 1);
      const f32x4 ca = wa, cb = wb;
      if (q + 1 < QKG / 2) {
        wa = *reinterpret_cast<const f32x4*>(frag + (2 * q + 2) * QFRAG);
        wb = *reinterpret_cast<const f32x4*>(frag + (2 * q + 3) * QFRAG);
      }
      __builtin_amdgcn_sched_barrier(0);
      mm_kk(ca, cb, xr[2 * q], xr[2 * q + 1], o0, o1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 post = {0.f, 0.f, 0.f, 0.f};
  if (a.res_post) { post = rs; rs = f32x4{0.f, 0.f, 0.f, 0.f}; }           // (kernel-uniform)
  o0.x += o1.x + (bz.x + rs.x); o0.y += o1.y + (bz.y + rs.y); o0.z += o1.z + (bz.z + rs.z); o0.w += o1.w + (bz.w + rs.w);
  if (a.ztab) {
    const f32x4 ca = *reinterpret_cast<const f32x4*>(a.zconst + c0);
    o0.x += ca.x; o0.y += ca.y; o0.z += ca.z; o0.w += ca.w;
    for (int j = 0; j < a.kz; ++j) {
      const float w = a.zw[(int64_t)rowc * a.kz + j];
      const f32x4 za = *reinterpret_cast<const f32x4*>(a.ztab + (int64_t)a.zidx[(int64_t)rowc * a.kz + j] * a.ldz + c0);
      o0.x = fmaf(w, za.x, o0.x); o0.y = fmaf(w, za.y, o0.y); o0.z = fmaf(w, za.z, o0.z); o0.w = fmaf(w, za.w, o0.w);
    }
  }
  if (a.mask) {
    o0.x = mk.x > 0.f ? o0.x : 0.f; o0.y = mk.y > 0.f ? o0.y : 0.f;
    o0.z = mk.z > 0.f ? o0.z : 0.f; o0.w = mk.w > 0.f ? o0.w : 0.f;
  }
  if (a.res_post) { o0.x += post.x; o0.y += post.y; o0.z += post.z; o0.w += post.w; }
  if (row < a.n) *reinterpret_cast<f32x4*>(a.y + (int64_t)row * a.ldy + c0) = o0;
}

__global__ __launch_bounds__(256, 2) void rowlin4_kernel(const Trunk4Args a) {
  __shared__ __attribute__((aligned(16))) float bufA[QSTAGE];
  __shared__ __attribute__((aligned(16))) float bufB[QSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const unsigned lane16 = lane * 16;
  int tile = blockIdx.x, s0 = 0, s1 = a.n_stages;
  if (a.tail_parts > 1 && (int)blockIdx.x >= a.full_tiles) {          // (workgroup-uniform)
    const int t = (int)blockIdx.x - a.full_tiles;
    const int per = (a.n_stages + a.tail_parts - 1) / a.tail_parts;
    tile = a.full_tiles + t / a.tail_parts;
    s0 = (t % a.tail_parts) * per;
    s1 = min(a.n_stages, s0 + per);
    if (s0 >= s1) return;
  }
  const int row = tile * QROWS + wave * 16 + r;
  const int rowc = min(row, a.n - 1);
  dma_stage_q(a.w0p + (int64_t)s0 * QSTAGE, bufA, wave, lane16);
  f32x4 xr[QKG];
  {
    const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * g;
#pragma unroll
    for (int t = 0; t < QKG; ++t) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 16 * t);
      xr[t] = a.relu_in ? relu4q(v) : v;
    }
  }
  dma_wait_q();
  __syncthreads();
  // the packed stream carries n_stages + 1 stages (the last repeats stage 0): prefetching is branch-free
#pragma clang loop unroll(disable)
  for (int s = s0; s < s1; s += 2) {
    rowlin4_stage(a, s, bufA + lane * 4, xr, row, rowc, g, a.w0p + (int64_t)(s + 1) * QSTAGE, bufB, wave, lane16);
    dma_wait_q();
    __syncthreads();
    if (s + 1 < s1)
      rowlin4_stage(a, s + 1, bufB + lane * 4, xr, row, rowc, g, a.w0p + (int64_t)(s + 2) * QSTAGE, bufA, wave, lane16);
    dma_wait_q();
    __syncthreads();
  }
}

int check_common4(const Trunk4Args& a, const char* who) {
  OCC4D_REQUIRE(a.x && a.y && a.w0p && a.b0, "%s: null pointer", who);
  OCC4D_REQUIRE(a.n >= 0, "%s: n = %d", who, a.n);
  OCC4D_REQUIRE(a.ldx >= QH && a.ldx % 4 == 0 && a.ldy % 4 == 0 && ((uintptr_t)a.x % 16) == 0 &&
                    ((uintptr_t)a.y % 16) == 0 && ((uintptr_t)a.w0p % 16) == 0 && ((uintptr_t)a.b0 % 16) == 0,
                "%s: x / y / weights / bias must be 16-byte aligned with row strides %% 4 == 0 (ldx >= %d)", who, QH);
  if (a.ztab) {
    OCC4D_REQUIRE(a.zconst && a.zidx && a.zw && a.kz >= 1 && a.ldz % 4 == 0 && ((uintptr_t)a.ztab % 16) == 0 &&
                      ((uintptr_t)a.zconst % 16) == 0,
                  "%s: interpolation term needs zconst / zidx / zw, kz >= 1 and a 16-byte aligned table", who);
  }
  return OCC4D_OK;
}

}  // namespace

extern "C" int64_t occ4d_trunk4_packed_floats(int n_out) { return (int64_t)(n_out / 16 + 1) * QSTAGE; }

extern "C" int occ4d_resblock4_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w0_packed,
                                   const float* b0, const float* w1_packed, const float* b1, const float* zconst,
                                   const float* ztab, int64_t ldz, const int32_t* zidx, const float* zw, int kz, int n,
                                   void* stream) {
  Trunk4Args a{x, ldx, y, ldy, w0_packed, b0, w1_packed, b1, nullptr, 0, zconst, ztab, ldz, zidx, zw, kz, n, QNC, 1};
  if (n == 0) return OCC4D_OK;               // (an empty batch has no storage: nothing to check)
  if (int rc = check_common4(a, "occ4d_resblock4_f32")) return rc;
  OCC4D_REQUIRE(w1_packed && b1 && ((uintptr_t)w1_packed % 16) == 0 && ((uintptr_t)b1 % 16) == 0 && ldy >= QH,
                "occ4d_resblock4_f32: second layer weights / bias missing or misaligned");
  resblock4_kernel<<<occ4d::cdiv(n, QROWS), 256, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_resblock4_f32");
}

// Launch geometry of rowlin4_kernel: whole rounds of 2 workgroups per CU as they are; when what remains is at most a
// quarter round, every remaining row tile is split over four workgroups by output stage range (measured at 68812 rows,
// 2.1 rounds: the 52 tail workgroups took half a round alone on their CUs; split, a sixth).
static int rowlin4_grid(Trunk4Args& a) {
  const int tiles = occ4d::cdiv(a.n, QROWS);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int round = 2 * cus;
  a.full_tiles = tiles / round * round;
  const int tail = tiles - a.full_tiles;
  a.tail_parts = (a.full_tiles > 0 && tail > 0 && 4 * tail <= round && a.n_stages >= 8) ? 4 : 1;
  return a.full_tiles + tail * a.tail_parts;
}

extern "C" int occ4d_rowlin4_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                 const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                 const float* zconst, const float* ztab, int64_t ldz, const int32_t* zidx,
                                 const float* zw, int kz, int n, void* stream) {
  Trunk4Args a{x, ldx, y, ldy, w_packed, b, nullptr, nullptr, res, ldr, zconst, ztab, ldz, zidx, zw, kz, n,
               n_out / 16, relu_in, nullptr, 0};
  if (n == 0) return OCC4D_OK;
  if (int rc = check_common4(a, "occ4d_rowlin4_f32")) return rc;
  OCC4D_REQUIRE(n_out >= 16 && n_out % 16 == 0 && ldy >= n_out, "occ4d_rowlin4_f32: n_out = %d must be a multiple of 16 <= ldy",
                n_out);
  OCC4D_REQUIRE(!res || (ldr % 4 == 0 && ((uintptr_t)res % 16) == 0 && ldr >= n_out),
                "occ4d_rowlin4_f32: residual rows must be 16-byte aligned with ldr %% 4 == 0");
  const int grid = rowlin4_grid(a);
  rowlin4_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_rowlin4_f32");
}

// occ4d_rowlin4_f32 with an output mask (the contract of occ4d_rowlin_masked_f32)
extern "C" int occ4d_rowlin4_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                        const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                        const float* mask, int64_t ldm, int n, void* stream) {
  Trunk4Args a{x, ldx, y, ldy, w_packed, b, nullptr, nullptr, res, ldr, nullptr, nullptr, 0, nullptr, nullptr, 0, n,
               n_out / 16, relu_in, mask, ldm};
  if (n == 0) return OCC4D_OK;
  if (int rc = check_common4(a, "occ4d_rowlin4_masked_f32")) return rc;
  OCC4D_REQUIRE(n_out >= 16 && n_out % 16 == 0 && ldy >= n_out,
                "occ4d_rowlin4_masked_f32: n_out = %d must be a multiple of 16 <= ldy", n_out);
  OCC4D_REQUIRE(!res || (ldr % 4 == 0 && ((uintptr_t)res % 16) == 0 && ldr >= n_out),
                "occ4d_rowlin4_masked_f32: residual rows must be 16-byte aligned with ldr %% 4 == 0");
  OCC4D_REQUIRE(mask && ldm % 4 == 0 && ((uintptr_t)mask % 16) == 0 && ldm >= n_out,
                "occ4d_rowlin4_masked_f32: mask rows must be 16-byte aligned with ldm %% 4 == 0 and ldm >= n_out");
  const int grid = rowlin4_grid(a);
  rowlin4_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_rowlin4_masked_f32");
}

// Data gradient of a residual block's first layer with the skip gradient folded in:
//   y = [mask > 0] (x W^T + b) + skip        (skip added AFTER the mask; b may be a zero vector)
// -- what occ4d_rowlin4_masked_f32 computes followed by an element-wise add of `skip`, in one launch.
extern "C" int occ4d_rowlin4_masked_skip_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                             const float* b, int n_out, int relu_in, const float* skip, int64_t lds,
                                             const float* mask, int64_t ldm, int n, void* stream) {
  Trunk4Args a{x, ldx, y, ldy, w_packed, b, nullptr, nullptr, skip, lds, nullptr, nullptr, 0, nullptr, nullptr, 0, n,
               n_out / 16, relu_in, mask, ldm};
  a.res_post = 1;
  if (n == 0) return OCC4D_OK;
  if (int rc = check_common4(a, "occ4d_rowlin4_masked_skip_f32")) return rc;
  OCC4D_REQUIRE(n_out >= 16 && n_out % 16 == 0 && ldy >= n_out,
                "occ4d_rowlin4_masked_skip_f32: n_out = %d must be a multiple of 16 <= ldy", n_out);
  OCC4D_REQUIRE(skip && lds % 4 == 0 && ((uintptr_t)skip % 16) == 0 && lds >= n_out,
                "occ4d_rowlin4_masked_skip_f32: skip rows must be 16-byte aligned with lds %% 4 == 0 and lds >= n_out");
  OCC4D_REQUIRE(mask && ldm % 4 == 0 && ((uintptr_t)mask % 16) == 0 && ldm >= n_out,
                "occ4d_rowlin4_masked_skip_f32: mask rows must be 16-byte aligned with ldm %% 4 == 0 and ldm >= n_out");
  const int grid = rowlin4_grid(a);
  rowlin4_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch("occ4d_rowlin4_masked_skip_f32");
}
