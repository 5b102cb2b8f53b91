// Weight gradient dW[n][k] = sum_m g[m][n] [relu](x)[m][k] for the wide layers of the decoder (N a multiple of 416,
// K a multiple of 32, M large: train.py:101-118, the backward of model/implicit.py:92-101 and of the attention MLP,
// model/point_transformer_layer.py:174-176) -- second generation.  backward.hip's wgrad_kernel stays for every other
// shape and for the rows behind the last full 16-row tile.
//
// What the first kernel left on the table (profiles/train_shapes.py: 0.28-0.33 of the fp32 MFMA peak at these shapes):
// one wave per SIMD (13 accumulator tiles of 32 x 32 = 208 registers), operand tiles staged through registers with
// per-element selects (VALU, which on gfx950 shares the vector issue with the fp32 MFMAs), 128-row blocks of N = 416
// (19 % of the rows idle).  Here:
//   * v_mfma_f32_16x16x4_f32; a workgroup of 4 waves covers ALL 416 rows of an N chunk (26 tile rows of 16) and 64
//     columns of K (4 tiles): wave (nh, kh) owns the tile rows 13 nh .. 13 nh + 12 and the column tiles 2 kh, 2 kh + 1 --
//     26 accumulator tiles = 104 registers, every MFMA row live (round 3 gave each wave 7 of 28 tile rows: 7 % of the
//     MFMAs computed rows 416 .. 447, which do not exist), so two workgroups share a CU and cover each other's barrier /
//     DMA waits (the attention kernels' phase skew between the two was tried here: no effect, profiles/r04_time_wgrad.txt);
//   * both operand tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no VALU), exactly as
//     they lie in memory ([16 m][416 n] and [16 m][64 k], rows contiguous), double buffered, one barrier per 16 rows
//     of M; per 4-row MFMA step a wave reads 13 + 2 operand dwords from LDS for 26 MFMAs;
//   * the bias gradient (column sums of g) is accumulated by the workgroups of K column 0 only (a workgroup-uniform
//     specialisation of the stage loop: 13 VALU adds per step there, none in the other columns);
//   * M is split over workgroups; the workgroups of one M slice (all K columns) are placed on the same XCD, so g is
//     read from HBM once per slice and from that XCD's L2 by the others.
// Partials [split][N][K] (+ [split][N] for the bias gradient) are summed by backward.hip's wgrad_reduce_kernel.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WN = 416;                  // rows of an N chunk
constexpr int WTA = 13;                  // tile rows (16 n) per wave: 2 wave rows x 13 x 16 = 416
constexpr int WTB = 2;                   // column tiles (16 k) per wave: 2 wave columns x 2 x 16 = 64
constexpr int WBM = 16;                  // rows of M per stage
constexpr int WGS = WBM * WN;            // floats of a g tile (26 KB = 26 DMA chunks of 1 KB)

struct W16Args {
  const float* g; int64_t ldg;
  const float* x; int64_t ldx;
  int M, N, K;
  int m_per_split, splits;               // m_per_split: a multiple of 16
  int ncol;                              // K column blocks of 64 (the last one 32 wide when K % 64 == 32)
  int nchunk;                            // N / 416
  float* part; float* part_b;
  int relu_x;
};

__device__ __forceinline__ unsigned lds_addr_w(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
// 1 KB, global -> LDS: scalar base + per-lane byte offset (precomputed once: no VALU in the stage loop)
__device__ __forceinline__ void dma_w(const float* __restrict__ base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ void dma_wait_w() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// NTB k tiles of 16 per workgroup column: 4 (64 columns) or 2 (the 32-wide remainder column)
template <int NTB, bool RELU, bool BIAS>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const W16Args a) {
  constexpr int XROW = 16 * NTB;                       // floats per x tile row
  __shared__ __attribute__((aligned(16))) float gs0[WGS];
  __shared__ __attribute__((aligned(16))) float gs1[WGS];
  __shared__ __attribute__((aligned(16))) float xs0[WBM * XROW];
  __shared__ __attribute__((aligned(16))) float xs1[WBM * XROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int nh = wave >> 1, kh = wave & 1;
  // ---- which (M slice, N chunk, K column): workgroup b runs on XCD b % 8; the ncol x nchunk workgroups of a slice
  // are consecutive multiples of 8 apart, i.e. on one XCD
  const int cc_n = a.ncol * a.nchunk;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int z = (j / cc_n) * 8 + xcd;
  if (z >= a.splits) return;
  const int cc = j % cc_n;
  const int chunk = cc / a.ncol, col = cc % a.ncol;
  const int n0 = chunk * WN;
  // the last column of a K that is not a multiple of 64 is shifted back to end at K: it recomputes (and does not
  // store) the columns it shares with its neighbour -- one uniform launch instead of a remainder launch behind it
  const int k0 = min(col * 64, a.K - 64), kskip = col * 64 - k0;
  const int m_begin = z * a.m_per_split;
  const int m_end = min(a.M, m_begin + a.m_per_split);          // (a.M: the rows in full 16-row tiles)
  const int n_stage = (m_end - m_begin) / WBM;
  float* P = a.part + (int64_t)z * a.N * a.K;
  if (n_stage <= 0) {                                   // (an empty slice still owns its partial: zeros)
    for (int e = tid; e < WN * XROW; e += 256)
      if (e % XROW >= kskip) P[(int64_t)(n0 + e / XROW) * a.K + k0 + e % XROW] = 0.f;
    if (BIAS && col == 0)
      for (int e = tid; e < WN; e += 256) a.part_b[(int64_t)z * a.N + n0 + e] = 0.f;
    return;
  }

  // ---- DMA plan.  A stage is 26 chunks (1 KB) of the g tile + 4 of the x tile; the slots of a wave are the SAME kind
  // in every wave (no per-slot branch): slots 0-5 = g chunks wave + 4 s, slot 6 = x chunk `wave`, slot 7 = g chunks
  // 24, 25 (waves 0, 1 only).  Lane offsets relative to the stage's first row are constant over the stages.
  // Only FULL 16-row stages run here (the host gives the rows behind the last full tile to tail_partial_kernel).
  // Everything that is uniform over a workgroup's life (relu, bias, the DMA slot kinds) is a template / by-construction
  // decision: a taken or not-taken scalar branch costs 24-34 cycles (profiles/micro/chain_latency.hip), and the first
  // version of this loop had ~28 of them per stage.
  static_assert(NTB == 4, "the DMA plan below is written for 64-column x tiles");
  constexpr int GSL = 6;
  unsigned voff[GSL + 2];
#pragma unroll
  for (int s = 0; s < GSL + 2; ++s) {
    const int c = s < GSL ? wave + 4 * s : (s == GSL ? wave : 24 + (wave & 1));
    const int flat = 256 * c + 4 * lane;                                  // float index inside the tile
    if (s != GSL) voff[s] = (unsigned)(((int64_t)(flat / WN) * a.ldg + flat % WN) * 4);
    else voff[s] = (unsigned)(((int64_t)(flat / XROW) * a.ldx + flat % XROW) * 4);
  }
  const float* gbase = a.g + (int64_t)m_begin * a.ldg + n0;
  const float* xbase = a.x + (int64_t)m_begin * a.ldx + k0;
  const int64_t gstep = (int64_t)WBM * a.ldg, xstep = (int64_t)WBM * a.ldx;
  const bool extra = wave < 2;
  const unsigned w1k = (unsigned)wave * 1024u;
  // slot `slot` of the stage whose first rows are at (gsrc, xsrc), into the buffers (gdst, xdst)
  auto dma_slot = [&](int slot, const float* gsrc, const float* xsrc, float* gdst, float* xdst) __attribute__((always_inline)) {
    if (slot < GSL) dma_w(gsrc, voff[slot], lds_addr_w(gdst) + w1k + (unsigned)slot * 4096u);
    else if (slot == GSL) dma_w(xsrc, voff[GSL], lds_addr_w(xdst) + w1k);
    else if (extra) dma_w(gsrc, voff[GSL + 1], lds_addr_w(gdst) + 24u * 1024u + w1k);
  };

  f32x4 acc[WTA][WTB];
#pragma unroll
  for (int ta = 0; ta < WTA; ++ta)
#pragma unroll
    for (int tb = 0; tb < WTB; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[WTA];
#pragma unroll
  for (int ta = 0; ta < WTA; ++ta) bsum[ta] = 0.f;

#pragma unroll
  for (int sl = 0; sl < GSL + 2; ++sl) dma_slot(sl, gbase, xbase, gs0, xs0);
  dma_wait_w();
  __syncthreads();

  // one stage = 4 MFMA steps of 4 rows of M: per step 13 A dwords (g, this wave's tile rows) + 2 B dwords (x) from
  // LDS, 26 MFMAs; the operands of step s + 1 are read before the MFMAs of step s (fenced)
  // (the next stage's DMA slots are issued INSIDE the step, between MFMAs: a burst of them at the top of a stage is
  // ~100 scalar instructions with the matrix pipe idle)
  auto stage = [&](auto with_bias, const float* __restrict__ gt, const float* __restrict__ xt, const float* gsrc,
                   const float* xsrc, float* gdst, float* xdst) __attribute__((always_inline)) {
    constexpr bool SUMS = decltype(with_bias)::value;
    const float* ga = gt + kq * WN + 16 * WTA * nh + li;
    const float* xb = xt + kq * XROW + 16 * WTB * kh + li;
    float av[2][WTA], bv[2][WTB];                         // operand registers of the current / next step (no copies)
#pragma unroll
    for (int ta = 0; ta < WTA; ++ta) av[0][ta] = ga[16 * ta];
#pragma unroll
    for (int tb = 0; tb < WTB; ++tb) bv[0][tb] = xb[16 * tb];
#pragma unroll
    for (int s = 0; s < WBM / 4; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < WBM / 4) {
#pragma unroll
        for (int ta = 0; ta < WTA; ++ta) av[nxt][ta] = ga[(4 * (s + 1)) * WN + 16 * ta];
#pragma unroll
        for (int tb = 0; tb < WTB; ++tb) bv[nxt][tb] = xb[(4 * (s + 1)) * XROW + 16 * tb];
      }
      if (RELU) {
#pragma unroll
        for (int tb = 0; tb < WTB; ++tb) bv[cur][tb] = fmaxf(bv[cur][tb], 0.f);
      }
      if (SUMS) {
#pragma unroll
        for (int ta = 0; ta < WTA; ++ta) bsum[ta] += av[cur][ta];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ta = 0; ta < WTA; ++ta) {
        // the next stage's 8 DMA slots in the first two steps (four each, three MFMA rows apart): the last of them
        // has two steps of MFMAs to land before the stage's closing wait
        if (s < 2 && ta % 3 == 1) dma_slot(4 * s + ta / 3, gsrc, xsrc, gdst, xdst);
#pragma unroll
        for (int tb = 0; tb < WTB; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][ta], bv[cur][tb], acc[ta][tb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  asm volatile("; OCC4D_MARK loop");
  // (the last stage has no successor: it re-loads itself into the idle buffer instead of branching around the DMA)
  auto run = [&](auto with_bias) __attribute__((always_inline)) {
    const float* gnext = gbase;
    const float* xnext = xbase;
#pragma clang loop unroll(disable)
    for (int st = 0; st < n_stage; st += 2) {
      if (st + 1 < n_stage) { gnext += gstep; xnext += xstep; }
      stage(with_bias, gs0, xs0, gnext, xnext, gs1, xs1);
      dma_wait_w();
      __syncthreads();
      if (st + 1 >= n_stage) break;
      if (st + 2 < n_stage) { gnext += gstep; xnext += xstep; }
      stage(with_bias, gs1, xs1, gnext, xnext, gs0, xs0);
      dma_wait_w();
      __syncthreads();
    }
  };
  if (BIAS && col == 0) run(std::true_type{});           // (workgroup-uniform: one scalar branch per workgroup)
  else run(std::false_type{});

  asm volatile("; OCC4D_MARK epilogue");
  // ---- partial tile to memory: C/D lane (column j = li, rows 4 kq + reg)
#pragma unroll
  for (int ta = 0; ta < WTA; ++ta) {
    const int nrow = 16 * (WTA * nh + ta) + 4 * kq;
#pragma unroll
    for (int tb = 0; tb < WTB; ++tb) {
      const int kc = 16 * (WTB * kh + tb);
      if (kc < kskip) continue;                           // (columns the neighbouring block owns)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(int64_t)(n0 + nrow + r) * a.K + k0 + kc + li] = acc[ta][tb][r];
    }
  }
  if (BIAS && col == 0 && kh == 0) {
    // bias gradient: this lane summed g[m][n] over its m residue class kq; add the four classes (lanes li + 16 kq)
#pragma unroll
    for (int ta = 0; ta < WTA; ++ta) {
      float v = bsum[ta];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int n = 16 * (WTA * nh + ta) + li;
      if (kq == 0) a.part_b[(int64_t)z * a.N + n0 + n] = v;
    }
  }
}


// the (< 16) rows behind the last full tile: one more partial, P[n][k] = sum_m g[m][n] [relu](x)[m][k], one thread per
// element (+ the bias partial)
__global__ __launch_bounds__(256) void tail_partial_kernel(const float* __restrict__ g, int64_t ldg,
                                                           const float* __restrict__ x, int64_t ldx, int m0, int m1, int N,
                                                           int K, int relu_x, float* __restrict__ P, float* __restrict__ Pb) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)N * K) return;
  const int k = (int)(e % K), n = (int)(e / K);
  float s = 0.f, sb = 0.f;
  for (int m = m0; m < m1; ++m) {
    const float gv = g[(int64_t)m * ldg + n];
    float xv = x[(int64_t)m * ldx + k];
    if (relu_x) xv = fmaxf(xv, 0.f);
    s += gv * xv;
    sb += gv;
  }
  P[e] = s;
  if (Pb && k == 0) Pb[n] = sb;
}

}  // namespace

namespace occ4d {

// Shapes the kernel takes.  splits: M slices (a multiple of 8: one XCD per residue class); m_per_split: rows per
// slice (a multiple of 16).
bool wgrad16_plan(int M, int N, int K, int* splits, int* m_per_split) {
  if (N % WN != 0 || K % 32 != 0 || K < 64 || M < 4096) return false;
  const int ncol = (K + 63) / 64, nchunk = N / WN, cc = ncol * nchunk;
  // Workgroups all take the same time and 512 run at once (2 per CU): the M split is chosen so that their number fills
  // whole rounds of 512 (a split that gives 1040 workgroups runs THREE rounds for 2.03 rounds of work: measured 0.60 of
  // the MFMA peak against 0.7 with 936).  Among the multiples of 8 up to M / 512: best fill, at most `rounds` rounds.
  static const int max_rounds = [] { const char* e = getenv("OCC4D_W16_ROUNDS"); return e ? atoi(e) : 3; }();
  const int cap = M / 512 > 8 ? M / 512 : 8;
  int s = 8;
  double best = -1.0;
  for (int cand = 8; cand <= cap; cand += 8) {
    const int wgs = cand * cc, rounds = (wgs + 511) / 512;
    if (rounds > max_rounds) break;
    const double fill = (double)wgs / (512.0 * rounds);
    if (fill > best + 1e-9) { best = fill; s = cand; }
  }
  int mps = cdiv(cdiv(M, s), WBM) * WBM;
  *splits = s;
  *m_per_split = mps;
  return true;
}

int wgrad16_launch(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K, int splits,
                   int m_per_split, float* part, float* part_b, int relu_x, hipStream_t st) {
  OCC4D_REQUIRE((int64_t)WBM * ldg * 4 < ((int64_t)1 << 31) && (int64_t)WBM * ldx * 4 < ((int64_t)1 << 31),
                "occ4d_linear_wgrad: row strides too large for 32-bit tile offsets");
  const int nchunk = N / WN;
  const int m_full = M / WBM * WBM;
  // partial `splits` (the last of the splits + 1): the rows behind the last full 16-row tile
  tail_partial_kernel<<<cdiv((int64_t)N * K, 256), 256, 0, st>>>(g, ldg, x, ldx, m_full, M, N, K, relu_x,
                                                                  part + (int64_t)splits * N * K,
                                                                  part_b ? part_b + (int64_t)splits * N : nullptr);
  const int ncol = (K + 63) / 64;
  W16Args a{g, ldg, x, ldx, m_full, N, K, m_per_split, splits, ncol, nchunk, part, part_b, relu_x};
  const int grid = (splits / 8) * ncol * nchunk * 8;
  if (relu_x && part_b) wgrad16_kernel<4, true, true><<<grid, 256, 0, st>>>(a);
  else if (relu_x) wgrad16_kernel<4, true, false><<<grid, 256, 0, st>>>(a);
  else if (part_b) wgrad16_kernel<4, false, true><<<grid, 256, 0, st>>>(a);
  else wgrad16_kernel<4, false, false><<<grid, 256, 0, st>>>(a);
  return check_launch("occ4d_linear_wgrad(16x16x4)");
}

}  // namespace occ4d
