// Backward-pass kernels (training step of BASELINE config 5).
// The reference trains through torch autograd over ATen ops (train.py:101-118); these are the
// native gradients of the ops in occ4d.h.  GEMM-shaped work (weight gradients) runs on the fp32
// MFMA; everything else is HBM / L2-bound element-wise, gather or scatter work.  Scatter
// reductions use fp32 atomicAdd (order-dependent rounding, ~1e-7 relative).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;
inline dim3 grid1d(int64_t total) { return dim3((unsigned)((total + TPB - 1) / TPB)); }

// ------------------------------------------------------------------------------------------
// dW[n][k] = sum_m g[m][n] * x[m][k]      (weight gradient of y = x W^T), split over m.
// Workgroup = 128 n-rows x 32*NT k-columns of dW for one m-chunk; g and x tiles ([16 m] x cols,
// exactly as they lie in memory) go through LDS; the MFMA A operand is read TRANSPOSED from the
// g tile (A[i=n][kk=m] = gs[m][n]).  Partials [split][N][K] are summed by wgrad_reduce_kernel.
// Fused: the bias gradient db[n] = sum_m g[m][n] (column sums of the g tile already staged in LDS; k-block 0
// only) and relu on the x operand (weight gradient of a relu_in Linear) -- both were separate HBM passes.
// ------------------------------------------------------------------------------------------
constexpr int WG_BM = 16;   // contraction (m) depth per tile

template <int NT>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const float* __restrict__ g, int64_t ldg,
                                                       const float* __restrict__ x, int64_t ldx, int M, int N,
                                                       int K, int m_per_split, float* __restrict__ part,
                                                       float* __restrict__ part_b, int relu_x) {
  constexpr int BNn = 128, BKk = 32 * NT;
  constexpr int LDG = BNn + 4, LDX = BKk + 4;
  __shared__ __attribute__((aligned(16))) float gs[2][WG_BM * LDG];
  constexpr int XL_ = (WG_BM * BKk / 4 + 255) / 256;
  constexpr int XROWS = (XL_ * 256 + BKk / 4 - 1) / (BKk / 4);   // >= WG_BM: every thread stores all its float4 (no branch)
  __shared__ __attribute__((aligned(16))) float xs[2][XROWS * LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BNn, k0 = blockIdx.y * BKk;
  const int m_begin = blockIdx.z * m_per_split, m_end = min(M, m_begin + m_per_split);

  constexpr int GL = (WG_BM * BNn / 4) / 256;                 // float4 per thread, g tile (= 2)
  constexpr int XL = (WG_BM * BKk / 4 + 255) / 256;           // float4 per thread, x tile
  f32x4 rg[GL], rx[XL];
  // unconditional clamped loads + select: the m-tile body below is one scheduling region (see linear.hip)
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int f = tid + 256 * i;
      const int mm = m0 + f / (BNn / 4), n = n0 + 4 * (f % (BNn / 4));
      const bool ok = mm < m_end && n < N;
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + (int64_t)min(mm, M - 1) * ldg + min(n, N - 4));
      rg[i].x = ok ? v.x : 0.f; rg[i].y = ok ? v.y : 0.f; rg[i].z = ok ? v.z : 0.f; rg[i].w = ok ? v.w : 0.f;
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int f = tid + 256 * i;
      const int mm = m0 + f / (BKk / 4), k = k0 + 4 * (f % (BKk / 4));
      const bool ok = f < WG_BM * BKk / 4 && mm < m_end && k < K;
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)min(mm, M - 1) * ldx + min(k, K - 4));
      if (relu_x) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      rx[i].x = ok ? v.x : 0.f; rx[i].y = ok ? v.y : 0.f; rx[i].z = ok ? v.z : 0.f; rx[i].w = ok ? v.w : 0.f;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&gs[buf][(f / (BNn / 4)) * LDG + 4 * (f % (BNn / 4))]) = rg[i];
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4*>(&xs[buf][(f / (BKk / 4)) * LDX + 4 * (f % (BKk / 4))]) = rx[i];
    }
  };
  f32x16 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const int nt = (m_end - m_begin + WG_BM - 1) / WG_BM;
  if (nt > 0) {
    gload(m_begin);
    sstore(0);
  }
  __syncthreads();
  const int col = lane & 31, kh = lane >> 5;
  const bool do_bias = part_b != nullptr && blockIdx.y == 0 && tid < BNn;
  float bsum = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    gload(t + 1 < nt ? m_begin + (t + 1) * WG_BM : m_begin);               // always (last: harmless re-load)
#pragma unroll
    for (int mm = 0; mm < WG_BM; ++mm) bsum += gs[buf][mm * LDG + (tid & (BNn - 1))];   // rows past m_end are zero-filled
#pragma unroll
    for (int s = 0; s < WG_BM / 2; ++s) {
      const float av = gs[buf][(2 * s + kh) * LDG + wave * 32 + col];      // A[i = n][kk = m]
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const float bv = xs[buf][(2 * s + kh) * LDX + 32 * c + col];       // B[kk = m][j = k]
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
      }
    }
    sstore(buf ^ 1);
    {   // prefetch loads in the shadow of the first MFMAs, LDS stores in the shadow of the last (see linear.hip)
      constexpr int NLD = GL + XL, NMFMA = (WG_BM / 2) * NT;
      constexpr int PER = NMFMA / (2 * NLD) >= 3 ? 3 : (NMFMA / (2 * NLD) >= 1 ? NMFMA / (2 * NLD) : 0);
      if (PER > 0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 2 * NLD * PER, 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    }
    __syncthreads();
  }
  if (do_bias && n0 + tid < N) part_b[(int64_t)blockIdx.z * N + n0 + tid] = bsum;
  float* P = part + (int64_t)blockIdx.z * N * K;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (n >= N) continue;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int k = k0 + 32 * c + col;
      if (k < K) P[(int64_t)n * K + k] = acc[c][r];
    }
  }
}

__global__ __launch_bounds__(TPB) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int64_t nk,
                                                           float* __restrict__ dw, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= nk) return;
  float s = accumulate ? dw[e] : 0.f;
  int z = 0;
  for (; z + 8 <= splits; z += 8) {            // (eight independent loads in flight: a serial loop is one HBM latency per partial)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(z + u) * nk + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z < splits; ++z) s += part[(int64_t)z * nk + e];
  dw[e] = s;
}
// The same sum for SHORT vectors (the bias gradient: N = 416 .. 832 elements, up to 113 partials): with one thread per
// element the launch is two workgroups walking 113 dependent-latency loads each (measured 28 us average, 121 us at 113
// partials, 70 launches per training step).  Here 64 elements per workgroup x 4 waves, wave w takes the partials
// z = w (mod 4) with 8 loads in flight; the four sums meet in LDS and are added in a fixed order (deterministic).
__global__ __launch_bounds__(TPB) void short_reduce_kernel(const float* __restrict__ part, int splits, int nk,
                                                           float* __restrict__ dw, int accumulate) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < nk) {
    int z = w;
    for (; z + 28 < splits; z += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(z + 4 * u) * nk + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; z += 4) s += part[(int64_t)z * nk + e];
  }
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && e < nk) {
    float t = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    if (accumulate) t += dw[e];
    dw[e] = t;
  }
}
static void launch_short_reduce(const float* part, int splits, int nk, float* dw, int accumulate, hipStream_t st) {
  short_reduce_kernel<<<(nk + 63) / 64, TPB, 0, st>>>(part, splits, nk, dw, accumulate);
}
__global__ __launch_bounds__(TPB) void wgrad_reduce4_kernel(const float* __restrict__ part, int splits, int64_t nk4,
                                                            float* __restrict__ dw, int accumulate) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= nk4) return;
  const f4* p = reinterpret_cast<const f4*>(part) + e;
  f4* out = reinterpret_cast<f4*>(dw) + e;
  f4 s = accumulate ? *out : f4{0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(z + u) * nk4];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z < splits; ++z) s += p[(int64_t)z * nk4];
  *out = s;
}

// column sums: out[c] (+)= sum_i x[i][c]; stage 1 per (row chunk, 64-column strip), stage 2 reduce
// (deterministic).  Block = 64 columns x 4 row lanes: 256-byte coalesced row segments, 4 rows in flight.
__global__ __launch_bounds__(TPB) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx, int n, int d,
                                                             int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < d)
    for (int i = r0 + rl; i < r1; i += 4) s += x[(int64_t)i * ldx + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < d) part[(int64_t)blockIdx.y * d + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// out = ref > 0 ? g : 0
__global__ __launch_bounds__(TPB) void relu_mask_kernel(const float* __restrict__ g, int64_t ldg,
                                                        const float* __restrict__ ref, int64_t ldr, int64_t total,
                                                        int d, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  out[i * ldo + c] = ref[i * ldr + c] > 0.f ? g[i * ldg + c] : 0.f;
}

// swish (model/implicit.py:46-64): y = x * sigmoid(x); dx = g * (s + x s (1 - s)), s = sigmoid(x) -- the linear kernel's
// own expression (csrc/linear.hip swish1), so that the training forward equals the inference forward
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + __expf(-v)); }
__global__ __launch_bounds__(TPB) void swish_kernel(const float* __restrict__ x, int64_t ldx, int64_t total, int d,
                                                    float* __restrict__ y, int64_t ldy) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  const float v = x[i * ldx + c];
  y[i * ldy + c] = v * sigmoid1(v);
}
__global__ __launch_bounds__(TPB) void swish_bwd_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ x,
                                                        int64_t ldx, int64_t total, int d, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  const float v = x[i * ldx + c], sg = sigmoid1(v);
  out[i * ldo + c] = g[i * ldg + c] * (sg + v * sg * (1.0f - sg));
}

// out[idx[i]][c] += scale * src[i][c]
__global__ __launch_bounds__(TPB) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                               const int32_t* __restrict__ idx, int64_t total, int d,
                                                               float scale, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  atomicAdd(out + (int64_t)idx[i] * ldo + c, scale * src[i * lds + c]);
}

// Deterministic counterpart of the atomic scatters (training.DETERMINISTIC / ops.DETERMINISTIC): the (source, target)
// pairs come pre-sorted by target (stable: original order inside a segment), one thread per (target row, channel) adds
// its segment in that fixed order:
//   out[r][c] = scale * sum_{t in [off[r], off[r + 1])} (w ? w[order[t]] : 1) * src[(order[t] / div) * lds + c]
__global__ __launch_bounds__(TPB) void segment_gather_sum_kernel(const float* __restrict__ src, int64_t lds,
                                                                 const int32_t* __restrict__ order,
                                                                 const int32_t* __restrict__ off,
                                                                 const float* __restrict__ w, int div, int64_t total, int d,
                                                                 float scale, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t r = e / d;
  float s = 0.f;
  for (int t = off[r]; t < off[r + 1]; ++t) {
    const int p = order[t];
    const float v = src[(int64_t)(p / div) * lds + c];
    s += w ? w[p] * v : v;
  }
  out[r * ldo + c] = scale * s;
}

// The same reduction for SPEED (not bit-reproducible): every target row's segment is cut into `parts` contiguous slices,
// one thread per (row, slice, channel quad) sums its slice with four independent float4 loads in flight and adds the
// result to the zero-initialised output with ONE atomic per element and slice -- parts x n_out x d atomics instead of one
// per (pair, channel), and slices instead of whole segments as the unit of work (segments are very uneven: an abstract
// point near many supervision queries collects thousands of pairs, most collect tens).
__global__ __launch_bounds__(TPB) void segment_sum_sorted_kernel(const float* __restrict__ src, int64_t lds,
                                                                 const int32_t* __restrict__ order,
                                                                 const int32_t* __restrict__ off, int64_t total, int d4,
                                                                 int parts, float scale, float* __restrict__ out,
                                                                 int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c4 = (int)(e % d4);
  const int64_t rp = e / d4;
  const int part = (int)(rp % parts);
  const int64_t r = rp / parts;
  const int lo = off[r], hi = off[r + 1];
  const int chunk = (hi - lo + parts - 1) / parts;
  const int a = lo + part * chunk, b = min(hi, a + chunk);
  if (a >= b) return;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const float* base = src + 4 * c4;
  f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int t = a;
  for (; t + 4 <= b; t += 4) {
    const int p0 = order[t], p1 = order[t + 1], p2 = order[t + 2], p3 = order[t + 3];
    const f4 v0 = *reinterpret_cast<const f4*>(base + (int64_t)p0 * lds);
    const f4 v1 = *reinterpret_cast<const f4*>(base + (int64_t)p1 * lds);
    const f4 v2 = *reinterpret_cast<const f4*>(base + (int64_t)p2 * lds);
    const f4 v3 = *reinterpret_cast<const f4*>(base + (int64_t)p3 * lds);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; t < b; ++t) s0 += *reinterpret_cast<const f4*>(base + (int64_t)order[t] * lds);
  const f4 s = (s0 + s1) + (s2 + s3);
  float* o = out + r * ldo + 4 * c4;
  atomicAdd(o + 0, scale * s.x);
  atomicAdd(o + 1, scale * s.y);
  atomicAdd(o + 2, scale * s.z);
  atomicAdd(o + 3, scale * s.w);
}

// Segments of a scatter (pair p -> target row idx[p]) by a multi-block counting sort, kernels only (capturable; torch.sort
// inside a captured graph clears its digit counters with memset nodes that this stack does not replay):
//   count: block b histograms its contiguous slice of the pairs in LDS            -> bh[b][row]
//   base : per row, exclusive prefix of bh[.][row] over the blocks                -> bh[b][row] = block b's base in the row; tot[row]
//   scan : exclusive prefix of tot over the rows (one workgroup)                  -> off[0 .. n_out]
//   fill : block b replays its slice: order[off[row] + bh[b][row] + (arrival rank in LDS)] = p
// Within a (block, row) pair the order is the LDS atomics' arrival order: a valid segmentation for sums whose order
// does not matter (the deterministic mode keeps its stable sort).
constexpr int SEG_BLOCKS = 128;
constexpr int SEG_MAX_ROWS = 16384;          // 64 KB of LDS counters
__global__ __launch_bounds__(TPB) void seg_count_kernel(const int32_t* __restrict__ idx, int64_t n, int n_out,
                                                        int32_t* __restrict__ bh) {
  extern __shared__ int s_cnt[];
  for (int r = threadIdx.x; r < n_out; r += TPB) s_cnt[r] = 0;
  __syncthreads();
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = min(n, lo + per);
  for (int64_t p = lo + threadIdx.x; p < hi; p += TPB) atomicAdd(&s_cnt[idx[p]], 1);
  __syncthreads();
  for (int r = threadIdx.x; r < n_out; r += TPB) bh[(int64_t)blockIdx.x * n_out + r] = s_cnt[r];
}
__global__ __launch_bounds__(TPB) void seg_base_kernel(int32_t* __restrict__ bh, int n_blocks, int n_out,
                                                       int32_t* __restrict__ tot) {
  const int r = blockIdx.x * TPB + threadIdx.x;
  if (r >= n_out) return;
  int run = 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int c = bh[(int64_t)b * n_out + r];
    bh[(int64_t)b * n_out + r] = run;
    run += c;
  }
  tot[r] = run;
}
__global__ __launch_bounds__(1024) void seg_scan_kernel(const int32_t* __restrict__ tot, int n_out, int32_t* __restrict__ off) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  const int per = (n_out + 1023) / 1024;
  const int lo = min(n_out, t * per), hi = min(n_out, lo + per);
  int sum = 0;
  for (int r = lo; r < hi; ++r) sum += tot[r];
  s_part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                 // inclusive Hillis-Steele scan of the thread sums
    const int v = t >= o ? s_part[t - o] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - sum;
  for (int r = lo; r < hi; ++r) {
    off[r] = run;
    run += tot[r];
  }
  if (t == 1023) off[n_out] = s_part[1023];
}
__global__ __launch_bounds__(TPB) void seg_fill_kernel(const int32_t* __restrict__ idx, int64_t n, int n_out,
                                                       const int32_t* __restrict__ bh, const int32_t* __restrict__ off,
                                                       int32_t* __restrict__ order) {
  extern __shared__ int s_cnt[];
  for (int r = threadIdx.x; r < n_out; r += TPB) s_cnt[r] = off[r] + bh[(int64_t)blockIdx.x * n_out + r];
  __syncthreads();
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = min(n, lo + per);
  for (int64_t p = lo + threadIdx.x; p < hi; p += TPB) order[atomicAdd(&s_cnt[idx[p]], 1)] = (int32_t)p;
}

// pos_hidden_bwd with the block partials written out (blocks x 4 slices x h x 4 floats) instead of atomics ...
__global__ __launch_bounds__(TPB) void pos_hidden_bwd_partials_kernel(const float* __restrict__ pos, int64_t ps,
                                                                      const float* __restrict__ pos2, int64_t p2s,
                                                                      const int32_t* __restrict__ idx, int64_t npairs, int k,
                                                                      int h, const float* __restrict__ r,
                                                                      const float* __restrict__ gr, float* __restrict__ ws) {
  const int m = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float ax = 0.f, ay = 0.f, az = 0.f, ac = 0.f;
  if (m < h) {
    for (int64_t p = (int64_t)blockIdx.x * 4 + sl; p < npairs; p += (int64_t)gridDim.x * 4) {
      const float gv = r[p * h + m] > 0.f ? gr[p * h + m] : 0.f;
      const float* a = pos + (p / k) * ps;
      const float* b = pos2 + (int64_t)idx[p] * p2s;
      ax += gv * (a[0] - b[0]); ay += gv * (a[1] - b[1]); az += gv * (a[2] - b[2]); ac += gv;
    }
    float* o = ws + (((int64_t)blockIdx.x * 4 + sl) * h + m) * 4;
    o[0] = ax; o[1] = ay; o[2] = az; o[3] = ac;
  }
}
// ... and added up in a fixed order: thread = (hidden unit, component)
__global__ void pos_hidden_bwd_reduce_kernel(const float* __restrict__ ws, int parts, int h, float* __restrict__ dP1,
                                             float* __restrict__ dc1) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 4 * h) return;
  const int m = e >> 2, comp = e & 3;
  float s = 0.f;
  for (int q = 0; q < parts; ++q) s += ws[((int64_t)q * h + m) * 4 + comp];
  if (comp < 3) dP1[3 * m + comp] = s;
  else dc1[m] = s;
}

// out[i][c] = sum_{j<k} src[i*k + j][c]
__global__ __launch_bounds__(TPB) void segment_sum_kernel(const float* __restrict__ src, int64_t total, int k, int d,
                                                          float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float s = 0.f;
  for (int j = 0; j < k; ++j) s += src[(i * k + j) * d + c];
  out[i * ldo + c] = s;
}
// the same on 16-byte lanes, the k <= 16 loads of a thread independent (the scalar kernel above: 3.5 TB/s on the 1.5 GB
// pair gradient of a training chunk)
__global__ __launch_bounds__(TPB) void segment_sum4_kernel(const float* __restrict__ src, int total4, int k, int d4,
                                                           float* __restrict__ out, int64_t ldo) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= total4) return;
  const int i = e / d4, c = e - i * d4;
  const f4* p = reinterpret_cast<const f4*>(src) + (int64_t)i * k * d4 + c;
  f4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = j < k ? p[(int64_t)j * d4] : f4{0.f, 0.f, 0.f, 0.f};
  f4 s = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) s += v[j];
  *reinterpret_cast<f4*>(out + (int64_t)i * ldo + 4 * c) = s;
}

// max pool backward: dy[idx[i][j*]][c] += dz[i][c], j* = first argmax_j y[idx[i][j]][c]
__global__ __launch_bounds__(TPB) void maxpool_bwd_kernel(const float* __restrict__ y, int64_t ldy,
                                                          const int32_t* __restrict__ idx, int64_t total, int k, int d,
                                                          const float* __restrict__ dz, int64_t ldz,
                                                          float* __restrict__ dy, int64_t ldd) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  int best = idx[i * k];
  float m = y[(int64_t)best * ldy + c];
  for (int j = 1; j < k; ++j) {
    const int r = idx[i * k + j];
    const float v = y[(int64_t)r * ldy + c];
    if (v > m) { m = v; best = r; }
  }
  atomicAdd(dy + (int64_t)best * ldd + c, dz[i * ldz + c]);
}

// LayerNorm backward, one wave per row: y = (x - mean) * rstd * gamma + beta;  g = dL/dy (already relu-masked)
// dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)), gh = g * gamma;  dgamma += g * xhat;  dbeta += g (atomics)
__global__ __launch_bounds__(TPB) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ g, int64_t ldg, float eps, int n,
                                                            int d, float* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // A wave walks rows with a grid stride and keeps its dgamma / dbeta contributions in registers (lane owns channels
  // lane + 64 i, i < LNC): one atomic per channel and wave at the end.  (One atomic per channel and ROW put n
  // serialised L2 atomics on each of the 2 d addresses: they were the kernel's time.)  d > 64 LNC: per-row atomics.
  constexpr int LNC = 8;
  const int wave0 = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (TPB / 64);
  const int lane = threadIdx.x & 63;
  const bool in_regs = d <= 64 * LNC;
  float ag[LNC], ab[LNC];
#pragma unroll
  for (int i = 0; i < LNC; ++i) ag[i] = ab[i] = 0.f;
  for (int row = wave0; row < n; row += nwaves) {
    const float* xr = x + (int64_t)row * ldx;
    const float* gr = g + (int64_t)row * ldg;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; q += t * t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)d + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float gh = gr[c] * (gamma ? gamma[c] : 1.f);
      const float xh = (xr[c] - mean) * rstd;
      a += gh; b += gh * xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    a /= (float)d; b /= (float)d;
    float* dr = dx + (int64_t)row * lddx;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < LNC; ++i) {
        const int c = lane + 64 * i;
        if (c < d) {
          const float xh = (xr[c] - mean) * rstd;
          const float gh = gr[c] * (gamma ? gamma[c] : 1.f);
          dr[c] = rstd * (gh - a - xh * b);
          ag[i] += gr[c] * xh;
          ab[i] += gr[c];
        }
      }
    } else {
      for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        const float gh = gr[c] * (gamma ? gamma[c] : 1.f);
        dr[c] = rstd * (gh - a - xh * b);
        if (dgamma) { atomicAdd(dgamma + c, gr[c] * xh); atomicAdd(dbeta + c, gr[c]); }
      }
    }
  }
  if (in_regs && dgamma) {
#pragma unroll
    for (int i = 0; i < LNC; ++i) {
      const int c = lane + 64 * i;
      if (c < d) { atomicAdd(dgamma + c, ag[i]); atomicAdd(dbeta + c, ab[i]); }
    }
  }
}

// softmax-aggregate backward (per (i, c)): a_j = softmax_j(l_j / div), val_j = v[idx_j] + pe_j, agg = sum a_j val_j
// dval_j = a_j dagg ; da_j = val_j dagg ; dl_j = a_j (da_j - sum_t a_t da_t) / div
template <int KMAX>
__global__ __launch_bounds__(TPB) void softmax_agg_bwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ v, int64_t ldv,
                                                              const float* __restrict__ pe,
                                                              const int32_t* __restrict__ idx, int64_t total, int k,
                                                              int d, float divisor, const float* __restrict__ dagg,
                                                              int64_t ldda, float* __restrict__ dlogits,
                                                              float* __restrict__ dpe, float* __restrict__ dv,
                                                              int64_t lddv) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float a[KMAX], val[KMAX];
  float mx = -__builtin_inff();
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < k) { a[j] = logits[(i * k + j) * d + c] / divisor; mx = fmaxf(mx, a[j]); }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < k) { a[j] = expf(a[j] - mx); den += a[j]; }
  const float go = dagg[i * ldda + c];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < k) {
      const int64_t p = i * k + j;
      a[j] = a[j] / den;
      val[j] = v[(int64_t)idx[p] * ldv + c] + (pe ? pe[p * d + c] : 0.f);
      dot += a[j] * val[j] * go;
    }
#pragma unroll
  for (int j = 0; j < KMAX; ++j)
    if (j < k) {
      const int64_t p = i * k + j;
      const float dval = a[j] * go;
      dlogits[p * d + c] = a[j] * (val[j] * go - dot) / divisor;
      if (dpe) dpe[p * d + c] = dval;
      if (dv) atomicAdd(dv + (int64_t)idx[p] * lddv + c, dval);
    }
}

// The same for d % 4 == 0 and a compile-time neighbour count: a thread owns FOUR consecutive channels of one query, so
// every pair-tensor access is 16 bytes per lane (1 KB per wave instruction) and all 2 K + 1 loads of a thread are
// independent and issued before the first use: 5.3-5.5 TB/s on the 3 GB of a training chunk (32768 queries x 14
// neighbours x 416 channels, four pair tensors) against 2.1 TB/s for the scalar kernel above.  Without the value-table
// atomics: 16-byte lanes make them sparse (2.5 ms with, 0.57 ms without; profiles/r04_time_softmax_bwd.txt) -- the caller
// reduces the per-pair value gradients `dval` (= dpe) itself (sorted-segment sum, occ4d_segment_sum_sorted_f32).
// exp through v_exp_f32 on pre-scaled arguments (as the forward kernels do).
typedef float sm_f4 __attribute__((ext_vector_type(4)));
template <int K, bool HAS_PE>
__global__ __launch_bounds__(256) void softmax_agg_bwd4_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ v, int ldv,
                                                               const float* __restrict__ pe,
                                                               const int32_t* __restrict__ idx, int total4, int d4,
                                                               float inv_div, const float* __restrict__ dagg, int ldda,
                                                               float* __restrict__ dlogits, float* __restrict__ dval) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int i = e / d4, c = 4 * (e - i * d4);
  const int64_t row0 = (int64_t)i * K;                                 // first pair row of the query
  const int64_t ld = 4 * (int64_t)d4;
  const float* lp = logits + row0 * ld + c;
  sm_f4 l[K], val[K];
  int id[K];
#pragma unroll
  for (int j = 0; j < K; ++j) l[j] = *reinterpret_cast<const sm_f4*>(lp + j * ld);
#pragma unroll
  for (int j = 0; j < K; ++j) id[j] = idx[row0 + j];
  const sm_f4 go = *reinterpret_cast<const sm_f4*>(dagg + (int64_t)i * ldda + c);
  if (HAS_PE) {
    const float* pp = pe + row0 * ld + c;
#pragma unroll
    for (int j = 0; j < K; ++j) val[j] = *reinterpret_cast<const sm_f4*>(pp + j * ld);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const sm_f4 vv = *reinterpret_cast<const sm_f4*>(v + (int64_t)id[j] * ldv + c);
    val[j] = HAS_PE ? val[j] + vv : vv;
  }
  constexpr float LOG2E = 1.44269504088896f;
  const float sc = inv_div * LOG2E;
  sm_f4 mx = l[0];
#pragma unroll
  for (int j = 1; j < K; ++j) {
    mx.x = fmaxf(mx.x, l[j].x); mx.y = fmaxf(mx.y, l[j].y); mx.z = fmaxf(mx.z, l[j].z); mx.w = fmaxf(mx.w, l[j].w);
  }
  sm_f4 den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < K; ++j) {
    l[j].x = __builtin_amdgcn_exp2f((l[j].x - mx.x) * sc); l[j].y = __builtin_amdgcn_exp2f((l[j].y - mx.y) * sc);
    l[j].z = __builtin_amdgcn_exp2f((l[j].z - mx.z) * sc); l[j].w = __builtin_amdgcn_exp2f((l[j].w - mx.w) * sc);
    den += l[j];
  }
  const sm_f4 rden = {1.f / den.x, 1.f / den.y, 1.f / den.z, 1.f / den.w};
  sm_f4 dot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < K; ++j) {
    l[j] *= rden;                                                      // a_j
    val[j] *= go;                                                      // da_j = val_j dagg
    dot += l[j] * val[j];
  }
  float* dlp = dlogits + row0 * ld + c;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const sm_f4 dl = l[j] * (val[j] - dot) * inv_div;
    *reinterpret_cast<sm_f4*>(dlp + j * ld) = dl;
  }
  float* dpp = dval + row0 * ld + c;
#pragma unroll
  for (int j = 0; j < K; ++j) *reinterpret_cast<sm_f4*>(dpp + j * ld) = l[j] * go;
}

template <int K>
static void softmax_agg_bwd4_launch(const float* logits, const float* v, int ldv, const float* pe, const int32_t* idx,
                                    int total4, int d4, float inv_div, const float* dagg, int ldda, float* dlogits,
                                    float* dval, hipStream_t st) {
  const dim3 grid((total4 + 255) / 256);
  if (pe) softmax_agg_bwd4_kernel<K, true><<<grid, 256, 0, st>>>(logits, v, ldv, pe, idx, total4, d4, inv_div, dagg, ldda, dlogits, dval);
  else softmax_agg_bwd4_kernel<K, false><<<grid, 256, 0, st>>>(logits, v, ldv, pe, idx, total4, d4, inv_div, dagg, ldda, dlogits, dval);
}

// pos-MLP first layer backward: r = relu(P1 delta + c1); gr = dL/dr (n*k, h)
// dP1[m][:] += sum_p [r>0] gr[p][m] delta_p ; dc1[m] += sum_p [r>0] gr[p][m]      (block partials + atomics)
__global__ __launch_bounds__(TPB) void pos_hidden_bwd_kernel(const float* __restrict__ pos, int64_t ps,
                                                             const float* __restrict__ pos2, int64_t p2s,
                                                             const int32_t* __restrict__ idx, int64_t npairs, int k,
                                                             int h, const float* __restrict__ r,
                                                             const float* __restrict__ gr, float* __restrict__ dP1,
                                                             float* __restrict__ dc1) {
  // A 64-lane slice covers pp = 64 / h pairs at a time (lane = pair-in-slice * h + hidden unit: every lane busy for
  // h = 32, consecutive lanes read consecutive floats of r / gr); four slices per block, eight independent pairs in
  // flight per lane (the loop is a chain of dependent loads: idx -> pos2 row).  The block's partial sums meet in LDS and
  // ONE thread per (unit, component) adds them to the result: atomics on the same 4 h addresses serialise in L2
  // (~90 ns each -- with one atomic per lane they, not the memory traffic, were the kernel's time).
  __shared__ float s_acc[4 * 64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int pp = 64 / h;
  const int m = lane % h, sub = lane / h;
  float ax = 0.f, ay = 0.f, az = 0.f, ac = 0.f;
  for (int i = threadIdx.x; i < 4 * 64; i += TPB) s_acc[i] = 0.f;
  __syncthreads();
  if (sub < pp) {
    const int64_t stride = (int64_t)gridDim.x * 4 * pp;
    int64_t p = ((int64_t)blockIdx.x * 4 + sl) * pp + sub;
    constexpr int U = 8;
    for (; p + (U - 1) * stride < npairs; p += U * stride) {
      float gv[U], dx[U], dy[U], dz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t q = p + u * stride;
        const float rv = r[q * h + m], gg = gr[q * h + m];
        const float* a = pos + (q / k) * ps;
        const float* b = pos2 + (int64_t)idx[q] * p2s;
        gv[u] = rv > 0.f ? gg : 0.f;
        dx[u] = a[0] - b[0]; dy[u] = a[1] - b[1]; dz[u] = a[2] - b[2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { ax += gv[u] * dx[u]; ay += gv[u] * dy[u]; az += gv[u] * dz[u]; ac += gv[u]; }
    }
    for (; p < npairs; p += stride) {
      const float gv = r[p * h + m] > 0.f ? gr[p * h + m] : 0.f;
      const float* a = pos + (p / k) * ps;
      const float* b = pos2 + (int64_t)idx[p] * p2s;
      ax += gv * (a[0] - b[0]); ay += gv * (a[1] - b[1]); az += gv * (a[2] - b[2]); ac += gv;
    }
    atomicAdd(&s_acc[4 * m + 0], ax); atomicAdd(&s_acc[4 * m + 1], ay); atomicAdd(&s_acc[4 * m + 2], az);
    atomicAdd(&s_acc[4 * m + 3], ac);
  }
  __syncthreads();
  if ((int)threadIdx.x < 4 * h) {
    const int mm = threadIdx.x >> 2, comp = threadIdx.x & 3;
    const float v = s_acc[threadIdx.x];
    if (comp < 3) atomicAdd(dP1 + 3 * mm + comp, v);
    else atomicAdd(dc1 + mm, v);
  }
}

// interpolation backward: y[i] = sum_j w[i][j] table[idx[i][j]]  ->  dtable[idx[i][j]] += w[i][j] dy[i]
__global__ __launch_bounds__(TPB) void interp_bwd_kernel(const float* __restrict__ dy, int64_t ldy,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                         int64_t total, int k, int d, float* __restrict__ dtable,
                                                         int64_t ldt) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  const float gv = dy[i * ldy + c];
  for (int j = 0; j < k; ++j) atomicAdd(dtable + (int64_t)idx[i * k + j] * ldt + c, w[i * k + j] * gv);
}

// out[i][c] = alpha * a[i][c] + beta * b[i][c]   (b may be NULL)
__global__ __launch_bounds__(TPB) void axpby_kernel(const float* __restrict__ a, int64_t lda, float alpha,
                                                    const float* __restrict__ b, int64_t ldb, float beta, int64_t total,
                                                    int d, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float v = alpha * a[i * lda + c];
  if (b) v += beta * b[i * ldb + c];
  out[i * ldo + c] = v;
}

// out[i][c] = scale * vec[c]  (broadcast a row vector to n rows: mean backward)
__global__ __launch_bounds__(TPB) void broadcast_rows_kernel(const float* __restrict__ vec, float scale, int64_t total,
                                                             int d, float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  out[(e / d) * ldo + (e % d)] = scale * vec[e % d];
}

template <int NT>
void launch_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K, int splits,
                  int mps, float* part, float* part_b, int relu_x, hipStream_t st) {
  dim3 grid(occ4d::cdiv(N, 128), occ4d::cdiv(K, 32 * NT), splits);
  wgrad_kernel<NT><<<grid, 256, 0, st>>>(g, ldg, x, ldx, M, N, K, mps, part, part_b, relu_x);
}

}  // namespace

extern "C" {

int occ4d_linear_wgrad_workspace(int M, int N, int K, int* splits_out, int64_t* floats_out) {
  OCC4D_REQUIRE(M >= 0 && N >= 1 && K >= 1 && splits_out && floats_out, "occ4d_linear_wgrad_workspace: bad arguments");
  int s16 = 0, mps16 = 0;
  if (occ4d::wgrad16_plan(M, N, K, &s16, &mps16)) {        // the wide decoder layers: csrc/wgrad16.hip
    *splits_out = s16;
    *floats_out = (int64_t)(s16 + 1) * N * K + (int64_t)(s16 + 1) * N;      // (+ 1: the partial of the rows behind the
    return OCC4D_OK;                                                            // last full 16-row tile)
  }
  // enough m-chunks to give every CU a workgroup, each at least 256 rows deep
  // (K <= 64: the launch is HBM-bound -- one 16-row tile in flight per workgroup -- and wants more workgroups in flight:
  // 458752 x 832 x 32 at 518 workgroups ran at 3 TB/s)
  const int tiles = occ4d::cdiv(N, 128) * occ4d::cdiv(K, 416);
  static const int narrow_target = [] { const char* e = getenv("OCC4D_WGRAD_NARROW_WGS"); return e ? atoi(e) : 2048; }();
  int splits = occ4d::cdiv(K <= 64 ? narrow_target : 512, tiles);
  const int cap = M / 256 > 1 ? M / 256 : 1;
  if (splits > cap) splits = cap;
  *splits_out = splits > 1 ? splits : 1;
  *floats_out = (int64_t)(*splits_out) * N * K + (int64_t)(*splits_out) * N;   // dW partials + db partials
  return OCC4D_OK;
}

int occ4d_linear_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K, float* dw,
                           int accumulate, float* workspace, int splits, void* stream) {
  return occ4d_linear_wgrad_bias_f32(g, ldg, x, ldx, M, N, K, 0, dw, nullptr, accumulate, workspace, splits, stream);
}

int occ4d_linear_wgrad_bias_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K,
                                int relu_x, float* dw, float* db, int accumulate, float* workspace, int splits,
                                void* stream) {
  OCC4D_REQUIRE(g && x && dw && workspace, "occ4d_linear_wgrad_f32: null pointer");
  OCC4D_REQUIRE(M >= 1 && N >= 1 && K >= 1 && splits >= 1, "occ4d_linear_wgrad_f32: bad sizes");
  OCC4D_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)g % 16) == 0 &&
                    ((uintptr_t)x % 16) == 0,
                "occ4d_linear_wgrad_f32: N, K, ldg, ldx must be multiples of 4 and g, x 16-byte aligned");
  OCC4D_REQUIRE(ldg >= N && ldx >= K, "occ4d_linear_wgrad_f32: leading dimension too small");
  hipStream_t st = (hipStream_t)stream;
  const int mps = occ4d::cdiv(occ4d::cdiv(M, splits), WG_BM) * WG_BM;
  const int64_t nk = (int64_t)N * K;
  float* part_b = db ? workspace + (int64_t)splits * nk : nullptr;
  int s16 = 0, mps16 = 0;
  if (occ4d::wgrad16_plan(M, N, K, &s16, &mps16) && s16 == splits) {
    float* pb16 = db ? workspace + (int64_t)(splits + 1) * nk : nullptr;
    if (int rc = occ4d::wgrad16_launch(g, ldg, x, ldx, M, N, K, splits, mps16, workspace, pb16, relu_x, st)) return rc;
    if (nk % 4 == 0 && ((uintptr_t)workspace % 16) == 0 && ((uintptr_t)dw % 16) == 0)
      wgrad_reduce4_kernel<<<grid1d(nk / 4), TPB, 0, st>>>(workspace, splits + 1, nk / 4, dw, accumulate);
    else
      wgrad_reduce_kernel<<<grid1d(nk), TPB, 0, st>>>(workspace, splits + 1, nk, dw, accumulate);
    if (db) launch_short_reduce(pb16, splits + 1, N, db, accumulate, st);
    return occ4d::check_launch("occ4d_linear_wgrad_f32(reduce)");
  }
#define OCC4D_WGRAD(NT) launch_wgrad<NT>(g, ldg, x, ldx, M, N, K, splits, mps, workspace, part_b, relu_x, st)
  if (K <= 32) OCC4D_WGRAD(1);
  else if (K <= 64) OCC4D_WGRAD(2);
  else if (K <= 96) OCC4D_WGRAD(3);
  else if (K <= 160) OCC4D_WGRAD(5);
  else if (K <= 288) OCC4D_WGRAD(9);
  else OCC4D_WGRAD(13);
#undef OCC4D_WGRAD
  int rc = occ4d::check_launch("occ4d_linear_wgrad_f32");
  if (rc) return rc;
  if (nk % 4 == 0 && ((uintptr_t)workspace % 16) == 0 && ((uintptr_t)dw % 16) == 0)
    wgrad_reduce4_kernel<<<grid1d(nk / 4), TPB, 0, st>>>(workspace, splits, nk / 4, dw, accumulate);
  else
    wgrad_reduce_kernel<<<grid1d(nk), TPB, 0, st>>>(workspace, splits, nk, dw, accumulate);
  if (db) launch_short_reduce(part_b, splits, N, db, accumulate, st);
  return occ4d::check_launch("occ4d_linear_wgrad_f32(reduce)");
}

int occ4d_colsum_f32(const float* x, int64_t ldx, int n, int d, float* out, int accumulate, float* workspace,
                     int chunks, void* stream) {
  OCC4D_REQUIRE(x && out && workspace && n >= 1 && d >= 1 && chunks >= 1 && ldx >= d, "occ4d_colsum_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int rpc = occ4d::cdiv(n, chunks);
  colsum_partial_kernel<<<dim3(occ4d::cdiv(d, 64), chunks), TPB, 0, st>>>(x, ldx, n, d, rpc, workspace);
  launch_short_reduce(workspace, chunks, d, out, accumulate, st);
  return occ4d::check_launch("occ4d_colsum_f32");
}

int occ4d_relu_mask_f32(const float* g, int64_t ldg, const float* ref, int64_t ldr, int n, int d, float* out,
                        int64_t ldo, void* stream) {
  OCC4D_REQUIRE(g && ref && out && n >= 0 && d >= 1, "occ4d_relu_mask_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  relu_mask_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(g, ldg, ref, ldr, total, d, out, ldo);
  return occ4d::check_launch("occ4d_relu_mask_f32");
}

int occ4d_swish_f32(const float* x, int64_t ldx, int n, int d, float* y, int64_t ldy, void* stream) {
  OCC4D_REQUIRE(x && y && n >= 0 && d >= 1, "occ4d_swish_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  swish_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(x, ldx, total, d, y, ldy);
  return occ4d::check_launch("occ4d_swish_f32");
}

int occ4d_swish_bwd_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int n, int d, float* out, int64_t ldo,
                        void* stream) {
  OCC4D_REQUIRE(g && x && out && n >= 0 && d >= 1, "occ4d_swish_bwd_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  swish_bwd_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(g, ldg, x, ldx, total, d, out, ldo);
  return occ4d::check_launch("occ4d_swish_bwd_f32");
}

int occ4d_scatter_add_rows_f32(const float* src, int64_t lds, const int32_t* idx, int n, int d, float scale,
                               float* out, int64_t ldo, void* stream) {
  OCC4D_REQUIRE(src && idx && out && n >= 0 && d >= 1, "occ4d_scatter_add_rows_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  scatter_add_rows_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(src, lds, idx, total, d, scale, out, ldo);
  return occ4d::check_launch("occ4d_scatter_add_rows_f32");
}

int occ4d_segment_sum_f32(const float* src, int n, int k, int d, float* out, int64_t ldo, void* stream) {
  OCC4D_REQUIRE(src && out && n >= 0 && k >= 1 && d >= 1 && ldo >= d, "occ4d_segment_sum_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  if (k <= 16 && d % 4 == 0 && ldo % 4 == 0 && total / 4 < ((int64_t)1 << 31) &&
      (((uintptr_t)src | (uintptr_t)out) % 16) == 0) {
    segment_sum4_kernel<<<grid1d(total / 4), TPB, 0, (hipStream_t)stream>>>(src, (int)(total / 4), k, d / 4, out, ldo);
    return occ4d::check_launch("occ4d_segment_sum_f32");
  }
  segment_sum_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(src, total, k, d, out, ldo);
  return occ4d::check_launch("occ4d_segment_sum_f32");
}

int occ4d_maxpool_gather_bwd_f32(const float* y, int64_t ldy, const int32_t* idx, int n_out, int k, int d,
                                 const float* dz, int64_t ldz, float* dy, int64_t ldd, void* stream) {
  OCC4D_REQUIRE(y && idx && dz && dy && n_out >= 0 && k >= 1 && d >= 1, "occ4d_maxpool_gather_bwd_f32: bad arguments");
  const int64_t total = (int64_t)n_out * d;
  if (!total) return OCC4D_OK;
  maxpool_bwd_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(y, ldy, idx, total, k, d, dz, ldz, dy, ldd);
  return occ4d::check_launch("occ4d_maxpool_gather_bwd_f32");
}

int occ4d_layernorm_bwd_f32(const float* x, int64_t ldx, const float* gamma, const float* g, int64_t ldg, float eps,
                            int n, int d, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* stream) {
  OCC4D_REQUIRE(x && g && dx && n >= 0 && d >= 1, "occ4d_layernorm_bwd_f32: bad arguments");
  OCC4D_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "occ4d_layernorm_bwd_f32: dgamma/dbeta both or neither");
  if (!n) return OCC4D_OK;
  const int ln_blocks = occ4d::cdiv(n, TPB / 64) < 2048 ? occ4d::cdiv(n, TPB / 64) : 2048;   // 8 K waves, rows by grid stride
  layernorm_bwd_kernel<<<ln_blocks, TPB, 0, (hipStream_t)stream>>>(x, ldx, gamma, g, ldg, eps, n, d, dx,
                                                                                    lddx, dgamma, dbeta);
  return occ4d::check_launch("occ4d_layernorm_bwd_f32");
}

int occ4d_pt_softmax_agg_bwd_f32(const float* logits, const float* v, int64_t ldv, const float* pe,
                                 const int32_t* idx, int n, int k, int d, float divisor, const float* dagg,
                                 int64_t ldda, float* dlogits, float* dpe, float* dv, int64_t lddv, void* stream) {
  OCC4D_REQUIRE(logits && v && idx && dagg && dlogits && (dv || dpe), "occ4d_pt_softmax_agg_bwd_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && k <= 16 && d >= 1 && divisor > 0.f, "occ4d_pt_softmax_agg_bwd_f32: bad sizes");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  static const bool wide = [] { const char* e = getenv("OCC4D_SOFTMAX_BWD4"); return !e || e[0] != '0'; }();
  const bool al = d % 4 == 0 && ldv % 4 == 0 && ldda % 4 == 0 && total / 4 < ((int64_t)1 << 31) &&
                  ldv < ((int64_t)1 << 31) && ldda < ((int64_t)1 << 31) &&
                  (((uintptr_t)logits | (uintptr_t)v | (uintptr_t)pe | (uintptr_t)dagg | (uintptr_t)dlogits |
                    (uintptr_t)dpe) % 16) == 0;
  if (wide && al && !dv && (k == 16 || k == 14 || k == 12 || k == 8)) {      // (dpe != null by the check above)
    const int total4 = (int)(total / 4), d4 = d / 4;
    const float inv_div = 1.f / divisor;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
      case 16: softmax_agg_bwd4_launch<16>(logits, v, (int)ldv, pe, idx, total4, d4, inv_div, dagg, (int)ldda, dlogits, dpe, st); break;
      case 14: softmax_agg_bwd4_launch<14>(logits, v, (int)ldv, pe, idx, total4, d4, inv_div, dagg, (int)ldda, dlogits, dpe, st); break;
      case 12: softmax_agg_bwd4_launch<12>(logits, v, (int)ldv, pe, idx, total4, d4, inv_div, dagg, (int)ldda, dlogits, dpe, st); break;
      default: softmax_agg_bwd4_launch<8>(logits, v, (int)ldv, pe, idx, total4, d4, inv_div, dagg, (int)ldda, dlogits, dpe, st); break;
    }
    return occ4d::check_launch("occ4d_pt_softmax_agg_bwd_f32");
  }
  softmax_agg_bwd_kernel<16><<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(logits, v, ldv, pe, idx, total, k, d,
                                                                             divisor, dagg, ldda, dlogits, dpe, dv, lddv);
  return occ4d::check_launch("occ4d_pt_softmax_agg_bwd_f32");
}

int occ4d_pt_pos_hidden_bwd_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s, const int32_t* idx,
                                int n, int k, int h, const float* r, const float* gr, float* dP1, float* dc1,
                                void* stream) {
  OCC4D_REQUIRE(pos && pos2 && idx && r && gr && dP1 && dc1, "occ4d_pt_pos_hidden_bwd_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && h >= 1 && h <= 64, "occ4d_pt_pos_hidden_bwd_f32: need h <= 64");
  const int64_t npairs = (int64_t)n * k;
  if (!npairs) return OCC4D_OK;
  const int64_t want = (npairs + 4 * (64 / h) - 1) / (4 * (64 / h));
  const int blocks = (int)(want < 512 ? want : 512);
  pos_hidden_bwd_kernel<<<blocks, TPB, 0, (hipStream_t)stream>>>(pos, ps, pos2, p2s, idx, npairs, k, h, r, gr, dP1, dc1);
  return occ4d::check_launch("occ4d_pt_pos_hidden_bwd_f32");
}

int occ4d_interp_bwd_f32(const float* dy, int64_t ldy, const int32_t* idx, const float* w, int n, int k, int d,
                         float* dtable, int64_t ldt, void* stream) {
  OCC4D_REQUIRE(dy && idx && w && dtable && n >= 0 && k >= 1 && d >= 1, "occ4d_interp_bwd_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  interp_bwd_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(dy, ldy, idx, w, total, k, d, dtable, ldt);
  return occ4d::check_launch("occ4d_interp_bwd_f32");
}

int occ4d_segment_gather_sum_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets,
                                 const float* weights, int div, int n_out, int d, float scale, float* out, int64_t ldo,
                                 void* stream) {
  OCC4D_REQUIRE(src && order && offsets && out && n_out >= 0 && d >= 1 && div >= 1 && lds >= d && ldo >= d,
                "occ4d_segment_gather_sum_f32: bad arguments");
  const int64_t total = (int64_t)n_out * d;
  if (!total) return OCC4D_OK;
  segment_gather_sum_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(src, lds, order, offsets, weights, div, total,
                                                                            d, scale, out, ldo);
  return occ4d::check_launch("occ4d_segment_gather_sum_f32");
}

int64_t occ4d_segments_workspace_ints(int n_out) { return (int64_t)(SEG_BLOCKS + 1) * n_out; }

int occ4d_segments_build_i32(const int32_t* idx, int64_t n, int n_out, int32_t* order, int32_t* offsets, int32_t* workspace,
                             void* stream) {
  OCC4D_REQUIRE(idx && order && offsets && workspace && n >= 0 && n < (int64_t)1 << 31 && n_out >= 1 && n_out <= SEG_MAX_ROWS,
                "occ4d_segments_build_i32: bad arguments (1 <= n_out <= %d)", SEG_MAX_ROWS);
  hipStream_t st = (hipStream_t)stream;
  int32_t* bh = workspace;
  int32_t* tot = workspace + (int64_t)SEG_BLOCKS * n_out;
  const size_t lds = sizeof(int) * (size_t)n_out;
  seg_count_kernel<<<SEG_BLOCKS, TPB, lds, st>>>(idx, n, n_out, bh);
  seg_base_kernel<<<occ4d::cdiv(n_out, TPB), TPB, 0, st>>>(bh, SEG_BLOCKS, n_out, tot);
  seg_scan_kernel<<<1, 1024, 0, st>>>(tot, n_out, offsets);
  seg_fill_kernel<<<SEG_BLOCKS, TPB, lds, st>>>(idx, n, n_out, bh, offsets, order);
  return occ4d::check_launch("occ4d_segments_build_i32");
}

int occ4d_segment_sum_sorted_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets, int n_out,
                                 int d, int parts, float scale, float* out, int64_t ldo, void* stream) {
  OCC4D_REQUIRE(src && order && offsets && out && n_out >= 0 && d >= 4 && d % 4 == 0 && parts >= 1 && lds >= d && ldo >= d &&
                    lds % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "occ4d_segment_sum_sorted_f32: bad arguments (d, lds, ldo multiples of 4; src, out 16-byte aligned)");
  if (!n_out) return OCC4D_OK;
  if (int rc = occ4d::zero_rows(out, ldo, n_out, d, (hipStream_t)stream)) return rc;
  const int64_t total = (int64_t)n_out * parts * (d / 4);
  segment_sum_sorted_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(src, lds, order, offsets, total, d / 4, parts,
                                                                            scale, out, ldo);
  return occ4d::check_launch("occ4d_segment_sum_sorted_f32");
}

int occ4d_pt_pos_hidden_bwd_det_workspace(int n, int k, int h, int64_t* floats) {
  OCC4D_REQUIRE(floats && n >= 0 && k >= 1 && h >= 1 && h <= 64, "occ4d_pt_pos_hidden_bwd_det_workspace: bad arguments");
  const int64_t want = ((int64_t)n * k + 3) / 4;
  const int64_t blocks = want < 1024 ? (want > 0 ? want : 1) : 1024;
  *floats = blocks * 4 * h * 4;
  return OCC4D_OK;
}

int occ4d_pt_pos_hidden_bwd_det_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s, const int32_t* idx,
                                    int n, int k, int h, const float* r, const float* gr, float* dP1, float* dc1,
                                    float* workspace, void* stream) {
  OCC4D_REQUIRE(pos && pos2 && idx && r && gr && dP1 && dc1 && workspace, "occ4d_pt_pos_hidden_bwd_det_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && h >= 1 && h <= 64, "occ4d_pt_pos_hidden_bwd_det_f32: need h <= 64");
  const int64_t npairs = (int64_t)n * k;
  if (!npairs) return OCC4D_OK;
  const int64_t want = (npairs + 3) / 4;
  const int blocks = (int)(want < 1024 ? want : 1024);
  pos_hidden_bwd_partials_kernel<<<blocks, TPB, 0, (hipStream_t)stream>>>(pos, ps, pos2, p2s, idx, npairs, k, h, r, gr,
                                                                          workspace);
  pos_hidden_bwd_reduce_kernel<<<occ4d::cdiv(4 * h, 64), 64, 0, (hipStream_t)stream>>>(workspace, blocks * 4, h, dP1, dc1);
  return occ4d::check_launch("occ4d_pt_pos_hidden_bwd_det_f32");
}

int occ4d_axpby_f32(const float* a, int64_t lda, float alpha, const float* b, int64_t ldb, float beta, int n, int d,
                    float* out, int64_t ldo, void* stream) {
  OCC4D_REQUIRE(a && out && n >= 0 && d >= 1, "occ4d_axpby_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  axpby_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(a, lda, alpha, b, ldb, beta, total, d, out, ldo);
  return occ4d::check_launch("occ4d_axpby_f32");
}

int occ4d_broadcast_rows_f32(const float* vec, float scale, int n, int d, float* out, int64_t ldo, void* stream) {
  OCC4D_REQUIRE(vec && out && n >= 0 && d >= 1 && ldo >= d, "occ4d_broadcast_rows_f32: bad arguments");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  broadcast_rows_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(vec, scale, total, d, out, ldo);
  return occ4d::check_launch("occ4d_broadcast_rows_f32");
}

}  // extern "C"
