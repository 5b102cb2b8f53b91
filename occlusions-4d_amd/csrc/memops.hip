// Row copy / row clear as KERNELS.  The library never calls hipMemset*Async / hipMemcpy*Async: under stream capture a
// 2-D memset was not replayed with the graph on this stack (round 4: the sorted-segment sum's output was cleared at
// capture time only and every replay accumulated into stale memory -- NaN gradients after two steps), and a kernel is a
// kernel node in every capture.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ dst, int64_t ld, int64_t total, int d) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < total) dst[(e / d) * ld + e % d] = 0.f;
}

__global__ __launch_bounds__(256) void copy_rows_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src,
                                                        int64_t lds, int64_t total, int d) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < total) dst[(e / d) * ldd + e % d] = src[(e / d) * lds + e % d];
}

__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ dst, int64_t ld, int64_t total, int d, float v) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < total) dst[(e / d) * ld + e % d] = v;
}

// Next level of a NESTED farthest-point chain: the level's subset = the first m picks of level 0's
// selection order; `orig` = ascending original (level-0) indices of the current cloud's n points.  One workgroup: every
// pick finds its position in the current cloud by binary search and sets a bit; the set bits, in ascending order, are the
// subset (positions in the current cloud) and, through `orig`, the next cloud's original indices.
constexpr int NEST_T = 1024;
constexpr int NEST_WORDS = 1024;             // n <= 32768
__global__ __launch_bounds__(NEST_T) void nested_level_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ orig,
                                                              int n, int m, int32_t* __restrict__ out_pos,
                                                              int32_t* __restrict__ out_orig) {
  __shared__ unsigned s_flags[NEST_WORDS];
  __shared__ int s_wave[NEST_T / 64];
  const int t = threadIdx.x;
  s_flags[t] = 0u;
  __syncthreads();
  for (int i = t; i < m; i += NEST_T) {
    const int pick = order[i];
    int lo = 0, hi = n;                       // lower bound of pick in orig[0, n)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (orig[mid] < pick) lo = mid + 1; else hi = mid;
    }
    lo = min(lo, n - 1);
    out_pos[i] = lo;                          // (stays behind the compacted prefix when picks repeat: a cloud with fewer
    atomicOr(&s_flags[lo >> 5], 1u << (lo & 31));   //  distinct points than samples, as in fps.hip)
  }
  __syncthreads();
  unsigned bits = s_flags[t];
  const int cnt = __popc(bits);
  int incl = cnt;                             // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((t & 63) >= o) incl += v;
  }
  if ((t & 63) == 63) s_wave[t >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (t >> 6); ++w) base += s_wave[w];
  int pos = base + incl - cnt;
  while (bits) {
    const int b = __ffs(bits) - 1;
    bits &= bits - 1;
    if (pos < m) out_pos[pos] = t * 32 + b;
    ++pos;
  }
  __syncthreads();
  // picks that repeat (a cloud with fewer distinct points than samples): the entries behind the `distinct` compacted
  // positions repeat the LAST of them -- out_pos stays ascending (non-decreasing), as the docstring of
  // ops.nested_fps_level promises (the raw search results used to stay there, unsorted)
  int distinct = 0;
  for (int w = 0; w < NEST_T / 64; ++w) distinct += s_wave[w];
  distinct = min(distinct, m);
  if (distinct < m) {
    const int last = out_pos[distinct - 1];
    __syncthreads();
    for (int i = distinct + t; i < m; i += NEST_T) out_pos[i] = last;
    __syncthreads();
  }
  for (int i = t; i < m; i += NEST_T) out_orig[i] = orig[out_pos[i]];
}

}  // namespace

extern "C" int occ4d_copy_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int n, int d, void* stream) {
  OCC4D_REQUIRE(dst && src && n >= 0 && d >= 0 && ldd >= d && lds >= d, "occ4d_copy_rows_f32: bad arguments (n = %d, d = %d)", n, d);
  return occ4d::copy_rows(dst, ldd, src, lds, n, d, (hipStream_t)stream);
}

extern "C" int occ4d_fill_rows_f32(float* dst, int64_t ld, int n, int d, float value, void* stream) {
  OCC4D_REQUIRE(dst && n >= 0 && d >= 0 && ld >= d, "occ4d_fill_rows_f32: bad arguments (n = %d, d = %d)", n, d);
  const int64_t total = (int64_t)n * d;
  if (total <= 0) return OCC4D_OK;
  fill_rows_kernel<<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(dst, ld, total, d, value);
  return occ4d::check_launch("occ4d_fill_rows_f32");
}

extern "C" int occ4d_nested_fps_level_i32(const int32_t* order, const int32_t* orig, int n, int m, int32_t* out_pos,
                                          int32_t* out_orig, void* stream) {
  OCC4D_REQUIRE(order && orig && out_pos && out_orig, "occ4d_nested_fps_level_i32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32 * NEST_WORDS && m >= 1 && m <= n, "occ4d_nested_fps_level_i32: n = %d (1 .. %d), m = %d", n,
                32 * NEST_WORDS, m);
  nested_level_kernel<<<1, NEST_T, 0, (hipStream_t)stream>>>(order, orig, n, m, out_pos, out_orig);
  return occ4d::check_launch("occ4d_nested_fps_level_i32");
}

namespace occ4d {

int zero_rows(float* dst, int64_t ld, int64_t n, int d, hipStream_t st) {
  const int64_t total = n * d;
  if (total <= 0) return OCC4D_OK;
  zero_rows_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst, ld, total, d);
  return check_launch("zero_rows");
}

int copy_rows(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t n, int d, hipStream_t st) {
  const int64_t total = n * d;
  if (total <= 0) return OCC4D_OK;
  copy_rows_kernel<<<cdiv(total, 256), 256, 0, st>>>(dst, ldd, src, lds, total, d);
  return check_launch("copy_rows");
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

}  // namespace occ4d
