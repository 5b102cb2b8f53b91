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

}  // namespace

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
