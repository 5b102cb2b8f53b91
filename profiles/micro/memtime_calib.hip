// Calibration: s_memtime ticks per v_mfma_f32_16x16x4_f32 issue slot (32 shader cycles), and ticks per wall ns.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float a = 1.f + threadIdx.x, b = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* t;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 256 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<<<256, 256>>>(out, t, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<<<256, 256>>>(out, t, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double n = (double)iters * 64;
  printf("ticks per MFMA issue slot: %.2f (block 0), %.2f (block 255); wall %.3f ms -> %.2f ns per MFMA -> %.3f ticks per ns\n",
         h[0] / n, h[255] / n, ms, ms * 1e6 / n, h[0] / (ms * 1e6));
  return 0;
}
