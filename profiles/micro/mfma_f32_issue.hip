// Microbenchmark: sustained issue rate of the fp32 MFMAs under the access patterns of csrc/trunk.hip.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/mfma_f32_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned& s) {      // uniform in [-1, 1): random mantissas, like real activations
  s = s * 1664525u + 1013904223u;
  return (float)(int)s * (1.0f / 2147483648.0f);
}

template <int NACC, bool LDS, int PF, bool RANDOM = false>
__global__ __launch_bounds__(512, 2) void k16(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[13312];
  const int lane = threadIdx.x & 63;
  unsigned seed = 12345u + 977u * threadIdx.x + 31u * blockIdx.x;
  for (int i = threadIdx.x; i < 13312; i += blockDim.x) buf[i] = RANDOM ? 0.05f * rnd(seed) : 1e-3f * (i & 7);
  __syncthreads();
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 x[8];
  for (int t = 0; t < 8; ++t)
    x[t] = RANDOM ? f32x4{rnd(seed), rnd(seed), rnd(seed), rnd(seed)} : f32x4{1.f + lane, 2.f, 3.f + t, 4.f};
  const float* fa = buf + lane * 4;
  for (int it = 0; it < iters; ++it) {
    f32x4 w[PF + 1][NACC];
    if (LDS) {
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int a = 0; a < NACC; ++a) w[p][a] = *reinterpret_cast<const f32x4*>(fa + (p * NACC + a) * 256);
    }
#pragma unroll
    for (int t = 0; t < 26; ++t) {
      f32x4 cur[NACC];
#pragma unroll
      for (int a = 0; a < NACC; ++a) cur[a] = LDS ? w[0][a] : x[(t + a) & 7];
      if (LDS) {
#pragma unroll
        for (int p = 0; p + 1 < PF; ++p)
#pragma unroll
          for (int a = 0; a < NACC; ++a) w[p][a] = w[p + 1][a];
        if (t + PF < 26) {
#pragma unroll
          for (int a = 0; a < NACC; ++a) w[PF - 1][a] = *reinterpret_cast<const f32x4*>(fa + (((t + PF) * NACC + a) % 52) * 256);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[a][e], x[t & 7][e], acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a].x + acc[a].y + acc[a].z + acc[a].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  float x[8];
  for (int t = 0; t < 8; ++t) x[t] = 1.f + lane + t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 104; ++t)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[t & 7], x[(t + a) & 7], acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_it(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4 * 4);
  const int iters = 2000;      // ~10 ms per launch: long enough for the power management to settle
  const double per16 = 2.0 * 16 * 16 * 4, per32 = 2.0 * 32 * 32 * 2;
#define RUN16(NACC, LDS, PF, THREADS, label) RUN16R(NACC, LDS, PF, THREADS, false, label)
#define RUN16R(NACC, LDS, PF, THREADS, RND, label)                                                                \
  {                                                                                                               \
    float ms = time_it([&] { k16<NACC, LDS, PF, RND><<<256, THREADS>>>(out, iters); });                           \
    double n = 256.0 * (THREADS / 64) * iters * 26 * 4 * NACC;                                                    \
    printf("%-58s %8.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", label, ms,             \
           n * per16 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));                                                 \
  }
  RUN16(2, false, 1, 512, "16x16x4, 2 accumulators, registers only, 2 waves/SIMD");
  RUN16(2, false, 1, 256, "16x16x4, 2 accumulators, registers only, 1 wave/SIMD");
  RUN16(4, false, 1, 512, "16x16x4, 4 accumulators, registers only, 2 waves/SIMD");
  RUN16(1, false, 1, 512, "16x16x4, 1 accumulator (dependent chain), 2 waves/SIMD");
  RUN16(1, false, 1, 256, "16x16x4, 1 accumulator (dependent chain), 1 wave/SIMD");
  RUN16(2, true, 1, 512, "16x16x4, 2 acc, A fragments from LDS (prefetch 1), 2 w/SIMD");
  RUN16(2, true, 2, 512, "16x16x4, 2 acc, A fragments from LDS (prefetch 2), 2 w/SIMD");
  RUN16(2, true, 1, 256, "16x16x4, 2 acc, A fragments from LDS (prefetch 1), 1 w/SIMD");
  RUN16(2, true, 2, 256, "16x16x4, 2 acc, A fragments from LDS (prefetch 2), 1 w/SIMD");
  RUN16R(2, false, 1, 512, true, "RANDOM DATA: 16x16x4, 2 acc, registers only, 2 w/SIMD");
  RUN16R(2, true, 1, 512, true, "RANDOM DATA: 16x16x4, 2 acc, A from LDS (prefetch 1), 2 w/SIMD");
  RUN16R(2, true, 1, 256, true, "RANDOM DATA: 16x16x4, 2 acc, A from LDS (prefetch 1), 1 w/SIMD");
  {
    float ms = time_it([&] { k32<1><<<256, 256>>>(out, iters); });
    double n = 256.0 * 4 * iters * 104;
    printf("%-58s %8.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD)\n", "32x32x2, 1 accumulator, 1 wave/SIMD", ms,
           n * per32 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
    ms = time_it([&] { k32<2><<<256, 256>>>(out, iters); });
    n = 256.0 * 4 * iters * 104 * 2;
    printf("%-58s %8.3f ms  %6.1f TFLOP/s  (%.1f cycles per MFMA per SIMD)\n", "32x32x2, 2 accumulators, 1 wave/SIMD", ms,
           n * per32 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
  }
  return 0;
}
