// Microbenchmark: the EFFECTIVE shader clock under different instruction mixes, measured inside the kernel as
// s_memtime (shader cycles) over s_memrealtime (100 MHz).  rocm-smi keeps reporting sclk = 2.39 GHz while the
// effective clock of MFMA-dense kernels is lower (power management), so roofline fractions priced at 2.4 GHz have a
// ceiling below 1 that depends on the kernel's power, not on its schedule.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/clock_under_load.hip -o /tmp/cul && /tmp/cul
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return (float)(int)s * (1.0f / 2147483648.0f);
}

// MODE 0: MFMA 16x16x4 f32 from registers; 1: + 2 ds_read_b128 per 8 MFMAs; 2: + LDS-DMA refill of 1 KB per 8 MFMAs per wave;
// 3: VALU fma only; 4: idle spin (s_sleep)
template <int MODE, bool RANDOM>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[2][8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = 12345u + 977u * threadIdx.x + 31u * blockIdx.x;
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) (&buf[0][0])[i] = RANDOM ? 0.01f * rnd(seed) : 0.f;
  __syncthreads();
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  f32x4 x = RANDOM ? f32x4{rnd(seed), rnd(seed), rnd(seed), rnd(seed)} : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 y = RANDOM ? f32x4{rnd(seed), rnd(seed), rnd(seed), rnd(seed)} : f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* f = &buf[0][0] + lane * 4;
  for (int it = 0; it < iters; ++it) {
    if (MODE <= 2) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        f32x4 wa = x, wb = y;
        if (MODE >= 1) {
          wa = *reinterpret_cast<const f32x4*>(f + ((2 * g + it) & 15) * 256);
          wb = *reinterpret_cast<const f32x4*>(f + ((2 * g + 1 + it) & 15) * 256);
        }
        if (MODE == 2) {
          const float* gsrc = src + ((it * 8 + g) & 1023) * 256 * 8 + wave * 256 + lane * 4;
          const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)&buf[1][wave * 256]);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
        }
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, x.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.x, y.x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, x.y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.y, y.y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, x.z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.z, y.z, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, x.w, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.w, y.w, a1, 0, 0, 0);
      }
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        a0.x = fmaf(a0.x, x.x, y.x); a0.y = fmaf(a0.y, x.y, y.y); a0.z = fmaf(a0.z, x.z, y.z); a0.w = fmaf(a0.w, x.w, y.w);
      }
    } else {
      __builtin_amdgcn_s_sleep(64);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    out[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
  }
  if (a0.x + a1.y == 123.456f) out[0] = 0;
}

template <int MODE, bool RANDOM>
void run(const char* name, unsigned long long* d, const float* src, int iters) {
  static unsigned long long h[256 * 8 * 2];
  for (int rep = 0; rep < 3; ++rep) { k<MODE, RANDOM><<<256, 512>>>(d, src, iters); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < 2048; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
  printf("%-66s %9.0f cycles in %8.1f us -> %.3f GHz", name, c / 2048, r / 2048 / 100.0, c / (r * 10.0));
  if (MODE <= 2) printf("   %.1f cycles per MFMA per SIMD", c / 2048 / (64.0 * iters) / 2);
  printf("\n");
}

int main() {
  unsigned long long* d;
  float* src;
  (void)hipMalloc(&d, 256 * 8 * 2 * 8);
  (void)hipMalloc(&src, 1024 * 256 * 8 * 4);
  (void)hipMemset(src, 0, 1024 * 256 * 8 * 4);
  const int it = 40000;     // ~35 ms per launch: long enough for the power management to settle
  run<4, false>("idle (s_sleep)", d, src, 4000);
  run<3, true>("VALU fma only, random data", d, src, it);
  run<0, false>("MFMA 16x16x4 f32 from registers, zeros", d, src, it);
  run<0, true>("MFMA 16x16x4 f32 from registers, random data", d, src, it);
  run<1, true>("MFMA + 2 ds_read_b128 per 8 MFMAs, random data", d, src, it);
  run<2, true>("MFMA + LDS reads + 1 KB LDS-DMA per wave per 8 MFMAs, random data", d, src, it);
  run<2, false>("MFMA + LDS reads + LDS-DMA, zeros", d, src, it);
  return 0;
}
