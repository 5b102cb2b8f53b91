// How fast does a CU pull an L2-resident stream into LDS with global_load_lds_dwordx4 when every CU pulls the SAME stream
// (the weight streams of csrc/resblock_f16x3.hip, crossattn_*.hip)?  No compute: 8 waves per workgroup, one workgroup per CU,
// each wave issues PARTS fragments (1 KB) per stage into one of DEPTH + 1 buffers, DEPTH stages ahead, waits, barrier.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int FW = 256;                       // words per fragment (1 KB)

__device__ __forceinline__ void dma(const unsigned* src, unsigned lds, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(lds), "s"(src) : "memory");
}

template <int SF, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned* w, int nstage, int stream_stages, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PARTS = (SF + 7) / 8, STAGE = SF * FW;
  const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned*)lds;
  auto issue = [&](int s) {
    const unsigned* src = w + (size_t)(s % stream_stages) * STAGE;
    const unsigned dst = base + (unsigned)(s % (DEPTH + 1)) * (STAGE * 4);
#pragma unroll
    for (int i = 0; i < PARTS; ++i) {
      const int f = min(wave + 8 * i, SF - 1);
      dma(src + f * FW, dst + f * (FW * 4), lane * 16);
    }
  };
  for (int s = 0; s < DEPTH; ++s) issue(s);
  unsigned acc = 0;
  for (int s = 0; s < nstage; ++s) {
    issue(s + DEPTH);
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PARTS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PARTS) : "memory");
    __builtin_amdgcn_s_barrier();
    acc += lds[(s % (DEPTH + 1)) * STAGE + threadIdx.x];           // (touch the stage that has landed)
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int SF, int DEPTH>
void run(const unsigned* w, unsigned* sink, int stream_stages, int cus) {
  const int nstage = 2000;
  const size_t lds = (size_t)(DEPTH + 1) * SF * FW * 4;
  hipFuncSetAttribute((const void*)stream_kernel<SF, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream_kernel<SF, DEPTH><<<cus, 512, lds>>>(w, nstage, stream_stages, sink);
  hipEventRecord(e0);
  stream_kernel<SF, DEPTH><<<cus, 512, lds>>>(w, nstage, stream_stages, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nstage * SF * 1024;
  printf("stage %2d KB, %d ahead, stream %5.0f KB: %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s on %d CUs  (%.1f B/clk/CU at 2.4 GHz)\n", SF, DEPTH,
         stream_stages * SF * 1.0, ms, bytes / ms / 1e6, bytes * cus / ms / 1e9, cus, bytes / ms / 1e6 / 2.4);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t words = (size_t)64 * 56 * FW;
  unsigned *w, *sink;
  hipMalloc(&w, words * 4);
  hipMalloc(&sink, 64);
  hipMemset(w, 1, words * 4);
  run<26, 1>(w, sink, 27, cus);
  run<30, 1>(w, sink, 27, cus);
  run<30, 2>(w, sink, 27, cus);
  run<52, 1>(w, sink, 27, cus);
  run<52, 2>(w, sink, 27, cus);
  run<52, 1>(w, sink, 1, cus);               // one 52 KB stage over and over
  run<28, 1>(w, sink, 54, cus);
  return 0;
}
