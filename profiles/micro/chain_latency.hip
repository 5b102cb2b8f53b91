// Latency of the building blocks of a serial, wave-uniform decision chain (the farthest-point-sampling step) on
// gfx950, in shader cycles per link (s_memtime: 1 tick = 1 cycle, memtime_calib.hip).  One workgroup; argv[1] = waves
// (1 = a wave alone on its SIMD, 8 = two waves per SIMD as in the FPS kernels).  Every test is a dependent chain of
// REP links inside a loop of ITERS iterations.
//   hipcc --offload-arch=gfx950 -O3 profiles/micro/chain_latency.hip -o /tmp/chain && /tmp/chain 1 && /tmp/chain 8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define SKIP8 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#define SKIP32 SKIP8 SKIP8 SKIP8 SKIP8
constexpr int ITERS = 2000;
constexpr int LINKS = 16;

template <int TEST>
__global__ __launch_bounds__(512) void k(unsigned long long* ticks, int* sink, int seed) {
  __shared__ int s_x[64];
  int v = threadIdx.x + seed, w = seed * 3 + 1;
  int s = seed;
  if (threadIdx.x < 64) s_x[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if (TEST == 0) {          // dependent VALU op
      REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));)
    } else if (TEST == 1) {   // dependent packed-fp32 op
      float2 a = {__int_as_float(v), __int_as_float(w)};
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a));)
      v = __float_as_int(a.x);
    } else if (TEST == 2) {   // GPR-indexed read: s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off, data-dependent chain
      REP16(asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\tv_mov_b32 %0, %1\n\ts_set_gpr_idx_off" : "=v"(v) : "v"(v), "s"(0));)
    } else if (TEST == 3) {   // VALU -> SGPR -> VALU: v_readlane then v_mov from the SGPR
      REP16(asm volatile("v_readlane_b32 %1, %0, 5\n\ts_nop 0\n\tv_add_u32 %0, %1, %0" : "+v"(v), "=s"(s));)
    } else if (TEST == 4) {   // VALU compare -> SALU select -> VALU: v_cmp, s_cmp_lg_u64, s_cselect, v_add
      REP16(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\ts_cmp_lg_u64 vcc, 0\n\ts_cselect_b32 %2, 3, 5\n\tv_add_u32 %0, %2, %0"
                         : "+v"(v), "+v"(w), "=s"(s) : : "vcc", "scc");)
    } else if (TEST == 5) {   // fused DPP max reduction (6 steps) + readlane + back to a VGPR
      REP4(asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                        "s_nop 0\n\tv_readlane_b32 %1, %0, 63\n\ts_nop 0\n\tv_xor_b32 %0, %1, %0" : "+v"(v), "=s"(s));)
    } else if (TEST == 6) {   // taken scalar branch
      REP16(asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n1:" : : "s"(s) : "scc");)
    } else if (TEST == 11) {  // taken scalar branch over 32 instructions (128 B: past the instruction buffer)
      REP16(asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\t" SKIP32 "1:" : : "s"(s) : "scc");)
    } else if (TEST == 12) {  // taken scalar branch over 256 instructions (1 KB)
      REP16(asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\t" SKIP32 SKIP32 SKIP32 SKIP32 SKIP32 SKIP32 SKIP32 SKIP32 "1:" : : "s"(s) : "scc");)
    } else if (TEST == 7) {   // not-taken scalar branch
      REP16(asm volatile("s_cmp_lg_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(s) : "scc");)
    } else if (TEST == 8) {   // LDS round trip of a uniform value: ds_write, wait, barrier, ds_read, wait
      REP4(s_x[threadIdx.x & 63] = v; __syncthreads(); v += s_x[(threadIdx.x + 1) & 63]; )
    } else if (TEST == 9) {   // dependent SALU op
      REP16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");)
    } else if (TEST == 10) {  // ballot -> s_ff1 -> v_readlane at that lane -> VALU
      REP16(asm volatile("v_cmp_ne_u32 vcc, 0, %0\n\ts_ff1_i32_b64 %1, vcc\n\ts_nop 0\n\tv_readlane_b32 %1, %0, %1\n\ts_nop 0\n\tv_or_b32 %0, %1, %0"
                         : "+v"(v), "=&s"(s) : : "vcc");)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[threadIdx.x] = v + s;
  if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

template <int TEST>
void run(const char* name, int links, int threads, unsigned long long* d_t, int* d_s) {
  k<TEST><<<1, threads>>>(d_t, d_s, 1);
  (void)hipDeviceSynchronize();
  k<TEST><<<1, threads>>>(d_t, d_s, 1);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-62s %7.1f cycles per link (wave 0), %7.1f (last wave)\n", name, (double)h[0] / ((double)ITERS * links),
         (double)h[threads / 64 - 1] / ((double)ITERS * links));
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 1;
  const int threads = 64 * waves;
  unsigned long long* d_t;
  int* d_s;
  (void)hipMalloc(&d_t, 64);
  (void)hipMalloc(&d_s, 512 * 4);
  printf("%d wave(s) in the workgroup\n", waves);
  run<0>("dependent v_add_u32", LINKS, threads, d_t, d_s);
  run<1>("dependent v_pk_mul_f32", LINKS, threads, d_t, d_s);
  run<9>("dependent s_add_u32", LINKS, threads, d_t, d_s);
  run<2>("s_set_gpr_idx_on + v_mov + s_set_gpr_idx_off (dependent)", LINKS, threads, d_t, d_s);
  run<3>("v_readlane -> SGPR -> v_add (VALU-SALU-VALU)", LINKS, threads, d_t, d_s);
  run<4>("v_cmp -> s_cmp_lg_u64 -> s_cselect -> v_add", LINKS, threads, d_t, d_s);
  run<10>("v_cmp -> s_ff1 -> v_readlane(lane) -> v_or", LINKS, threads, d_t, d_s);
  run<5>("6-step fused DPP max + readlane + v_xor", 4, threads, d_t, d_s);
  run<6>("taken s_cbranch (skips 4 s_nop)", LINKS, threads, d_t, d_s);
  run<11>("taken s_cbranch over 32 instructions (128 B)", LINKS, threads, d_t, d_s);
  run<12>("taken s_cbranch over 256 instructions (1 KB)", LINKS, threads, d_t, d_s);
  run<7>("not-taken s_cbranch", LINKS, threads, d_t, d_s);
  run<8>("ds_write + barrier + ds_read", 4, threads, d_t, d_s);
  return 0;
}
