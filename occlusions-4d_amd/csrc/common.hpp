// Shared helpers for libocc4d.so (gfx950 only; no CUDA dual path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "occ4d.h"

namespace occ4d {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return OCC4D_ELAUNCH;
  }
  return OCC4D_OK;
}

#define OCC4D_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      occ4d::set_error(__VA_ARGS__);    \
      return OCC4D_EINVAL;              \
    }                                   \
  } while (0)

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// fps_bucket.hip: the pruned single-workgroup FPS (FPS_BUCKET_MIN_POINTS <= n <= 16384); -1 = n outside that range.
// Since a round of the pruned kernel accepts several samples (round 3) it also wins on the small encoder levels, whose
// step is bound by the serial argmax chain (profiles/time_fps.py, pruned vs exhaustive: 9558 points 1.84 vs 3.54 ms,
// 4779 points 0.91 vs 1.15 ms, 1593 points 0.27 vs 0.31 ms; 2049 uniform points in a half-empty second bucket row
// 0.42 vs 0.40 ms).  OCC4D_FPS_BUCKET_MIN overrides the threshold (ablation).
constexpr int FPS_BUCKET_MIN_POINTS = 1536;
int fps_bucket_launch(const float* xyz, int64_t stride, int n, int m, int start, int32_t* out_sorted,
                      int32_t* out_order, hipStream_t stream);


// wgrad16.hip: weight gradient for N a multiple of 416, K a multiple of 32 (>= 64), M >= 4096 (the wide decoder
// layers); false = the shape is not taken (backward.hip's wgrad_kernel then runs).
bool wgrad16_plan(int M, int N, int K, int* splits, int* m_per_split);
int wgrad16_launch(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K, int splits,
                   int m_per_split, float* part, float* part_b, int relu_x, hipStream_t stream);

// memops.hip: dst[i][0 .. d) = 0 / = src[i][0 .. d) for n rows with row strides (kernels, never hipMemset / hipMemcpy: those
// are not reliably replayed from a captured graph on this stack)
int zero_rows(float* dst, int64_t ld, int64_t n, int d, hipStream_t st);
int copy_rows(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t n, int d, hipStream_t st);

// memops.hip: compute units of the current device (cached; 256 if the query fails)
int cu_count();

// path.hip: OCC4D_F16W=1 (default 0): the fp16 scheme's attention layers on csrc/crossattn_f16w.hip (A/B)
bool f16w_enabled();
// path.hip: OCC4D_F16_RESBLOCK=0 (default 1): the fp16 scheme's residual blocks as two row-kernel launches instead of the fused
// block of csrc/resblock_f16x3.hip (A/B)
bool f16_resblock_enabled();

// path.hip: phase offset of the paired attention workgroups (units of s_sleep(127); OCC4D_CA16P_SKEW, default 6)
int attn16p_skew();

}  // namespace occ4d
