// Farthest point sampling (K5): one 1024-thread workgroup, points and their running
// min-distance resident in VGPRs (interleaved: point i lives in thread i % 1024,
// slot i / 1024), one barrier per selected point.  The dependent chain
// (m - 1 argmax steps) is the reason this is a single-CU kernel: a grid barrier
// costs more than a whole step does here (MI355X_MICROARCH.md barrier-xcd row).
//
// Arithmetic pinned to oracle/cluster.py: d = ((dx*dx + dy*dy) + dz*dz) (no FMA:
// -ffp-contract=off), running min, first (lowest-index) argmax.
#include "common.hpp"

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / 64;

// wave64 all-lanes -> lane 63 reduction on the DPP network (no LDS traffic).
template <bool IS_MAX>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned v) {
#define OCC4D_DPP_STEP(ctrl, rmask)                                                        \
  {                                                                                        \
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); \
    v = IS_MAX ? max(v, o) : min(v, o);                                                    \
  }
  OCC4D_DPP_STEP(0xB1, 0xf)   // quad_perm [1,0,3,2]
  OCC4D_DPP_STEP(0x4E, 0xf)   // quad_perm [2,3,0,1]
  OCC4D_DPP_STEP(0x141, 0xf)  // row_half_mirror
  OCC4D_DPP_STEP(0x140, 0xf)  // row_mirror      -> every lane holds its row's result
  OCC4D_DPP_STEP(0x142, 0xa)  // row_bcast15 into rows 1,3
  OCC4D_DPP_STEP(0x143, 0xc)  // row_bcast31 into rows 2,3 -> lane 63 holds the wave result
#undef OCC4D_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// PPT (even) points per thread, held as PPT/2 float2 pairs so that the distance update maps to
// packed fp32 VALU ops (v_pk_add_f32 / v_pk_mul_f32: two points per instruction).
template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int64_t stride, int n,
                                                          int m, int32_t* __restrict__ out_sorted,
                                                          int32_t* __restrict__ out_order) {
  static_assert(PPT % 2 == 0, "PPT must be even");
  constexpr int PP = PPT / 2;
  __shared__ unsigned s_d[2][FPS_WAVES];
  __shared__ unsigned s_i[2][FPS_WAVES];
  __shared__ float s_p[2][FPS_WAVES][4];
  __shared__ unsigned s_flags[FPS_THREADS];  // bitmask of selected points (n <= 32768)
  __shared__ int s_scan[FPS_THREADS];

  const int t = threadIdx.x;
  const int wave = t >> 6;
  f32x2 px[PP], py[PP], pz[PP], md[PP];
#pragma unroll
  for (int u = 0; u < PP; ++u) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = t + FPS_THREADS * (2 * u + v);
      float x = 0.f, y = 0.f, z = 0.f, d0 = -1.f;   // d0 = -1: never wins (live running mins are >= 0)
      if (i < n) {
        const float* p = xyz + (int64_t)i * stride;
        x = p[0]; y = p[1]; z = p[2];
        d0 = __builtin_inff();
      }
      px[u][v] = x; py[u][v] = y; pz[u][v] = z; md[u][v] = d0;
    }
  }
  s_flags[t] = 0u;
  __syncthreads();
  float cx = xyz[0], cy = xyz[1], cz = xyz[2];
  if (t == 0) {
    s_flags[0] = 1u;
    if (out_order) out_order[0] = 0;
  }

  int par = 0;
  for (int it = 1; it < m; ++it) {
    float bd = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
    unsigned bi = 0xffffffffu;
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const f32x2 dx = px[u] - c2x, dy = py[u] - c2y, dz = pz[u] - c2z;
      const f32x2 d = (dx * dx + dy * dy) + dz * dz;          // -ffp-contract=off: no FMA
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float old = md[u][v];
        const float nv = d[v] < old ? d[v] : old;              // dead slots: old = -1 stays
        md[u][v] = nv;
        if (nv > bd) {  // strict: slots ascend in index, so the lowest index wins ties
          bd = nv; bi = (unsigned)(t + FPS_THREADS * (2 * u + v));
          bx = px[u][v]; by = py[u][v]; bz = pz[u][v];
        }
      }
    }
    // bd >= 0 for live candidates, so its bit pattern orders like the float; dead lanes map to 0.
    const unsigned dbits = bd >= 0.f ? __float_as_uint(bd) + 1u : 0u;
    const unsigned wmax = wave_reduce_u32<true>(dbits);
    const unsigned cand = (dbits == wmax) ? bi : 0xffffffffu;
    const unsigned wmin = wave_reduce_u32<false>(cand);
    if (dbits == wmax && bi == wmin) {  // exactly one lane per wave (or a dead wave: bi = ~0)
      s_d[par][wave] = wmax; s_i[par][wave] = wmin;
      s_p[par][wave][0] = bx; s_p[par][wave][1] = by; s_p[par][wave][2] = bz;
    }
    __syncthreads();
    unsigned gd = s_d[par][0], gi = s_i[par][0];
    int gw = 0;
#pragma unroll
    for (int w = 1; w < FPS_WAVES; ++w) {
      const unsigned d2 = s_d[par][w], i2 = s_i[par][w];
      if (d2 > gd || (d2 == gd && i2 < gi)) { gd = d2; gi = i2; gw = w; }
    }
    cx = s_p[par][gw][0]; cy = s_p[par][gw][1]; cz = s_p[par][gw][2];
    if (t == 0) {
      s_flags[gi >> 5] |= 1u << (gi & 31);
      if (out_order) out_order[it] = (int)gi;
    }
    par ^= 1;
  }
  __syncthreads();

  // stream-compact the selection mask into ascending indices (block-wide scan of popcounts)
  const unsigned word = s_flags[t];
  const int cnt = __popc(word);
  s_scan[t] = cnt;
  __syncthreads();
  for (int off = 1; off < FPS_THREADS; off <<= 1) {
    const int add = (t >= off) ? s_scan[t - off] : 0;
    __syncthreads();
    s_scan[t] += add;
    __syncthreads();
  }
  int pos = s_scan[t] - cnt;
  unsigned wbits = word;
  while (wbits) {
    const int b = __ffs(wbits) - 1;
    wbits &= wbits - 1;
    if (pos < m) out_sorted[pos] = t * 32 + b;
    ++pos;
  }
}

}  // namespace

extern "C" int occ4d_fps_f32(const float* xyz, int64_t stride, int n, int m, int32_t* out_sorted,
                             int32_t* out_order, void* stream) {
  OCC4D_REQUIRE(xyz && out_sorted, "occ4d_fps_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32768, "occ4d_fps_f32: n=%d outside [1,32768]", n);
  OCC4D_REQUIRE(m >= 1 && m <= n, "occ4d_fps_f32: m=%d outside [1,n=%d]", m, n);
  OCC4D_REQUIRE(stride >= 3, "occ4d_fps_f32: stride=%lld < 3", (long long)stride);
  hipStream_t st = (hipStream_t)stream;
  const int ppt = occ4d::cdiv(n, FPS_THREADS);
#define OCC4D_FPS(P) fps_kernel<P><<<1, FPS_THREADS, 0, st>>>(xyz, stride, n, m, out_sorted, out_order)
  if (ppt <= 2) OCC4D_FPS(2);
  else if (ppt <= 6) OCC4D_FPS(6);
  else if (ppt <= 10) OCC4D_FPS(10);
  else if (ppt <= 14) OCC4D_FPS(14);
  else if (ppt <= 28) OCC4D_FPS(28);
  else OCC4D_FPS(32);
#undef OCC4D_FPS
  return occ4d::check_launch("occ4d_fps_f32");
}
