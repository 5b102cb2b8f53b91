// Farthest point sampling (K5): ONE workgroup, points and their running min-distance resident
// in VGPRs (interleaved: point i lives in thread i % T, slot i / T), one barrier per selected
// point.  The m - 1 dependent argmax steps are why this is a single-CU kernel: a grid barrier
// costs more than a whole step does (MI355X_MICROARCH.md, barrier-xcd row).
//
// Per step: packed-fp32 distance update of the thread's points, per-thread best, two DPP wave
// reductions (max distance bits, then min index among the maxima), the wave winner publishes
// (key, xyz) to LDS, one barrier, every thread picks the block winner with a max tree over the
// per-wave 64-bit keys (distance bits << 32 | ~index : larger distance first, then lower index).
//
// Arithmetic pinned to oracle/cluster.py: d = ((dx*dx + dy*dy) + dz*dz) (no FMA:
// -ffp-contract=off), running min, first (lowest-index) argmax.
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FLAG_WORDS = 1024;  // selection bitmask, n <= 32768

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

// PPT (even) points per thread as PPT/2 float2 pairs -> v_pk_add_f32 / v_pk_mul_f32.
template <int PPT, int T>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int64_t stride, int n, int m,
                                                int32_t* __restrict__ out_sorted, int32_t* __restrict__ out_order) {
  static_assert(PPT % 2 == 0, "PPT must be even");
  constexpr int PP = PPT / 2;
  constexpr int NW = T / 64;
  __shared__ u64 s_key[2][NW];
  __shared__ unsigned s_flags[FLAG_WORDS];
  __shared__ int s_cnt[T];

  const int t = threadIdx.x;
  const int wave = t >> 6;
  f32x2 px[PP], py[PP], pz[PP], md[PP];
#pragma unroll
  for (int u = 0; u < PP; ++u) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = t + T * (2 * u + v);
      float x = 0.f, y = 0.f, z = 0.f, d0 = -1.f;   // d0 = -1: never wins (live running mins are >= 0)
      if (i < n) {
        const float* p = xyz + (int64_t)i * stride;
        x = p[0]; y = p[1]; z = p[2];
        d0 = __builtin_inff();
      }
      px[u][v] = x; py[u][v] = y; pz[u][v] = z; md[u][v] = d0;
    }
  }
  for (int w = t; w < FLAG_WORDS; w += T) s_flags[w] = 0u;
  __syncthreads();
  float cx = xyz[0], cy = xyz[1], cz = xyz[2];
  if (t == 0) {
    s_flags[0] = 1u;
    if (out_order) out_order[0] = 0;
  }

  int par = 0;
  for (int it = 1; it < m; ++it) {
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
    // (1) running-min update + the thread's maximum (value only: min / max, no selects)
    f32x2 bd2 = {-1.f, -1.f};
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const f32x2 dx = px[u] - c2x, dy = py[u] - c2y, dz = pz[u] - c2z;
      const f32x2 d = (dx * dx + dy * dy) + dz * dz;          // -ffp-contract=off: no FMA
      // (d is never NaN for finite input; dead slots hold -1 and stay: min(d, -1) = -1)
      md[u] = f32x2{fminf(d[0], md[u][0]), fminf(d[1], md[u][1])};
      bd2 = f32x2{fmaxf(bd2[0], md[u][0]), fmaxf(bd2[1], md[u][1])};
    }
    const float bd = fmaxf(bd2[0], bd2[1]);
    // bd >= 0 for live candidates, so its bit pattern orders like the float; dead lanes map to 0.
    const unsigned dbits = bd >= 0.f ? __float_as_uint(bd) + 1u : 0u;
    const unsigned wmax = wave_reduce_u32<true>(dbits);
    // (2) the wave's lowest INDEX among the points at that maximum (index = t + T slot: lowest slot first, then the
    // lowest lane): one compare per slot into a wave-wide ballot, the search over the ballots is scalar work.  (The
    // sequential "better than the best so far" scan with its index / xyz selects was 60 % of the step's VALU work.)
    const float wv = __uint_as_float(wmax - 1u);               // the maximum as a float (unused when wmax == 0)
    unsigned long long hit = 0ull;
    int hslot = 0;
#pragma unroll
    for (int u = PP - 1; u >= 0; --u) {
#pragma unroll
      for (int v = 1; v >= 0; --v) {
        const unsigned long long m = __ballot(md[u][v] == wv);
        hit = m ? m : hit;                                      // descending slots: the last assignment is the lowest slot
        hslot = m ? 2 * u + v : hslot;
      }
    }
    const unsigned wmin = (wmax != 0u && hit != 0ull)
                              ? (unsigned)((t & ~63) + (int)__builtin_ctzll(hit) + T * hslot) : 0xffffffffu;
    if ((t & 63) == 0) s_key[par][wave] = ((u64)wmax << 32) | (u64)(0xffffffffu - wmin);   // (wave-uniform values)
    __syncthreads();
    u64 k[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) k[w] = s_key[par][w];
#pragma unroll
    for (int span = NW / 2; span > 0; span >>= 1)
#pragma unroll
      for (int w = 0; w < span; ++w) k[w] = k[w] > k[w + span] ? k[w] : k[w + span];
    const unsigned gi = __builtin_amdgcn_readfirstlane(0xffffffffu - (unsigned)(k[0] & 0xffffffffu));
    {
      const float* pw = xyz + (int64_t)min(gi, (unsigned)(n - 1)) * stride;     // wave-uniform address: scalar loads
      cx = pw[0]; cy = pw[1]; cz = pw[2];
    }
    if (t == 0) {
      s_flags[gi >> 5] |= 1u << (gi & 31);
      if (out_order) out_order[it] = (int)gi;
    }
    par ^= 1;
  }
  __syncthreads();

  // stream-compact the selection mask into ascending indices: each thread owns a contiguous
  // chunk of mask words; chunk totals are scanned serially (executed once, T <= 1024 adds)
  constexpr int CH = FLAG_WORDS / T;
  int cnt = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) cnt += __popc(s_flags[t * CH + c]);
  s_cnt[t] = cnt;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < T; ++i) {
      const int c = s_cnt[i];
      s_cnt[i] = run;
      run += c;
    }
  }
  __syncthreads();
  int pos = s_cnt[t];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    unsigned wbits = s_flags[t * CH + c];
    while (wbits) {
      const int b = __ffs(wbits) - 1;
      wbits &= wbits - 1;
      if (pos < m) out_sorted[pos] = (t * CH + c) * 32 + b;
      ++pos;
    }
  }
}

template <int T>
int launch_threads(const float* xyz, int64_t stride, int n, int m, int32_t* os, int32_t* oo, hipStream_t st) {
  const int ppt = occ4d::cdiv(n, T);
#define OCC4D_FPS(P) fps_kernel<P, T><<<1, T, 0, st>>>(xyz, stride, n, m, os, oo)
  if (ppt <= 2) OCC4D_FPS(2);
  else if (ppt <= 6) OCC4D_FPS(6);
  else if (ppt <= 10) OCC4D_FPS(10);
  else if (ppt <= 14) OCC4D_FPS(14);
  else if (ppt <= 20) OCC4D_FPS(20);
  else if (ppt <= 28) OCC4D_FPS(28);
  else if (ppt <= 32) OCC4D_FPS(32);
  else if (ppt <= 56 && T <= 512) OCC4D_FPS(56);
  else return -1;
#undef OCC4D_FPS
  return 0;
}

}  // namespace

extern "C" int occ4d_fps_f32(const float* xyz, int64_t stride, int n, int m, int32_t* out_sorted,
                             int32_t* out_order, void* stream) {
  OCC4D_REQUIRE(xyz && out_sorted, "occ4d_fps_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32768, "occ4d_fps_f32: n=%d outside [1,32768]", n);
  OCC4D_REQUIRE(m >= 1 && m <= n, "occ4d_fps_f32: m=%d outside [1,n=%d]", m, n);
  OCC4D_REQUIRE(stride >= 3, "occ4d_fps_f32: stride=%lld < 3", (long long)stride);
  hipStream_t st = (hipStream_t)stream;
  // Threads per workgroup: fewer waves = cheaper per-step reduce/broadcast, more points per thread.
  // (OCC4D_FPS_THREADS overrides for experiments.)
  // measured: 14336 pts 1024/512/256 threads = 10.7/9.4/15.3 ms; above 56 points per thread only 1024 threads fit
  int threads = n <= 2048 ? 256 : (n <= 56 * 512 ? 512 : 1024);
  static const int forced_threads = [] { const char* e = getenv("OCC4D_FPS_THREADS"); return e ? atoi(e) : 0; }();   // read once
  if (forced_threads > 0) threads = forced_threads;
  int rc;
  if (threads == 256) rc = launch_threads<256>(xyz, stride, n, m, out_sorted, out_order, st);
  else if (threads == 512) rc = launch_threads<512>(xyz, stride, n, m, out_sorted, out_order, st);
  else rc = launch_threads<1024>(xyz, stride, n, m, out_sorted, out_order, st);
  OCC4D_REQUIRE(rc == 0, "occ4d_fps_f32: n=%d does not fit %d threads", n, threads);
  return occ4d::check_launch("occ4d_fps_f32");
}
