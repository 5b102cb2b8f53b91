// Farthest point sampling (K5): ONE workgroup, points and their running min-distance resident
// in VGPRs (interleaved: point i lives in thread i % T, slot i / T), one barrier per selected
// point.  The m - 1 dependent argmax steps are why this is a single-CU kernel: a grid barrier
// costs more than a whole step does (MI355X_MICROARCH.md, barrier-xcd row).
//
// Per step (cycle accounting: profiles/stamp_fps.py):
//   1. packed-fp32 distance update of the thread's points; running mins are kept as bit patterns (>= +0 or exactly
//      -1.0f for padding: those order like signed integers, so v_min_i32 / v_max_i32 need no canonicalize), per-group
//      maxima (groups of GS consecutive slots) and the thread's maximum;
//   2. wave maximum on the DPP network (fused v_max_i32_dpp);
//   3. the wave's lowest INDEX at that maximum (index = t + T slot: lowest slot first, then the lowest lane), found
//      hierarchically: one compare + ballot per GROUP, then one per slot of the lowest hit group (the slots of a
//      wave-uniform group are read with a GPR-indexed v_mov: the running mins live in one register tuple);
//   4. the winner's coordinates come out of the register tuples the same way (uniform slot, v_readlane of the lane):
//      re-loading xyz[winner] from global memory after the barrier cost 650 .. 870 cycles of every step;
//   5. the wave publishes (key, xyz) to LDS, one barrier, every wave picks the block winner with a max tree over the
//      64-bit keys (distance bits << 32 | ~index << 8 | wave: larger distance first, then the lower index).
//
// Arithmetic pinned to oracle/cluster.py: d = ((dx*dx + dy*dy) + dz*dz) (no FMA:
// -ffp-contract=off), running min, first (lowest-index) argmax.
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FLAG_WORDS = 1024;  // selection bitmask, n <= 32768

// wave64 signed-max reduction on the DPP network, one fused v_max_i32_dpp per step (the compiler's update_dpp
// lowering is mov + mov_dpp + max); s_nop 1 = the VALU-write -> DPP-read hazard.  Returned wave-uniform.
__device__ __forceinline__ int wave_max_i32(int v) {
#define OCC4D_DPP_STEP(ctrl) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl : "+v"(v));
  OCC4D_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")     // every lane holds its row's result
  OCC4D_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")   // into rows 1, 3
  OCC4D_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")   // into rows 2, 3: lane 63 holds the wave result
#undef OCC4D_DPP_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// The thread's coordinates and running mins live in ONE register tuple each (ext_vector of VL elements, VL >= PPT a
// tuple size the register file has: 2, 8, 16, 32), so that "slot s" with a wave-uniform s is a GPR-indexed v_mov
// (s_set_gpr_idx_on) instead of a branch tree over the slots.
template <typename E, int VL> struct vec_of { typedef E type __attribute__((ext_vector_type(VL))); };
constexpr int tuple_len(int ppt) { return ppt <= 16 ? 16 : 32; }   // (tuples of 8 are indexed with a compare-select chain instead)

// PPT (even) points per thread; pairs of slots -> v_pk_add_f32 / v_pk_mul_f32.
template <int PPT, int T>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int64_t stride, int n, int m,
                                                int start, int32_t* __restrict__ out_sorted,
                                                int32_t* __restrict__ out_order, u64* __restrict__ gate) {
  static_assert(PPT % 2 == 0, "PPT must be even");
  // repair launch behind the cooperative kernel (fps_coop.hip): nothing to do unless its status word says "timed out"
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return;
  constexpr int NW = T / 64;
  constexpr int NV = PPT <= 32 ? 1 : 2;                 // 56 points per thread: two tuples of 28 (in 32)
  constexpr int PV = PPT / NV;                          // slots per tuple
  constexpr int VL = tuple_len(PV);
  constexpr int GS = PPT % 4 == 0 ? 4 : 2;              // slots per group
  constexpr int NG = PPT / GS;
  static_assert(PV % GS == 0, "a group never straddles the two tuples");
  typedef typename vec_of<float, VL>::type fvec;
  typedef typename vec_of<int, VL>::type ivec;
  __shared__ u64 s_key[2][NW];
  __shared__ float4 s_c[2][NW];
  __shared__ unsigned s_flags[FLAG_WORDS];
  __shared__ int s_cnt[T];

  const int t = threadIdx.x;
  const int wave = t >> 6;
  fvec px[NV], py[NV], pz[NV];
  ivec md[NV];                                          // running min-distances (bit patterns)
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const int i = t + T * s;
    float x = 0.f, y = 0.f, z = 0.f, d0 = -1.f;         // d0 = -1: never wins, never changes (min(d, -1) = -1)
    if (i < n) {
      const float* p = xyz + (int64_t)i * stride;
      x = p[0]; y = p[1]; z = p[2];
      d0 = __builtin_inff();
    }
    px[s / PV][s % PV] = x; py[s / PV][s % PV] = y; pz[s / PV][s % PV] = z;
    md[s / PV][s % PV] = __float_as_int(d0);
  }
  for (int w = t; w < FLAG_WORDS; w += T) s_flags[w] = 0u;
  __syncthreads();
  const float* first = xyz + (int64_t)start * stride;
  float cx = first[0], cy = first[1], cz = first[2];
  if (t == 0) {
    out_sorted[0] = start;                          // (out_sorted carries the picks in selection order until the end)
    if (out_order) out_order[0] = start;
  }

#ifdef OCC4D_FPS_STAMP
  // per-wave cycle accounting (debug build): [0] distance update, [1] wave max + index search + winner coordinates,
  // [2] publish + barrier, [3] block winner
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define STAMP(i)
#endif
  int par = 0;
  for (int it = 1; it < m; ++it) {
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
    // (1) running-min update, group maxima, the thread's maximum
    int gm[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int top = (int)0x80000000;
#pragma unroll
      for (int s = g * GS; s < (g + 1) * GS; s += 2) {
        const int q = s / PV, e = s % PV;
        const f32x2 dx = f32x2{px[q][e], px[q][e + 1]} - c2x;
        const f32x2 dy = f32x2{py[q][e], py[q][e + 1]} - c2y;
        const f32x2 dz = f32x2{pz[q][e], pz[q][e + 1]} - c2z;
        const f32x2 d = (dx * dx + dy * dy) + dz * dz;          // -ffp-contract=off: no FMA; d >= +0 for finite input
        md[q][e] = min(__float_as_int(d[0]), md[q][e]);
        md[q][e + 1] = min(__float_as_int(d[1]), md[q][e + 1]);
        top = max(top, max(md[q][e], md[q][e + 1]));            // (v_max3_i32)
      }
      gm[g] = top;
    }
    int bd = gm[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) bd = max(bd, gm[g]);
    STAMP(0)
    // (2) wave maximum; < 0 = the wave holds padding only
    const int wtop = wave_max_i32(bd);
    // (3) lowest slot, then lowest lane, at the maximum.  Descending loops: the last assignment is the lowest.
    int hgroup = 0;
#pragma unroll
    for (int g = NG - 1; g >= 0; --g) hgroup = __ballot(gm[g] == wtop) ? g : hgroup;
    unsigned long long hit = 0ull;
    int hslot = 0;
#pragma unroll
    for (int s = GS - 1; s >= 0; --s) {
      const int slot = hgroup * GS + s;
      const int val = (NV == 1 || slot < PV) ? md[0][slot % PV] : md[NV - 1][slot - PV];     // (uniform register index)
      const unsigned long long mk = __ballot(val == wtop);
      hit = mk ? mk : hit;
      hslot = mk ? slot : hslot;
    }
    const int hlane = (int)__builtin_ctzll(hit | (1ull << 63));
    const unsigned wmin = (unsigned)((t & ~63) + hlane + T * hslot);
    // (4) the candidate's coordinates: slot hslot of every lane, then lane hlane of that
    float vx, vy, vz;
    if (NV == 1 || hslot < PV) {
      vx = px[0][hslot % PV]; vy = py[0][hslot % PV]; vz = pz[0][hslot % PV];
    } else {
      vx = px[NV - 1][hslot - PV]; vy = py[NV - 1][hslot - PV]; vz = pz[NV - 1][hslot - PV];
    }
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx), hlane));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy), hlane));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vz), hlane));
    STAMP(1)
    // (5) key: distance bits + 1 (0 = the wave holds padding only: loses against every real candidate, whatever its
    // low word says), then the LOWER index, then the wave (indices are unique)
    if ((t & 63) == 0) {   // (wave-uniform values)
      s_key[par][wave] = ((u64)(unsigned)max(wtop + 1, 0) << 32) | (u64)((0x7fff00u | (unsigned)wave) - ((wmin & 0x7fffu) << 8));
      s_c[par][wave] = float4{wx, wy, wz, 0.f};
    }
    __syncthreads();
    STAMP(2)
    u64 k[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) k[w] = s_key[par][w];
    const float4 mine = s_c[par][t & (NW - 1)];              // lane l holds wave l % NW's candidate
#pragma unroll
    for (int span = NW / 2; span > 0; span >>= 1)
#pragma unroll
      for (int w = 0; w < span; ++w) k[w] = k[w] > k[w + span] ? k[w] : k[w + span];
    const unsigned low = __builtin_amdgcn_readfirstlane((unsigned)(k[0] & 0xffffffffu));
    const unsigned gi = 0x7fffu - (low >> 8);
    const int ww = (int)(low & 0xffu);
    cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), ww));
    cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), ww));
    cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.z), ww));
    STAMP(3)
    if (t == 0) {
      // The pick goes to global memory only (fire-and-forget stores): an LDS flag update here put a full LDS round
      // trip in front of wave 0's next step (the loop header waits for lgkmcnt(0)), and wave 0 is the wave the
      // others then wait for at the barrier.  The selection mask is rebuilt from out_sorted after the loop.
      const unsigned g = min(gi, (unsigned)(n - 1));
      out_sorted[it] = (int)g;
      if (out_order) out_order[it] = (int)g;
    }
    par ^= 1;
  }
  __syncthreads();
#ifdef OCC4D_FPS_STAMP
  if ((t & 63) == 0 && out_order) {   // debug build: out_order has 8 * NW * 2 spare ints behind the (even-rounded) m entries
    unsigned long long* o = (unsigned long long*)(out_order + ((m + 1) & ~1)) + wave * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
  }
#endif

  // selection mask from the picks (written by thread 0 of this workgroup: __syncthreads orders them), then
  // stream-compact it into ascending indices: each thread owns a contiguous chunk of mask words; chunk totals are
  // scanned serially (executed once, T <= 1024 adds)
  for (int i = t; i < m; i += T) {
    const unsigned g = (unsigned)out_sorted[i];
    atomicOr(&s_flags[g >> 5], 1u << (g & 31));
  }
  __syncthreads();
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
  if (gate && t == 0) __hip_atomic_store(gate, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "repaired"
}

template <int T>
int launch_threads(const float* xyz, int64_t stride, int n, int m, int start, int32_t* os, int32_t* oo, hipStream_t st,
                   u64* gate = nullptr) {
  const int ppt = occ4d::cdiv(n, T);
#define OCC4D_FPS(P) fps_kernel<P, T><<<1, T, 0, st>>>(xyz, stride, n, m, start, os, oo, gate)
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
  return occ4d_fps_start_f32(xyz, stride, n, m, 0, out_sorted, out_order, stream);
}

extern "C" int occ4d_fps_start_f32(const float* xyz, int64_t stride, int n, int m, int start, int32_t* out_sorted,
                                   int32_t* out_order, void* stream) {
  OCC4D_REQUIRE(xyz && out_sorted, "occ4d_fps_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32768, "occ4d_fps_f32: n=%d outside [1,32768]", n);
  OCC4D_REQUIRE(m >= 1 && m <= n, "occ4d_fps_f32: m=%d outside [1,n=%d]", m, n);
  OCC4D_REQUIRE(start >= 0 && start < n, "occ4d_fps_f32: start=%d outside [0,n=%d)", start, n);
  OCC4D_REQUIRE(stride >= 3, "occ4d_fps_f32: stride=%lld < 3", (long long)stride);
  hipStream_t st = (hipStream_t)stream;
  // Threads per workgroup: fewer waves = cheaper per-step reduce/broadcast, more points per thread.
  // (OCC4D_FPS_THREADS overrides for experiments.)
  // measured (profiles/time_fps.py): 4779 pts 256 / 512 threads = 0.76 / 0.90 us per step, 9558 pts 1.38 / 1.16,
  // 14336 pts 1.39 / 1.34 / 1.78 (1024); above 56 points per thread only 1024 threads fit
  static const int forced_threads = [] { const char* e = getenv("OCC4D_FPS_THREADS"); return e ? atoi(e) : 0; }();   // read once
  // 1536 .. 16384 points: the spatially pruned kernel (fps_bucket.hip), same indices.  OCC4D_FPS_PRUNE=0 keeps the
  // exhaustive kernel below (experiments: profiles/time_fps.py, profiles/stamp_fps.py).
  static const int prune = [] { const char* e = getenv("OCC4D_FPS_PRUNE"); return e ? atoi(e) : 1; }();
  if (prune && forced_threads <= 0 && occ4d::fps_bucket_launch(xyz, stride, n, m, start, out_sorted, out_order, st) == 0)
    return occ4d::check_launch("occ4d_fps_f32");
  int threads = n <= 28 * 256 ? 256 : (n <= 56 * 512 ? 512 : 1024);
  if (forced_threads > 0) threads = forced_threads;
  int rc;
  if (threads == 256) rc = launch_threads<256>(xyz, stride, n, m, start, out_sorted, out_order, st);
  else if (threads == 512) rc = launch_threads<512>(xyz, stride, n, m, start, out_sorted, out_order, st);
  else rc = launch_threads<1024>(xyz, stride, n, m, start, out_sorted, out_order, st);
  OCC4D_REQUIRE(rc == 0, "occ4d_fps_f32: n=%d does not fit %d threads", n, threads);
  return occ4d::check_launch("occ4d_fps_f32");
}

// The exhaustive single-workgroup kernel as a CONDITIONAL launch: runs only when *status != 0 (the cooperative kernel's
// "an inter-workgroup wait timed out"), overwrites out_sorted / out_order with the selection (same arithmetic, same tie
// rule: the indices the cooperative kernel would have produced) and leaves *status = 2.  Costs one empty launch otherwise.
extern "C" int occ4d_fps_repair_f32(const float* xyz, int64_t stride, int n, int m, int start, int32_t* out_sorted,
                                    int32_t* out_order, void* status, void* stream) {
  OCC4D_REQUIRE(xyz && out_sorted && status, "occ4d_fps_repair_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32768 && m >= 1 && m <= n && start >= 0 && start < n && stride >= 3,
                "occ4d_fps_repair_f32: n=%d (1 .. 32768), m=%d, start=%d", n, m, start);
  hipStream_t st = (hipStream_t)stream;
  const int threads = n <= 28 * 256 ? 256 : (n <= 56 * 512 ? 512 : 1024);
  int rc;
  if (threads == 256) rc = launch_threads<256>(xyz, stride, n, m, start, out_sorted, out_order, st, (u64*)status);
  else if (threads == 512) rc = launch_threads<512>(xyz, stride, n, m, start, out_sorted, out_order, st, (u64*)status);
  else rc = launch_threads<1024>(xyz, stride, n, m, start, out_sorted, out_order, st, (u64*)status);
  OCC4D_REQUIRE(rc == 0, "occ4d_fps_repair_f32: n=%d does not fit %d threads", n, threads);
  return occ4d::check_launch("occ4d_fps_repair_f32");
}
