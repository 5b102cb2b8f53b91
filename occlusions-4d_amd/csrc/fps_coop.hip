// Farthest point sampling over MANY workgroups (the dataloader's reduction of a
// ~172 K-point clip to n_points, utils/geometry.py:353-364; also the 28 672-point training clouds).
// fps.hip keeps a cloud in ONE workgroup's registers (n <= 32768); here up to 16 workgroups each own a
// contiguous index range in registers and agree on every sample through a tagged all-to-all in global memory:
//
//   per step   update own running min-distances (packed fp32) -> per-wave DPP argmax -> LDS -> wave 0 reduces the
//              workgroup's candidate and PUBLISHES it as four self-validating 8-byte words
//              {tag | dist bits | ~index}, {tag | x}, {tag | y}, {tag | z} with relaxed agent-scope atomic stores;
//              wave 0 POLLS all 4 G words with one relaxed agent-scope 8-byte load per lane until every word
//              carries this step's tag, reduces the G candidates on the DPP network, broadcasts the winner
//              through LDS.  No fence, no counter, no read-modify-write: 8-byte agent atomics on both sides are
//              a valid hand-off on gfx950 (MI355X_MICROARCH.md, inter-workgroup visibility), ~1 us per exchange.
//   rounds     (round 4) an exchange yields SEVERAL samples: a workgroup also publishes a floor (the second largest of
//              its running minima) and every workgroup runs the same candidate round on the G candidates -- see the
//              loop below; 2.1 -> 1.1 us per sample at 28672 points with 16 workgroups, indices unchanged.
//   tags       two slot buffers alternate by step; the 1-bit tag flips each time a buffer is reused.  A slot
//              can only ever hold this use's value or the previous use's (opposite tag): a workgroup cannot
//              publish step j+2 before every workgroup has finished polling step j (it needs all of j+1).
//              The workspace is reset (tag = 1 everywhere, status = 0) by a small kernel per call.
//   safety     every spin is bounded; on time-out the status word is set to 1 and the kernel finishes quickly with
//              undefined indices.  Workgroups that are not yet resident only delay the others (no deadlock as long
//              as other kernels on the device terminate).  Round 5: NO consumer ever sees those indices -- for clouds
//              the single-workgroup kernel can hold (n <= 32768: every training cloud) the entry point enqueues
//              occ4d_fps_repair_f32 right behind the cooperative kernel: a one-workgroup launch that reads the status
//              word, returns at once when it is 0 and otherwise recomputes the whole selection (fps.hip: the same
//              arithmetic and tie rule, identical indices) and sets the status to 2 ("repaired": a warning on the
//              host, never an error).  In stream order, so it also works inside a captured graph.  Larger clouds
//              (the dataloader's whole clips) keep status 1; their host wrapper retries the launch.
//
// Arithmetic and tie rule identical to fps.hip / oracle/cluster.py: d = ((dx*dx + dy*dy) + dz*dz), running
// min, first (lowest-index) argmax -- the selected indices are bit-identical for any workgroup count.
#include "common.hpp"

namespace {

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int MAX_WG = 16;
constexpr int SLOT_WORDS = 4 * MAX_WG;          // u64 words per buffer
constexpr int WS_WORDS = 2 * SLOT_WORDS + 1;    // two buffers + status
constexpr int FLAG_WORDS = 8192;                // selection bitmask (workgroup 0), n <= 262144
constexpr unsigned SPIN_LIMIT = 1u << 22;
// tests: occ4d_fps_coop_debug(spin_limit, fail_round) shortens the bounded spin / declares a time-out in a chosen round
unsigned g_spin_limit = SPIN_LIMIT;
int g_fail_round = -1;

template <bool IS_MAX>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned v) {
#define OCC4D_DPP_STEP(ctrl, rmask)                                                              \
  {                                                                                              \
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); \
    v = IS_MAX ? max(v, o) : min(v, o);                                                          \
  }
  OCC4D_DPP_STEP(0xB1, 0xf)
  OCC4D_DPP_STEP(0x4E, 0xf)
  OCC4D_DPP_STEP(0x141, 0xf)
  OCC4D_DPP_STEP(0x140, 0xf)
  OCC4D_DPP_STEP(0x142, 0xa)
  OCC4D_DPP_STEP(0x143, 0xc)
#undef OCC4D_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int PPT, int T>
__global__ __launch_bounds__(T) void fps_coop_kernel(const float* __restrict__ xyz, int64_t stride, int n, int m,
                                                     int start, int chunk, int32_t* __restrict__ out_sorted,
                                                     int32_t* __restrict__ out_order, u64* __restrict__ ws,
                                                     unsigned spin_limit, int fail_round) {
  static_assert(PPT % 2 == 0, "PPT must be even");
  constexpr int PP = PPT / 2;
  constexpr int NW = T / 64;
  __shared__ u64 s_key[NW];
  __shared__ float s_p[NW][4];
  __shared__ unsigned s_flags[FLAG_WORDS];
  __shared__ int s_cnt[T];

  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const int G = gridDim.x, g = blockIdx.x;
  const int base = g * chunk;
  f32x2 px[PP], py[PP], pz[PP], md[PP];
#pragma unroll
  for (int u = 0; u < PP; ++u) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int li = t + T * (2 * u + v);
      const int i = base + li;
      float x = 0.f, y = 0.f, z = 0.f, d0 = -1.f;   // dead slots never win
      if (li < chunk && i < n) {
        const float* p = xyz + (int64_t)i * stride;
        x = p[0]; y = p[1]; z = p[2];
        d0 = __builtin_inff();
      }
      px[u][v] = x; py[u][v] = y; pz[u][v] = z; md[u][v] = d0;
    }
  }
  if (g == 0)
    for (int w = t; w < FLAG_WORDS; w += T) s_flags[w] = 0u;
  __syncthreads();
  // ---- rounds.  A round = one all-to-all; it yields the next sample and every FURTHER sample that can be proven from
  // what was exchanged (round 4; csrc/fps_bucket.hip's rule): beside its candidate (its point of largest running
  // minimum, exact value) a workgroup publishes a FLOOR -- an upper bound of the running minima of all its OTHER points
  // (the second largest, exactly).  After the exchange every workgroup holds the same G candidates and F = the largest
  // floor, and runs the same candidate round: the best candidate is the next sample (as before); the others are lowered by
  // their distance to it (the very expression the point update will compute), and the best of them is accepted too iff it
  // is STRICTLY above F -- then it is above every point nobody has seen, whose minima only shrink -- and so on, lowest
  // index first among equal candidates.  A tie with F ends the round (an unseen point of lower index could sit there).
  // The accepted samples are applied to the points together at the start of the next round.
  __shared__ float s_acc[2][MAX_WG][4];      // the samples accepted by a round (x, y, z), read by the next one
  __shared__ int s_nacc[2];
  __shared__ unsigned s_fl[NW];
  if (t == 0) {
    const float* p = xyz + (int64_t)start * stride;
    s_acc[0][0][0] = p[0]; s_acc[0][0][1] = p[1]; s_acc[0][0][2] = p[2];
    s_nacc[0] = 1;
    if (g == 0) {
      s_flags[start >> 5] = 1u << (start & 31);
      if (out_order) out_order[0] = start;
    }
  }
  __syncthreads();
  bool dead = false;   // wave-0 uniform: a poll timed out
  auto enc = [](float d) { return d >= 0.f ? __float_as_uint(d) + 1u : 0u; };      // <= 0x7f800001: bit 31 is free (tag)
  auto dec = [](unsigned b) { return b ? __uint_as_float(b - 1u) : -1.f; };

  int it = 1;                                   // samples chosen so far
  for (int r = 0; it < m; ++r) {
    const int par = r & 1;
    const int na = s_nacc[par];
    float b1 = -1.f, b2 = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
    unsigned bi = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < PP; ++u) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const float x = px[u][v], y = py[u][v], z = pz[u][v];
        float nv = md[u][v];
        for (int a = 0; a < na; ++a) {
          const float dx = x - s_acc[par][a][0], dy = y - s_acc[par][a][1], dz = z - s_acc[par][a][2];
          const float d = (dx * dx + dy * dy) + dz * dz;          // -ffp-contract=off: no FMA
          nv = d < nv ? d : nv;
        }
        md[u][v] = nv;
        const bool better = nv > b1;                              // strict: slots ascend in index
        b2 = better ? b1 : (nv > b2 ? nv : b2);
        b1 = better ? nv : b1;
        bi = better ? (unsigned)(base + t + T * (2 * u + v)) : bi;
        bx = better ? x : bx;
        by = better ? y : by;
        bz = better ? z : bz;
      }
    }
    const unsigned dbits = enc(b1);
    const unsigned wmax = wave_reduce_u32<true>(dbits);
    const unsigned cand = (dbits == wmax) ? bi : 0xffffffffu;
    const unsigned wmin = wave_reduce_u32<false>(cand);
    const bool iswin = dbits == wmax && bi == wmin;
    const unsigned wfl = wave_reduce_u32<true>(enc(iswin ? b2 : b1));    // everything of this wave but its winner
    if (iswin) {
      s_key[wave] = ((u64)wmax << 32) | (u64)(0xffffffffu - wmin);
      s_p[wave][0] = bx; s_p[wave][1] = by; s_p[wave][2] = bz;
    }
    if (lane == 0) s_fl[wave] = wfl;
    __syncthreads();
    if (wave == 0) {
      // this workgroup's candidate and floor
      const u64 k = lane < NW ? s_key[lane] : 0ull;
      const unsigned khi = (unsigned)(k >> 32), klo = (unsigned)k;
      const unsigned m1 = wave_reduce_u32<true>(khi);
      const unsigned m2 = wave_reduce_u32<true>(khi == m1 ? klo : 0u);
      const u64 hit = __ballot(lane < NW && khi == m1 && klo == m2);
      const int wv = hit ? __ffsll((long long)hit) - 1 : 0;
      const unsigned fw = lane < NW ? max(s_fl[lane], lane == wv ? 0u : khi) : 0u;
      const unsigned floor_wg = wave_reduce_u32<true>(fw);
      const u64 tag = (u64)((r >> 1) & 1) << 63;
      u64* slots = ws + (r & 1) * SLOT_WORDS;
      if (lane < 4) {
        u64 payload = lane == 0 ? (((u64)m1 << 32) | (u64)m2) : (u64)__float_as_uint(s_p[wv][lane - 1]);
        if (lane == 1) payload |= (u64)floor_wg << 32;             // (31 free bits above the x coordinate)
        __hip_atomic_store(slots + 4 * g + lane, payload | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // all-to-all: lane L watches word (L & 3) of workgroup (L >> 2)
      const bool watch = lane < 4 * G;
      u64 v = 0ull;
      if (!dead && r == fail_round) {              // (tests: a declared time-out)
        dead = true;
        if (lane == 0) __hip_atomic_store(ws + 2 * SLOT_WORDS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!dead) {
        unsigned spins = 0;
        for (;;) {
          if (watch) v = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ok = !watch || ((v ^ tag) >> 63) == 0ull;
          if (__ballot(ok) == ~0ull) break;
          if (++spins > spin_limit) {
            dead = true;
            if (lane == 0) __hip_atomic_store(ws + 2 * SLOT_WORDS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      // candidate round: lane i < G holds candidate i
      const unsigned vhi32 = (unsigned)(v >> 32) & 0x7fffffffu, vlo32 = (unsigned)v;
      const int src = min(4 * lane, 63);
      unsigned ckey = (unsigned)__shfl((int)vhi32, src);            // encoded running minimum
      const unsigned cidx = (unsigned)__shfl((int)vlo32, src);      // ~index
      const unsigned cfl = (unsigned)__shfl((int)vhi32, min(src + 1, 63));
      const float ccx = __int_as_float(__shfl((int)vlo32, min(src + 1, 63)));
      const float ccy = __int_as_float(__shfl((int)vlo32, min(src + 2, 63)));
      const float ccz = __int_as_float(__shfl((int)vlo32, min(src + 3, 63)));
      const bool isc = lane < G;
      if (!isc) ckey = 0u;
      const unsigned F = wave_reduce_u32<true>(isc ? cfl : 0u);
      float cm = dec(ckey);
      int nacc = 0;
      for (;;) {
        const unsigned M1 = wave_reduce_u32<true>(ckey);
        const unsigned M2 = wave_reduce_u32<true>((isc && ckey == M1) ? cidx : 0u);
        if (nacc > 0 && !(M1 > F)) break;                           // (M1 == 0: no candidate left)
        const u64 win = __ballot(isc && ckey == M1 && cidx == M2);
        const int lw = win ? __ffsll((long long)win) - 1 : 0;
        const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccx), lw));
        const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccy), lw));
        const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccz), lw));
        if (lane == 0) {
          s_acc[par ^ 1][nacc][0] = sx; s_acc[par ^ 1][nacc][1] = sy; s_acc[par ^ 1][nacc][2] = sz;
          if (g == 0) {
            const unsigned gi = min(0xffffffffu - M2, (unsigned)(n - 1));   // clamp: only reachable after a time-out
            s_flags[gi >> 5] |= 1u << (gi & 31);
            if (out_order) out_order[it + nacc] = (int)gi;
          }
        }
        ++nacc;
        if (it + nacc >= m || nacc >= G) break;
        // the remaining candidates against the sample just accepted (the point update's own expression)
        const float dx = ccx - sx, dy = ccy - sy, dz = ccz - sz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        cm = d < cm ? d : cm;
        ckey = (isc && lane != lw && ckey != 0u) ? enc(cm) : 0u;
        if (lane == lw) cm = -1.f;
      }
      if (lane == 0) s_nacc[par ^ 1] = nacc;
    }
    __syncthreads();
    it += s_nacc[par ^ 1];
  }
  if (g != 0) return;
  __syncthreads();

  // ascending indices from the selection mask (workgroup 0)
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

__global__ void fps_coop_reset_kernel(u64* __restrict__ ws) {
  for (int i = threadIdx.x; i < WS_WORDS; i += blockDim.x) ws[i] = i < 2 * SLOT_WORDS ? ~0ull : 0ull;
}

template <int T>
int launch(int G, int ppt, const float* xyz, int64_t stride, int n, int m, int start, int chunk, int32_t* os,
           int32_t* oo, u64* ws, hipStream_t st) {
#define OCC4D_FPSC(P) \
  fps_coop_kernel<P, T><<<G, T, 0, st>>>(xyz, stride, n, m, start, chunk, os, oo, ws, g_spin_limit, g_fail_round)
  if (ppt <= 2) OCC4D_FPSC(2);
  else if (ppt <= 4) OCC4D_FPSC(4);
  else if (ppt <= 8) OCC4D_FPSC(8);
  else if (ppt <= 16) OCC4D_FPSC(16);
  else return -1;
#undef OCC4D_FPSC
  return 0;
}

}  // namespace

extern "C" {

int64_t occ4d_fps_coop_workspace_bytes(void) { return (int64_t)WS_WORDS * 8; }

int occ4d_fps_coop_f32(const float* xyz, int64_t stride, int n, int m, int start, int n_workgroups,
                       int32_t* out_sorted, int32_t* out_order, void* workspace, void* stream) {
  OCC4D_REQUIRE(xyz && out_sorted && workspace, "occ4d_fps_coop_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && n <= 32 * FLAG_WORDS, "occ4d_fps_coop_f32: n=%d outside [1,%d]", n, 32 * FLAG_WORDS);
  OCC4D_REQUIRE(m >= 1 && m <= n, "occ4d_fps_coop_f32: m=%d outside [1,n=%d]", m, n);
  OCC4D_REQUIRE(start >= 0 && start < n, "occ4d_fps_coop_f32: start=%d outside [0,n=%d)", start, n);
  OCC4D_REQUIRE(stride >= 3, "occ4d_fps_coop_f32: stride=%lld < 3", (long long)stride);
  OCC4D_REQUIRE(n_workgroups >= 0 && n_workgroups <= MAX_WG, "occ4d_fps_coop_f32: n_workgroups=%d outside [0,%d]",
                n_workgroups, MAX_WG);
  OCC4D_REQUIRE(((uintptr_t)workspace % 8) == 0, "occ4d_fps_coop_f32: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int T = n <= 65536 ? 512 : 1024;
  int G = n_workgroups ? n_workgroups : occ4d::cdiv(n, 2 * T);   // 0 = automatic: >= 2 points per thread
  if (G > MAX_WG) G = MAX_WG;
  if (G < 1) G = 1;
  const int chunk = occ4d::cdiv(n, G);
  const int ppt = occ4d::cdiv(chunk, T);
  u64* ws = (u64*)workspace;
  // every tag = 1 (nothing published yet), status = 0 (ok; 1 = a poll timed out) -- by a kernel, not hipMemsetAsync: a
  // memset node is not replayed reliably from a captured graph on this runtime
  fps_coop_reset_kernel<<<1, 256, 0, st>>>(ws);
  int rc = T == 512 ? launch<512>(G, ppt, xyz, stride, n, m, start, chunk, out_sorted, out_order, ws, st)
                    : launch<1024>(G, ppt, xyz, stride, n, m, start, chunk, out_sorted, out_order, ws, st);
  OCC4D_REQUIRE(rc == 0, "occ4d_fps_coop_f32: n=%d does not fit %d workgroups of %d threads", n, G, T);
  rc = occ4d::check_launch("occ4d_fps_coop_f32");
  if (rc) return rc;
  // no consumer sees a timed-out selection: the single-workgroup kernel, gated on the status word (see the header)
  if (n <= 32768) return occ4d_fps_repair_f32(xyz, stride, n, m, start, out_sorted, out_order, ws + 2 * SLOT_WORDS, stream);
  return OCC4D_OK;
}

int occ4d_fps_coop_debug(unsigned spin_limit, int fail_round) {
  g_spin_limit = spin_limit ? spin_limit : SPIN_LIMIT;
  g_fail_round = fail_round;
  return OCC4D_OK;
}

}  // extern "C"
