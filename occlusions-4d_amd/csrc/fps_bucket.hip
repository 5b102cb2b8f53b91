// Farthest point sampling with spatial pruning (K5b): ONE workgroup, 1536 .. 16384 points.
//
// Same greedy selection as fps.hip (argmax of the running min-distance, lowest index on ties; distances
// ((dx*dx + dy*dy) + dz*dz) without FMA: oracle/cluster.py), but a step only touches the points that the new
// sample can change.  A prologue orders the points along a Morton curve (counting sort over 16^3 cells in LDS: the
// order inside a cell does not matter) and deals consecutive runs of 256 ordered points ("buckets") round-robin to
// the waves: bucket b is slots 4 (b / NW) .. + 3 of wave b % NW, four points per lane.  Per wave, lane u of seven VGPRs
// holds the bounding box of the wave's bucket u and an upper bound `bm` of the bucket's running min-distances.
//
// Per step, with c the last sample:
//   1. every wave tests its buckets at once, one per lane: box distance db = ((ex*ex + ey*ey) + ez*ez) with
//      e = max(lo - c, c - hi, 0).  fl() is monotone, so db <= the computed distance of every point in the box:
//      when db >= bm no running min in the bucket can change and the bucket is skipped (wave-uniform branch).
//      Typically 7 % of the buckets survive (profiles/README.md), spread over the waves by the round-robin deal.
//   2. a touched bucket updates its 128 running mins and recomputes bm (DPP wave reduction).
//   3. wave maximum = DPP reduction over the bm lanes; the buckets at that maximum are searched for the lowest
//      ORIGINAL index at that distance (the sort permutes the points, the tie rule does not change).
//   4. wave winners (key, xyz) go to LDS, one barrier, every wave picks the block winner.
//
// Several samples per round.  The serial chain of one round (wave maximum -> index search -> LDS -> barrier -> block
// winner) costs about as much as the arithmetic, so a round accepts every following sample it can PROVE: beside its
// winner a wave publishes a floor (an upper bound of the running mins of all its other points).  After the barrier
// every wave holds the NW candidates c_w with their exact running mins and F = the largest floor.  The block winner A
// is the next sample as before.  The candidates' running mins are then lowered by their distance to A (the same
// ((dx*dx + dy*dy) + dz*dz) expression the point update uses, so bit-identical to what the update will store); if the
// best of them is STRICTLY above F it is above every other point's running min (those only shrink), so it is exactly the
// sample the next sequential step would pick (ties between candidates: lowest index, as always; a tie with F stops the
// round because an unseen point of lower index could sit at that distance).  Repeat until the test fails, up to CMAX
// samples.  The next round then applies all accepted samples: lane 8 c + u tests bucket u against sample c in one pass.
#include "common.hpp"

#include <cstdlib>

namespace {

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int FLAG_WORDS = 512;   // selection bitmask, n <= 16384
constexpr int IDX_BITS = 14;
constexpr unsigned IDX_MASK = (1u << IDX_BITS) - 1u;

// wave64 all-lanes reduction on the DPP network, one fused v_{max_i32,min_u32}_dpp per step (the compiler's
// update_dpp lowering is mov + mov_dpp + op); s_nop 1 = the VALU-write -> DPP-read hazard.  The result is returned
// wave-uniform.  OP 0: unsigned min, 1: signed max.
template <int OP>
__device__ __forceinline__ int wave_reduce(int v) {
#define OCC4D_DPP_STEP(ctrl)                                                                      \
  if (OP == 0) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 " ctrl : "+v"(v));               \
  else asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl : "+v"(v));
  OCC4D_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")     // every lane holds its row's result
  OCC4D_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")   // into rows 1, 3
  OCC4D_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")   // into rows 2, 3: lane 63 holds the wave result
#undef OCC4D_DPP_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// wave64 float min / max on the DPP network (fused), returned wave-uniform.
template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce_f32(float v) {
#define OCC4D_DPP_STEP(ctrl)                                                                      \
  if (IS_MAX) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl : "+v"(v));               \
  else asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl : "+v"(v));
  OCC4D_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
  OCC4D_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
  OCC4D_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef OCC4D_DPP_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <typename E, int VL> struct vec_of { typedef E type __attribute__((ext_vector_type(VL))); };
constexpr int tuple_len(int len) { return len <= 2 ? 2 : len <= 4 ? 4 : len <= 8 ? 8 : len <= 16 ? 16 : 32; }

__device__ __forceinline__ unsigned spread3(unsigned x) {   // 6 bits -> every third bit
  x = (x | (x << 8)) & 0x0300Fu;
  x = (x | (x << 4)) & 0x030C3u;
  x = (x | (x << 2)) & 0x09249u;
  return x;
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* s_red, int t, int nw) {
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, w) : fminf(v, w);
  }
  __syncthreads();
  if ((t & 63) == 0) s_red[t >> 6] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, s_red[w]) : fminf(r, s_red[w]);
  return r;
}

// NB buckets of 256 points per wave (four per lane), T threads.
template <int NB, int T>
__global__ __launch_bounds__(T) void fps_bucket_kernel(const float* __restrict__ xyz, int64_t stride, int n, int m,
                                                       int start, int32_t* __restrict__ out_sorted,
                                                       int32_t* __restrict__ out_order) {
  static_assert(NB >= 1 && NB <= 8, "buckets per wave: at most 8 (32 slots per lane)");
  constexpr int NW = T / 64;
  static_assert(NW == 8, "the candidate round below keeps candidate l % 8 in lane l");
  constexpr int CMAX = 8;                        // samples per round: lane 8 c + u pairs sample c with bucket u
  constexpr int REFRESH = 8;                     // rounds between exact bucket bounds (power of two)
  constexpr int SORT_MAX = 16384;
  constexpr int CELL_BITS = 4, CELLS = 1 << (3 * CELL_BITS);
  static_assert(CELLS == 8 * T, "the scan below gives every thread eight cells");
  __shared__ unsigned short s_sort[SORT_MAX];    // Morton-ordered position -> original index
  __shared__ unsigned s_tmp[SORT_MAX];           // original index -> (cell << 14 | rank inside the cell)
  __shared__ unsigned s_hist[CELLS];
  __shared__ unsigned s_wsum[NW];
  __shared__ u64 s_key[2][NW];
  __shared__ float4 s_c[2][NW];
  __shared__ unsigned s_flags[FLAG_WORDS];
  __shared__ int s_cnt[T];
  __shared__ float s_red[NW];

  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;

  // ---- prologue 1: bounding box of the cloud, Morton cells, counting sort ---------------------------------------
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = t; i < n; i += T) {
    const float* p = xyz + (int64_t)i * stride;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], p[a]);
      hi[a] = fmaxf(hi[a], p[a]);
    }
  }
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = block_reduce(lo[a], false, s_red, t, NW);
    hi[a] = block_reduce(hi[a], true, s_red, t, NW);
    const float ext = hi[a] - lo[a];
    scale[a] = ext > 0.f ? (float)(1 << CELL_BITS) / ext : 0.f;   // (NaN / inf extents only cost bucket quality)
  }
  for (int c = t; c < CELLS; c += T) s_hist[c] = 0u;
  __syncthreads();
  for (int i = t; i < n; i += T) {
    const float* p = xyz + (int64_t)i * stride;
    unsigned cell = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int q = max(0, min((1 << CELL_BITS) - 1, (int)((p[a] - lo[a]) * scale[a])));
      cell |= spread3((unsigned)q) << a;
    }
    const unsigned rank = atomicAdd(&s_hist[cell], 1u);    // (arrival order: any order inside a cell will do)
    s_tmp[i] = (cell << IDX_BITS) | rank;
  }
  __syncthreads();
  {
    // exclusive scan of the cell counts: eight cells per thread, wave scan, wave totals through LDS
    unsigned cnt[8], sum = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      cnt[k] = s_hist[8 * t + k];
      sum += cnt[k];
    }
    unsigned incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s_hist[8 * t + k] = base;
      base += cnt[k];
    }
  }
  __syncthreads();
  for (int i = t; i < n; i += T) {
    const unsigned cr = s_tmp[i];
    s_sort[s_hist[cr >> IDX_BITS] + (cr & IDX_MASK)] = (unsigned short)i;
  }
  __syncthreads();

  // ---- prologue 2: deal the buckets, per-bucket boxes -----------------------------------------------------------
  // One register tuple per array: a wave-uniform slot index is then a GPR-indexed v_mov (s_set_gpr_idx_on).
  constexpr int PPT = 4 * NB;
  typedef typename vec_of<float, tuple_len(PPT)>::type fvec;
  typedef typename vec_of<int, tuple_len(PPT)>::type ivec;
  typedef typename vec_of<int, tuple_len(2 * NB)>::type idvec;
  fvec px, py, pz;                               // slot 4 u + v = point v of bucket u
  ivec md;                                       // running min-distances as bit patterns: >= +0 or exactly -1.0f, and
                                                 // those order like signed integers (v_min_i32 needs no canonicalize)
  idvec id;                                      // original indices, 16 bits each: element s / 2 holds slots s, s + 1
  float bminx = 0.f, bminy = 0.f, bminz = 0.f, bmaxx = 0.f, bmaxy = 0.f, bmaxz = 0.f, bm = -1.f;
  const unsigned last = s_sort[n - 1];           // padding slots repeat a real point: boxes stay tight
  // (a rolled loop: the tuple elements are written with a GPR-indexed v_mov.  Unrolled, the register allocator held two
  // copies of every tuple across this prologue and spilled into the main loop.)
#pragma clang loop unroll(disable)
  for (int u = 0; u < NB; ++u) {
    const int pos = ((u * NW + wave) << 8) + 4 * lane;
    float c[4][3], d0[4];
    unsigned ids[2] = {0u, 0u};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool live = pos + v < n;
      const unsigned i = live ? (unsigned)s_sort[pos + v] : last;
      const float* p = xyz + (int64_t)i * stride;
      c[v][0] = p[0]; c[v][1] = p[1]; c[v][2] = p[2];
      d0[v] = live ? __builtin_inff() : -1.f;     // -1: never wins, never changes (min(d, -1) = -1)
      ids[v >> 1] |= (live ? i : 0xffffu) << (16 * (v & 1));
    }
    float bl[3], bh[3];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      px[4 * u + v] = c[v][0]; py[4 * u + v] = c[v][1]; pz[4 * u + v] = c[v][2];
      md[4 * u + v] = __float_as_int(d0[v]);
    }
    asm volatile("" : "+v"(ids[0]), "+v"(ids[1]));   // keep the indices packed
    id[2 * u] = (int)ids[0]; id[2 * u + 1] = (int)ids[1];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      bl[a] = wave_reduce_f32<false>(fminf(fminf(c[0][a], c[1][a]), fminf(c[2][a], c[3][a])));
      bh[a] = wave_reduce_f32<true>(fmaxf(fmaxf(c[0][a], c[1][a]), fmaxf(c[2][a], c[3][a])));
    }
    const bool any_live = ((u * NW + wave) << 8) < n;
    if ((lane & 7) == u) {                           // lane 8 c + u: bucket u (against accepted sample c, step 1)
      bminx = bl[0]; bminy = bl[1]; bminz = bl[2];
      bmaxx = bh[0]; bmaxy = bh[1]; bmaxz = bh[2];
      bm = any_live ? __builtin_inff() : -1.f;
    }
  }
  for (int w = t; w < FLAG_WORDS; w += T) s_flags[w] = 0u;
  __syncthreads();
  int gm[NB];                                    // the lane's maximum over a bucket's four slots
#pragma unroll
  for (int g = 0; g < NB; ++g) gm[g] = max(max(md[4 * g], md[4 * g + 1]), max(md[4 * g + 2], md[4 * g + 3]));
  const float* first = xyz + (int64_t)start * stride;
  float ccx = first[0], ccy = first[1], ccz = first[2];   // lane l: accepted sample l / 8 of the last round
  int nacc = 1;                                           // how many of them are live
  if (t == 0) {
    out_sorted[0] = start;                          // (out_sorted carries the picks in selection order until the end)
    if (out_order) out_order[0] = start;
  }

#ifdef OCC4D_FPSB_STAMP
  // per-wave cycle accounting (debug build): [0] box test, [1] bucket updates, [2] wave max, [3] index search +
  // coordinates + floor, [4] publish + barrier, [5] candidates from LDS, [6] candidate round, [7] accepted samples
  // from LDS; [8] rounds
  unsigned long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define STAMP(i)
#endif
  // 8-lane group reductions (candidate l % 8 in lane l): quad, quad, half-row mirror
#define OCC4D_DPP8(op, v)                                                                          \
  asm volatile("s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
               "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
               "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v))
  int par = 0, round = 0;
  for (int it = 1; it < m;) {
    // (0) every REFRESH rounds the bounds become exact again (in between they only go stale upwards: running mins
    // never grow, so a stale bound still is an upper bound and the skip test stays conservative)
    if (round != 0 && (round & (REFRESH - 1)) == 0) {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int top = wave_reduce<1>(gm[u]);
        bm = (lane & 7) == u ? __int_as_float(top) : bm;
      }
    }
    ++round;
    // (1) which of this wave's buckets can change: lane 8 c + u tests bucket u against accepted sample c
    const float ex = fmaxf(fmaxf(bminx - ccx, ccx - bmaxx), 0.f);
    const float ey = fmaxf(fmaxf(bminy - ccy, ccy - bmaxy), 0.f);
    const float ez = fmaxf(fmaxf(bminz - ccz, ccz - bmaxz), 0.f);
    const float db = (ex * ex + ey * ey) + ez * ez;
    const u64 touch = __ballot(db < bm && (lane >> 3) < nacc);   // lanes with l % 8 >= NB hold bm = -1: never set
    STAMP(0)
    // (2) update the touched buckets
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      u64 mu = (touch >> u) & 0x0101010101010101ull;
      if (mu) {
        do {
          const int src = (int)__builtin_ctzll(mu) ;            // lane 8 c (+ 0): sample c
          mu &= mu - 1;
          const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccx), src));
          const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccy), src));
          const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccz), src));
          const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
          for (int e = 4 * u; e < 4 * u + 4; e += 2) {
            const f32x2 dx = f32x2{px[e], px[e + 1]} - c2x;
            const f32x2 dy = f32x2{py[e], py[e + 1]} - c2y;
            const f32x2 dz = f32x2{pz[e], pz[e + 1]} - c2z;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;      // -ffp-contract=off: no FMA; d >= +0 for finite input
            md[e] = min(__float_as_int(d[0]), md[e]);
            md[e + 1] = min(__float_as_int(d[1]), md[e + 1]);
          }
        } while (mu);
        gm[u] = max(max(md[4 * u], md[4 * u + 1]), max(md[4 * u + 2], md[4 * u + 3]));
      }
    }
    STAMP(1)
    // (3) the wave's maximum; < 0 = the wave holds padding only
    int bd = gm[0];
#pragma unroll
    for (int g = 1; g < NB; ++g) bd = max(bd, gm[g]);
    const int wtop = wave_reduce<1>(bd);
    STAMP(2)
    // (4) the lowest ORIGINAL index at the maximum.  One compare + ballot per bucket, then per slot of the hit bucket;
    // a single hit (the usual case) is the answer, several hits (exact ties) take the exhaustive scan below.
    // (the hit COUNT is kept per lane on the vector side, off the scalar select chain: a dependent SALU op costs
    // 8 cycles, profiles/micro/chain_latency.hip)
    // Beside it, per lane, the largest running min that is NOT at the maximum (og over the buckets' lane maxima, sn
    // over the hit bucket's slots): with a single hit that covers every point of the wave but the winner.
    constexpr int NEG1 = (int)0xbf800000;           // -1.0f
    int hgroup = 0, lane_hits = 0, og = NEG1, sn = NEG1;
#pragma unroll
    for (int g = NB - 1; g >= 0; --g) {
      const bool e = gm[g] == wtop;
      hgroup = __ballot(e) ? g : hgroup;
      lane_hits += e ? 1 : 0;
      og = max(og, e ? NEG1 : gm[g]);
    }
    unsigned long long hit = 0ull;
    int hslot = 0, slot_hits = 0;
#pragma unroll
    for (int sl = 3; sl >= 0; --sl) {
      const int slot = hgroup * 4 + sl;
      const int v = md[slot];                                             // (uniform register index)
      const bool e = v == wtop;
      const unsigned long long mk = __ballot(e);
      hit = mk ? mk : hit;
      hslot = mk ? slot : hslot;
      slot_hits += e ? 1 : 0;
      sn = max(sn, e ? NEG1 : v);
    }
    // exactly one lane with exactly one bucket and one slot at the maximum?
    const bool single = __popcll(__ballot(lane_hits != 0)) == 1 && __ballot(lane_hits > 1 || slot_hits > 1) == 0ull;
    int hlane = (int)__builtin_ctzll(hit | (1ull << 63));
    unsigned widx;
    int wfloor = wtop;                             // ties: another point of the wave sits at the maximum itself
    if (wtop >= 0 && !single) {
      // ties: the lowest index over every slot at the maximum (the sort permuted the points, so slot / lane order is
      // not index order)
      unsigned cand = 0xffffffffu;
      int cslot = 0;
#pragma unroll
      for (int sl = 0; sl < PPT; ++sl) {
        const unsigned ids = (sl & 1) ? ((unsigned)id[sl >> 1] >> 16) : ((unsigned)id[sl >> 1] & 0xffffu);
        const bool better = md[sl] == wtop && ids < cand;
        cand = better ? ids : cand;
        cslot = better ? sl : cslot;
      }
      widx = (unsigned)wave_reduce<0>((int)cand);
      hlane = (int)__builtin_ctzll(__ballot(cand == widx) | (1ull << 63));
      hslot = __builtin_amdgcn_readlane(cslot, hlane);
    } else {
      const unsigned pair = (unsigned)__builtin_amdgcn_readlane(id[hslot >> 1], hlane);
      widx = (hslot & 1) ? (pair >> 16) : (pair & 0xffffu);
      if (wtop >= 0) wfloor = wave_reduce<1>(max(og, sn));
    }
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[hslot]), hlane));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[hslot]), hlane));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[hslot]), hlane));
    STAMP(3)
    // (5) key: distance bits + 1 (0 = the wave holds padding only: loses against every real candidate, whatever its
    // low word says), then the LOWER index, then the wave (indices are unique); the floor rides in the fourth float
    if (lane == 0) {
      s_key[par][wave] = ((u64)(unsigned)max(wtop + 1, 0) << 32) |
                         (u64)(((IDX_MASK << 8) | (unsigned)wave) - ((widx & IDX_MASK) << 8));
      s_c[par][wave] = float4{wx, wy, wz, __int_as_float(wfloor)};
    }
    __syncthreads();
    STAMP(4)
    // (6) the candidate round.  Lane l = 8 k + i holds candidate i (cmd: its running min) and D = its distance to
    // candidate k (the point update's expression), so that an accepted sample a lowers every candidate with one short
    // lane reduction: cmd_i = min(cmd_i, min over k of (k == a ? D : inf)).
    const int ci = lane & (NW - 1), ck = lane >> 3;
    const u64 key = s_key[par][ci];
    const float4 mine = s_c[par][ci], oth = s_c[par][ck];
    constexpr int TAKEN = (int)0x80000000, FAR = 0x7fffffff;
    const int khi = (int)(unsigned)(key >> 32);
    int cmd = khi ? khi - 1 : TAKEN;                               // (key 0: padding only, never a sample)
    const int cidx = (int)(IDX_MASK - (((unsigned)key >> 8) & IDX_MASK));
    int fl = __float_as_int(mine.w);
    OCC4D_DPP8("v_max_i32_dpp", fl);
    const int F = __builtin_amdgcn_readfirstlane(fl);
    int D;
    {
      const float dx = mine.x - oth.x, dy = mine.y - oth.y, dz = mine.z - oth.z;
      D = __float_as_int((dx * dx + dy * dy) + dz * dz);
    }
    STAMP(5)
    const int room = min(CMAX, m - it);
    unsigned alist = 0u;                                          // accepted candidates, four bits each
    int nnew = 0;
    // (fully unrolled, one forward exit per sample: the rolled loop's exits cost five taken branches per sample)
#pragma unroll
    for (int j = 0; j < CMAX; ++j) {
      int best = cmd;
      OCC4D_DPP8("v_max_i32_dpp", best);
      bool eq = cmd == best;
      unsigned m8 = (unsigned)__ballot(eq) & 0xffu;
      if (j == 0) {
        // the block winner, as always: lowest index among equals
        int low = eq ? cidx : FAR;
        OCC4D_DPP8("v_min_i32_dpp", low);
        eq = eq && cidx == low;
        m8 = (unsigned)__ballot(eq) & 0xffu;
      } else if (j >= room || !(__builtin_amdgcn_readfirstlane(best) > F) || (m8 & (m8 - 1u)) != 0u) {
        break;        // not provably the next sample (or two candidates tie: the next round resolves it by index)
      }
      const int a = __builtin_ctz(m8);
      alist |= (unsigned)a << (4 * j);
      nnew = j + 1;
      if (j == CMAX - 1) break;
      int red = ck == a ? D : FAR;
      asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(red));
      {
        const auto r16 = __builtin_amdgcn_permlane16_swap((unsigned)red, (unsigned)red, false, false);
        red = min((int)r16[0], (int)r16[1]);
        const auto r32 = __builtin_amdgcn_permlane32_swap((unsigned)red, (unsigned)red, false, false);
        red = min((int)r32[0], (int)r32[1]);                          // over k, for this lane's i
      }
      cmd = eq ? TAKEN : min(cmd, red);
    }
    STAMP(6)
    // the accepted samples for the next round's updates: lane 8 c + u takes sample c
    {
      const int asel = (int)((alist >> (4 * ck)) & 7u);
      const float4 acc = s_c[par][asel];
      ccx = acc.x; ccy = acc.y; ccz = acc.z;
      if (wave == 0 && ci == 0 && ck < nnew) {
        // The picks go to global memory only (fire-and-forget stores): an LDS flag update here put a full LDS round
        // trip in front of wave 0's next step (the loop header waits for lgkmcnt(0)), and wave 0 is the wave the
        // others then wait for at the barrier.  The selection mask is rebuilt from out_sorted after the loop.
        const unsigned gi = IDX_MASK - (((unsigned)s_key[par][asel] >> 8) & IDX_MASK);
        const unsigned g = min(gi, (unsigned)(n - 1));
        out_sorted[it + ck] = (int)g;
        if (out_order) out_order[it + ck] = (int)g;
      }
    }
    nacc = nnew;
    it += nnew;
    STAMP(7)
    par ^= 1;
  }
#undef OCC4D_DPP8
  __syncthreads();
#ifdef OCC4D_FPSB_STAMP
  if (lane == 0 && out_order) {   // debug build: out_order has 16 * NW * 2 spare ints behind the (even-rounded) m entries
    unsigned long long* o = (unsigned long long*)(out_order + ((m + 1) & ~1)) + wave * 16;
    tacc[8] = (unsigned long long)round;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = tacc[i];
  }
#endif

  // selection mask from the picks (written by thread 0 of this workgroup: __syncthreads orders them), then
  // stream-compact it into ascending indices
  for (int i = t; i < m; i += T) {
    const unsigned g = (unsigned)out_sorted[i];
    atomicOr(&s_flags[g >> 5], 1u << (g & 31));
  }
  __syncthreads();
  constexpr int CH = FLAG_WORDS / T;
  static_assert(CH >= 1, "T <= FLAG_WORDS");
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

}  // namespace

namespace occ4d {

// FPS_BUCKET_MIN_POINTS <= n <= 16384.  Returns -1 when n is outside that range.
int fps_bucket_launch(const float* xyz, int64_t stride, int n, int m, int start, int32_t* os, int32_t* oo,
                      hipStream_t st) {
  static const int min_points = [] { const char* e = getenv("OCC4D_FPS_BUCKET_MIN"); return e ? atoi(e) : FPS_BUCKET_MIN_POINTS; }();
  if (n < min_points || n > 16384) return -1;
  constexpr int T = 512;
  const int nb = cdiv(n, 256 * (T / 64));
#define OCC4D_FPSB(B) fps_bucket_kernel<B, T><<<1, T, 0, st>>>(xyz, stride, n, m, start, os, oo)
  if (nb <= 1) OCC4D_FPSB(1);
  else if (nb <= 2) OCC4D_FPSB(2);
  else if (nb <= 3) OCC4D_FPSB(3);
  else if (nb <= 4) OCC4D_FPSB(4);
  else if (nb <= 5) OCC4D_FPSB(5);
  else if (nb <= 6) OCC4D_FPSB(6);
  else if (nb <= 7) OCC4D_FPSB(7);
  else OCC4D_FPSB(8);
#undef OCC4D_FPSB
  return 0;
}

}  // namespace occ4d
