// "Is any target point within radius r of this query?" on a uniform grid -- the air / solid gap filter of the
// training-time point sampler (utils/geometry.py:1164-1196 filter_air_solid_gap, called six times per frame by
// GuidedImplicitPointSampler, :692, :956): the reference takes the 1-NN distance of every candidate to the WHOLE target
// cloud (my_knn_torch(..., 1), in <= 2^27-pair slices) and keeps the candidates whose distance exceeds the radius.  The
// sampler only uses the kept ROWS, i.e. the decision  min_j |q - p_j| > r  <=>  no p_j with |q - p_j| <= r, and that
// decision only needs the targets in the 27 cells around the query when the cell edge exceeds r: 1.1 G pair distances
// per call (20 K candidates x 57 K targets) become a few hundred per candidate.
//
// Exactness: every distance that is evaluated uses the streaming kNN kernel's METRIC 1 expression
// (sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))), d = q - p: csrc/knn.hip), so a target within r is found with exactly
// the value the brute-force minimum would have seen; a target that is not visited lies at least one whole cell away
// along some axis: the cell edge is r / 0.95 or more and cell coordinates are floor((p - origin) * inv_h) clamped to
// the grid (clamping is monotone: points within r of each other stay in adjacent cells), the fp32 rounding of a cell
// coordinate (~1e-5 cells) is far inside the 5 % margin.  Points outside the grid's box therefore need no special case.
//
// Build (per target cloud): bounding box + cell size on the device (no host round trip), cell histogram with atomics,
// one-workgroup exclusive scan, scatter of the points (as float4) into cell order.  GD^3 cells at most: a cloud larger
// than GD cells along an axis gets coarser cells (more candidates per query, same answers).
#include <stdlib.h>

#include "common.hpp"

namespace {

constexpr int GD = 64;                       // cells per axis at most
constexpr int GCELLS = GD * GD * GD;
constexpr int GT = 1024;                     // threads of the single-workgroup kernels

struct GridPlan {
  float ox, oy, oz, inv_h;
  int nx, ny, nz, n;
};

// workspace layout (bytes): [0, 32) GridPlan | counts / cursors (GCELLS ints) | starts (GCELLS + 1 ints) | points (n float4)
__host__ __device__ inline size_t off_counts() { return 32; }
__host__ __device__ inline size_t off_starts() { return off_counts() + (size_t)GCELLS * 4; }
__host__ __device__ inline size_t off_points() { return (off_starts() + (size_t)(GCELLS + 1) * 4 + 15) / 16 * 16; }

__device__ __forceinline__ int cell_of(const GridPlan& g, float x, float y, float z) {
  const int cx = max(0, min(g.nx - 1, (int)floorf((x - g.ox) * g.inv_h)));
  const int cy = max(0, min(g.ny - 1, (int)floorf((y - g.oy) * g.inv_h)));
  const int cz = max(0, min(g.nz - 1, (int)floorf((z - g.oz) * g.inv_h)));
  return (cz * g.ny + cy) * g.nx + cx;
}

__device__ __forceinline__ float block_minmax(float v, bool is_max, float* red, int t) {
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, w) : fminf(v, w);
  }
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < GT / 64; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
  return r;
}

// h_min > 0: the cell edge is at least h_min (radius tests).  per_cell > 0 (exact kNN): the edge that puts ~per_cell
// points into a cell of the cloud's box, cbrt(volume x per_cell / n) with degenerate extents widened to 0.1 % of the
// longest one -- a guess that only sets the speed: the search below widens its ring until the answer is proven.
__global__ __launch_bounds__(GT) void grid_plan_kernel(const float* __restrict__ xyz, int64_t stride, int n, float h_min,
                                                       float per_cell, GridPlan* __restrict__ plan,
                                                       int* __restrict__ counts) {
  __shared__ float red[GT / 64];
  const int t = threadIdx.x;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = t; i < n; i += GT) {
    const float* p = xyz + (int64_t)i * stride;
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
  }
  float ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = block_minmax(lo[a], false, red, t);
    hi[a] = block_minmax(hi[a], true, red, t);
    ext = fmaxf(ext, hi[a] - lo[a]);
  }
  // cell edge: at least h_min, and large enough that the longest axis fits GD cells (every thread computes the same plan)
  if (per_cell > 0.f) {
    const float eps = fmaxf(ext, 1e-30f) * 1e-3f;
    const float vol = fmaxf(hi[0] - lo[0], eps) * fmaxf(hi[1] - lo[1], eps) * fmaxf(hi[2] - lo[2], eps);
    h_min = cbrtf(vol * per_cell / (float)n);
  }
  float h = fmaxf(h_min, ext / (float)(GD - 1));
  if (!(h > 0.f) || !(h < __builtin_inff())) h = 1.f;            // (degenerate / non-finite clouds: one coarse grid)
  GridPlan g;
  g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
  g.inv_h = 1.0f / h;
  g.nx = max(1, min(GD, (int)floorf((hi[0] - lo[0]) * g.inv_h) + 1));
  g.ny = max(1, min(GD, (int)floorf((hi[1] - lo[1]) * g.inv_h) + 1));
  g.nz = max(1, min(GD, (int)floorf((hi[2] - lo[2]) * g.inv_h) + 1));
  g.n = n;
  for (int c = t; c < g.nx * g.ny * g.nz; c += GT) counts[c] = 0;    // (only the live cells are ever touched)
  if (t == 0) *plan = g;
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float* __restrict__ xyz, int64_t stride, int n,
                                                         const GridPlan* __restrict__ plan, int* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const GridPlan g = *plan;
  const float* p = xyz + (int64_t)i * stride;
  atomicAdd(counts + cell_of(g, p[0], p[1], p[2]), 1);
}

// exclusive scan of the live cells' counts (one workgroup, a contiguous chunk of cells per thread); the counts become
// the fill cursors (0)
__global__ __launch_bounds__(GT) void grid_scan_kernel(const GridPlan* __restrict__ plan, int* __restrict__ counts,
                                                       int* __restrict__ starts) {
  __shared__ int wsum[GT / 64];
  const int ncell = plan->nx * plan->ny * plan->nz;
  const int per = (ncell + GT - 1) / GT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c_lo = min(ncell, t * per), c_hi = min(ncell, c_lo + per);
  int local = 0;
  for (int c = c_lo; c < c_hi; ++c) local += counts[c];
  int incl = local;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - local;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int c = c_lo; c < c_hi; ++c) {
    const int v = counts[c];
    starts[c] = base;
    counts[c] = 0;
    base += v;
  }
  if (t == GT - 1) starts[ncell] = base;
}

__global__ __launch_bounds__(256) void grid_fill_kernel(const float* __restrict__ xyz, int64_t stride, int n,
                                                        const GridPlan* __restrict__ plan, int* __restrict__ cursors,
                                                        const int* __restrict__ starts, float4* __restrict__ pts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const GridPlan g = *plan;
  const float* p = xyz + (int64_t)i * stride;
  const int c = cell_of(g, p[0], p[1], p[2]);
  const int slot = atomicAdd(cursors + c, 1);
  pts[starts[c] + slot] = float4{p[0], p[1], p[2], __int_as_float(i)};        // (.w: the point's index, for the kNN search)
}

// far[i] = 1.0f when no target lies within r of query i (the rows the gap filter keeps), else 0.0f.
// Sixteen lanes per query: lanes 0 .. 8 fetch the point ranges of the nine (z, y) cell rows around the query (the up to
// three cells of a row are contiguous in cell order), then all sixteen stride through every range together -- the loads
// of a step are independent and coalesced; one lane per query ran a chain of dependent L2 round trips.
constexpr int FAR_TPQ = 16;
__global__ __launch_bounds__(256) void grid_far_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                       const GridPlan* __restrict__ plan, const int* __restrict__ starts,
                                                       const float4* __restrict__ pts, float r, float* __restrict__ far) {
  const int i = (blockIdx.x * 256 + threadIdx.x) / FAR_TPQ, sub = threadIdx.x % FAR_TPQ;
  const bool live = i < nq;
  const GridPlan g = *plan;
  const float* q = query + (int64_t)(live ? i : 0) * qs;
  const float qx = q[0], qy = q[1], qz = q[2];
  const int cx = max(0, min(g.nx - 1, (int)floorf((qx - g.ox) * g.inv_h)));
  const int cy = max(0, min(g.ny - 1, (int)floorf((qy - g.oy) * g.inv_h)));
  const int cz = max(0, min(g.nz - 1, (int)floorf((qz - g.oz) * g.inv_h)));
  int rs = 0, re = 0;                                  // this lane's row (sub < 9): z = cz - 1 + sub / 3, y = cy - 1 + sub % 3
  if (sub < 9) {
    const int z = cz - 1 + sub / 3, y = cy - 1 + sub % 3;
    if (z >= 0 && z < g.nz && y >= 0 && y < g.ny) {
      const int row = (z * g.ny + y) * g.nx;
      rs = starts[row + max(0, cx - 1)];
      re = starts[row + min(g.nx - 1, cx + 1) + 1];
    }
  }
  // (all eighteen exchanges before the first divergent loop: a shuffle reads nothing from a lane that is masked off)
  int ss[9], ee[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { ss[j] = __shfl(rs, j, FAR_TPQ); ee[j] = __shfl(re, j, FAR_TPQ); }
  bool found = false;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int s = ss[j], e = ee[j];
    for (int t = s + sub; t < e; t += FAR_TPQ) {
      const float4 p = pts[t];
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      const float d = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));        // csrc/knn.hip, METRIC 1
      found = found || d <= r;
    }
  }
  int any = found ? 1 : 0;                             // (plain |: every lane must take part in every exchange)
#pragma unroll
  for (int o = FAR_TPQ / 2; o > 0; o >>= 1) any |= __shfl_xor(any, o, FAR_TPQ);
  if (live && sub == 0) far[i] = any ? 0.f : 1.f;
}

// ---------------------------------------------------------------------------------------------------------------
// Exact k nearest neighbours on the same grid (csrc/knn.hip's contract: the k smallest (distance, index) pairs in
// lexicographic order, distances by that file's METRIC 0 / 1 expressions, so the lists are bit-identical to the
// brute-force scan's).  A query walks the cells around its own in rings of growing Chebyshev radius and keeps its k best in
// registers; after ring rho every unvisited point lies beyond one of the faces of the visited block that
// still has cells behind it, i.e. at least lb = (smallest distance from the query to such a face) away.  The search ends
// when the k-th best distance is STRICTLY below lb minus a margin of 1e-2 cell edges (cell coordinates are
// floor((p - origin) / h) in fp32: their rounding, ~1e-5 cells for coordinates of this size, is far inside the margin; strict, so that an unvisited
// point at exactly the k-th distance with a lower index cannot be missed), or when the block covers the grid.  A query
// outside the cloud's box starts from the clamped cell; its distances to the faces are then simply larger.
// Order inside a cell is arbitrary (atomics at build time): the comparison is lexicographic, so it does not matter.
// The brute-force kernel scans n_query x n_data pairs (28672 x 28672 of a training cloud: 2 ms on the whole chip); here
// a query sees a few hundred candidates.
// The same search with ONE thread per query, for searches with so many queries that the machine is full without
// splitting them (the decoder's 68812 queries of a training step): exact termination (the thread knows its own k-th
// distance), no merge.  profiles/r04_time_knn_grid.txt has both on every shape.
// Rounding slack of the face distances `ox + c h - q` (and of the cell a point was binned into): 8 ulp of the largest
// magnitude involved.  1 % of a cell (the fixed margin) covers it for normalised clouds; for coordinates that are large
// against their extent (|origin| / h >~ 1e5: un-normalised world coordinates) ulp(origin) approaches the cell
// edge, so the slack scales with the coordinates.  A larger margin can only cost a ring, never an error.
__device__ __forceinline__ float coord_slack(const GridPlan& g, float h, float qx, float qy, float qz) {
  const float big = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz)) +
                    fmaxf(fmaxf(fabsf(g.ox) + h * (float)g.nx, fabsf(g.oy) + h * (float)g.ny), fabsf(g.oz) + h * (float)g.nz);
  return 1e-6f * big;
}

template <int KT, int METRIC>
__global__ __launch_bounds__(64) void knn_grid1_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                      const GridPlan* __restrict__ plan, const int* __restrict__ starts,
                                                      const float4* __restrict__ pts, int k, int32_t* __restrict__ out_idx,
                                                      float* __restrict__ out_dist) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nq) return;
  const GridPlan g = *plan;
  const float h = 1.0f / g.inv_h;
  const float* q = query + (int64_t)i * qs;
  const float qx = q[0], qy = q[1], qz = q[2];
  const int cx = max(0, min(g.nx - 1, (int)floorf((qx - g.ox) * g.inv_h)));
  const int cy = max(0, min(g.ny - 1, (int)floorf((qy - g.oy) * g.inv_h)));
  const int cz = max(0, min(g.nz - 1, (int)floorf((qz - g.oz) * g.inv_h)));
  float bd[KT];
  int bi[KT];
#pragma unroll
  for (int s = 0; s < KT; ++s) { bd[s] = __builtin_inff(); bi[s] = 0x7fffffff; }
  auto visit = [&](int s0, int s1) {
    for (int t = s0; t < s1; ++t) {
      const float4 p = pts[t];
      const int id = __float_as_int(p.w);
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      const float d = METRIC == 0 ? (dx * dx + dy * dy) + dz * dz : sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
      if (d < bd[KT - 1] || (d == bd[KT - 1] && id < bi[KT - 1])) {
        bd[KT - 1] = d;
        bi[KT - 1] = id;
#pragma unroll
        for (int s = KT - 1; s > 0; --s) {
          if (bd[s] < bd[s - 1] || (bd[s] == bd[s - 1] && bi[s] < bi[s - 1])) {
            const float td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
            const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
          }
        }
      }
    }
  };
  const int rings = max(g.nx, max(g.ny, g.nz));
  for (int rho = 0; rho < rings; ++rho) {
    const int z0 = max(0, cz - rho), z1 = min(g.nz - 1, cz + rho);
    const int y0 = max(0, cy - rho), y1 = min(g.ny - 1, cy + rho);
    const int x0 = max(0, cx - rho), x1 = min(g.nx - 1, cx + rho);
    for (int z = z0; z <= z1; ++z) {
      for (int y = y0; y <= y1; ++y) {
        const int row = (z * g.ny + y) * g.nx;
        if (z - cz == rho || cz - z == rho || y - cy == rho || cy - y == rho) {
          visit(starts[row + x0], starts[row + x1 + 1]);              // a face of the block: the whole x range
        } else {                                                        // inside: only the two end cells of the row
          if (cx - rho >= 0) visit(starts[row + cx - rho], starts[row + cx - rho + 1]);
          if (rho > 0 && cx + rho <= g.nx - 1) visit(starts[row + cx + rho], starts[row + cx + rho + 1]);
        }
      }
    }
    float lb = __builtin_inff();
    if (cx - rho > 0) lb = fminf(lb, qx - (g.ox + (float)(cx - rho) * h));
    if (cx + rho < g.nx - 1) lb = fminf(lb, (g.ox + (float)(cx + rho + 1) * h) - qx);
    if (cy - rho > 0) lb = fminf(lb, qy - (g.oy + (float)(cy - rho) * h));
    if (cy + rho < g.ny - 1) lb = fminf(lb, (g.oy + (float)(cy + rho + 1) * h) - qy);
    if (cz - rho > 0) lb = fminf(lb, qz - (g.oz + (float)(cz - rho) * h));
    if (cz + rho < g.nz - 1) lb = fminf(lb, (g.oz + (float)(cz + rho + 1) * h) - qz);
    if (lb == __builtin_inff()) break;                                // the block covers the grid
    lb = fmaxf(0.f, lb - (1e-2f * h + coord_slack(g, h, qx, qy, qz)));
    const float kth = bd[k - 1];
    if (METRIC == 0 ? kth < lb * lb : kth < lb) break;
  }
#pragma unroll
  for (int s = 0; s < KT; ++s) {
    if (s < k) {
      out_idx[(int64_t)i * k + s] = min(bi[s], g.n - 1);               // (unfilled slot -- NaN / inf input --: in bounds)
      if (out_dist) out_dist[(int64_t)i * k + s] = bd[s];
    }
  }
}

constexpr int KG_TPQ = 16;                   // lanes per query
constexpr int KG_BLOCK = 256;                // 16 queries per workgroup
template <int KT, int METRIC>
__global__ __launch_bounds__(KG_BLOCK) void knn_grid_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                            const GridPlan* __restrict__ plan, const int* __restrict__ starts,
                                                            const float4* __restrict__ pts, int k, int32_t* __restrict__ out_idx,
                                                            float* __restrict__ out_dist) {
  // Sixteen lanes per query (one thread per query is a chain of ~100 dependent L2 round trips with < 2 waves per CU to
  // hide them: 0.3 ms whatever the size).  The lanes fetch the cell-row ranges of a ring in parallel and stride through
  // every range together; each keeps the best KT of ITS points, sorted.  After a ring the k-th distance of the union is
  // bounded from above without merging: by the smallest lane-level k-th, and -- every lane holding at least one point
  // means sixteen points no farther than the largest lane minimum -- by that maximum (k <= 16).  The bound is
  // conservative (a ring too many at worst), the answer exact: the sixteen lists are merged through LDS at the end.
  __shared__ float m_d[KG_BLOCK * (KT + 1)];
  __shared__ int m_i[KG_BLOCK * (KT + 1)];
  const int i = (blockIdx.x * KG_BLOCK + threadIdx.x) / KG_TPQ, sub = threadIdx.x % KG_TPQ;
  const bool live = i < nq;
  const GridPlan g = *plan;
  const float h = 1.0f / g.inv_h;
  const float* q = query + (int64_t)(live ? i : 0) * qs;
  const float qx = q[0], qy = q[1], qz = q[2];
  const int cx = max(0, min(g.nx - 1, (int)floorf((qx - g.ox) * g.inv_h)));
  const int cy = max(0, min(g.ny - 1, (int)floorf((qy - g.oy) * g.inv_h)));
  const int cz = max(0, min(g.nz - 1, (int)floorf((qz - g.oz) * g.inv_h)));
  float bd[KT];
  int bi[KT];
#pragma unroll
  for (int s = 0; s < KT; ++s) { bd[s] = __builtin_inff(); bi[s] = 0x7fffffff; }
  auto insert = [&](const float4 p) {
    const int id = __float_as_int(p.w);
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    const float d = METRIC == 0 ? (dx * dx + dy * dy) + dz * dz : sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    if (d < bd[KT - 1] || (d == bd[KT - 1] && id < bi[KT - 1])) {
      bd[KT - 1] = d;
      bi[KT - 1] = id;
#pragma unroll
      for (int s = KT - 1; s > 0; --s) {
        if (bd[s] < bd[s - 1] || (bd[s] == bd[s - 1] && bi[s] < bi[s - 1])) {
          const float td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
          const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
        }
      }
    }
  };
  auto visit = [&](int s0, int s1) {
    for (int t = s0 + sub; t < s1; t += KG_TPQ) insert(pts[t]);
  };
  const int rings = live ? max(g.nx, max(g.ny, g.nz)) : 0;
  for (int rho = 0; rho < rings; ++rho) {              // (uniform over the sixteen lanes of a query)
    const int z0 = max(0, cz - rho), z1 = min(g.nz - 1, cz + rho);
    const int y0 = max(0, cy - rho), y1 = min(g.ny - 1, cy + rho);
    const int x0 = max(0, cx - rho), x1 = min(g.nx - 1, cx + rho);
    const int ny_cnt = y1 - y0 + 1, n_rows = (z1 - z0 + 1) * ny_cnt;
    for (int base = 0; base < n_rows; base += KG_TPQ) {
      const int r = base + sub;
      int a0 = 0, a1 = 0, b0 = 0, b1 = 0;              // this lane's row: one range (a face of the block) or its two end cells
      if (r < n_rows) {
        const int z = z0 + r / ny_cnt, y = y0 + r % ny_cnt;
        const int row = (z * g.ny + y) * g.nx;
        if (z - cz == rho || cz - z == rho || y - cy == rho || cy - y == rho) {
          a0 = starts[row + x0]; a1 = starts[row + x1 + 1];
        } else {
          if (cx - rho >= 0) { a0 = starts[row + cx - rho]; a1 = starts[row + cx - rho + 1]; }
          if (rho > 0 && cx + rho <= g.nx - 1) { b0 = starts[row + cx + rho]; b1 = starts[row + cx + rho + 1]; }
        }
      }
      const int cnt = min(KG_TPQ, n_rows - base);
      for (int j = 0; j < cnt; ++j) {
        const int sa = __shfl(a0, j, KG_TPQ), ea = __shfl(a1, j, KG_TPQ);
        const int sb = __shfl(b0, j, KG_TPQ), eb = __shfl(b1, j, KG_TPQ);
        visit(sa, ea);
        visit(sb, eb);
      }
    }
    float lb = __builtin_inff();
    if (cx - rho > 0) lb = fminf(lb, qx - (g.ox + (float)(cx - rho) * h));
    if (cx + rho < g.nx - 1) lb = fminf(lb, (g.ox + (float)(cx + rho + 1) * h) - qx);
    if (cy - rho > 0) lb = fminf(lb, qy - (g.oy + (float)(cy - rho) * h));
    if (cy + rho < g.ny - 1) lb = fminf(lb, (g.oy + (float)(cy + rho + 1) * h) - qy);
    if (cz - rho > 0) lb = fminf(lb, qz - (g.oz + (float)(cz - rho) * h));
    if (cz + rho < g.nz - 1) lb = fminf(lb, (g.oz + (float)(cz + rho + 1) * h) - qz);
    if (lb == __builtin_inff()) break;                                // the block covers the grid
    lb = fmaxf(0.f, lb - (1e-2f * h + coord_slack(g, h, qx, qy, qz)));
    float lo = bd[k - 1], hi = bd[0];                                 // upper bounds of the union's k-th distance
#pragma unroll
    for (int o = KG_TPQ / 2; o > 0; o >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, o, KG_TPQ));
      hi = fmaxf(hi, __shfl_xor(hi, o, KG_TPQ));
    }
    const float kth = fminf(lo, hi);
    if (METRIC == 0 ? kth < lb * lb : kth < lb) break;
  }
  // ---- merge the sixteen sorted lists of each query (lexicographic on (distance, index)), as csrc/knn.hip does
#pragma unroll
  for (int s = 0; s < KT; ++s) {
    m_d[threadIdx.x * (KT + 1) + s] = bd[s];
    m_i[threadIdx.x * (KT + 1) + s] = bi[s];
  }
  __syncthreads();
  if (live && sub == 0) {
    int head[KG_TPQ];
#pragma unroll
    for (int u = 0; u < KG_TPQ; ++u) head[u] = 0;
    const int t0 = threadIdx.x;
    for (int s = 0; s < k; ++s) {
      float best_d = __builtin_inff();
      int best_i = 0x7fffffff, best_u = 0;
#pragma unroll
      for (int u = 0; u < KG_TPQ; ++u) {
        const int hd = head[u];
        const float d = hd < KT ? m_d[(t0 + u) * (KT + 1) + hd] : __builtin_inff();
        const int id = hd < KT ? m_i[(t0 + u) * (KT + 1) + hd] : 0x7fffffff;
        if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; best_u = u; }
      }
#pragma unroll
      for (int u = 0; u < KG_TPQ; ++u) head[u] += (u == best_u) ? 1 : 0;
      out_idx[(int64_t)i * k + s] = min(best_i, g.n - 1);              // (unfilled slot -- NaN / inf input --: in bounds)
      if (out_dist) out_dist[(int64_t)i * k + s] = best_d;
    }
  }
}

}  // namespace

extern "C" int64_t occ4d_radius_grid_workspace_bytes(int n) { return (int64_t)off_points() + (int64_t)(n > 0 ? n : 0) * 16; }

extern "C" int occ4d_radius_grid_build_f32(const float* xyz, int64_t stride, int n, float radius_max, void* workspace,
                                           void* stream) {
  OCC4D_REQUIRE(xyz && workspace && n >= 1 && stride >= 3, "occ4d_radius_grid_build_f32: bad arguments");
  OCC4D_REQUIRE(radius_max > 0.f && ((uintptr_t)workspace % 16) == 0,
                "occ4d_radius_grid_build_f32: radius_max must be > 0 and the workspace 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  GridPlan* plan = (GridPlan*)ws;
  int* counts = (int*)(ws + off_counts());
  int* starts = (int*)(ws + off_starts());
  float4* pts = (float4*)(ws + off_points());
  grid_plan_kernel<<<1, GT, 0, st>>>(xyz, stride, n, radius_max / 0.95f, 0.f, plan, counts);
  grid_count_kernel<<<occ4d::cdiv(n, 256), 256, 0, st>>>(xyz, stride, n, plan, counts);
  grid_scan_kernel<<<1, GT, 0, st>>>(plan, counts, starts);
  grid_fill_kernel<<<occ4d::cdiv(n, 256), 256, 0, st>>>(xyz, stride, n, plan, counts, starts, pts);
  return occ4d::check_launch("occ4d_radius_grid_build_f32");
}

extern "C" int occ4d_radius_far_f32(const float* query, int64_t qs, int nq, const void* workspace, float radius,
                                    float* far, void* stream) {
  OCC4D_REQUIRE(query && workspace && far && nq >= 0 && qs >= 3, "occ4d_radius_far_f32: bad arguments");
  OCC4D_REQUIRE(radius >= 0.f, "occ4d_radius_far_f32: radius must be >= 0 (and <= the radius_max the grid was built for)");
  if (nq == 0) return OCC4D_OK;
  const char* ws = (const char*)workspace;
  grid_far_kernel<<<occ4d::cdiv((int64_t)nq * FAR_TPQ, 256), 256, 0, (hipStream_t)stream>>>(
      query, qs, nq, (const GridPlan*)ws, (const int*)(ws + off_starts()), (const float4*)(ws + off_points()), radius, far);
  return occ4d::check_launch("occ4d_radius_far_f32");
}

// Exact kNN through the grid: builds the grid of `data` in `workspace` (occ4d_radius_grid_workspace_bytes(n_data)) and
// searches it.  Same results as occ4d_knn_f32 (int32 indices), bit for bit.
extern "C" int occ4d_knn_grid_f32(const float* query, int64_t q_stride, int n_query, const float* data, int64_t d_stride,
                                  int n_data, int k, int metric, int32_t* out_idx, float* out_dist, void* workspace,
                                  void* stream) {
  OCC4D_REQUIRE(k >= 1 && k <= 16, "occ4d_knn_grid_f32: k=%d outside [1,16]", k);
  OCC4D_REQUIRE(n_data >= k, "occ4d_knn_grid_f32: n_data=%d < k=%d", n_data, k);
  OCC4D_REQUIRE(metric == 0 || metric == 1, "occ4d_knn_grid_f32: metric=%d", metric);
  OCC4D_REQUIRE(n_query >= 0 && q_stride >= 3 && d_stride >= 3, "occ4d_knn_grid_f32: bad sizes/strides");
  OCC4D_REQUIRE(query && data && out_idx && workspace && ((uintptr_t)workspace % 16) == 0,
                "occ4d_knn_grid_f32: null pointer or workspace not 16-byte aligned");
  if (n_query == 0) return OCC4D_OK;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  GridPlan* plan = (GridPlan*)ws;
  int* counts = (int*)(ws + off_counts());
  int* starts = (int*)(ws + off_starts());
  float4* pts = (float4*)(ws + off_points());
  static const int one_from = [] { const char* e = getenv("OCC4D_KNN_GRID_ONE_THREAD_FROM"); return e ? atoi(e) : 32768; }();
  // points per cell (measured, profiles/r04_time_knn_grid.txt): sixteen lanes share a query's candidates, so coarser cells
  // (fewer row visits) win there; a single thread wants few candidates
  static const float forced = [] { const char* e = getenv("OCC4D_KNN_GRID_PER_CELL"); return e ? (float)atof(e) : 0.f; }();
  const float per_cell = forced > 0.f ? forced : (n_query >= one_from ? 2.f : 8.f);
  grid_plan_kernel<<<1, GT, 0, st>>>(data, d_stride, n_data, 0.f, per_cell, plan, counts);
  grid_count_kernel<<<occ4d::cdiv(n_data, 256), 256, 0, st>>>(data, d_stride, n_data, plan, counts);
  grid_scan_kernel<<<1, GT, 0, st>>>(plan, counts, starts);
  grid_fill_kernel<<<occ4d::cdiv(n_data, 256), 256, 0, st>>>(data, d_stride, n_data, plan, counts, starts, pts);
  if (n_query >= one_from) {
    const int grid = occ4d::cdiv(n_query, 64);
#define OCC4D_KG(KT, M) knn_grid1_kernel<KT, M><<<grid, 64, 0, st>>>(query, q_stride, n_query, plan, starts, pts, k, out_idx, out_dist)
    if (k <= 8) { if (metric == 0) OCC4D_KG(8, 0); else OCC4D_KG(8, 1); }
    else { if (metric == 0) OCC4D_KG(16, 0); else OCC4D_KG(16, 1); }
#undef OCC4D_KG
  } else {
    const int grid = occ4d::cdiv((int64_t)n_query * KG_TPQ, KG_BLOCK);
#define OCC4D_KG(KT, M) knn_grid_kernel<KT, M><<<grid, KG_BLOCK, 0, st>>>(query, q_stride, n_query, plan, starts, pts, k, out_idx, out_dist)
    if (k <= 8) { if (metric == 0) OCC4D_KG(8, 0); else OCC4D_KG(8, 1); }
    else { if (metric == 0) OCC4D_KG(16, 0); else OCC4D_KG(16, 1); }
#undef OCC4D_KG
  }
  return occ4d::check_launch("occ4d_knn_grid_f32");
}
