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

__global__ __launch_bounds__(GT) void grid_plan_kernel(const float* __restrict__ xyz, int64_t stride, int n, float h_min,
                                                       GridPlan* __restrict__ plan, int* __restrict__ counts) {
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
  pts[starts[c] + slot] = float4{p[0], p[1], p[2], 0.f};
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
  grid_plan_kernel<<<1, GT, 0, st>>>(xyz, stride, n, radius_max / 0.95f, plan, counts);
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
