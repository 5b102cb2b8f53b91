// Device-side pre/post steps of perform_inference: query-grid generation
// (utils/geometry.py:1257-1283), density-threshold split with order-preserving stream compaction
// (eval/inference.py:279-287) and the compress_air argmax (:299-305).  HBM-bound, one pass each.
#include "common.hpp"

namespace {

constexpr int TPB = 256;

// points[i] = ((ix + 0.5) * sx + x0, (iy + 0.5) * sy + y0, (iz + 0.5) * sz + z0, t), x slowest, z fastest;
// fp32 arithmetic in numpy's order (float32 arange + 0.5, times float32 spacing, plus float32 minimum; no FMA).
__global__ __launch_bounds__(TPB) void grid_points_kernel(int nx, int ny, int nz, float x0, float sx, float y0, float sy,
                                                          float z0, float sz, float t, int64_t total,
                                                          float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int iz = (int)(i % nz);
  const int iy = (int)((i / nz) % ny);
  const int ix = (int)(i / ((int64_t)nz * ny));
  float4 v;
  v.x = ((float)ix + 0.5f) * sx + x0;
  v.y = ((float)iy + 0.5f) * sy + y0;
  v.z = ((float)iz + 0.5f) * sz + z0;
  v.w = t;
  reinterpret_cast<float4*>(out)[i] = v;
}

// pass 1: per block of 256 rows, number of kept rows (key >= threshold, or key > threshold when strict)
__global__ __launch_bounds__(TPB) void split_count_kernel(const float* __restrict__ dens, int64_t ld, int n,
                                                          float threshold, int strict, int* __restrict__ block_counts) {
  __shared__ int s_cnt[TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  const float kv = i < n ? dens[(int64_t)i * ld] : 0.f;
  const bool solid = i < n && (strict ? kv > threshold : kv >= threshold);
  const unsigned long long m = __ballot(solid);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// pass 2: exclusive scan of the block counts (single workgroup; nblocks <= a few thousand)
__global__ __launch_bounds__(1024) void split_scan_kernel(int* __restrict__ block_counts, int nblocks,
                                                          int* __restrict__ total_solid) {
  __shared__ int s[1024];
  int carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_counts[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) block_counts[i] = carry + s[threadIdx.x] - v;   // exclusive prefix
    const int chunk_total = s[1023];
    __syncthreads();
    carry += chunk_total;
  }
  if (threadIdx.x == 0) *total_solid = carry;
}

// pass 3: write rows in original order: solid -> (x,y,z,t, out[0..g)), air -> either the same layout or the
// compressed (x,y,z, density, argmax of the last n_cls channels)
__global__ __launch_bounds__(TPB) void split_write_kernel(const float* __restrict__ pts, const float* __restrict__ outp,
                                                          int64_t ld, int n, int g, float threshold,
                                                          const int* __restrict__ block_offsets, int compress,
                                                          int n_cls, float* __restrict__ solid,
                                                          float* __restrict__ air) {
  __shared__ int s_pre[TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = i < n;
  const bool is_solid = live && outp[(int64_t)i * ld] >= threshold;
  const unsigned long long m = __ballot(is_solid);
  if (lane == 0) s_pre[wave] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_pre[w];
  const int rank_solid = block_offsets[blockIdx.x] + before + __popcll(m & ((1ull << lane) - 1ull));
  if (!live) return;
  const float* p = pts + (int64_t)i * 4;
  const float* o = outp + (int64_t)i * ld;
  if (is_solid) {
    float* d = solid + (int64_t)rank_solid * (4 + g);
    d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3];
    for (int c = 0; c < g; ++c) d[4 + c] = o[c];
  } else {
    const int64_t rank_air = (int64_t)i - rank_solid;          // rows before i that are not solid
    if (compress) {
      float* d = air + rank_air * 5;
      // numpy argmax (first maximum) over the last n_cls columns of the concatenated row (x,y,z,t, out...); like
      // the reference's negative slice, a window wider than the outputs reaches into the coordinates
      const int width = 4 + g;
      const int start = width > n_cls ? width - n_cls : 0;
      int best = 0;
      float bv = start < 4 ? p[start] : o[start - 4];
      for (int c = start + 1; c < width; ++c) {
        const float v = c < 4 ? p[c] : o[c - 4];
        if (v > bv) { bv = v; best = c - start; }
      }
      d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = o[0]; d[4] = (float)best;
    } else {
      float* d = air + rank_air * (4 + g);
      d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = p[3];
      for (int c = 0; c < g; ++c) d[4 + c] = o[c];
    }
  }
}

// keep the rows whose key passes the threshold, in order: out_rows[rank] = src[i][0..d), out_key[rank] = key[i]
// (filter_air_solid_gap, utils/geometry.py:1190-1194)
__global__ __launch_bounds__(TPB) void compact_rows_kernel(const float* __restrict__ src, int64_t ld, int n, int d,
                                                           const float* __restrict__ key, int64_t ldk, float threshold,
                                                           int strict, const int* __restrict__ block_offsets,
                                                           float* __restrict__ out_rows, float* __restrict__ out_key) {
  __shared__ int s_pre[TPB / 64];
  const int i = blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float kv = i < n ? key[(int64_t)i * ldk] : 0.f;
  const bool keep = i < n && (strict ? kv > threshold : kv >= threshold);
  const unsigned long long m = __ballot(keep);
  if (lane == 0) s_pre[wave] = __popcll(m);
  __syncthreads();
  if (!keep) return;
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_pre[w];
  const int64_t rank = block_offsets[blockIdx.x] + before + __popcll(m & ((1ull << lane) - 1ull));
  const float* p = src + (int64_t)i * ld;
  float* o = out_rows + rank * d;
  for (int c = 0; c < d; ++c) o[c] = p[c];
  if (out_key) out_key[rank] = kv;
}

}  // namespace

extern "C" {

int occ4d_grid_points_f32(int nx, int ny, int nz, float x0, float sx, float y0, float sy, float z0, float sz, float t,
                          float* out, void* stream) {
  OCC4D_REQUIRE(out && nx >= 1 && ny >= 1 && nz >= 1, "occ4d_grid_points_f32: bad arguments");
  OCC4D_REQUIRE(((uintptr_t)out % 16) == 0, "occ4d_grid_points_f32: out must be 16-byte aligned");
  const int64_t total = (int64_t)nx * ny * nz;
  grid_points_kernel<<<occ4d::cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(nx, ny, nz, x0, sx, y0, sy, z0, sz, t, total, out);
  return occ4d::check_launch("occ4d_grid_points_f32");
}

int occ4d_split_count_f32(const float* implicit_output, int64_t ld, int n, float threshold, int* block_counts,
                          int* total_solid, void* stream) {
  return occ4d_compact_count_f32(implicit_output, ld, n, threshold, 0, block_counts, total_solid, stream);
}

int occ4d_compact_count_f32(const float* key, int64_t ld, int n, float threshold, int strict, int* block_counts,
                            int* total_kept, void* stream) {
  OCC4D_REQUIRE(key && block_counts && total_kept && n >= 0 && ld >= 1, "occ4d_compact_count_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nb = occ4d::cdiv(n, TPB);
  if (nb > 0) split_count_kernel<<<nb, TPB, 0, st>>>(key, ld, n, threshold, strict, block_counts);
  split_scan_kernel<<<1, 1024, 0, st>>>(block_counts, nb, total_kept);
  return occ4d::check_launch("occ4d_compact_count_f32");
}

int occ4d_compact_rows_f32(const float* src, int64_t ld, int n, int d, const float* key, int64_t ld_key,
                           float threshold, int strict, const int* block_offsets, float* out_rows, float* out_key,
                           void* stream) {
  OCC4D_REQUIRE(src && key && block_offsets && n >= 0 && d >= 1 && ld >= d && ld_key >= 1,
                "occ4d_compact_rows_f32: bad arguments");
  if (n == 0) return OCC4D_OK;
  compact_rows_kernel<<<occ4d::cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(src, ld, n, d, key, ld_key, threshold, strict,
                                                                           block_offsets, out_rows, out_key);
  return occ4d::check_launch("occ4d_compact_rows_f32");
}

int occ4d_split_write_f32(const float* points_query, const float* implicit_output, int64_t ld, int n, int g,
                          float threshold, const int* block_offsets, int compress_air, int n_classes, float* solid,
                          float* air, void* stream) {
  OCC4D_REQUIRE(points_query && implicit_output && block_offsets && n >= 0 && g >= 1 && ld >= g,
                "occ4d_split_write_f32: bad arguments");
  OCC4D_REQUIRE(!compress_air || n_classes >= 1, "occ4d_split_write_f32: bad n_classes");
  if (n == 0) return OCC4D_OK;
  split_write_kernel<<<occ4d::cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(points_query, implicit_output, ld, n, g, threshold,
                                                                          block_offsets, compress_air, n_classes, solid, air);
  return occ4d::check_launch("occ4d_split_write_f32");
}

}  // extern "C"
