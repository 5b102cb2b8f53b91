// Streaming brute-force top-k kNN (K1/K6/K8 of SURVEY.md §2.1).
// One thread per query; data points staged through LDS in tiles; the k best
// (distance, index) pairs live sorted in registers.  Distance arithmetic is
// pinned to the reference's CPU results (see occ4d.h); the translation unit is
// built with -ffp-contract=off so that only explicit fmaf() fuses.
#include "common.hpp"

namespace {

constexpr int KNN_BLOCK = 256;
constexpr int KNN_TILE = 1024;

template <int METRIC>
__device__ __forceinline__ float point_dist(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = qx - px, dy = qy - py, dz = qz - pz;
  if (METRIC == 0) {
    return (dx * dx + dy * dy) + dz * dz;
  } else {
    return sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  }
}

template <int K, int METRIC, typename IdxT>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                        const float* __restrict__ data, int64_t ds, int nd,
                                                        IdxT* __restrict__ out_idx, float* __restrict__ out_dist) {
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
  const bool live = qi < nq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    const float* q = query + (int64_t)qi * qs;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  float bd[K];
  int bi[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { bd[s] = __builtin_inff(); bi[s] = 0x7fffffff; }

  for (int base = 0; base < nd; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, nd - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += KNN_BLOCK) {
      const float* p = data + (int64_t)(base + t) * ds;
      sx[t] = p[0]; sy[t] = p[1]; sz[t] = p[2];
    }
    __syncthreads();
    if (live) {
      for (int t = 0; t < cnt; ++t) {
        const float d = point_dist<METRIC>(qx, qy, qz, sx[t], sy[t], sz[t]);
        if (d < bd[K - 1]) {  // strict: an equal distance keeps the earlier (lower) index
          bd[K - 1] = d;
          bi[K - 1] = base + t;
#pragma unroll
          for (int s = K - 1; s > 0; --s) {
            if (bd[s] < bd[s - 1]) {
              float td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
              int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
            }
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      out_idx[(int64_t)qi * K + s] = (IdxT)bi[s];
      if (out_dist) out_dist[(int64_t)qi * K + s] = bd[s];
    }
  }
}

template <int K>
int launch_k(const float* q, int64_t qs, int nq, const float* d, int64_t ds, int nd, int metric, void* oi,
             int i64, float* od, hipStream_t st) {
  dim3 grid(occ4d::cdiv(nq, KNN_BLOCK)), block(KNN_BLOCK);
  if (metric == 0) {
    if (i64) knn_kernel<K, 0, int64_t><<<grid, block, 0, st>>>(q, qs, nq, d, ds, nd, (int64_t*)oi, od);
    else knn_kernel<K, 0, int32_t><<<grid, block, 0, st>>>(q, qs, nq, d, ds, nd, (int32_t*)oi, od);
  } else {
    if (i64) knn_kernel<K, 1, int64_t><<<grid, block, 0, st>>>(q, qs, nq, d, ds, nd, (int64_t*)oi, od);
    else knn_kernel<K, 1, int32_t><<<grid, block, 0, st>>>(q, qs, nq, d, ds, nd, (int32_t*)oi, od);
  }
  return occ4d::check_launch("occ4d_knn_f32");
}

}  // namespace

extern "C" int occ4d_knn_f32(const float* query, int64_t q_stride, int n_query, const float* data,
                             int64_t d_stride, int n_data, int k, int metric, void* out_idx,
                             int idx_is_i64, float* out_dist, void* stream) {
  OCC4D_REQUIRE(k >= 1 && k <= 16, "occ4d_knn_f32: k=%d outside [1,16]", k);
  OCC4D_REQUIRE(n_data >= k, "occ4d_knn_f32: n_data=%d < k=%d", n_data, k);
  OCC4D_REQUIRE(metric == 0 || metric == 1, "occ4d_knn_f32: metric=%d", metric);
  OCC4D_REQUIRE(n_query >= 0 && q_stride >= 3 && d_stride >= 3, "occ4d_knn_f32: bad sizes/strides");
  OCC4D_REQUIRE(query && data && out_idx, "occ4d_knn_f32: null pointer");
  if (n_query == 0) return OCC4D_OK;
  hipStream_t st = (hipStream_t)stream;
#define OCC4D_KNN_CASE(KK) \
  case KK: return launch_k<KK>(query, q_stride, n_query, data, d_stride, n_data, metric, out_idx, idx_is_i64, out_dist, st);
  switch (k) {
    OCC4D_KNN_CASE(1) OCC4D_KNN_CASE(2) OCC4D_KNN_CASE(3) OCC4D_KNN_CASE(4)
    OCC4D_KNN_CASE(5) OCC4D_KNN_CASE(6) OCC4D_KNN_CASE(7) OCC4D_KNN_CASE(8)
    OCC4D_KNN_CASE(9) OCC4D_KNN_CASE(10) OCC4D_KNN_CASE(11) OCC4D_KNN_CASE(12)
    OCC4D_KNN_CASE(13) OCC4D_KNN_CASE(14) OCC4D_KNN_CASE(15) OCC4D_KNN_CASE(16)
  }
#undef OCC4D_KNN_CASE
  return OCC4D_EINVAL;
}
