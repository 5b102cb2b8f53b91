// Streaming brute-force top-k kNN (K1/K6/K8 of SURVEY.md §2.1).
// Data points are staged through LDS in tiles; each query's k best (distance, index)
// pairs live sorted in registers.  TPQ threads cooperate on one query (each scans a
// strided share of every tile, the TPQ sorted lists are merged through LDS at the end):
// the encoder's self-kNN has only 14 336 queries, which would fill 56 of 256 CUs with
// one thread per query.  Distance arithmetic is pinned to the reference's CPU results
// (see occ4d.h); the translation unit is built with -ffp-contract=off so that only the
// explicit fmaf() fuses.  Ties: (distance, index) lexicographic, lowest index first.
#include <stdlib.h>

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

// KT = compile-time list length (8 or 16) >= runtime k: the first k of a sorted top-KT list
// are exactly the top-k.
template <int KT, int METRIC, typename IdxT, int TPQ>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                        const float* __restrict__ data, int64_t ds, int nd, int k,
                                                        IdxT* __restrict__ out_idx, float* __restrict__ out_dist) {
  constexpr int QPB = KNN_BLOCK / TPQ;   // queries per block
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  // merge scratch (TPQ > 1): per thread KT (dist, idx) pairs; +1 pad against bank conflicts
  __shared__ float m_d[TPQ > 1 ? KNN_BLOCK * (KT + 1) : 1];
  __shared__ int m_i[TPQ > 1 ? KNN_BLOCK * (KT + 1) : 1];

  const int ql = threadIdx.x / TPQ, sub = threadIdx.x % TPQ;
  const int qi = blockIdx.x * QPB + ql;
  const bool live = qi < nq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    const float* q = query + (int64_t)qi * qs;
    qx = q[0]; qy = q[1]; qz = q[2];
  }
  float bd[KT];
  int bi[KT];
#pragma unroll
  for (int s = 0; s < KT; ++s) { bd[s] = __builtin_inff(); bi[s] = 0x7fffffff; }

  for (int base = 0; base < nd; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, nd - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += KNN_BLOCK) {
      const float* p = data + (int64_t)(base + t) * ds;
      sx[t] = p[0]; sy[t] = p[1]; sz[t] = p[2];
    }
    __syncthreads();
    if (live) {
      for (int t = sub; t < cnt; t += TPQ) {
        const float d = point_dist<METRIC>(qx, qy, qz, sx[t], sy[t], sz[t]);
        if (d < bd[KT - 1]) {  // strict: an equal distance keeps the earlier (lower) index
          bd[KT - 1] = d;
          bi[KT - 1] = base + t;
#pragma unroll
          for (int s = KT - 1; s > 0; --s) {
            if (bd[s] < bd[s - 1]) {
              float td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
              int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
            }
          }
        }
      }
    }
  }

  if (TPQ == 1) {
    if (live) {
#pragma unroll
      for (int s = 0; s < KT; ++s) {
        if (s < k) {
          out_idx[(int64_t)qi * k + s] = (IdxT)min(bi[s], nd - 1);   // unfilled slot (NaN / inf input): in bounds
          if (out_dist) out_dist[(int64_t)qi * k + s] = bd[s];
        }
      }
    }
    return;
  }
  // ---- merge the TPQ sorted lists of each query (lexicographic on (distance, index))
  __syncthreads();
#pragma unroll
  for (int s = 0; s < KT; ++s) {
    m_d[threadIdx.x * (KT + 1) + s] = bd[s];
    m_i[threadIdx.x * (KT + 1) + s] = bi[s];
  }
  __syncthreads();
  if (live && sub == 0) {
    int head[TPQ];
#pragma unroll
    for (int u = 0; u < TPQ; ++u) head[u] = 0;
    const int t0 = ql * TPQ;
    for (int s = 0; s < k; ++s) {
      float best_d = __builtin_inff();
      int best_i = 0x7fffffff, best_u = 0;
#pragma unroll
      for (int u = 0; u < TPQ; ++u) {
        const int h = head[u];
        const float d = h < KT ? m_d[(t0 + u) * (KT + 1) + h] : __builtin_inff();
        const int i = h < KT ? m_i[(t0 + u) * (KT + 1) + h] : 0x7fffffff;
        if (d < best_d || (d == best_d && i < best_i)) { best_d = d; best_i = i; best_u = u; }
      }
#pragma unroll
      for (int u = 0; u < TPQ; ++u) head[u] += (u == best_u) ? 1 : 0;
      out_idx[(int64_t)qi * k + s] = (IdxT)min(best_i, nd - 1);
      if (out_dist) out_dist[(int64_t)qi * k + s] = best_d;
    }
  }
}

// distances of caller-supplied neighbour lists, in the search kernel's own expressions
template <int METRIC>
__global__ __launch_bounds__(KNN_BLOCK) void knn_dists_kernel(const float* __restrict__ query, int64_t qs, int nq,
                                                              const float* __restrict__ data, int64_t ds, int nd,
                                                              const int32_t* __restrict__ idx, int k,
                                                              float* __restrict__ out_dist) {
  const int64_t e = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (e >= (int64_t)nq * k) return;
  const float* q = query + (e / k) * qs;
  const float* p = data + (int64_t)min(max(idx[e], 0), nd - 1) * ds;
  out_dist[e] = point_dist<METRIC>(q[0], q[1], q[2], p[0], p[1], p[2]);
}

template <int KT, int METRIC, typename IdxT>
void launch_t(int tpq, const float* q, int64_t qs, int nq, const float* d, int64_t ds, int nd, int k, IdxT* oi,
              float* od, hipStream_t st) {
  dim3 block(KNN_BLOCK);
  if (tpq == 1) knn_kernel<KT, METRIC, IdxT, 1><<<occ4d::cdiv(nq, KNN_BLOCK), block, 0, st>>>(q, qs, nq, d, ds, nd, k, oi, od);
  else if (tpq == 4) knn_kernel<KT, METRIC, IdxT, 4><<<occ4d::cdiv(nq, KNN_BLOCK / 4), block, 0, st>>>(q, qs, nq, d, ds, nd, k, oi, od);
  else knn_kernel<KT, METRIC, IdxT, 16><<<occ4d::cdiv(nq, KNN_BLOCK / 16), block, 0, st>>>(q, qs, nq, d, ds, nd, k, oi, od);
}

template <int KT>
int launch_k(const float* q, int64_t qs, int nq, const float* d, int64_t ds, int nd, int k, int metric, void* oi,
             int i64, float* od, hipStream_t st) {
  // Threads per query (measured, profiles/time_knn.py): 4 threads per query halve the time even with plenty of
  // queries (32256 x 531, k 14: 128 -> 60 us: shorter per-thread scans, 4x the blocks); 16 per query when 64
  // queries per block would leave CUs idle (< 384 blocks): 4779 x 14336: 664 -> 325 us, 20000 x 57344 (the
  // sampler's 1-NN filter): 8.1 ms -> 1.15 ms.  Tiny data sets are never split.
  int tpq = 1;
  if (nd >= 256) tpq = (nq / 64 >= 384) ? 4 : 16;
  static const int forced_tpq = [] { const char* e = getenv("OCC4D_KNN_TPQ"); return e ? atoi(e) : 0; }();   // experiments; read once
  if (forced_tpq > 0) tpq = forced_tpq;
  if (metric == 0) {
    if (i64) launch_t<KT, 0, int64_t>(tpq, q, qs, nq, d, ds, nd, k, (int64_t*)oi, od, st);
    else launch_t<KT, 0, int32_t>(tpq, q, qs, nq, d, ds, nd, k, (int32_t*)oi, od, st);
  } else {
    if (i64) launch_t<KT, 1, int64_t>(tpq, q, qs, nq, d, ds, nd, k, (int64_t*)oi, od, st);
    else launch_t<KT, 1, int32_t>(tpq, q, qs, nq, d, ds, nd, k, (int32_t*)oi, od, st);
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
  if (k <= 8) return launch_k<8>(query, q_stride, n_query, data, d_stride, n_data, k, metric, out_idx, idx_is_i64, out_dist, st);
  return launch_k<16>(query, q_stride, n_query, data, d_stride, n_data, k, metric, out_idx, idx_is_i64, out_dist, st);
}

extern "C" int occ4d_knn_dists_f32(const float* query, int64_t q_stride, int n_query, const float* data,
                                   int64_t d_stride, int n_data, const int32_t* idx, int k, int metric, float* out_dist,
                                   void* stream) {
  OCC4D_REQUIRE(k >= 1 && n_data >= 1 && (metric == 0 || metric == 1) && n_query >= 0 && q_stride >= 3 && d_stride >= 3,
                "occ4d_knn_dists_f32: bad sizes (k = %d, n_data = %d, metric = %d)", k, n_data, metric);
  OCC4D_REQUIRE(query && data && idx && out_dist, "occ4d_knn_dists_f32: null pointer");
  if (n_query == 0) return OCC4D_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)occ4d::cdiv((int64_t)n_query * k, KNN_BLOCK);
  if (metric == 0) knn_dists_kernel<0><<<blocks, KNN_BLOCK, 0, st>>>(query, q_stride, n_query, data, d_stride, n_data, idx, k, out_dist);
  else knn_dists_kernel<1><<<blocks, KNN_BLOCK, 0, st>>>(query, q_stride, n_query, data, d_stride, n_data, idx, k, out_dist);
  return occ4d::check_launch("occ4d_knn_dists_f32");
}
