// CPU twin of the C ABI (include/occ4d.h), compiled by g++ -- SURVEY.md 8(b), last line of "What a C-ABI replacement
// must export": "each with a CPU twin compiled by g++ for config 1" (BASELINE configs[0]: "runs without a GPU").
//
// What this is: the SAME entry points, prototypes and contracts as libocc4d.so for the inference path (kNN, FPS, Linear,
// PointTransformerLayer / Block, DownTransition pooling, the LocalPclResnetFC decoder, post-ops, grid, split), written as
// plain host loops in the reference's AS-WRITTEN op order (model/point_transformer_layer.py:167-179,
// model/implicit.py:328-443, model/modules.py:126-158) -- no merged weights, no packed streams, no tables beyond the
// per-scene to_k / to_v rows.  `prepared` / `workspace` buffers are unused (their size queries return a token size);
// `stream` is ignored (every call is synchronous).  Pointers are HOST pointers.
//
// What this is NOT: a fallback.  libocc4d_cpu.so is never loaded unless the caller asks for it by name
// (occlusions4d_amd.cpu_twin.enable(), tests only); the product on a GPU box fails loudly without libocc4d.so and rejects
// CPU tensors.  Its job is to let BASELINE configs[0] run without a GPU and to let the REFERENCE's own caller
// (eval/inference.py:perform_inference) drive the product modules in the build container (tests/test_cpu_twin.py).
// Entry points outside the inference path (training kernels, packers, the A/B kernels) are not here; a call through the
// twin to one of them raises in Python.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "occ4d.h"

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return OCC4D_EINVAL;
}
#define REQ(cond, ...) \
  do {                 \
    if (!(cond)) return fail(__VA_ARGS__); \
  } while (0)
#define TRY(expr)              \
  do {                         \
    const int rc_ = (expr);    \
    if (rc_ != OCC4D_OK) return rc_; \
  } while (0)

inline float act(float v, int code) {      // 0 identity, 1 relu, 2 swish (model/implicit.py:46-64)
  if (code == 1) return v > 0.f ? v : 0.f;
  if (code == 2) return v * (1.f / (1.f + std::exp(-v)));
  return v;
}

inline float dot(const float* __restrict__ a, const float* __restrict__ b, int k) {
  float s = 0.f;
#pragma omp simd reduction(+ : s)
  for (int i = 0; i < k; ++i) s += a[i] * b[i];
  return s;
}

// y[0..n) = W (n, k; row stride ldw) x + b
inline void matvec(const float* w, int64_t ldw, const float* b, const float* x, int n, int k, float* y) {
  for (int o = 0; o < n; ++o) y[o] = dot(w + (int64_t)o * ldw, x, k) + (b ? b[o] : 0.f);
}

// the two distance expressions of occ4d_knn_f32
inline float dist_metric(const float* q, const float* p, int metric) {
  const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
  if (metric == 0) {
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;     // ((dx*dx + dy*dy) + dz*dz), fp32, no FMA (-ffp-contract=off)
    return (xx + yy) + zz;
  }
  return std::sqrt(std::fma(dz, dz, std::fma(dy, dy, dx * dx)));
}

// k nearest of one query: nearest first, lowest index on equal distances
void knn_one(const float* q, const float* data, int64_t ds, int n_data, int k, int metric, int32_t* idx, float* dist) {
  float bd[16];
  int32_t bi[16];
  int have = 0;
  for (int j = 0; j < n_data; ++j) {
    const float d = dist_metric(q, data + (int64_t)j * ds, metric);
    if (have == k && !(d < bd[k - 1])) continue;               // (ties keep the earlier index)
    int p = have < k ? have++ : k - 1;
    while (p > 0 && d < bd[p - 1]) {
      bd[p] = bd[p - 1];
      bi[p] = bi[p - 1];
      --p;
    }
    bd[p] = d;
    bi[p] = j;
  }
  for (int j = 0; j < k; ++j) {
    idx[j] = bi[j];
    if (dist) dist[j] = bd[j];
  }
}

// ---- one vector-attention query, as written (model/point_transformer_layer.py:168-179): q_i (D), its K neighbours'
// keys / values (rows of kf / vf) and positions -> agg (D)
struct AttnScratch {
  std::vector<float> r, pe, a, hid, logit;
  AttnScratch(int k, int d, int h) : r(h), pe((size_t)k * d), a(d), hid(2 * d), logit((size_t)k * d) {}
};
void attn_one(const occ4d_pt_layer_weights& w, const float* qi, const float* pi, const float* pos2, int64_t p2s,
              const float* kf, const float* vf, const int32_t* idx, int k, float divisor, AttnScratch& s, float* agg) {
  const int D = w.dim, h = w.pos_hidden;
  for (int j = 0; j < k; ++j) {
    const float* pj = pos2 + (int64_t)idx[j] * p2s;
    const float rel[3] = {pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2]};
    for (int u = 0; u < h; ++u) {
      const float v = (w.pos0_w[3 * u] * rel[0] + w.pos0_w[3 * u + 1] * rel[1]) + w.pos0_w[3 * u + 2] * rel[2] + w.pos0_b[u];
      s.r[u] = v > 0.f ? v : 0.f;
    }
    float* pe = s.pe.data() + (size_t)j * D;
    matvec(w.pos2_w, h, w.pos2_b, s.r.data(), D, h, pe);                     // pos_enc = pos_mlp(rel)          (:174)
    const float* kj = kf + (int64_t)idx[j] * D;
    for (int c = 0; c < D; ++c) s.a[c] = qi[c] - kj[c] + pe[c];              // q - k + pos_enc                 (:176)
    matvec(w.attn0_w, D, w.attn0_b, s.a.data(), 2 * D, D, s.hid.data());
    for (int c = 0; c < 2 * D; ++c) s.hid[c] = s.hid[c] > 0.f ? s.hid[c] : 0.f;
    matvec(w.attn2_w, 2 * D, w.attn2_b, s.hid.data(), D, 2 * D, s.logit.data() + (size_t)j * D);
  }
  for (int c = 0; c < D; ++c) {                                              // softmax over the neighbours, per channel (:177)
    float mx = -INFINITY;
    for (int j = 0; j < k; ++j) mx = std::max(mx, s.logit[(size_t)j * D + c] / divisor);
    float den = 0.f, num = 0.f;
    for (int j = 0; j < k; ++j) {
      const float e = std::exp(s.logit[(size_t)j * D + c] / divisor - mx);
      den += e;
      num += e * (vf[(int64_t)idx[j] * D + c] + s.pe[(size_t)j * D + c]);    // attn * (v + pos_enc)            (:179)
    }
    agg[c] = num / den;
  }
}

int check_layer(const occ4d_pt_layer_weights* w, const char* who) {
  REQ(w, "%s: null weights", who);
  REQ(w->dim >= 1 && w->pos_hidden >= 1 && w->dim2 >= 1, "%s: bad dimensions", who);
  REQ(w->to_q && w->to_k && w->to_v && w->pos0_w && w->pos0_b && w->pos2_w && w->pos2_b && w->attn0_w && w->attn0_b &&
          w->attn2_w && w->attn2_b,
      "%s: null parameter pointer", who);
  REQ(!w->post_w || (w->post_b && w->d_out == (w->pre_w ? w->d_in : w->dim)), "%s: layer3 + residual needs d_out == d_in", who);
  return OCC4D_OK;
}

int64_t up4(int64_t n) { return (n + 3) / 4 * 4; }

// scene tables of a cross layer in the twin: kf = to_k(x2) (m, D), vf = to_v(x2) (m, D)
void layer_tables(const occ4d_pt_layer_weights& w, const float* x2, int64_t ldx2, int m, float* kf, float* vf) {
  const int D = w.dim, D2 = w.dim2;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; ++j) {
    matvec(w.to_k, D2, nullptr, x2 + (int64_t)j * ldx2, D, D2, kf + (int64_t)j * D);
    matvec(w.to_v, D2, nullptr, x2 + (int64_t)j * ldx2, D, D2, vf + (int64_t)j * D);
  }
}

struct DecScene { int64_t xyz, feats, fglobal, layer[OCC4D_MAX_CROSS], total; };
DecScene dec_scene(const occ4d_decoder_weights& w, int m) {
  DecScene s{};
  int64_t o = 0;
  s.xyz = o; o += up4((int64_t)m * 3);
  s.feats = o; o += up4((int64_t)m * w.d_latent_local);
  s.fglobal = o; o += up4(w.d_latent - w.d_latent_local);
  for (int j = 0; j < w.n_cross; ++j) { s.layer[j] = o; o += up4((int64_t)2 * m * w.cross[j].dim); }
  s.total = o;
  return s;
}

}  // namespace

extern "C" {

int occ4d_abi_version(void) { return OCC4D_ABI_VERSION; }
const char* occ4d_last_error(void) { return g_err; }
int occ4d_is_cpu_twin(void) { return 1; }      // (only this library exports it)

// ---------------------------------------------------------------------------------------------------------------- geometry
int occ4d_knn_f32(const float* query, int64_t q_stride, int n_query, const float* data, int64_t d_stride, int n_data, int k,
                  int metric, void* out_idx, int idx_is_i64, float* out_dist, void*) {
  REQ(query && data && out_idx && n_query >= 0, "occ4d_knn_f32: null pointer");
  REQ(k >= 1 && k <= 16 && n_data >= k, "occ4d_knn_f32: k = %d must be in 1 .. 16 and <= n_data = %d", k, n_data);
  REQ(metric == 0 || metric == 1, "occ4d_knn_f32: metric %d", metric);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_query; ++i) {
    int32_t idx[16];
    knn_one(query + (int64_t)i * q_stride, data, d_stride, n_data, k, metric, idx, out_dist ? out_dist + (int64_t)i * k : nullptr);
    for (int j = 0; j < k; ++j) {
      if (idx_is_i64) static_cast<int64_t*>(out_idx)[(int64_t)i * k + j] = idx[j];
      else static_cast<int32_t*>(out_idx)[(int64_t)i * k + j] = idx[j];
    }
  }
  return OCC4D_OK;
}

int occ4d_knn_dists_f32(const float* query, int64_t q_stride, int n_query, const float* data, int64_t d_stride, int n_data,
                        const int32_t* idx, int k, int metric, float* out_dist, void*) {
  REQ(query && data && idx && out_dist && n_data >= 1 && k >= 1, "occ4d_knn_dists_f32: bad arguments");
  for (int64_t p = 0; p < (int64_t)n_query * k; ++p) {
    const int j = std::min(std::max(idx[p], 0), n_data - 1);
    out_dist[p] = dist_metric(query + (p / k) * q_stride, data + (int64_t)j * d_stride, metric);
  }
  return OCC4D_OK;
}

int occ4d_fps_start_f32(const float* xyz, int64_t stride, int n, int m, int start, int32_t* out_sorted, int32_t* out_order,
                        void*) {
  REQ(xyz && out_sorted && n >= 1 && m >= 1 && m <= n && start >= 0 && start < n, "occ4d_fps_f32: bad arguments");
  std::vector<float> best((size_t)n, INFINITY);
  std::vector<int32_t> order((size_t)m);
  int cur = start;
  for (int t = 0; t < m; ++t) {
    order[t] = cur;
    const float* pc = xyz + (int64_t)cur * stride;
    float bv = -1.f;
    int bi = 0;
    for (int i = 0; i < n; ++i) {
      const float d = dist_metric(xyz + (int64_t)i * stride, pc, 0);
      if (d < best[i]) best[i] = d;
      if (best[i] > bv) { bv = best[i]; bi = i; }          // first argmax: lowest index on ties
    }
    cur = bi;
  }
  if (out_order) std::copy(order.begin(), order.end(), out_order);
  std::sort(order.begin(), order.end());
  order.erase(std::unique(order.begin(), order.end()), order.end());
  for (int t = 0; t < m; ++t) out_sorted[t] = order[std::min<size_t>(t, order.size() - 1)];
  return OCC4D_OK;
}
int occ4d_fps_f32(const float* xyz, int64_t stride, int n, int m, int32_t* out_sorted, int32_t* out_order, void* st) {
  return occ4d_fps_start_f32(xyz, stride, n, m, 0, out_sorted, out_order, st);
}

int occ4d_nested_fps_level_i32(const int32_t* order, const int32_t* orig, int n, int m, int32_t* out_pos, int32_t* out_orig,
                               void*) {
  REQ(order && orig && out_pos && out_orig && n >= 1 && m >= 1 && m <= n, "occ4d_nested_fps_level_i32: bad arguments");
  std::vector<int32_t> pos((size_t)m);
  for (int t = 0; t < m; ++t) pos[t] = (int32_t)(std::lower_bound(orig, orig + n, order[t]) - orig);
  std::sort(pos.begin(), pos.end());
  pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
  for (int t = 0; t < m; ++t) {
    out_pos[t] = pos[std::min<size_t>(t, pos.size() - 1)];
    out_orig[t] = orig[out_pos[t]];
  }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- rows
int occ4d_copy_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int n, int d, void*) {
  REQ(dst && src && n >= 0 && d >= 1, "occ4d_copy_rows_f32: bad arguments");
  for (int i = 0; i < n; ++i) std::memmove(dst + (int64_t)i * ldd, src + (int64_t)i * lds, sizeof(float) * d);
  return OCC4D_OK;
}
int occ4d_fill_rows_f32(float* dst, int64_t ld, int n, int d, float value, void*) {
  REQ(dst && n >= 0 && d >= 1, "occ4d_fill_rows_f32: bad arguments");
  for (int i = 0; i < n; ++i) std::fill(dst + (int64_t)i * ld, dst + (int64_t)i * ld + d, value);
  return OCC4D_OK;
}
int occ4d_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int n_out, int d, float* out, int64_t ldo, void*) {
  REQ(src && idx && out && n_out >= 0 && d >= 1, "occ4d_gather_rows_f32: bad arguments");
  for (int i = 0; i < n_out; ++i) std::memcpy(out + (int64_t)i * ldo, src + (int64_t)idx[i] * lds, sizeof(float) * d);
  return OCC4D_OK;
}
int occ4d_mean_rows_f32(const float* x, int64_t ldx, int n, int d, float* out, void*) {
  REQ(x && out && n >= 1 && d >= 1, "occ4d_mean_rows_f32: bad arguments");
  for (int c = 0; c < d; ++c) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[(int64_t)i * ldx + c];
    out[c] = (float)(s / n);
  }
  return OCC4D_OK;
}
int occ4d_maxpool_gather_f32(const float* y, int64_t ldy, const int32_t* idx, int n_out, int k, int d, float* z, int64_t ldz,
                             void*) {
  REQ(y && idx && z && k >= 1 && d >= 1, "occ4d_maxpool_gather_f32: bad arguments");
  for (int i = 0; i < n_out; ++i)
    for (int c = 0; c < d; ++c) {
      float m = -INFINITY;
      for (int j = 0; j < k; ++j) m = std::max(m, y[(int64_t)idx[(int64_t)i * k + j] * ldy + c]);
      z[(int64_t)i * ldz + c] = m;
    }
  return OCC4D_OK;
}
int occ4d_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int relu_out, float* y,
                        int64_t ldy, int n, int d, void*) {
  REQ(x && y && n >= 0 && d >= 1, "occ4d_layernorm_f32: bad arguments");
  for (int i = 0; i < n; ++i) {
    const float* r = x + (int64_t)i * ldx;
    double mu = 0.0, var = 0.0;
    for (int c = 0; c < d; ++c) mu += r[c];
    mu /= d;
    for (int c = 0; c < d; ++c) var += (r[c] - mu) * (r[c] - mu);
    var /= d;                                                            // biased, as torch.nn.LayerNorm
    const float inv = (float)(1.0 / std::sqrt(var + (double)eps));
    for (int c = 0; c < d; ++c) {
      float v = (float)(r[c] - mu) * inv;
      if (gamma) v = v * gamma[c] + beta[c];
      y[(int64_t)i * ldy + c] = relu_out && v < 0.f ? 0.f : v;
    }
  }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- Linear
int occ4d_linear_f32(const occ4d_linear_args* a, void*) {
  REQ(a && a->x && a->w && a->y && a->M >= 0 && a->K >= 1 && a->N >= 1, "occ4d_linear_f32: bad arguments");
  REQ(a->relu_in >= 0 && a->relu_in <= 2, "occ4d_linear_f32: Unknown activation: %d", a->relu_in);
  const int M = a->M, K = a->K, N = a->N;
#pragma omp parallel
  {
    std::vector<float> xin((size_t)K), t((size_t)N);
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) {
      const float* xr = a->x + (int64_t)i * a->ldx;
      for (int c = 0; c < K; ++c) xin[c] = act(xr[c], a->relu_in);
      for (int o = 0; o < N; ++o) {
        float v = dot(a->w + (int64_t)o * a->ldw, xin.data(), K) + (a->bias ? a->bias[o] : 0.f);
        if (a->add_rows) v += a->add_rows[(int64_t)(i / a->add_div) * a->ld_add + o];
        if (a->sub_rows) v -= a->sub_rows[(int64_t)a->sub_idx[i] * a->ld_sub + o];
        if (a->relu_out && v < 0.f) v = 0.f;
        t[o] = v;
      }
      for (int o = 0; o < N; ++o)
        a->y[(int64_t)i * a->ldy + o] = t[o] + (a->residual ? a->residual[(int64_t)i * a->ldr + o] : 0.f);
    }
  }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- decoder pieces
int occ4d_posenc_f32(const float* pts, int64_t stride, int n, int c, int n_freq, double base_freq, float* out, int64_t ldo,
                     void*) {
  REQ(pts && out && n >= 0 && c >= 1 && n_freq >= 0, "occ4d_posenc_f32: bad arguments");
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (int64_t)i * stride;
    float* o = out + (int64_t)i * ldo;
    for (int ch = 0; ch < c; ++ch) o[ch] = p[ch];
    for (int f = 0; f < n_freq; ++f) {
      const float w = (float)(2.0 * M_PI * base_freq * std::ldexp(1.0, f));   // double, rounded once (model/implicit.py:33-36)
      for (int ch = 0; ch < c; ++ch) {
        const float arg = p[ch] * w;
        o[c + 2 * c * f + ch] = std::sin(arg);
        o[c + 2 * c * f + c + ch] = std::cos(arg);
      }
    }
  }
  return OCC4D_OK;
}
int occ4d_interp_weights_f32(const float* dist, int n, int k, float* w, void*) {
  REQ(dist && w && n >= 0 && k >= 1, "occ4d_interp_weights_f32: bad arguments");
  for (int i = 0; i < n; ++i) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
      w[(int64_t)i * k + j] = 1.f / (dist[(int64_t)i * k + j] + 1e-4f);
      s += std::fabs(w[(int64_t)i * k + j]);
    }
    s = std::max(s, 1e-12f);
    for (int j = 0; j < k; ++j) w[(int64_t)i * k + j] /= s;
  }
  return OCC4D_OK;
}
int occ4d_interp_add_f32(float* x, int64_t ldx, const float* cvec, const float* table, int64_t ldt, const int32_t* idx,
                         const float* w, int n, int k, int d, void*) {
  REQ(x && table && idx && w && n >= 0 && k >= 1 && d >= 1, "occ4d_interp_add_f32: bad arguments");
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < d; ++c) {
      float s = cvec ? cvec[c] : 0.f;
      for (int j = 0; j < k; ++j) s += w[(int64_t)i * k + j] * table[(int64_t)idx[(int64_t)i * k + j] * ldt + c];
      x[(int64_t)i * ldx + c] += s;
    }
  return OCC4D_OK;
}
int occ4d_squash_f32(float* out, int64_t ld, int n, int g, const int32_t* ops_host, void*) {
  REQ(out && ops_host && n >= 0 && g >= 1, "occ4d_squash_f32: bad arguments");
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < g; ++c) {
      float& v = out[(int64_t)i * ld + c];
      if (ops_host[c] == 1) v = 1.f / (1.f + std::exp(-v));
      else if (ops_host[c] == 2) v = std::min(std::max(v, 0.f), 1.f);
    }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- pre / post steps
int occ4d_grid_points_f32(int nx, int ny, int nz, float x0, float sx, float y0, float sy, float z0, float sz, float t, float* out,
                          void*) {
  REQ(out && nx >= 1 && ny >= 1 && nz >= 1, "occ4d_grid_points_f32: bad arguments");
  int64_t i = 0;
  for (int ix = 0; ix < nx; ++ix)
    for (int iy = 0; iy < ny; ++iy)
      for (int iz = 0; iz < nz; ++iz, ++i) {
        const float ax = (float)ix + 0.5f, ay = (float)iy + 0.5f, az = (float)iz + 0.5f;   // numpy's order, no FMA
        const float mx = ax * sx, my = ay * sy, mz = az * sz;
        out[4 * i + 0] = mx + x0;
        out[4 * i + 1] = my + y0;
        out[4 * i + 2] = mz + z0;
        out[4 * i + 3] = t;
      }
  return OCC4D_OK;
}
int occ4d_compact_count_f32(const float* key, int64_t ld, int n, float threshold, int strict, int* block_counts, int* total_kept,
                            void*) {
  REQ(key && block_counts && total_kept && n >= 0 && ld >= 1, "occ4d_compact_count_f32: bad arguments");
  int total = 0;
  for (int b = 0; b * 256 < n; ++b) {
    block_counts[b] = total;                                              // exclusive prefix
    for (int i = b * 256; i < std::min(n, (b + 1) * 256); ++i) {
      const float kv = key[(int64_t)i * ld];
      total += strict ? kv > threshold : kv >= threshold;
    }
  }
  *total_kept = total;
  return OCC4D_OK;
}
int occ4d_split_count_f32(const float* implicit_output, int64_t ld, int n, float threshold, int* block_counts, int* total_solid,
                          void* st) {
  return occ4d_compact_count_f32(implicit_output, ld, n, threshold, 0, block_counts, total_solid, st);
}
int occ4d_split_write_f32(const float* pts, const float* outp, int64_t ld, int n, int g, float threshold, const int*, int compress,
                          int n_cls, float* solid, float* air, void*) {
  REQ(pts && outp && n >= 0 && g >= 1 && ld >= g, "occ4d_split_write_f32: bad arguments");
  REQ(!compress || n_cls >= 1, "occ4d_split_write_f32: bad n_classes");
  int64_t ns = 0, na = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (int64_t)i * 4;
    const float* o = outp + (int64_t)i * ld;
    if (o[0] >= threshold) {
      float* d = solid + ns++ * (4 + g);
      std::memcpy(d, p, 16);
      std::memcpy(d + 4, o, sizeof(float) * g);
    } else if (compress) {
      float* d = air + na++ * 5;
      const int width = 4 + g, start = width > n_cls ? width - n_cls : 0;   // numpy negative-slice semantics (:299-305)
      int best = 0;
      float bv = start < 4 ? p[start] : o[start - 4];
      for (int c = start + 1; c < width; ++c) {
        const float v = c < 4 ? p[c] : o[c - 4];
        if (v > bv) { bv = v; best = c - start; }
      }
      d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; d[3] = o[0]; d[4] = (float)best;
    } else {
      float* d = air + na++ * (4 + g);
      std::memcpy(d, p, 16);
      std::memcpy(d + 4, o, sizeof(float) * g);
    }
  }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- E3 / E2
int64_t occ4d_pt_layer_prepared_floats(const occ4d_pt_layer_weights* w, int) { return w ? 64 : -1; }
int occ4d_pt_layer_prepare_f32(const occ4d_pt_layer_weights* w, float* prepared, int, void*) {
  TRY(check_layer(w, "occ4d_pt_layer_prepare_f32"));
  REQ(prepared, "occ4d_pt_layer_prepare_f32: null buffer");
  return OCC4D_OK;                       // (as written: nothing is derived from the weights)
}
int64_t occ4d_pt_layer_scene_floats(const occ4d_pt_layer_weights* w, int m) { return w ? up4((int64_t)2 * m * w->dim) : -1; }
int occ4d_pt_layer_scene_f32(const occ4d_pt_layer_weights* w, const float*, const float* x2, int64_t ldx2, int m, float* scene,
                             int, void*) {
  TRY(check_layer(w, "occ4d_pt_layer_scene_f32"));
  REQ(x2 && scene && m >= 1 && w->cross, "occ4d_pt_layer_scene_f32: bad arguments");
  layer_tables(*w, x2, ldx2, m, scene, scene + (int64_t)m * w->dim);
  return OCC4D_OK;
}
int64_t occ4d_pt_layer_workspace_floats(const occ4d_pt_layer_weights* w, int, int, int, int) { return w ? 64 : -1; }

int occ4d_pt_layer_fwd_f32(const occ4d_pt_layer_weights* w, const float*, const float* x, int64_t ldx, const float* pos,
                           int64_t ps, int n, const float* x2, int64_t ldx2, const float* pos2, int64_t p2s, int m, int k,
                           const int32_t* knn_idx, const float* scene, float* out, int64_t ldo, float*, int,
                           occ4d_launch_events* ev, void*) {
  const char* who = "occ4d_pt_layer_fwd_f32";
  TRY(check_layer(w, who));
  REQ(x && pos && out && n >= 0 && k >= 1 && k <= 16, "%s: bad arguments (k = %d)", who, k);
  REQ(!w->cross || (pos2 && m >= k && (x2 || scene)), "%s: cross-attention needs x2 / pos2 with m >= k", who);
  if (ev) ev->used = 0;
  const int D = w->dim, d_in = w->pre_w ? w->d_in : D;
  if (n == 0) return OCC4D_OK;
  // layer1 (model/modules.py:61)
  std::vector<float> y((size_t)n * D);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    if (w->pre_w) matvec(w->pre_w, d_in, w->pre_b, x + (int64_t)i * ldx, D, d_in, y.data() + (size_t)i * D);
    else std::memcpy(y.data() + (size_t)i * D, x + (int64_t)i * ldx, sizeof(float) * D);
  }
  if (!w->cross) { pos2 = pos; p2s = ps; m = n; REQ(m >= k, "%s: n = %d < k = %d", who, n, k); }
  // to_k / to_v of the key cloud (:171-172), to_q of the queries (:170)
  std::vector<float> tab;
  const float *kf, *vf;
  if (w->cross && scene) {
    kf = scene; vf = scene + (int64_t)m * D;
  } else {
    tab.resize((size_t)2 * m * D);
    layer_tables(*w, w->cross ? x2 : y.data(), w->cross ? ldx2 : D, m, tab.data(), tab.data() + (size_t)m * D);
    kf = tab.data(); vf = tab.data() + (size_t)m * D;
  }
  const float divisor = std::sqrt((float)D);
  const int d_out = w->post_w ? w->d_out : D;
#pragma omp parallel
  {
    AttnScratch s(k, D, w->pos_hidden);
    std::vector<float> q((size_t)D), agg((size_t)D);
    int32_t idx[16];
#pragma omp for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i) {
      const float* pi = pos + (int64_t)i * ps;
      if (knn_idx) std::copy(knn_idx + (int64_t)i * k, knn_idx + (int64_t)i * k + k, idx);
      else knn_one(pi, pos2, p2s, m, k, 0, idx, nullptr);                  // kNN_torch(pos, pos2, k)          (:167)
      matvec(w->to_q, D, nullptr, y.data() + (size_t)i * D, D, D, q.data());
      attn_one(*w, q.data(), pi, pos2, p2s, kf, vf, idx, k, divisor, s, agg.data());
      float* o = out + (int64_t)i * ldo;
      if (w->post_w) {                                                       // z = x + layer3(agg)            (model/modules.py:64-66)
        for (int c = 0; c < d_out; ++c) o[c] = x[(int64_t)i * ldx + c] + dot(w->post_w + (int64_t)c * D, agg.data(), D) + w->post_b[c];
      } else {
        std::memcpy(o, agg.data(), sizeof(float) * D);
      }
    }
  }
  return OCC4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------- E6
int occ4d_down_pool_fwd_f32(const float* x, int64_t ldx, int n, int d_in, const float* w, const float* b, int d_out, int norm,
                            const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            const int32_t* nn_idx, int n_new, int k, float* z, int64_t ldz, float* workspace, void* st) {
  const char* who = "occ4d_down_pool_fwd_f32";
  REQ(x && w && b && nn_idx && z && workspace && n >= 1 && n_new >= 1 && k >= 1, "%s: bad arguments", who);
  REQ(norm >= 0 && norm <= 2, "%s: Unknown norm type: %d", who, norm);
  REQ(norm == 0 || (gamma && beta), "%s: norm needs gamma / beta", who);
  REQ(norm != 2 || (mean && var), "%s: batch norm needs running statistics", who);
  float* y = workspace;                                                      // (n, d_out): the MLP on ALL points   (model/modules.py:152)
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) matvec(w, d_in, b, x + (int64_t)i * ldx, d_out, d_in, y + (int64_t)i * d_out);
  if (norm == 1) {
    TRY(occ4d_layernorm_f32(y, d_out, gamma, beta, eps, 1, y, d_out, n, d_out, st));
  } else {
    for (int64_t i = 0; i < (int64_t)n; ++i)
      for (int c = 0; c < d_out; ++c) {
        float v = y[i * d_out + c];
        if (norm == 2) v = (v - mean[c]) / std::sqrt(var[c] + eps) * gamma[c] + beta[c];
        y[i * d_out + c] = v > 0.f ? v : 0.f;
      }
  }
  return occ4d_maxpool_gather_f32(y, d_out, nn_idx, n_new, k, d_out, z, ldz, st);   // max over the k neighbours   (:156-158)
}

// ---------------------------------------------------------------------------------------------------------------- D1-D7
static int check_decoder(const occ4d_decoder_weights* w, const char* who) {
  REQ(w, "%s: null weights", who);
  REQ(w->n_blocks >= 1 && w->n_blocks <= OCC4D_MAX_BLOCKS && w->n_cross >= 0 && w->n_cross <= OCC4D_MAX_CROSS,
      "%s: n_blocks = %d, n_cross = %d", who, w->n_blocks, w->n_cross);
  REQ(w->activation == 0 || w->activation == 1, "%s: Unknown activation: %d", who, w->activation);
  REQ(w->k_local >= 1 && w->k_local <= 16 && (w->n_cross == 0 || (w->k_cross >= 1 && w->k_cross <= 16)),
      "%s: k_local = %d, k_cross = %d (1 .. 16)", who, w->k_local, w->k_cross);
  REQ(w->lin_in_w && w->lin_in_b && w->lin_out_w && w->lin_out_b, "%s: lin_in / lin_out missing", who);
  for (int j = 0; j < w->n_cross; ++j) TRY(check_layer(&w->cross[j], who));
  return OCC4D_OK;
}
int64_t occ4d_decoder_prepared_floats(const occ4d_decoder_weights* w, int) { return w ? 64 : -1; }
int occ4d_decoder_prepare_f32(const occ4d_decoder_weights* w, float* prepared, int, void*) {
  TRY(check_decoder(w, "occ4d_decoder_prepare_f32"));
  REQ(prepared, "occ4d_decoder_prepare_f32: null buffer");
  return OCC4D_OK;
}
int64_t occ4d_decoder_scene_floats(const occ4d_decoder_weights* w, int m) { return w && m >= 1 ? dec_scene(*w, m).total : -1; }
int occ4d_decoder_prepare_scene_f32(const occ4d_decoder_weights* w, const float*, const float* xyz, int64_t xyz_stride,
                                    const float* feats, int64_t ld_feats, const float* fglobal, int m, float* scene, int, void* st) {
  const char* who = "occ4d_decoder_prepare_scene_f32";
  TRY(check_decoder(w, who));
  REQ(xyz && feats && fglobal && scene && m >= 1, "%s: bad arguments", who);
  const DecScene S = dec_scene(*w, m);
  TRY(occ4d_copy_rows_f32(scene + S.xyz, 3, xyz, xyz_stride, m, 3, st));
  TRY(occ4d_copy_rows_f32(scene + S.feats, w->d_latent_local, feats, ld_feats, m, w->d_latent_local, st));
  std::memcpy(scene + S.fglobal, fglobal, sizeof(float) * (w->d_latent - w->d_latent_local));
  for (int j = 0; j < w->n_cross; ++j)      // to_k / to_v rows of the abstract cloud: once per scene instead of once per call (D7)
    layer_tables(w->cross[j], scene + S.feats, w->d_latent_local, m, scene + S.layer[j], scene + S.layer[j] + (int64_t)m * w->cross[j].dim);
  return OCC4D_OK;
}
int64_t occ4d_decoder_query_workspace_floats(const occ4d_decoder_weights* w, int, int, int) { return w ? 64 : -1; }

int occ4d_decoder_query_fwd_f32(const occ4d_decoder_weights* w, const float*, const float* scene, int m, const float* queries,
                                int64_t qs, int n, const int32_t* knn_local, const int32_t* knn_cross, float* out, int64_t ld_out,
                                float* penult, int64_t ld_pen, float*, int, occ4d_launch_events* ev, void*) {
  const char* who = "occ4d_decoder_query_fwd_f32";
  TRY(check_decoder(w, who));
  REQ(scene && queries && out && n >= 0 && m >= w->k_local && (w->n_cross == 0 || m >= w->k_cross), "%s: bad arguments", who);
  if (ev) ev->used = 0;
  const DecScene S = dec_scene(*w, m);
  const float *xyz = scene + S.xyz, *feats = scene + S.feats, *fglobal = scene + S.fglobal;
  const int H = w->d_hidden, E = w->d_latent_local, dg = w->d_latent - E, DL = w->d_latent;
  const int P = w->d_in * (2 * w->n_freq + 1), A = w->activation == 1 ? 2 : 1;
  const int kl = w->k_local, kc = w->k_cross;
#pragma omp parallel
  {
    std::vector<float> fq((size_t)DL), pe((size_t)std::max(P, 1)), x((size_t)H), h((size_t)H), t((size_t)H), y, agg, q;
    std::vector<AttnScratch> scr;
    for (int j = 0; j < w->n_cross; ++j) scr.emplace_back(kc, w->cross[j].dim, w->cross[j].pos_hidden);
    int maxd = 1;
    for (int j = 0; j < w->n_cross; ++j) maxd = std::max(maxd, (int)w->cross[j].dim);
    y.resize(maxd); agg.resize(maxd); q.resize(maxd);
#pragma omp for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
      const float* qi = queries + (int64_t)i * qs;
      // D2 + D3: 8 nearest abstract points by Euclidean norm, inverse-distance interpolation (model/implicit.py:328-342)
      int32_t i8[16], ia[16];
      float d8[16], w8[16];
      if (knn_local) {
        for (int j = 0; j < kl; ++j) {
          i8[j] = std::min(std::max(knn_local[(int64_t)i * kl + j], 0), m - 1);
          d8[j] = dist_metric(qi, xyz + 3 * (int64_t)i8[j], 1);
        }
      } else {
        knn_one(qi, xyz, 3, m, kl, 1, i8, d8);
      }
      float ws = 0.f;
      for (int j = 0; j < kl; ++j) { w8[j] = 1.f / (d8[j] + 1e-4f); ws += std::fabs(w8[j]); }
      ws = std::max(ws, 1e-12f);
      for (int c = 0; c < dg; ++c) fq[c] = fglobal[c];                     // [global | local]                (:342)
      for (int c = 0; c < E; ++c) {
        float s = 0.f;
        for (int j = 0; j < kl; ++j) s += (w8[j] / ws) * feats[(int64_t)i8[j] * E + c];
        fq[dg + c] = s;
      }
      // one kNN_torch serves every cross layer (same coordinates, same K)   (model/point_transformer_layer.py:167)
      if (w->n_cross) {
        if (knn_cross) for (int j = 0; j < kc; ++j) ia[j] = std::min(std::max(knn_cross[(int64_t)i * kc + j], 0), m - 1);
        else knn_one(qi, xyz, 3, m, kc, 0, ia, nullptr);
      }
      // D5 + lin_in (:405-408)
      if (w->n_freq > 0) occ4d_posenc_f32(qi, qs, 1, w->d_in, w->n_freq, (double)w->base_frequency, pe.data(), P, nullptr);
      else std::copy(qi, qi + w->d_in, pe.begin());
      matvec(w->lin_in_w, w->lin_in_ld, w->lin_in_b, pe.data(), H, P, x.data());
      int next_cross = 0;
      for (int bl = 0; bl < w->n_blocks; ++bl) {
        matvec(w->lin_z_w[bl], DL, w->lin_z_b[bl], fq.data(), H, DL, t.data());                    // x += lin_z[i](features_query)   (:416-417)
        for (int c = 0; c < H; ++c) x[c] += t[c];
        for (int c = 0; c < H; ++c) t[c] = act(x[c], A);                                            // ResnetBlockFC                   (:92-101)
        matvec(w->fc0_w[bl], H, w->fc0_b[bl], t.data(), H, H, h.data());
        for (int c = 0; c < H; ++c) h[c] = act(h[c], A);
        matvec(w->fc1_w[bl], H, w->fc1_b[bl], h.data(), H, H, t.data());
        for (int c = 0; c < H; ++c) x[c] += t[c];
        if (next_cross < w->n_cross && w->cross_after[next_cross] == bl) {                          // PointTransformerBlock           (:421-439)
          const int j = next_cross++;
          const occ4d_pt_layer_weights& cw = w->cross[j];
          const int D = cw.dim;
          matvec(cw.pre_w, H, cw.pre_b, x.data(), D, H, y.data());
          matvec(cw.to_q, D, nullptr, y.data(), D, D, q.data());
          const float* kf = scene + S.layer[j];
          attn_one(cw, q.data(), qi, xyz, 3, kf, kf + (int64_t)m * D, ia, kc, std::sqrt((float)D), scr[j], agg.data());
          matvec(cw.post_w, D, cw.post_b, agg.data(), H, D, t.data());
          for (int c = 0; c < H; ++c) x[c] += t[c];
        }
      }
      if (penult) std::copy(x.begin(), x.end(), penult + (int64_t)i * ld_pen);
      for (int c = 0; c < H; ++c) t[c] = act(x[c], A);
      matvec(w->lin_out_w, H, w->lin_out_b, t.data(), w->d_out, H, out + (int64_t)i * ld_out);     // lin_out(act(x))                 (:441-443)
    }
  }
  return OCC4D_OK;
}

}  // extern "C"
