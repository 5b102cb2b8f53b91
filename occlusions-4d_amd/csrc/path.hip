// Path-level entry points of the C ABI (include/occ4d.h, last section): the launch sequence of one reference forward
// -- PointTransformerLayer / PointTransformerBlock (model/point_transformer_layer.py:148-183, model/modules.py:45-67),
// the feature half of DownTransition (model/modules.py:152-158) and LocalPclResnetFC (model/implicit.py:271-445) -- driven
// from C++ on the caller's stream, with the reference's parameters in the reference's layout.  The merged-weight algebra
// of DESIGN.md 4 (i) (fp64 products, rounded once) and every stage packing of the MFMA kernels are device kernels here,
// so that a binder needs nothing but this library.  No allocation, no synchronisation: three caller-provided buffers
// (prepared / scene / workspace), laid out by the host-only *_floats() functions below.
#include <algorithm>
#include <stdlib.h>

#include "common.hpp"

namespace {

using occ4d::cdiv;

#define TRY(expr)                  \
  do {                             \
    const int rc_ = (expr);        \
    if (rc_ != OCC4D_OK) return rc_; \
  } while (0)

constexpr int TRUNK = 416;            // width the row-resident kernels are built for (occ4d_trunk_width)
constexpr int ROW_CHUNK = 32768;      // most query rows per pass (bounds the per-pair workspace of the unfused chain)
// Rows per pass for n rows: passes of EQUAL size (a multiple of the attention kernels' 9 queries per workgroup) instead
// of full passes + a short one -- 68812 training queries = 3 x 22941, not 2 x 32768 + 3276 whose last launch fills a
// third of the machine.  n <= ROW_CHUNK: one pass.
inline int row_step(int n) {
  static const bool balance = [] { const char* e = getenv("OCC4D_ROW_BALANCE"); return !e || e[0] != '0'; }();
  const int passes = (n + ROW_CHUNK - 1) / ROW_CHUNK;
  return passes <= 1 || !balance ? ROW_CHUNK : ((n + passes - 1) / passes + 8) / 9 * 9;
}
constexpr int64_t ALIGN = 64;         // floats: every sub-buffer starts on a 256-byte boundary

inline int64_t up(int64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// Bump allocator over a caller-provided buffer; with base == nullptr it only counts (the *_floats() queries run the
// very code path of the forward in `dry` mode, so sizes and use cannot drift apart).
struct Bump {
  float* base;
  int64_t off = 0, peak = 0;
  explicit Bump(float* b) : base(b) {}
  float* take(int64_t n) {
    float* p = base ? base + off : nullptr;
    off += up(n);
    if (off > peak) peak = off;
    return p;
  }
  int64_t mark() const { return off; }
  void release(int64_t m) { off = m; }
};

struct Events {
  occ4d_launch_events* ev;
  hipStream_t st;
  bool on(int kind) const { return ev && ev->events && ev->kernel == kind && ev->used < ev->capacity; }
  void before(int kind) const {
    if (on(kind)) (void)hipEventRecord((hipEvent_t)ev->events[2 * ev->used], st);
  }
  void after(int kind) const {
    if (on(kind)) {
      (void)hipEventRecord((hipEvent_t)ev->events[2 * ev->used + 1], st);
      ev->used += 1;
    }
  }
};

// ----------------------------------------------------------------------------------------------------------------
// packers: one thread per output float, the index formulas of include/occ4d.h
// ----------------------------------------------------------------------------------------------------------------
__global__ void pack_trunk_rows_kernel(const float* __restrict__ w, int64_t ldw, int n_stages, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)(n_stages + 1) * 13312) return;
  const int s = (int)(i / 13312) % n_stages;               // the extra stage repeats stage 0
  const int rem = (int)(i % 13312);
  const int frag = rem >> 8, in = rem & 255;
  const int nt = frag / 26, t = frag % 26, g = in >> 6, r = (in >> 2) & 15, e = in & 3;
  out[i] = w[(int64_t)(32 * s + 16 * nt + r) * ldw + 16 * t + 4 * g + e];
}

__global__ void pack_trunk_cols_kernel(const float* __restrict__ w, int64_t ldw, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)14 * 13312) return;
  const int j = (int)(i / 13312) % 13;
  const int rem = (int)(i % 13312);
  const int frag = rem >> 8, in = rem & 255;
  const int nt = frag >> 1, tt = frag & 1, g = in >> 6, r = (in >> 2) & 15, e = in & 3;
  out[i] = w[(int64_t)(16 * nt + r) * ldw + 32 * j + 16 * tt + 4 * g + e];
}

__global__ void pack_trunk4_rows_kernel(const float* __restrict__ w, int64_t ldw, int n_stages, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)(n_stages + 1) * 6656) return;
  const int s = (int)(i / 6656) % n_stages;
  const int rem = (int)(i % 6656);
  const int t = rem >> 8, in = rem & 255;
  const int g = in >> 6, r = (in >> 2) & 15, e = in & 3;
  out[i] = w[(int64_t)(16 * s + r) * ldw + 16 * t + 4 * g + e];
}

__global__ void pack_trunk4_cols_kernel(const float* __restrict__ w, int64_t ldw, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)27 * 6656) return;
  const int j = (int)(i / 6656) % 26;
  const int rem = (int)(i % 6656);
  const int nt = rem >> 8, in = rem & 255;
  const int g = in >> 6, r = (in >> 2) & 15, e = in & 3;
  out[i] = w[(int64_t)(16 * nt + r) * ldw + 16 * j + 4 * g + e];
}

// 54 stages x 28 fragments x 256 floats (occ4d_pt_cross_attn16p_f32)
__global__ void pack_attn16p_kernel(const float* __restrict__ w2, const float* __restrict__ wp,
                                    const float* __restrict__ p2, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 54 * 28 * 256) return;
  const int s = i / (28 * 256), rem = i % (28 * 256);
  const int frag = rem >> 8, in = rem & 255;
  const int g = in >> 6, c = (in >> 2) & 15, e = in & 3;
  float v = 0.f;
  if (s < 52) {
    if (frag < 26) v = w2[(16 * frag + c) * 832 + 16 * s + 4 * g + e];
    else v = wp[(16 * s + c) * 32 + 16 * (frag - 26) + 4 * e + g];
  } else {
    const int f = (s - 52) * 28 + frag;               // fragment 2 t + kh over the 26 channel tiles; 4 zero ones last
    if (f < 52) v = p2[(16 * (f >> 1) + c) * 32 + 16 * (f & 1) + 4 * e + g];
  }
  out[i] = v;
}

// ----------------------------------------------------------------------------------------------------------------
// fp64 helpers of the merged-weight algebra
// ----------------------------------------------------------------------------------------------------------------
__global__ void cvt_f32_f64_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols, double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  dst[i] = (double)src[(i / cols) * ld + i % cols];
}
__global__ void cvt_f64_f32_kernel(const double* __restrict__ src, int64_t n, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
__global__ void add_f64_kernel(double* __restrict__ dst, const double* __restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = dst[i] + a[i];
}
// y = gamma (x - mean) / sqrt(var + eps) + beta, then ReLU: BatchNorm1d in eval mode (torch's op order)
__global__ void bn_eval_relu_kernel(float* __restrict__ y, int64_t ld, int n, int d, const float* __restrict__ mean,
                                    const float* __restrict__ var, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * d) return;
  const int c = (int)(i % d);
  float* p = y + (i / d) * ld + c;
  const float inv = 1.0f / sqrtf(var[c] + eps);
  float v = (*p - mean[c]) * inv;
  v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  *p = fmaxf(v, 0.f);
}

int to_f64(const float* src, int64_t ld, int rows, int cols, double* dst, hipStream_t st) {
  const int64_t n = (int64_t)rows * cols;
  cvt_f32_f64_kernel<<<cdiv(n, 256), 256, 0, st>>>(src, ld, rows, cols, dst);
  return occ4d::check_launch("merged weights: f32 -> f64");
}
int to_f32(const double* src, int64_t n, float* dst, hipStream_t st) {
  cvt_f64_f32_kernel<<<cdiv(n, 256), 256, 0, st>>>(src, n, dst);
  return occ4d::check_launch("merged weights: f64 -> f32");
}
int add64(double* dst, const double* a, int64_t n, hipStream_t st) {
  add_f64_kernel<<<cdiv(n, 256), 256, 0, st>>>(dst, a, n);
  return occ4d::check_launch("merged weights: f64 add");
}
__global__ void scale_f32_kernel(float* __restrict__ x, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= s;
}
int scale_f32(float* x, int64_t n, float s, hipStream_t st) {
  scale_f32_kernel<<<cdiv(n, 256), 256, 0, st>>>(x, n, s);
  return occ4d::check_launch("merged weights: scale");
}
int mm64(const double* a, const double* b, double* c, int m, int n, int k, hipStream_t st) {   // contiguous operands
  return occ4d_matmul_f64(a, k, 1, b, n, 1, c, m, n, k, st);
}

// ----------------------------------------------------------------------------------------------------------------
// thin wrappers over the piece-level entry points
// ----------------------------------------------------------------------------------------------------------------
int lin(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, float* y, int64_t ldy, int M, int K,
        int N, int act_in, int relu_out, const float* res, int64_t ldr, hipStream_t st) {
  occ4d_linear_args a{};
  a.x = x; a.ldx = ldx; a.w = w; a.ldw = ldw; a.bias = b; a.residual = res; a.ldr = ldr; a.y = y; a.ldy = ldy;
  a.M = M; a.K = K; a.N = N; a.relu_in = act_in; a.relu_out = relu_out;
  return occ4d_linear_f32(&a, st);
}

bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

// the split-precision row kernel in either scheme (csrc/trunk_bf16x6.hip): bf16 x 3 pieces / fp16 x 2 pieces
int64_t split_packed_floats(bool f16, int n_out) {
  return f16 ? occ4d_rowlin_f16x3_packed_floats(n_out) : occ4d_rowlin_bf16x6_packed_floats(n_out);
}
int split_pack_rowlin(bool f16, const float* w, int64_t ldw, int n_out, float* packed, hipStream_t st) {
  return f16 ? occ4d_pack_rowlin_f16x3_f32(w, ldw, n_out, packed, st) : occ4d_pack_rowlin_bf16x6_f32(w, ldw, n_out, packed, st);
}
int split_rowlin(bool f16, const float* x, int64_t ldx, float* y, int64_t ldy, const float* wpk, const float* b, int n_out,
                 int relu_in, const float* res, int64_t ldr, int n, hipStream_t st) {
  return f16 ? occ4d_rowlin_f16x3_f32(x, ldx, y, ldy, wpk, b, n_out, relu_in, res, ldr, n, st)
             : occ4d_rowlin_bf16x6_f32(x, ldx, y, ldy, wpk, b, n_out, relu_in, res, ldr, n, st);
}

// ----------------------------------------------------------------------------------------------------------------
// one vector-attention layer (+ optional layer1 / layer3 of the PointTransformerBlock around it)
// ----------------------------------------------------------------------------------------------------------------
struct LayerLayout {
  int D, D2, h, Kq;                 // Kq: input width of the merged query projection (d_in when layer1 is folded in)
  bool fold_pre, fused16p, fused_first, fused_self16, bf16x6, wq_rows, w3_rows, trunk4, x6rows;
  bool f16;                         // the split kernels' scheme: fp16 x 2 pieces instead of bf16 x 3 (OCC4D_PATH_SPLIT_F16)
  bool f16w;                        // ... its attention kernel on 32 x 32 x 16 instructions (csrc/crossattn_f16w.hip)
  int64_t wq, bq, wk, wp, wq_packed, stream, stream6, w3_packed, wq_x6, w3_x6, scratch, total;
  int64_t s_A, s_B, s_C, s_C2, s_v, s_bq;      // doubles, inside the scratch region
};

int check_layer(const occ4d_pt_layer_weights* w, const char* who) {
  OCC4D_REQUIRE(w, "%s: null weights", who);
  OCC4D_REQUIRE(w->dim >= 4 && w->dim % 4 == 0 && w->pos_hidden >= 4 && w->pos_hidden % 4 == 0,
                "%s: dim = %d and pos_hidden = %d must be multiples of 4", who, w->dim, w->pos_hidden);
  OCC4D_REQUIRE(w->cross ? (w->dim2 >= 4 && w->dim2 % 4 == 0) : (w->dim2 == w->dim),
                "%s: dim2 = %d (a multiple of 4; equal to dim = %d for self-attention)", who, w->dim2, w->dim);
  OCC4D_REQUIRE(w->to_q && w->to_k && w->to_v && w->pos0_w && w->pos0_b && w->pos2_w && w->pos2_b && w->attn0_w &&
                    w->attn0_b && w->attn2_w && w->attn2_b,
                "%s: null parameter pointer", who);
  OCC4D_REQUIRE(!w->pre_w || (w->pre_b && w->d_in >= 4 && w->d_in % 4 == 0), "%s: layer1 needs a bias and d_in %% 4 == 0", who);
  OCC4D_REQUIRE(w->pre_w || w->d_in == w->dim || w->d_in == 0, "%s: d_in = %d without layer1 (dim = %d)", who, w->d_in, w->dim);
  OCC4D_REQUIRE(!w->post_w || (w->post_b && w->d_out == (w->pre_w ? w->d_in : w->dim)),
                "%s: layer3 + residual needs d_out == d_in", who);
  const void* ps[] = {w->to_q, w->to_k, w->to_v, w->pos2_w, w->attn0_w, w->attn2_w, w->pre_w, w->post_w, w->attn2_b,
                      w->pos2_b, w->attn0_b, w->pre_b, w->post_b};
  for (const void* p : ps) OCC4D_REQUIRE(al16(p), "%s: parameters must be 16-byte aligned", who);
  return OCC4D_OK;
}

LayerLayout layer_layout(const occ4d_pt_layer_weights& w, int flags) {
  LayerLayout L{};
  L.D = w.dim; L.D2 = w.dim2; L.h = w.pos_hidden;
  L.fold_pre = w.cross && w.pre_w;
  L.Kq = L.fold_pre ? w.d_in : L.D;
  const bool fusable = (L.D == 288 || L.D == 416) && L.h == 32 && !(flags & OCC4D_PATH_UNFUSED);
  L.bf16x6 = fusable && L.D == 416 && (flags & OCC4D_PATH_BF16X6) && !(flags & OCC4D_PATH_FIRST_GEN);
  L.f16 = flags & OCC4D_PATH_SPLIT_F16;
  L.fused16p = fusable && L.D == 416 && !L.bf16x6 && !(flags & OCC4D_PATH_FIRST_GEN);
  L.fused_first = fusable && !L.fused16p && !L.bf16x6;
  L.fused_self16 = L.h == 32 && L.D % 4 == 0 && L.D <= 288 && !(flags & OCC4D_PATH_UNFUSED);   // (used when k == 16)
  L.trunk4 = flags & OCC4D_PATH_TRUNK4;
  const bool trunk = !(flags & OCC4D_PATH_GENERIC_LINEAR);
  L.wq_rows = trunk && w.cross && L.Kq == TRUNK && (2 * L.D) % 32 == 0;
  L.w3_rows = trunk && w.post_w && L.D == TRUNK && w.d_out % 32 == 0 && w.d_out == (w.pre_w ? w.d_in : L.D);
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += up(n); return at; };
  L.wq = take((int64_t)2 * L.D * L.Kq);
  L.bq = take(2 * L.D);
  L.wk = take((int64_t)2 * L.D * L.D2);
  L.wp = take((int64_t)2 * L.D * L.h);
  const auto packed = [&](int n_out) {
    return L.trunk4 ? occ4d_trunk4_packed_floats(n_out) : occ4d_trunk_packed_floats(n_out);
  };
  // split-precision trunk rows (csrc/trunk_bf16x6.hip): the merged query projection and layer3 of a d = 416 cross layer
  L.x6rows = (flags & OCC4D_PATH_BF16X6_TRUNK) && L.wq_rows && L.w3_rows && !L.trunk4;
  L.wq_packed = L.wq_rows ? take(packed(2 * L.D)) : -1;
  L.wq_x6 = L.x6rows ? take(split_packed_floats(L.f16, 2 * L.D)) : -1;
  L.w3_x6 = L.x6rows ? take(split_packed_floats(L.f16, w.d_out)) : -1;
  L.stream = L.fused16p ? take(occ4d_pt_cross_attn16p_stream_floats()) : -1;
  L.f16w = L.bf16x6 && L.f16 && occ4d::f16w_enabled();
  L.stream6 = L.bf16x6 ? take(L.f16w ? occ4d_pt_cross_attn_f16w_stream_floats()
                                     : L.f16 ? occ4d_pt_cross_attn_f16x3_stream_floats() : occ4d_pt_cross_attn_bf16x6_stream_floats()) : -1;
  L.w3_packed = L.w3_rows ? take(packed(w.d_out)) : -1;
  // fp64 scratch (doubles): A = W1, B = right factor, C = W1 Wq, C2 = C L1, v / bq vectors
  int64_t d = 0;
  auto take64 = [&](int64_t n) { const int64_t at = d; d += (n + 31) / 32 * 32; return at; };
  const int kb = std::max(std::max(L.D, L.D2), std::max(L.h, L.Kq));
  L.s_A = take64((int64_t)2 * L.D * L.D);
  L.s_B = take64((int64_t)L.D * kb);
  L.s_C = take64((int64_t)2 * L.D * L.D);
  L.s_C2 = take64((int64_t)2 * L.D * kb);
  L.s_v = take64(2 * L.D + kb);
  L.s_bq = take64(2 * L.D);
  L.scratch = take(2 * d);
  L.total = o;
  return L;
}

int layer_prepare(const occ4d_pt_layer_weights& w, const LayerLayout& L, float* prep, hipStream_t st) {
  const int D = L.D, D2 = L.D2, h = L.h;
  double* S = reinterpret_cast<double*>(prep + L.scratch);
  double *A = S + L.s_A, *B = S + L.s_B, *C = S + L.s_C, *C2 = S + L.s_C2, *v = S + L.s_v, *bq = S + L.s_bq;
  TRY(to_f64(w.attn0_w, D, 2 * D, D, A, st));                         // W1 (2D, D)
  TRY(to_f64(w.to_q, D, D, D, B, st));
  TRY(mm64(A, B, C, 2 * D, D, D, st));                                // W1 Wq
  TRY(to_f64(w.pos2_b, 1, D, 1, v, st));
  TRY(occ4d_matmul_f64(A, D, 1, v, 1, 1, bq, 2 * D, 1, D, st));       // W1 c2
  TRY(to_f64(w.attn0_b, 1, 2 * D, 1, v, st));
  TRY(add64(bq, v, 2 * D, st));                                       // + b1
  if (L.fold_pre) {
    TRY(to_f64(w.pre_b, 1, D, 1, v, st));
    TRY(occ4d_matmul_f64(C, D, 1, v, 1, 1, v + D, 2 * D, 1, D, st));  // (W1 Wq) l1_b
    TRY(add64(bq, v + D, 2 * D, st));
    TRY(to_f64(w.pre_w, w.d_in, D, w.d_in, B, st));
    TRY(mm64(C, B, C2, 2 * D, w.d_in, D, st));                        // (W1 Wq) L1
    TRY(to_f32(C2, (int64_t)2 * D * w.d_in, prep + L.wq, st));
  } else {
    TRY(to_f32(C, (int64_t)2 * D * D, prep + L.wq, st));
  }
  TRY(to_f32(bq, 2 * D, prep + L.bq, st));
  TRY(to_f64(w.to_k, D2, D, D2, B, st));
  TRY(mm64(A, B, C2, 2 * D, D2, D, st));
  TRY(to_f32(C2, (int64_t)2 * D * D2, prep + L.wk, st));              // W1 Wk
  TRY(to_f64(w.pos2_w, h, D, h, B, st));
  TRY(mm64(A, B, C2, 2 * D, h, D, st));
  TRY(to_f32(C2, (int64_t)2 * D * h, prep + L.wp, st));               // W1 P2
  if (L.bf16x6 && L.f16) {
    // the fp16 attention kernel keeps its hidden activations at HSCALE x their value: the matrices behind its init term
    // (Aq = wq x + bq, Kt = wk f) are multiplied by that power of two here, once (exact), BEFORE they are packed
    const float hs = occ4d_pt_cross_attn_f16x3_hidden_scale();
    TRY(scale_f32(prep + L.wq, (int64_t)2 * D * L.Kq, hs, st));
    TRY(scale_f32(prep + L.bq, 2 * D, hs, st));
    TRY(scale_f32(prep + L.wk, (int64_t)2 * D * D2, hs, st));
  }
  if (L.wq_rows) {
    if (L.trunk4) TRY(occ4d_pack_trunk4_rows_f32(prep + L.wq, L.Kq, 2 * D, prep + L.wq_packed, st));
    else TRY(occ4d_pack_trunk_rows_f32(prep + L.wq, L.Kq, 2 * D, prep + L.wq_packed, st));
  }
  if (L.x6rows) {
    TRY(split_pack_rowlin(L.f16, prep + L.wq, L.Kq, 2 * D, prep + L.wq_x6, st));
    TRY(split_pack_rowlin(L.f16, w.post_w, D, w.d_out, prep + L.w3_x6, st));
  }
  if (L.fused16p) TRY(occ4d_pack_attn16p_stream_f32(w.attn2_w, prep + L.wp, w.pos2_w, prep + L.stream, st));
  if (L.bf16x6)
    TRY(L.f16w ? occ4d_pack_attn_f16w_stream_f32(w.attn2_w, prep + L.wp, w.pos2_w, prep + L.stream6, st)
        : L.f16 ? occ4d_pack_attn_f16x3_stream_f32(w.attn2_w, prep + L.wp, w.pos2_w, prep + L.stream6, st)
                : occ4d_pack_attn_bf16x6_stream_f32(w.attn2_w, prep + L.wp, w.pos2_w, prep + L.stream6, st));
  if (L.w3_rows) {
    if (L.trunk4) TRY(occ4d_pack_trunk4_rows_f32(w.post_w, D, w.d_out, prep + L.w3_packed, st));
    else TRY(occ4d_pack_trunk_rows_f32(w.post_w, D, w.d_out, prep + L.w3_packed, st));
  }
  return OCC4D_OK;
}

// scene tables of a cross layer: kt (m, 2D), vt (m, D), vtc (m, D) = vt + c2
struct SceneTables { int64_t kt, vt, vtc, total; };
SceneTables scene_layout(const occ4d_pt_layer_weights& w, int m) {
  SceneTables s{};
  s.kt = 0;
  s.vt = up((int64_t)m * 2 * w.dim);
  s.vtc = s.vt + up((int64_t)m * w.dim);
  s.total = s.vtc + up((int64_t)m * w.dim);
  return s;
}
int layer_scene(const occ4d_pt_layer_weights& w, const LayerLayout& L, const float* prep, const float* x2, int64_t ldx2,
                int m, float* scene, hipStream_t st) {
  const SceneTables s = scene_layout(w, m);
  const int D = L.D, D2 = L.D2;
  TRY(lin(x2, ldx2, prep + L.wk, D2, nullptr, scene + s.kt, 2 * D, m, D2, 2 * D, 0, 0, nullptr, 0, st));
  TRY(lin(x2, ldx2, w.to_v, D2, nullptr, scene + s.vt, D, m, D2, D, 0, 0, nullptr, 0, st));
  TRY(lin(x2, ldx2, w.to_v, D2, w.pos2_b, scene + s.vtc, D, m, D2, D, 0, 0, nullptr, 0, st));
  return OCC4D_OK;
}

int rowlin_any(bool trunk4, const float* x, int64_t ldx, float* y, int64_t ldy, const float* wpk, const float* b, int n_out,
               int relu_in, const float* res, int64_t ldr, int n, const Events& E, hipStream_t st) {
  E.before(OCC4D_PROFILE_ROWLIN);
  const int rc = trunk4
      ? occ4d_rowlin4_f32(x, ldx, y, ldy, wpk, b, n_out, relu_in, res, ldr, nullptr, nullptr, 0, nullptr, nullptr, 0, n, st)
      : occ4d_rowlin_f32(x, ldx, y, ldy, wpk, b, n_out, relu_in, res, ldr, nullptr, nullptr, 0, nullptr, nullptr, 0, n, st);
  E.after(OCC4D_PROFILE_ROWLIN);
  return rc;
}

int rowlin_x6(bool f16, const float* x, int64_t ldx, float* y, int64_t ldy, const float* wpk, const float* b, int n_out,
              int relu_in, const float* res, int64_t ldr, int n, const Events& E, hipStream_t st) {
  E.before(OCC4D_PROFILE_ROWLIN);
  const int rc = split_rowlin(f16, x, ldx, y, ldy, wpk, b, n_out, relu_in, res, ldr, n, st);
  E.after(OCC4D_PROFILE_ROWLIN);
  return rc;
}

// The forward of one layer / block.  dry: only the workspace is counted (no pointer is dereferenced, nothing launched).
int layer_forward(const occ4d_pt_layer_weights& w, const LayerLayout& L, const float* prep, const float* x, int64_t ldx,
                  const float* pos, int64_t ps, int n, const float* x2, int64_t ldx2, const float* pos2, int64_t p2s, int m,
                  int k, const int32_t* knn_idx, const float* scene, float* out, int64_t ldo, Bump& ws,
                  const Events& E, hipStream_t st, bool dry, float* logits_out = nullptr, float* a_out = nullptr,
                  float* pe_out = nullptr) {
  const int D = L.D, h = L.h;
  const int64_t mark = ws.mark();
  const float *kt, *vt, *vtc, *aq_all = nullptr, *yfeat = x;
  int64_t ld_y = ldx;
  if (!w.cross) {
    // self-attention: queries, keys and values all come from the (post-layer1) features
    if (w.pre_w) {
      float* y = ws.take((int64_t)n * D);
      if (!dry) TRY(lin(x, ldx, w.pre_w, w.d_in, w.pre_b, y, D, n, w.d_in, D, 0, 0, nullptr, 0, st));
      yfeat = y; ld_y = D;
    }
    float* ktb = ws.take((int64_t)n * 2 * D);
    float* vtb = ws.take((int64_t)n * D);
    float* vcb = (L.fused16p || L.bf16x6) ? ws.take((int64_t)n * D) : nullptr;
    float* aqb = ws.take((int64_t)n * 2 * D);
    if (!dry) {
      TRY(lin(yfeat, ld_y, prep + L.wk, D, nullptr, ktb, 2 * D, n, D, 2 * D, 0, 0, nullptr, 0, st));
      TRY(lin(yfeat, ld_y, w.to_v, D, nullptr, vtb, D, n, D, D, 0, 0, nullptr, 0, st));
      if (vcb) TRY(lin(yfeat, ld_y, w.to_v, D, w.pos2_b, vcb, D, n, D, D, 0, 0, nullptr, 0, st));
      TRY(lin(yfeat, ld_y, prep + L.wq, D, prep + L.bq, aqb, 2 * D, n, D, 2 * D, 0, 0, nullptr, 0, st));
    }
    kt = ktb; vt = vtb; vtc = vcb; aq_all = aqb;
    pos2 = pos; p2s = ps; m = n;
  } else if (scene) {
    const SceneTables s = scene_layout(w, m);
    kt = scene + s.kt; vt = scene + s.vt; vtc = scene + s.vtc;
  } else {
    const SceneTables s = scene_layout(w, m);
    float* tb = ws.take(s.total);
    if (!dry) TRY(layer_scene(w, L, prep, x2, ldx2, m, tb, st));
    kt = tb + s.kt; vt = tb + s.vt; vtc = tb + s.vtc;
  }
  float* agg = out;
  int64_t ld_agg = ldo;
  if (w.post_w) { agg = ws.take((int64_t)n * D); ld_agg = D; }
  const bool fused = (L.fused16p || L.fused_first || L.bf16x6) && k <= 14;
  OCC4D_REQUIRE(fused || !(L.bf16x6 && L.f16), "OCC4D_PATH_SPLIT_F16: the fp16 attention kernel takes k <= 14 neighbours (k = %d); "
                "its prepared matrices are scaled for it", k);
  const float divisor = sqrtf((float)D);          // fp32(sqrt(d)), as torch.tensor(math.sqrt(d), float32)
  const int step = row_step(n);
  for (int lo = 0; lo < n; lo += step) {
    const int c = std::min(step, n - lo);
    const int64_t cmark = ws.mark();
    const int32_t* idx = knn_idx ? knn_idx + (int64_t)lo * k : nullptr;
    if (!idx) {
      int32_t* ib = reinterpret_cast<int32_t*>(ws.take((int64_t)c * k));
      // large searches (the self-kNN of a 14336- / 28672-point cloud) on the exact grid search: the same lists
      const bool grid = m >= 1024 && (int64_t)c * m >= ((int64_t)1 << 26);
      float* gws = grid ? ws.take((occ4d_radius_grid_workspace_bytes(m) + 3) / 4) : nullptr;
      if (!dry) {
        if (grid) TRY(occ4d_knn_grid_f32(pos + (int64_t)lo * ps, ps, c, pos2, p2s, m, k, 0, ib, nullptr, gws, st));
        else TRY(occ4d_knn_f32(pos + (int64_t)lo * ps, ps, c, pos2, p2s, m, k, 0, ib, 0, nullptr, st));
      }
      idx = ib;
    }
    const float* aq;
    if (aq_all) {
      aq = aq_all + (int64_t)lo * 2 * D;
    } else {
      float* ab = ws.take((int64_t)c * 2 * D);
      if (!dry) {
        if (L.x6rows)
          TRY(rowlin_x6(L.f16, x + (int64_t)lo * ldx, ldx, ab, 2 * D, prep + L.wq_x6, prep + L.bq, 2 * D, 0, nullptr, 0, c, E, st));
        else if (L.wq_rows)
          TRY(rowlin_any(L.trunk4, x + (int64_t)lo * ldx, ldx, ab, 2 * D, prep + L.wq_packed, prep + L.bq, 2 * D, 0, nullptr, 0,
                         c, E, st));
        else
          TRY(lin(x + (int64_t)lo * ldx, ldx, prep + L.wq, L.Kq, prep + L.bq, ab, 2 * D, c, L.Kq, 2 * D, 0, 0, nullptr, 0, st));
      }
      aq = ab;
    }
    float* agg_c = agg + (int64_t)lo * ld_agg;
    const float* qp = pos + (int64_t)lo * ps;
    if (fused) {
      if (!dry) {
        E.before(OCC4D_PROFILE_CROSS_ATTN);
        int rc;
        if (L.f16w)
          rc = occ4d_pt_cross_attn_f16w_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w, w.pos0_b,
                                            prep + L.stream6, agg_c, ld_agg, c, m, k, D, divisor, st);
        else if (L.bf16x6 && L.f16)
          rc = occ4d_pt_cross_attn_f16x3_prescaled_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w,
                                                       w.pos0_b, prep + L.stream6, agg_c, ld_agg, c, m, k, D, divisor, st);
        else if (L.bf16x6 && logits_out)
          rc = occ4d_pt_cross_attn_bf16x6_logits_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w, w.pos0_b,
                                                     prep + L.stream6, agg_c, ld_agg, logits_out + (int64_t)lo * k * D,
                                                     a_out ? a_out + (int64_t)lo * k * 2 * D : nullptr,
                                                     pe_out ? pe_out + (int64_t)lo * k * D : nullptr, a_out ? w.pos2_b : nullptr,
                                                     c, m, k, D, divisor, st);
        else if (L.bf16x6)
          rc = occ4d_pt_cross_attn_bf16x6_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w, w.pos0_b,
                                              prep + L.stream6, agg_c, ld_agg, c, m, k, D, divisor, st);
        else if (L.fused16p && logits_out)          // training forward: the logits of rows lo k .. stay in HBM for backward
          rc = occ4d_pt_cross_attn16p_logits_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w, w.pos0_b,
                                                 prep + L.stream, agg_c, ld_agg, logits_out + (int64_t)lo * k * D,
                                                 a_out ? a_out + (int64_t)lo * k * 2 * D : nullptr,
                                                 pe_out ? pe_out + (int64_t)lo * k * D : nullptr, a_out ? w.pos2_b : nullptr, c, m,
                                                 k, D, divisor, occ4d::attn16p_skew(), st);
        else if (L.fused16p)
          rc = occ4d_pt_cross_attn16p_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vtc, D, w.pos0_w, w.pos0_b,
                                          prep + L.stream, agg_c, ld_agg, c, m, k, D, divisor, occ4d::attn16p_skew(), st);
        else
          rc = occ4d_pt_cross_attn_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vt, D, w.pos0_w, w.pos0_b, prep + L.wp,
                                       w.attn2_w, w.attn2_b, w.pos2_w, w.pos2_b, agg_c, ld_agg, c, m, k, D, divisor, st);
        E.after(OCC4D_PROFILE_CROSS_ATTN);
        TRY(rc);
      }
    } else if (L.fused_self16 && k == 16) {
      // the encoder's widths, 16 neighbours: one kernel, no pair tensor in HBM (csrc/selfattn16.hip)
      if (!dry)
        TRY(occ4d_pt_self_attn16_f32(aq, 2 * D, qp, ps, pos2, p2s, idx, kt, 2 * D, vt, D, w.pos0_w, w.pos0_b, prep + L.wp,
                                     w.attn2_w, w.pos2_w, w.pos2_b, agg_c, ld_agg, c, m, k, D, divisor, st));
    } else {
      // unfused chain: r = relu(P1 (p_i - p_j) + c1); hid = relu(aq_i - kt_j + Wp r); logits = W2 hid + b2; pe = P2 r + c2
      const int64_t rows = (int64_t)c * k;
      float* r = ws.take(rows * h);
      float* hid = ws.take(rows * 2 * D);
      float* logits = ws.take(rows * D);
      float* pe = ws.take(rows * D);
      if (!dry) {
        TRY(occ4d_pt_pos_hidden_f32(qp, ps, pos2, p2s, idx, c, k, w.pos0_w, w.pos0_b, h, r, st));
        occ4d_linear_args a{};
        a.x = r; a.ldx = h; a.w = prep + L.wp; a.ldw = h; a.y = hid; a.ldy = 2 * D;
        a.M = (int)rows; a.K = h; a.N = 2 * D; a.relu_out = 1;
        a.add_rows = aq; a.ld_add = 2 * D; a.add_div = k; a.sub_rows = kt; a.ld_sub = 2 * D; a.sub_idx = idx;
        TRY(occ4d_linear_f32(&a, st));
        TRY(lin(hid, 2 * D, w.attn2_w, 2 * D, w.attn2_b, logits, D, (int)rows, 2 * D, D, 0, 0, nullptr, 0, st));
        TRY(lin(r, h, w.pos2_w, h, w.pos2_b, pe, D, (int)rows, h, D, 0, 0, nullptr, 0, st));
        TRY(occ4d_pt_softmax_agg_f32(logits, vt, D, pe, idx, c, k, D, divisor, agg_c, ld_agg, st));
      }
    }
    ws.release(cmark);
  }
  if (w.post_w && !dry) {
    if (L.x6rows)
      TRY(rowlin_x6(L.f16, agg, D, out, ldo, prep + L.w3_x6, w.post_b, w.d_out, 0, x, ldx, n, E, st));
    else if (L.w3_rows)
      TRY(rowlin_any(L.trunk4, agg, D, out, ldo, prep + L.w3_packed, w.post_b, w.d_out, 0, x, ldx, n, E, st));
    else
      TRY(lin(agg, D, w.post_w, D, w.post_b, out, ldo, n, D, w.d_out, 0, 0, x, ldx, st));
  }
  ws.release(mark);
  return OCC4D_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// decoder
// ----------------------------------------------------------------------------------------------------------------
struct DecoderLayout {
  int H, P, P4, dg, E, nB, nC;
  bool trunk, trunk4, resblock;              // row-resident Linear kernels usable / half-CU variant / fused residual block
  bool x6trunk;                              // residual blocks as two split-precision launches (csrc/trunk_bf16x6.hip)
  bool f16;                                  // ... on the fp16 two-piece scheme (OCC4D_PATH_SPLIT_F16)
  bool f16block;                             // ... as ONE launch per block (csrc/resblock_f16x3.hip): w0x holds both weights
  int64_t w0p[OCC4D_MAX_BLOCKS], w1p[OCC4D_MAX_BLOCKS];
  int64_t w0x[OCC4D_MAX_BLOCKS], w1x[OCC4D_MAX_BLOCKS];
  int64_t cross[OCC4D_MAX_CROSS];
  LayerLayout cl[OCC4D_MAX_CROSS];
  int64_t total;
};

int check_decoder(const occ4d_decoder_weights* w, const char* who) {
  OCC4D_REQUIRE(w, "%s: null weights", who);
  OCC4D_REQUIRE(w->n_blocks >= 1 && w->n_blocks <= OCC4D_MAX_BLOCKS && w->n_cross >= 0 && w->n_cross <= OCC4D_MAX_CROSS,
                "%s: n_blocks = %d (1 .. %d), n_cross = %d (0 .. %d)", who, w->n_blocks, OCC4D_MAX_BLOCKS, w->n_cross,
                OCC4D_MAX_CROSS);
  OCC4D_REQUIRE(w->d_hidden % 4 == 0 && w->d_hidden >= 4 && w->d_out >= 1 && w->d_in >= 3 && w->n_freq >= 0,
                "%s: d_hidden = %d must be a multiple of 4, d_in = %d >= 3 (x, y, z first)", who, w->d_hidden, w->d_in);
  const int P = w->d_in * (2 * w->n_freq + 1);
  OCC4D_REQUIRE(w->lin_in_ld >= P && w->lin_in_ld % 4 == 0, "%s: lin_in_ld = %d must be a multiple of 4 >= %d (zero-pad "
                "lin_in.weight's rows)", who, w->lin_in_ld, P);
  OCC4D_REQUIRE(w->d_latent_local >= 4 && w->d_latent_local % 4 == 0 && w->d_latent >= w->d_latent_local &&
                    (w->d_latent - w->d_latent_local) % 4 == 0,
                "%s: d_latent = %d, d_latent_local = %d (both parts multiples of 4; local_mode 'feature' / 'attention')",
                who, w->d_latent, w->d_latent_local);
  OCC4D_REQUIRE(w->k_local >= 1 && w->k_local <= 16 && (w->n_cross == 0 || (w->k_cross >= 1 && w->k_cross <= 16)),
                "%s: k_local = %d, k_cross = %d (1 .. 16)", who, w->k_local, w->k_cross);
  OCC4D_REQUIRE(w->activation == 0 || w->activation == 1, "%s: Unknown activation: %d", who, w->activation);
  OCC4D_REQUIRE(w->lin_in_w && w->lin_in_b && w->lin_out_w && w->lin_out_b && al16(w->lin_in_w) && al16(w->lin_out_w),
                "%s: lin_in / lin_out missing or misaligned", who);
  for (int i = 0; i < w->n_blocks; ++i)
    OCC4D_REQUIRE(w->lin_z_w[i] && w->lin_z_b[i] && w->fc0_w[i] && w->fc0_b[i] && w->fc1_w[i] && w->fc1_b[i] &&
                      al16(w->lin_z_w[i]) && al16(w->fc0_w[i]) && al16(w->fc1_w[i]) && al16(w->fc0_b[i]) &&
                      al16(w->fc1_b[i]) && al16(w->lin_z_b[i]),
                  "%s: block %d: parameter missing or not 16-byte aligned", who, i);
  for (int j = 0; j < w->n_cross; ++j) {
    TRY(check_layer(&w->cross[j], who));
    const occ4d_pt_layer_weights& c = w->cross[j];
    OCC4D_REQUIRE(c.cross && c.pre_w && c.post_w && c.d_in == w->d_hidden && c.d_out == w->d_hidden &&
                      c.dim2 == w->d_latent_local,
                  "%s: cross layer %d must be a PointTransformerBlock (layer1, layer3) over d_hidden with dim2 = d_latent_local",
                  who, j);
    OCC4D_REQUIRE(w->cross_after[j] >= 0 && w->cross_after[j] < w->n_blocks && (j == 0 || w->cross_after[j] > w->cross_after[j - 1]),
                  "%s: cross_after[%d] = %d", who, j, w->cross_after[j]);
  }
  return OCC4D_OK;
}

DecoderLayout decoder_layout(const occ4d_decoder_weights& w, int flags) {
  DecoderLayout L{};
  L.H = w.d_hidden; L.P = w.d_in * (2 * w.n_freq + 1); L.P4 = w.lin_in_ld;
  L.E = w.d_latent_local; L.dg = w.d_latent - w.d_latent_local; L.nB = w.n_blocks; L.nC = w.n_cross;
  L.trunk = L.H == TRUNK && !(flags & OCC4D_PATH_GENERIC_LINEAR);
  L.trunk4 = flags & OCC4D_PATH_TRUNK4;
  L.resblock = L.trunk && w.activation == 0;
  L.x6trunk = L.resblock && !L.trunk4 && (flags & OCC4D_PATH_BF16X6_TRUNK);
  L.f16 = flags & OCC4D_PATH_SPLIT_F16;
  L.f16block = L.x6trunk && L.f16 && occ4d::f16_resblock_enabled();
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += up(n); return at; };
  const int64_t pk = L.trunk4 ? occ4d_trunk4_packed_floats(TRUNK) : occ4d_trunk_packed_floats(TRUNK);
  for (int i = 0; i < L.nB; ++i) {
    L.w0p[i] = L.resblock ? take(pk) : -1;
    L.w1p[i] = L.resblock ? take(pk) : -1;
    L.w0x[i] = L.f16block ? take(occ4d_resblock_f16x3_packed_floats()) : L.x6trunk ? take(split_packed_floats(L.f16, TRUNK)) : -1;
    L.w1x[i] = L.x6trunk && !L.f16block ? take(split_packed_floats(L.f16, TRUNK)) : -1;
  }
  for (int j = 0; j < L.nC; ++j) {
    L.cl[j] = layer_layout(w.cross[j], flags);
    L.cross[j] = take(L.cl[j].total);
  }
  L.total = o;
  return L;
}

struct DecoderScene { int64_t xyz, ztab, zconst, layer[OCC4D_MAX_CROSS], total; };
DecoderScene decoder_scene_layout(const occ4d_decoder_weights& w, int m) {
  DecoderScene s{};
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += up(n); return at; };
  s.xyz = take((int64_t)m * 3);
  s.ztab = take((int64_t)m * w.n_blocks * w.d_hidden);
  s.zconst = take((int64_t)w.n_blocks * w.d_hidden);
  for (int j = 0; j < w.n_cross; ++j) s.layer[j] = take(scene_layout(w.cross[j], m).total);
  s.total = o;
  return s;
}

// caller-supplied neighbour lists are copied into the workspace with every entry forced into [0, m): the gathers behind
// them are unchecked
__global__ void clamp_indices_kernel(const int32_t* __restrict__ src, int64_t n, int m, int32_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = min(max(src[i], 0), m - 1);
}
int clamp_indices(const int32_t* src, int64_t n, int m, int32_t* dst, hipStream_t st) {
  clamp_indices_kernel<<<cdiv(n, 256), 256, 0, st>>>(src, n, m, dst);
  return occ4d::check_launch("occ4d_decoder_query_fwd_f32 (neighbour lists)");
}

int decoder_forward(const occ4d_decoder_weights& w, const DecoderLayout& L, const float* prep, const float* scene, int m,
                    const float* queries, int64_t qs, int n, const int32_t* knn_local, const int32_t* knn_cross, float* out,
                    int64_t ld_out, float* penult, int64_t ld_pen, Bump& ws, int flags, const Events& E, hipStream_t st,
                    bool dry) {
  const int H = L.H;
  const DecoderScene S = decoder_scene_layout(w, m);
  const float* xyz = scene ? scene + S.xyz : nullptr;
  const int act = w.activation == 1 ? 2 : 1;           // linear's act_in code: 1 relu, 2 swish
  const int step = row_step(n);
  for (int lo = 0; lo < n; lo += step) {
    const int c = std::min(step, n - lo);
    const int64_t mark = ws.mark();
    const float* q = queries + (int64_t)lo * qs;
    float* x = penult ? penult + (int64_t)lo * ld_pen : ws.take((int64_t)c * H);
    const int64_t ldx = penult ? ld_pen : H;
    int32_t* idx8 = reinterpret_cast<int32_t*>(ws.take((int64_t)c * w.k_local));
    float* w8 = ws.take((int64_t)c * w.k_local);
    int32_t* idx_att = L.nC ? reinterpret_cast<int32_t*>(ws.take((int64_t)c * w.k_cross)) : nullptr;
    float* pe = ws.take((int64_t)c * L.P4);
    float* hbuf = (L.resblock && (!L.x6trunk || L.f16block)) ? nullptr : ws.take((int64_t)c * H);
    if (!dry) {
      // D2 + D3 (model/implicit.py:328-342): 8 nearest abstract points by Euclidean norm, inverse-distance weights
      // (caller-supplied lists: the reference's own tie order; distances recomputed with the search's expression)
      if (knn_local) {
        TRY(clamp_indices(knn_local + (int64_t)lo * w.k_local, (int64_t)c * w.k_local, m, idx8, st));
        TRY(occ4d_knn_dists_f32(q, qs, c, xyz, 3, m, idx8, w.k_local, 1, w8, st));
      } else {
        TRY(occ4d_knn_f32(q, qs, c, xyz, 3, m, w.k_local, 1, idx8, 0, w8, st));
      }
      TRY(occ4d_interp_weights_f32(w8, c, w.k_local, w8, st));
      // one kNN_torch (model/point_transformer_layer.py:167) serves every cross-attention layer: same xyz, same K
      if (L.nC && knn_cross) TRY(clamp_indices(knn_cross + (int64_t)lo * w.k_cross, (int64_t)c * w.k_cross, m, idx_att, st));
      else if (L.nC) TRY(occ4d_knn_f32(q, qs, c, xyz, 3, m, w.k_cross, 0, idx_att, 0, nullptr, st));
      // D5 + lin_in (:405-408)
      const float* emb = q;
      int64_t ld_emb = qs;
      if (w.n_freq > 0) {
        if (L.P4 != L.P) TRY(occ4d::zero_rows(pe, L.P4, c, L.P4, st));
        TRY(occ4d_posenc_f32(q, qs, c, w.d_in, w.n_freq, (double)w.base_frequency, pe, L.P4, st));
        emb = pe; ld_emb = L.P4;
      } else {
        TRY(occ4d::zero_rows(pe, L.P4, c, L.P4, st));
        TRY(occ4d::copy_rows(pe, L.P4, q, qs, c, w.d_in, st));
        emb = pe; ld_emb = L.P4;
      }
      TRY(lin(emb, ld_emb, w.lin_in_w, w.lin_in_ld, w.lin_in_b, x, ldx, c, L.P4, H, 0, 0, nullptr, 0, st));
    }
    int next_cross = 0;
    for (int i = 0; i < L.nB; ++i) {
      if (!dry) {
        // x += lin_z[i](features_query) in the exact-in-R form.  OCC4D_PATH_FUSED_INTERP (A/B only,
        // measured slower: DESIGN.md 6e): the term of block i + 1 is added in block i's epilogue instead, wherever no
        // cross-attention layer sits between the two blocks.
        const bool fuse_ok = (flags & OCC4D_PATH_FUSED_INTERP) && L.resblock && !L.trunk4;
        const bool cross_behind = next_cross < L.nC && w.cross_after[next_cross] == i;
        const bool cross_before = i > 0 && next_cross > 0 && w.cross_after[next_cross - 1] == i - 1;
        const bool had_it = fuse_ok && i > 0 && !cross_before;           // block i - 1's epilogue added this block's term
        const bool give_next = fuse_ok && i + 1 < L.nB && !cross_behind;
        if (!had_it)
          TRY(occ4d_interp_add_f32(x, ldx, scene + S.zconst + (int64_t)i * H, scene + S.ztab + (int64_t)i * H,
                                   (int64_t)L.nB * H, idx8, w8, c, w.k_local, H, st));
        if (L.f16block) {
          // x = x + fc_1(relu(fc_0(relu(x)))) in place, one launch, h in registers
          E.before(OCC4D_PROFILE_RESBLOCK);
          const int rc = occ4d_resblock_f16x3_f32(x, ldx, x, ldx, prep + L.w0x[i], w.fc0_b[i], w.fc1_b[i], c, st);
          E.after(OCC4D_PROFILE_RESBLOCK);
          TRY(rc);
        } else if (L.x6trunk) {
          // h = fc_0(relu(x)); x = x + fc_1(relu(h)): two split-precision launches, h through the workspace
          E.before(OCC4D_PROFILE_RESBLOCK);
          int rc = split_rowlin(L.f16, x, ldx, hbuf, H, prep + L.w0x[i], w.fc0_b[i], H, 1, nullptr, 0, c, st);
          if (!rc) rc = split_rowlin(L.f16, hbuf, H, x, ldx, prep + L.w1x[i], w.fc1_b[i], H, 1, x, ldx, c, st);
          E.after(OCC4D_PROFILE_RESBLOCK);
          TRY(rc);
        } else if (L.resblock) {
          E.before(OCC4D_PROFILE_RESBLOCK);
          const float* zc = give_next ? scene + S.zconst + (int64_t)(i + 1) * H : nullptr;
          const float* zt = give_next ? scene + S.ztab + (int64_t)(i + 1) * H : nullptr;
          const int rc = L.trunk4
              ? occ4d_resblock4_f32(x, ldx, x, ldx, prep + L.w0p[i], w.fc0_b[i], prep + L.w1p[i], w.fc1_b[i], nullptr,
                                    nullptr, 0, nullptr, nullptr, 0, c, st)
              : occ4d_resblock_f32(x, ldx, x, ldx, prep + L.w0p[i], w.fc0_b[i], prep + L.w1p[i], w.fc1_b[i], zc, zt,
                                   (int64_t)L.nB * H, give_next ? idx8 : nullptr, give_next ? w8 : nullptr,
                                   give_next ? w.k_local : 0, c, st);
          E.after(OCC4D_PROFILE_RESBLOCK);
          TRY(rc);
        } else {
          TRY(lin(x, ldx, w.fc0_w[i], H, w.fc0_b[i], hbuf, H, c, H, H, act, 0, nullptr, 0, st));
          TRY(lin(hbuf, H, w.fc1_w[i], H, w.fc1_b[i], x, ldx, c, H, H, act, 0, x, ldx, st));
        }
      }
      if (next_cross < L.nC && w.cross_after[next_cross] == i) {
        const int j = next_cross++;
        TRY(layer_forward(w.cross[j], L.cl[j], prep ? prep + L.cross[j] : nullptr, x, ldx, q, qs, c, nullptr, 0, xyz, 3, m,
                          w.k_cross, idx_att, scene ? scene + S.layer[j] : nullptr, x, ldx, ws, E, st, dry));
      }
    }
    if (!dry)
      TRY(lin(x, ldx, w.lin_out_w, H, w.lin_out_b, out + (int64_t)lo * ld_out, ld_out, c, H, w.d_out, act, 0, nullptr, 0, st));
    ws.release(mark);
  }
  (void)flags;
  return OCC4D_OK;
}

}  // namespace

namespace occ4d {
// fp16 scheme: OCC4D_F16W=1 selects the 32 x 32 x 16 attention kernel (csrc/crossattn_f16w.hip) for A/B runs; the default is
// the 16 x 16 x 32 kernel of csrc/crossattn_bf16x6.hip
bool f16_resblock_enabled() {
  static const bool v = [] {
    const char* e = getenv("OCC4D_F16_RESBLOCK");
    return !e || atoi(e) != 0;
  }();
  return v;
}
bool f16w_enabled() {
  static const bool v = [] {
    const char* e = getenv("OCC4D_F16W");              // (opt-in: measured slower than the 16 x 16 x 32 kernel)
    return e && atoi(e) != 0;
  }();
  return v;
}
// phase offset of the paired attention workgroups (units of s_sleep(127)); OCC4D_CA16P_SKEW overrides (performance only)
int attn16p_skew() {
  static const int v = [] {
    const char* e = getenv("OCC4D_CA16P_SKEW");
    return e ? atoi(e) : 6;
  }();
  return v;
}
}  // namespace occ4d

// ================================================================================================================
// extern "C"
// ================================================================================================================
extern "C" int occ4d_pack_trunk_rows_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream) {
  OCC4D_REQUIRE(w && packed && n_out >= 32 && n_out % 32 == 0 && ldw >= TRUNK,
                "occ4d_pack_trunk_rows_f32: (%d, %d) weight with n_out %% 32 == 0 expected", n_out, TRUNK);
  const int64_t total = occ4d_trunk_packed_floats(n_out);
  pack_trunk_rows_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, ldw, n_out / 32, packed);
  return occ4d::check_launch("occ4d_pack_trunk_rows_f32");
}
extern "C" int occ4d_pack_trunk_cols_f32(const float* w, int64_t ldw, float* packed, void* stream) {
  OCC4D_REQUIRE(w && packed && ldw >= TRUNK, "occ4d_pack_trunk_cols_f32: (416, 416) weight expected");
  pack_trunk_cols_kernel<<<cdiv((int64_t)14 * 13312, 256), 256, 0, (hipStream_t)stream>>>(w, ldw, packed);
  return occ4d::check_launch("occ4d_pack_trunk_cols_f32");
}
extern "C" int occ4d_pack_trunk4_rows_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream) {
  OCC4D_REQUIRE(w && packed && n_out >= 16 && n_out % 16 == 0 && ldw >= TRUNK,
                "occ4d_pack_trunk4_rows_f32: (%d, %d) weight with n_out %% 16 == 0 expected", n_out, TRUNK);
  const int64_t total = occ4d_trunk4_packed_floats(n_out);
  pack_trunk4_rows_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, ldw, n_out / 16, packed);
  return occ4d::check_launch("occ4d_pack_trunk4_rows_f32");
}
extern "C" int occ4d_pack_trunk4_cols_f32(const float* w, int64_t ldw, float* packed, void* stream) {
  OCC4D_REQUIRE(w && packed && ldw >= TRUNK, "occ4d_pack_trunk4_cols_f32: (416, 416) weight expected");
  pack_trunk4_cols_kernel<<<cdiv((int64_t)27 * 6656, 256), 256, 0, (hipStream_t)stream>>>(w, ldw, packed);
  return occ4d::check_launch("occ4d_pack_trunk4_cols_f32");
}
extern "C" int occ4d_pack_attn16p_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream,
                                             void* stream) {
  OCC4D_REQUIRE(w2 && wp && p2 && wstream, "occ4d_pack_attn16p_stream_f32: null pointer");
  pack_attn16p_kernel<<<cdiv(54 * 28 * 256, 256), 256, 0, (hipStream_t)stream>>>(w2, wp, p2, wstream);
  return occ4d::check_launch("occ4d_pack_attn16p_stream_f32");
}

extern "C" int64_t occ4d_pt_layer_prepared_floats(const occ4d_pt_layer_weights* w, int flags) {
  if (check_layer(w, "occ4d_pt_layer_prepared_floats")) return -1;
  return layer_layout(*w, flags).total;
}
extern "C" int occ4d_pt_layer_prepare_f32(const occ4d_pt_layer_weights* w, float* prepared, int flags, void* stream) {
  TRY(check_layer(w, "occ4d_pt_layer_prepare_f32"));
  OCC4D_REQUIRE(prepared && al16(prepared), "occ4d_pt_layer_prepare_f32: prepared buffer missing or misaligned");
  return layer_prepare(*w, layer_layout(*w, flags), prepared, (hipStream_t)stream);
}
extern "C" int64_t occ4d_pt_layer_scene_floats(const occ4d_pt_layer_weights* w, int m) {
  if (check_layer(w, "occ4d_pt_layer_scene_floats") || m < 0) return -1;
  return scene_layout(*w, m).total;
}
extern "C" int occ4d_pt_layer_scene_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x2,
                                        int64_t ldx2, int m, float* scene, int flags, void* stream) {
  TRY(check_layer(w, "occ4d_pt_layer_scene_f32"));
  OCC4D_REQUIRE(w->cross && prepared && x2 && scene && m >= 1, "occ4d_pt_layer_scene_f32: cross layer, x2 and buffers required");
  return layer_scene(*w, layer_layout(*w, flags), prepared, x2, ldx2, m, scene, (hipStream_t)stream);
}
extern "C" int64_t occ4d_pt_layer_workspace_floats(const occ4d_pt_layer_weights* w, int n, int m, int k, int flags) {
  if (check_layer(w, "occ4d_pt_layer_workspace_floats") || n < 0 || m < 0) return -1;
  Bump ws(nullptr);
  Events E{nullptr, nullptr};
  // counted without caller-provided kNN lists / scene tables (the larger case)
  if (layer_forward(*w, layer_layout(*w, flags), nullptr, nullptr, w->pre_w ? w->d_in : w->dim, nullptr, 3, n, nullptr,
                    w->dim2, nullptr, 3, m, k, nullptr, nullptr, nullptr, w->post_w ? w->d_out : w->dim, ws, E, nullptr, true))
    return -1;
  return ws.peak + ALIGN;
}
extern "C" int occ4d_pt_layer_fwd_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x, int64_t ldx,
                                      const float* pos, int64_t pos_stride, int n, const float* x2, int64_t ldx2,
                                      const float* pos2, int64_t pos2_stride, int m, int k, const int32_t* knn_idx,
                                      const float* scene, float* out, int64_t ldo, float* workspace, int flags,
                                      occ4d_launch_events* ev, void* stream) {
  const char* who = "occ4d_pt_layer_fwd_f32";
  TRY(check_layer(w, who));
  if (ev) ev->used = 0;
  OCC4D_REQUIRE(n >= 0 && k >= 1 && k <= 16, "%s: n = %d, k = %d (1 .. 16)", who, n, k);
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(prepared && x && pos && out && workspace && al16(workspace) && al16(x) && al16(out) && ldx % 4 == 0 &&
                    ldo % 4 == 0,
                "%s: null or misaligned buffer (x, out, workspace 16-byte aligned; ldx, ldo multiples of 4)", who);
  if (w->cross) {
    OCC4D_REQUIRE(pos2 && m >= k && (scene || (x2 && al16(x2) && ldx2 % 4 == 0)),
                  "%s: cross-attention needs pos2, m >= k and x2 (or the scene tables)", who);
  } else {
    OCC4D_REQUIRE(n >= k, "%s: self-attention over %d points cannot serve k = %d", who, n, k);
  }
  OCC4D_REQUIRE(out != x || w->post_w, "%s: out may alias x only with layer3 (residual form)", who);
  Bump ws(workspace);
  const Events E{ev, (hipStream_t)stream};
  return layer_forward(*w, layer_layout(*w, flags), prepared, x, ldx, pos, pos_stride, n, x2, ldx2, pos2, pos2_stride, m, k,
                       knn_idx, scene, out, ldo, ws, E, (hipStream_t)stream, false);
}

// ... and the pre-softmax logits W2 relu(a) of every (query, neighbour) pair into logits_out (n * k, dim): the training
// forward (backward then recomputes only the hidden pre-activations: occ4d_pt_pair_mlp_f32 with logits = NULL).  Only the
// layers the fp32 paired-workgroup kernel or the bf16 x 3 split kernel serves (dim = 416, k <= 14): OCC4D_ERR_INVALID otherwise.
extern "C" int occ4d_pt_layer_fwd_logits_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x, int64_t ldx,
                                             const float* pos, int64_t pos_stride, int n, const float* x2, int64_t ldx2,
                                             const float* pos2, int64_t pos2_stride, int m, int k, const int32_t* knn_idx,
                                             const float* scene, float* out, int64_t ldo, float* logits_out, float* a_out,
                                             float* pe_out, float* workspace, int flags, occ4d_launch_events* ev,
                                             void* stream) {
  const char* who = "occ4d_pt_layer_fwd_logits_f32";
  TRY(check_layer(w, who));
  if (ev) ev->used = 0;
  OCC4D_REQUIRE(n >= 0 && k >= 1 && k <= 14, "%s: n = %d, k = %d (1 .. 14)", who, n, k);
  if (n == 0) return OCC4D_OK;
  const LayerLayout L = layer_layout(*w, flags);
  OCC4D_REQUIRE((L.fused16p || (L.bf16x6 && !L.f16)) && w->cross, "%s: only the cross-attention layers of the fp32 paired-workgroup "
                "kernel or of the bf16 x 3 split kernel (dim = 416, fused) store their logits", who);
  OCC4D_REQUIRE(prepared && x && pos && out && logits_out && workspace && al16(workspace) && al16(x) && al16(out) &&
                    al16(logits_out) && ldx % 4 == 0 && ldo % 4 == 0,
                "%s: null or misaligned buffer (x, out, logits_out, workspace 16-byte aligned; ldx, ldo multiples of 4)", who);
  OCC4D_REQUIRE(pos2 && m >= k && (scene || (x2 && al16(x2) && ldx2 % 4 == 0)),
                "%s: cross-attention needs pos2, m >= k and x2 (or the scene tables)", who);
  OCC4D_REQUIRE(out != x || w->post_w, "%s: out may alias x only with layer3 (residual form)", who);
  OCC4D_REQUIRE((a_out != nullptr) == (pe_out != nullptr) && (!a_out || (al16(a_out) && al16(pe_out))),
                "%s: a_out and pe_out come together, 16-byte aligned", who);
  Bump ws(workspace);
  const Events E{ev, (hipStream_t)stream};
  return layer_forward(*w, L, prepared, x, ldx, pos, pos_stride, n, x2, ldx2, pos2, pos2_stride, m, k, knn_idx, scene, out, ldo,
                       ws, E, (hipStream_t)stream, false, logits_out, a_out, pe_out);
}

extern "C" int occ4d_down_pool_fwd_f32(const float* x, int64_t ldx, int n, int d_in, const float* w, const float* b,
                                       int d_out, int norm, const float* gamma, const float* beta, const float* mean,
                                       const float* var, float eps, const int32_t* nn_idx, int n_new, int k, float* z,
                                       int64_t ldz, float* workspace, void* stream) {
  const char* who = "occ4d_down_pool_fwd_f32";
  OCC4D_REQUIRE(x && w && b && nn_idx && z && workspace, "%s: null pointer", who);
  OCC4D_REQUIRE(n >= 1 && n_new >= 1 && k >= 1 && d_in % 4 == 0 && d_out % 4 == 0, "%s: bad sizes n = %d, n_new = %d, k = %d, "
                "d_in = %d, d_out = %d", who, n, n_new, k, d_in, d_out);
  hipStream_t st = (hipStream_t)stream;
  float* y = workspace;
  if (norm == 0) {
    TRY(lin(x, ldx, w, d_in, b, y, d_out, n, d_in, d_out, 0, 1, nullptr, 0, st));                 // Linear + ReLU
  } else if (norm == 1) {
    TRY(lin(x, ldx, w, d_in, b, y, d_out, n, d_in, d_out, 0, 0, nullptr, 0, st));
    TRY(occ4d_layernorm_f32(y, d_out, gamma, beta, eps, 1, y, d_out, n, d_out, st));              // LayerNorm + ReLU
  } else if (norm == 2) {
    OCC4D_REQUIRE(mean && var, "%s: BatchNorm (eval) needs the running mean and variance", who);
    TRY(lin(x, ldx, w, d_in, b, y, d_out, n, d_in, d_out, 0, 0, nullptr, 0, st));
    bn_eval_relu_kernel<<<cdiv((int64_t)n * d_out, 256), 256, 0, st>>>(y, d_out, n, d_out, mean, var, gamma, beta, eps);
    TRY(occ4d::check_launch(who));
  } else {
    OCC4D_REQUIRE(false, "%s: norm = %d (0 none, 1 layer, 2 batch in eval mode)", who, norm);
  }
  return occ4d_maxpool_gather_f32(y, d_out, nn_idx, n_new, k, d_out, z, ldz, st);
}

extern "C" int64_t occ4d_decoder_prepared_floats(const occ4d_decoder_weights* w, int flags) {
  if (check_decoder(w, "occ4d_decoder_prepared_floats")) return -1;
  return decoder_layout(*w, flags).total + ALIGN;
}
extern "C" int occ4d_decoder_prepare_f32(const occ4d_decoder_weights* w, float* prepared, int flags, void* stream) {
  TRY(check_decoder(w, "occ4d_decoder_prepare_f32"));
  OCC4D_REQUIRE(prepared && al16(prepared), "occ4d_decoder_prepare_f32: prepared buffer missing or misaligned");
  const DecoderLayout L = decoder_layout(*w, flags);
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < L.nB && L.resblock; ++i) {
    if (L.trunk4) {
      TRY(occ4d_pack_trunk4_rows_f32(w->fc0_w[i], TRUNK, TRUNK, prepared + L.w0p[i], st));
      TRY(occ4d_pack_trunk4_cols_f32(w->fc1_w[i], TRUNK, prepared + L.w1p[i], st));
    } else {
      TRY(occ4d_pack_trunk_rows_f32(w->fc0_w[i], TRUNK, TRUNK, prepared + L.w0p[i], st));
      TRY(occ4d_pack_trunk_cols_f32(w->fc1_w[i], TRUNK, prepared + L.w1p[i], st));
    }
  }
  for (int i = 0; i < L.nB && L.f16block; ++i)
    TRY(occ4d_pack_resblock_f16x3_f32(w->fc0_w[i], TRUNK, w->fc1_w[i], TRUNK, prepared + L.w0x[i], st));
  for (int i = 0; i < L.nB && L.x6trunk && !L.f16block; ++i) {
    TRY(split_pack_rowlin(L.f16, w->fc0_w[i], TRUNK, TRUNK, prepared + L.w0x[i], st));
    TRY(split_pack_rowlin(L.f16, w->fc1_w[i], TRUNK, TRUNK, prepared + L.w1x[i], st));
  }
  for (int j = 0; j < L.nC; ++j) TRY(layer_prepare(w->cross[j], L.cl[j], prepared + L.cross[j], st));
  return OCC4D_OK;
}
extern "C" int64_t occ4d_decoder_scene_floats(const occ4d_decoder_weights* w, int m) {
  if (check_decoder(w, "occ4d_decoder_scene_floats") || m < 0) return -1;
  return decoder_scene_layout(*w, m).total + ALIGN;
}
extern "C" int occ4d_decoder_prepare_scene_f32(const occ4d_decoder_weights* w, const float* prepared, const float* xyz,
                                               int64_t xyz_stride, const float* feats, int64_t ld_feats,
                                               const float* fglobal, int m, float* scene, int flags, void* stream) {
  const char* who = "occ4d_decoder_prepare_scene_f32";
  TRY(check_decoder(w, who));
  OCC4D_REQUIRE(prepared && xyz && feats && scene && al16(scene) && al16(feats) && ld_feats % 4 == 0 && xyz_stride >= 3,
                "%s: null or misaligned buffer", who);
  OCC4D_REQUIRE(m >= w->k_local && (w->n_cross == 0 || m >= w->k_cross), "%s: %d abstract points cannot serve the %d / %d "
                "neighbours the decoder asks for", who, m, w->k_local, w->k_cross);
  const DecoderLayout L = decoder_layout(*w, flags);
  const DecoderScene S = decoder_scene_layout(*w, m);
  hipStream_t st = (hipStream_t)stream;
  const int H = L.H;
  OCC4D_REQUIRE(L.dg == 0 || (fglobal && al16(fglobal)), "%s: features_global missing or misaligned", who);
  TRY(occ4d::copy_rows(scene + S.xyz, 3, xyz, xyz_stride, m, 3, st));
  for (int i = 0; i < L.nB; ++i) {
    // Z[:, i] = F (W_z_i^local)^T (m, H): the local columns of lin_z[i].weight are a strided view of the parameter
    TRY(lin(feats, ld_feats, w->lin_z_w[i] + L.dg, w->d_latent, nullptr, scene + S.ztab + (int64_t)i * H, (int64_t)L.nB * H, m,
            L.E, H, 0, 0, nullptr, 0, st));
    // c_i = W_z_i^global g + b_i (H)
    if (L.dg > 0)
      TRY(lin(fglobal, L.dg, w->lin_z_w[i], w->d_latent, w->lin_z_b[i], scene + S.zconst + (int64_t)i * H, H, 1, L.dg, H, 0, 0,
              nullptr, 0, st));
    else
      TRY(occ4d::copy_rows(scene + S.zconst + (int64_t)i * H, H, w->lin_z_b[i], H, 1, H, st));
  }
  for (int j = 0; j < L.nC; ++j)
    TRY(layer_scene(w->cross[j], L.cl[j], prepared + L.cross[j], feats, ld_feats, m, scene + S.layer[j], st));
  return OCC4D_OK;
}
extern "C" int64_t occ4d_decoder_query_workspace_floats(const occ4d_decoder_weights* w, int n, int m, int flags) {
  if (check_decoder(w, "occ4d_decoder_query_workspace_floats") || n < 0 || m < 0) return -1;
  Bump ws(nullptr);
  Events E{nullptr, nullptr};
  if (decoder_forward(*w, decoder_layout(*w, flags), nullptr, nullptr, m, nullptr, w->d_in, n, nullptr, nullptr, nullptr,
                      w->d_out, nullptr, 0, ws, flags, E, nullptr, true))
    return -1;
  return ws.peak + ALIGN;
}
extern "C" int occ4d_decoder_query_fwd_f32(const occ4d_decoder_weights* w, const float* prepared, const float* scene,
                                           int m, const float* queries, int64_t q_stride, int n,
                                           const int32_t* knn_local, const int32_t* knn_cross, float* out,
                                           int64_t ld_out, float* penult, int64_t ld_pen, float* workspace, int flags,
                                           occ4d_launch_events* ev, void* stream) {
  const char* who = "occ4d_decoder_query_fwd_f32";
  TRY(check_decoder(w, who));
  if (ev) ev->used = 0;
  OCC4D_REQUIRE(n >= 0, "%s: n = %d", who, n);
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(prepared && scene && queries && out && workspace && al16(workspace) && q_stride >= w->d_in && ld_out >= w->d_out,
                "%s: null or misaligned buffer", who);
  OCC4D_REQUIRE(!penult || (al16(penult) && ld_pen % 4 == 0 && ld_pen >= w->d_hidden), "%s: penult rows must be 16-byte aligned "
                "with ld_pen %% 4 == 0", who);
  OCC4D_REQUIRE(m >= w->k_local && (w->n_cross == 0 || m >= w->k_cross), "%s: m = %d abstract points", who, m);
  Bump ws(workspace);
  const Events E{ev, (hipStream_t)stream};
  return decoder_forward(*w, decoder_layout(*w, flags), prepared, scene, m, queries, q_stride, n, knn_local, knn_cross, out,
                         ld_out, penult, ld_pen, ws, flags, E, (hipStream_t)stream, false);
}
