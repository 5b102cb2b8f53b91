// Element-wise / gather / small-reduction kernels around the MFMA Linear:
// K2, K4, K7, K9, K10, K12, K13 of SURVEY.md §2.1.  All are HBM/L2-bound; every
// kernel maps the fastest-varying output dimension (channel) to consecutive lanes
// so that stores and the gathered row reads are coalesced 256-byte wave accesses.
#include "common.hpp"

namespace {

constexpr int TPB = 256;

inline dim3 grid1d(int64_t total) { return dim3((unsigned)((total + TPB - 1) / TPB)); }

// ---- r = relu(P1 (pos_i - pos2_j) + c1) --------------------------------------------------
__global__ __launch_bounds__(TPB) void pos_hidden_kernel(const float* __restrict__ pos, int64_t ps,
                                                         const float* __restrict__ pos2, int64_t p2s,
                                                         const int32_t* __restrict__ idx, int64_t total, int k, int h,
                                                         const float* __restrict__ P1, const float* __restrict__ c1,
                                                         float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int m = (int)(e % h);
  const int64_t p = e / h;
  const int64_t i = p / k;
  const int j = idx[p];
  const float* a = pos + i * ps;
  const float* b = pos2 + (int64_t)j * p2s;
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  const float* w = P1 + 3 * m;
  float v = fmaf(dz, w[2], fmaf(dy, w[1], dx * w[0])) + c1[m];
  out[e] = fmaxf(v, 0.f);
}

// ---- attn_in = (q_i - k_j) + pe --------------------------------------------------------
__global__ __launch_bounds__(TPB) void attn_in_kernel(const float* __restrict__ q, int64_t ldq,
                                                      const float* __restrict__ kf, int64_t ldk,
                                                      const float* __restrict__ pe, const int32_t* __restrict__ idx,
                                                      int64_t total, int k, int d, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t p = e / d;
  const int64_t i = p / k;
  const int j = idx[p];
  out[e] = (q[i * ldq + c] - kf[(int64_t)j * ldk + c]) + pe[e];
}

// ---- per-channel softmax over the k neighbours + weighted sum ------------------------------
template <int KMAX>
__global__ __launch_bounds__(TPB) void softmax_agg_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ v, int64_t ldv,
                                                          const float* __restrict__ pe,
                                                          const int32_t* __restrict__ idx, int64_t total, int k, int d,
                                                          float divisor, float* __restrict__ agg, int64_t ld_agg) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float a[KMAX];
  float mx = -__builtin_inff();
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      a[j] = logits[(i * k + j) * d + c] / divisor;
      mx = fmaxf(mx, a[j]);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      a[j] = expf(a[j] - mx);
      den += a[j];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      const int64_t p = i * k + j;
      float val = v[(int64_t)idx[p] * ldv + c];
      if (pe) val += pe[p * d + c];
      s += (a[j] / den) * val;
    }
  }
  agg[i * ld_agg + c] = s;
}

// ---- LayerNorm (+ReLU): one wave per row ------------------------------------------------------
__global__ __launch_bounds__(TPB) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int relu_out,
                                                        float* __restrict__ y, int64_t ldy, int n, int d) {
  const int row = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* xr = x + (int64_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = xr[c] - mean;
    q += t * t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)d + eps);
  float* yr = y + (int64_t)row * ldy;
  for (int c = lane; c < d; c += 64) {
    float t = (xr[c] - mean) * rstd;
    if (gamma) t = t * gamma[c] + beta[c];
    yr[c] = relu_out ? fmaxf(t, 0.f) : t;
  }
}

// ---- k-way max pool of gathered rows --------------------------------------------------------
__global__ __launch_bounds__(TPB) void maxpool_gather_kernel(const float* __restrict__ y, int64_t ldy,
                                                             const int32_t* __restrict__ idx, int64_t total, int k,
                                                             int d, float* __restrict__ z, int64_t ldz) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float m = y[(int64_t)idx[i * k] * ldy + c];
  for (int j = 1; j < k; ++j) m = fmaxf(m, y[(int64_t)idx[i * k + j] * ldy + c]);
  z[i * ldz + c] = m;
}

__global__ __launch_bounds__(TPB) void gather_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                          const int32_t* __restrict__ idx, int64_t total, int d,
                                                          float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  out[i * ldo + c] = src[(int64_t)idx[i] * lds + c];
}

// ---- column mean: 32 channels x 8 row groups per workgroup; a thread sums the rows of its group in order, the 8
// partials are added in a fixed order (deterministic).  (One thread per channel over all rows took 108 us for the
// encoder's (531, 288) global pool, on the critical path behind the last FPS.)
__global__ __launch_bounds__(TPB) void mean_rows_kernel(const float* __restrict__ x, int64_t ldx, int n, int d,
                                                        float* __restrict__ out) {
  static_assert(TPB == 256, "mean_rows_kernel assumes 256 threads");
  __shared__ float part[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < d)
    for (int i = g; i < n; i += 8) s += x[(int64_t)i * ldx + c];
  part[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < d) {
    float t = part[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += part[k][cl];
    out[c] = t / (float)n;
  }
}

// ---- Fourier features -------------------------------------------------------------------------
struct PosencFreqs { float w[16]; };

__global__ __launch_bounds__(TPB) void posenc_kernel(const float* __restrict__ pts, int64_t stride, int64_t total,
                                                     int c, int width, PosencFreqs fr, float* __restrict__ out,
                                                     int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int col = (int)(e % width);
  const int64_t i = e / width;
  const float* p = pts + i * stride;
  float v;
  if (col < c) {
    v = p[col];
  } else {
    const int f = (col - c) / (2 * c), r = (col - c) % (2 * c);
    const float arg = p[r % c] * fr.w[f];
    v = r < c ? sinf(arg) : cosf(arg);
  }
  out[i * ldo + col] = v;
}

__global__ __launch_bounds__(TPB) void interp_weights_kernel(const float* __restrict__ dist, int n, int k,
                                                             float* __restrict__ w) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int j = 0; j < k; ++j) {
    const float t = 1.0f / (dist[(int64_t)i * k + j] + 1e-4f);
    w[(int64_t)i * k + j] = t;
    s += fabsf(t);
  }
  s = fmaxf(s, 1e-12f);
  for (int j = 0; j < k; ++j) w[(int64_t)i * k + j] = w[(int64_t)i * k + j] / s;
}

__global__ __launch_bounds__(TPB) void interp_add_kernel(float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ cvec,
                                                         const float* __restrict__ table, int64_t ldt,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                         int64_t total, int k, int d) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % d);
  const int64_t i = e / d;
  float s = 0.f;
  for (int j = 0; j < k; ++j) s += w[i * k + j] * table[(int64_t)idx[i * k + j] * ldt + c];
  if (cvec) s += cvec[c];
  x[i * ldx + c] += s;
}

// float4 variant: one lane owns 4 consecutive channels -> 16-byte gathers from the L2-resident table.
// KC > 0: compile-time neighbour count, fully unrolled -- the KC index / weight loads and then the KC table
// gathers are all in flight before the first add (the runtime-k loop was a chain of dependent loads).
// Same operation order in both: s = (...((w0 t0) + w1 t1) + ...), then + cvec, then + x.
template <int KC>
__global__ __launch_bounds__(TPB) void interp_add4_kernel(float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ cvec,
                                                          const float* __restrict__ table, int64_t ldt,
                                                          const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                          int64_t total, int k, int d4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = 4 * (int)(e % d4);
  const int64_t i = e / d4;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (KC > 0) {
    int jj[KC > 0 ? KC : 1];
    float wj[KC > 0 ? KC : 1];
    f4 t[KC > 0 ? KC : 1];
#pragma unroll
    for (int j = 0; j < KC; ++j) { jj[j] = idx[i * KC + j]; wj[j] = w[i * KC + j]; }
#pragma unroll
    for (int j = 0; j < KC; ++j) t[j] = *reinterpret_cast<const f4*>(table + (int64_t)jj[j] * ldt + c);
#pragma unroll
    for (int j = 0; j < KC; ++j) { s.x += wj[j] * t[j].x; s.y += wj[j] * t[j].y; s.z += wj[j] * t[j].z; s.w += wj[j] * t[j].w; }
  } else {
    for (int j = 0; j < k; ++j) {
      const float wj = w[i * k + j];
      const f4 t = *reinterpret_cast<const f4*>(table + (int64_t)idx[i * k + j] * ldt + c);
      s.x += wj * t.x; s.y += wj * t.y; s.z += wj * t.z; s.w += wj * t.w;
    }
  }
  if (cvec) {
    const f4 cv = *reinterpret_cast<const f4*>(cvec + c);
    s.x += cv.x; s.y += cv.y; s.z += cv.z; s.w += cv.w;
  }
  f4* xp = reinterpret_cast<f4*>(x + i * ldx + c);
  f4 xv = *xp;
  xv.x += s.x; xv.y += s.y; xv.z += s.z; xv.w += s.w;
  *xp = xv;
}

struct SquashOps { int32_t op[32]; };

__global__ __launch_bounds__(TPB) void squash_kernel(float* __restrict__ out, int64_t ld, int64_t total, int g,
                                                     SquashOps ops) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % g);
  const int64_t i = e / g;
  float v = out[i * ld + c];
  const int op = ops.op[c];
  if (op == 1) v = 1.0f / (1.0f + expf(-v));
  else if (op == 2) v = fminf(fmaxf(v, 0.f), 1.f);
  out[i * ld + c] = v;
}


// C[m][n] = sum_k A[m sam + k sak] * B[k sbk + n sbn] in fp64 (element strides: either operand may be a transposed
// view).  The merged weights of refactoring (i) (W1 Wq, W1 Wk, W1 P2, W1 c2 + b1 -- a few hundred MFLOP
// once per weight update) were torch fp64 matmuls, i.e. the one vendor-BLAS call on the path; 16 x 16 output tiles
// through LDS, one thread per output.
__global__ __launch_bounds__(256) void matmul_f64_kernel(const double* __restrict__ A, int64_t sam, int64_t sak,
                                                         const double* __restrict__ B, int64_t sbk, int64_t sbn,
                                                         double* __restrict__ Cm, int M, int N, int K) {
  __shared__ double sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  double acc = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    sa[ty][tx] = (m < M && ka < K) ? A[(int64_t)m * sam + (int64_t)ka * sak] : 0.0;
    sb[ty][tx] = (kb < K && n < N) ? B[(int64_t)kb * sbk + (int64_t)n * sbn] : 0.0;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = fma(sa[ty][kk], sb[kk][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) Cm[(int64_t)m * N + n] = acc;
}

}  // namespace

extern "C" {

int occ4d_pt_pos_hidden_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s, const int32_t* idx,
                            int n, int k, const float* P1, const float* c1, int h, float* out, void* stream) {
  OCC4D_REQUIRE(pos && pos2 && idx && P1 && c1 && out, "occ4d_pt_pos_hidden_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && h >= 1 && ps >= 3 && p2s >= 3, "occ4d_pt_pos_hidden_f32: bad sizes");
  const int64_t total = (int64_t)n * k * h;
  if (!total) return OCC4D_OK;
  pos_hidden_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(pos, ps, pos2, p2s, idx, total, k, h, P1, c1, out);
  return occ4d::check_launch("occ4d_pt_pos_hidden_f32");
}

int occ4d_pt_attn_in_f32(const float* q, int64_t ldq, const float* kfeat, int64_t ldk, const float* pe,
                         const int32_t* idx, int n, int k, int d, float* out, void* stream) {
  OCC4D_REQUIRE(q && kfeat && pe && idx && out, "occ4d_pt_attn_in_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && d >= 1 && ldq >= d && ldk >= d, "occ4d_pt_attn_in_f32: bad sizes");
  const int64_t total = (int64_t)n * k * d;
  if (!total) return OCC4D_OK;
  attn_in_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(q, ldq, kfeat, ldk, pe, idx, total, k, d, out);
  return occ4d::check_launch("occ4d_pt_attn_in_f32");
}

int occ4d_pt_softmax_agg_f32(const float* logits, const float* v, int64_t ldv, const float* pe, const int32_t* idx,
                             int n, int k, int d, float divisor, float* agg, int64_t ld_agg, void* stream) {
  OCC4D_REQUIRE(logits && v && idx && agg, "occ4d_pt_softmax_agg_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && k <= 16 && d >= 1 && ldv >= d && ld_agg >= d, "occ4d_pt_softmax_agg_f32: bad sizes");
  OCC4D_REQUIRE(divisor > 0.f, "occ4d_pt_softmax_agg_f32: divisor must be > 0");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  softmax_agg_kernel<16><<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(logits, v, ldv, pe, idx, total, k, d, divisor, agg, ld_agg);
  return occ4d::check_launch("occ4d_pt_softmax_agg_f32");
}

int occ4d_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int relu_out,
                        float* y, int64_t ldy, int n, int d, void* stream) {
  OCC4D_REQUIRE(x && y, "occ4d_layernorm_f32: null pointer");
  OCC4D_REQUIRE((gamma == nullptr) == (beta == nullptr), "occ4d_layernorm_f32: gamma/beta must both be set or NULL");
  OCC4D_REQUIRE(n >= 0 && d >= 1 && ldx >= d && ldy >= d, "occ4d_layernorm_f32: bad sizes");
  if (!n) return OCC4D_OK;
  layernorm_kernel<<<occ4d::cdiv(n, TPB / 64), TPB, 0, (hipStream_t)stream>>>(x, ldx, gamma, beta, eps, relu_out, y, ldy, n, d);
  return occ4d::check_launch("occ4d_layernorm_f32");
}

int occ4d_maxpool_gather_f32(const float* y, int64_t ldy, const int32_t* idx, int n_out, int k, int d, float* z,
                             int64_t ldz, void* stream) {
  OCC4D_REQUIRE(y && idx && z, "occ4d_maxpool_gather_f32: null pointer");
  OCC4D_REQUIRE(n_out >= 0 && k >= 1 && d >= 1 && ldy >= d && ldz >= d, "occ4d_maxpool_gather_f32: bad sizes");
  const int64_t total = (int64_t)n_out * d;
  if (!total) return OCC4D_OK;
  maxpool_gather_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(y, ldy, idx, total, k, d, z, ldz);
  return occ4d::check_launch("occ4d_maxpool_gather_f32");
}

int occ4d_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int n_out, int d, float* out,
                          int64_t ldo, void* stream) {
  OCC4D_REQUIRE(src && idx && out, "occ4d_gather_rows_f32: null pointer");
  OCC4D_REQUIRE(n_out >= 0 && d >= 1 && lds >= d && ldo >= d, "occ4d_gather_rows_f32: bad sizes");
  const int64_t total = (int64_t)n_out * d;
  if (!total) return OCC4D_OK;
  gather_rows_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(src, lds, idx, total, d, out, ldo);
  return occ4d::check_launch("occ4d_gather_rows_f32");
}

int occ4d_mean_rows_f32(const float* x, int64_t ldx, int n, int d, float* out, void* stream) {
  OCC4D_REQUIRE(x && out, "occ4d_mean_rows_f32: null pointer");
  OCC4D_REQUIRE(n >= 1 && d >= 1 && ldx >= d, "occ4d_mean_rows_f32: bad sizes");
  mean_rows_kernel<<<occ4d::cdiv(d, 32), TPB, 0, (hipStream_t)stream>>>(x, ldx, n, d, out);
  return occ4d::check_launch("occ4d_mean_rows_f32");
}

int occ4d_posenc_f32(const float* pts, int64_t stride, int n, int c, int n_freq, double base_freq, float* out,
                     int64_t ldo, void* stream) {
  OCC4D_REQUIRE(pts && out, "occ4d_posenc_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && c >= 1 && n_freq >= 0 && n_freq <= 16 && stride >= c, "occ4d_posenc_f32: bad sizes");
  const int width = c * (2 * n_freq + 1);
  OCC4D_REQUIRE(ldo >= width, "occ4d_posenc_f32: ldo too small");
  PosencFreqs fr;
  for (int f = 0; f < 16; ++f) {
    // model/implicit.py:33-34: cur_freq = base * 2**p; omega = cur_freq * pi * 2.0 (double), then fp32
    const double omega = base_freq * (double)(1u << f) * 3.141592653589793 * 2.0;
    fr.w[f] = (float)omega;
  }
  const int64_t total = (int64_t)n * width;
  if (!total) return OCC4D_OK;
  posenc_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(pts, stride, total, c, width, fr, out, ldo);
  return occ4d::check_launch("occ4d_posenc_f32");
}

int occ4d_interp_weights_f32(const float* dist, int n, int k, float* w, void* stream) {
  OCC4D_REQUIRE(dist && w, "occ4d_interp_weights_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1, "occ4d_interp_weights_f32: bad sizes");
  if (!n) return OCC4D_OK;
  interp_weights_kernel<<<occ4d::cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(dist, n, k, w);
  return occ4d::check_launch("occ4d_interp_weights_f32");
}

int occ4d_interp_add_f32(float* x, int64_t ldx, const float* cvec, const float* table, int64_t ldt,
                         const int32_t* idx, const float* w, int n, int k, int d, void* stream) {
  OCC4D_REQUIRE(x && table && idx && w, "occ4d_interp_add_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && k >= 1 && d >= 1 && ldx >= d && ldt >= d, "occ4d_interp_add_f32: bad sizes");
  const int64_t total = (int64_t)n * d;
  if (!total) return OCC4D_OK;
  const bool vec = d % 4 == 0 && ldx % 4 == 0 && ldt % 4 == 0 && ((uintptr_t)x % 16) == 0 &&
                   ((uintptr_t)table % 16) == 0 && (!cvec || ((uintptr_t)cvec % 16) == 0);
  if (vec) {
    const int64_t total4 = (int64_t)n * (d / 4);
    if (k == 8)
      interp_add4_kernel<8><<<grid1d(total4), TPB, 0, (hipStream_t)stream>>>(x, ldx, cvec, table, ldt, idx, w, total4, k, d / 4);
    else
      interp_add4_kernel<0><<<grid1d(total4), TPB, 0, (hipStream_t)stream>>>(x, ldx, cvec, table, ldt, idx, w, total4, k, d / 4);
  } else {
    interp_add_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(x, ldx, cvec, table, ldt, idx, w, total, k, d);
  }
  return occ4d::check_launch("occ4d_interp_add_f32");
}

int occ4d_squash_f32(float* out, int64_t ld, int n, int g, const int32_t* ops_host, void* stream) {
  OCC4D_REQUIRE(out && ops_host, "occ4d_squash_f32: null pointer");
  OCC4D_REQUIRE(n >= 0 && g >= 1 && g <= 32 && ld >= g, "occ4d_squash_f32: need 1 <= g <= 32, ld >= g");
  SquashOps ops;
  for (int i = 0; i < 32; ++i) ops.op[i] = i < g ? ops_host[i] : 0;
  for (int i = 0; i < g; ++i) OCC4D_REQUIRE(ops.op[i] >= 0 && ops.op[i] <= 2, "occ4d_squash_f32: op code %d", ops.op[i]);
  const int64_t total = (int64_t)n * g;
  if (!total) return OCC4D_OK;
  squash_kernel<<<grid1d(total), TPB, 0, (hipStream_t)stream>>>(out, ld, total, g, ops);
  return occ4d::check_launch("occ4d_squash_f32");
}

}  // extern "C"

extern "C" int occ4d_matmul_f64(const double* a, int64_t sam, int64_t sak, const double* b, int64_t sbk, int64_t sbn,
                                double* c, int m, int n, int k, void* stream) {
  OCC4D_REQUIRE(a && b && c && m >= 0 && n >= 0 && k >= 1, "occ4d_matmul_f64: bad arguments");
  if (m == 0 || n == 0) return OCC4D_OK;
  matmul_f64_kernel<<<dim3(occ4d::cdiv(n, 16), occ4d::cdiv(m, 16)), 256, 0, (hipStream_t)stream>>>(a, sam, sak, b, sbk,
                                                                                                   sbn, c, m, n, k);
  return occ4d::check_launch("occ4d_matmul_f64");
}
