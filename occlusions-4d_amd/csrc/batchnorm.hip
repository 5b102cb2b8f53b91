// BatchNorm1d in TRAINING mode for DownTransition(norm_type='batch') (model/modules.py:98-102, 152: Linear -> BatchNorm1d
// (eps 1e-3, momentum 0.1) -> ReLU over the B N rows of a level): batch statistics, normalise + ReLU, and the backward
// through both.  Column statistics are accumulated in fp64 (two-stage: per row chunk, then per column), so the result
// does not depend on the chunking beyond fp64 rounding.
#include "common.hpp"

namespace {

constexpr int BN_COLS = 64;       // columns per workgroup
constexpr int BN_LANES = 4;       // row lanes per column (256 threads)
constexpr int BN_CHUNK = 512;     // rows per workgroup

// partial[chunk][0 / 1][d] (doubles): sum_r a(r, c), sum_r b(r, c) over the chunk's rows, where
//   MODE 0 (forward statistics): a = y, b = y^2
//   MODE 1 (backward):           a = gm, b = gm * xhat,  gm = out > 0 ? g : 0, xhat = (y - mean) * rstd
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ g,
                                                         int64_t ldg, const float* __restrict__ out, int64_t ldo,
                                                         const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                         int n, int d, double* __restrict__ partial) {
  __shared__ double sa[BN_LANES][BN_COLS], sb[BN_LANES][BN_COLS];
  const int cl = threadIdx.x % BN_COLS, rl = threadIdx.x / BN_COLS;
  const int c = blockIdx.y * BN_COLS + cl;
  const int r0 = blockIdx.x * BN_CHUNK;
  double a = 0.0, b = 0.0;
  if (c < d) {
    float mu = 0.f, rs = 0.f;
    if (MODE == 1) { mu = mean[c]; rs = 1.0f / sqrtf(var[c] + eps); }
    for (int r = r0 + rl; r < min(r0 + BN_CHUNK, n); r += BN_LANES) {
      const float v = y[(int64_t)r * ldy + c];
      if (MODE == 0) {
        a += (double)v;
        b += (double)v * (double)v;
      } else {
        const float gm = out[(int64_t)r * ldo + c] > 0.f ? g[(int64_t)r * ldg + c] : 0.f;
        a += (double)gm;
        b += (double)gm * (double)((v - mu) * rs);
      }
    }
  }
  sa[rl][cl] = a;
  sb[rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < d) {
#pragma unroll
    for (int i = 1; i < BN_LANES; ++i) { a += sa[i][cl]; b += sb[i][cl]; }
    partial[((int64_t)blockIdx.x * 2 + 0) * d + c] = a;
    partial[((int64_t)blockIdx.x * 2 + 1) * d + c] = b;
  }
}

// MODE 0: mean = S1 / n, var (biased) = S2 / n - mean^2;  MODE 1: o0 = S1 (dbeta), o1 = S2 (dgamma)
template <int MODE>
__global__ void bn_final_kernel(const double* __restrict__ partial, int chunks, int n, int d, float* __restrict__ o0,
                                float* __restrict__ o1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < chunks; ++k) {
    a += partial[((int64_t)k * 2 + 0) * d + c];
    b += partial[((int64_t)k * 2 + 1) * d + c];
  }
  if (MODE == 0) {
    const double m = a / n;
    o0[c] = (float)m;
    o1[c] = (float)fmax(b / n - m * m, 0.0);
  } else {
    o0[c] = (float)a;
    o1[c] = (float)b;
  }
}

// out = relu(gamma (y - mean) / sqrt(var + eps) + beta)   (torch's op order, as the eval kernel of csrc/path.hip)
__global__ void bn_relu_kernel(const float* __restrict__ y, int64_t ldy, int n, int d, const float* __restrict__ mean,
                               const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float eps, float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * d) return;
  const int c = (int)(i % d);
  const int64_t r = i / d;
  const float inv = 1.0f / sqrtf(var[c] + eps);
  const float v = (y[r * ldy + c] - mean[c]) * inv * gamma[c] + beta[c];
  out[r * ldo + c] = fmaxf(v, 0.f);
}

// dx = gamma rstd (gm - dbeta / n - xhat dgamma / n)
__global__ void bn_relu_bwd_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ g, int64_t ldg,
                                   const float* __restrict__ out, int64_t ldo, const float* __restrict__ mean,
                                   const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                                   const float* __restrict__ dbeta, const float* __restrict__ dgamma, int n, int d,
                                   float* __restrict__ dx, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * d) return;
  const int c = (int)(i % d);
  const int64_t r = i / d;
  const float rs = 1.0f / sqrtf(var[c] + eps);
  const float xhat = (y[r * ldy + c] - mean[c]) * rs;
  const float gm = out[r * ldo + c] > 0.f ? g[r * ldg + c] : 0.f;
  const float inv_n = 1.0f / (float)n;
  dx[r * ldx + c] = gamma[c] * rs * (gm - dbeta[c] * inv_n - xhat * dgamma[c] * inv_n);
}

}  // namespace

extern "C" int64_t occ4d_bn_workspace_doubles(int n, int d) { return (int64_t)occ4d::cdiv(n, BN_CHUNK) * 2 * d; }

extern "C" int occ4d_bn_train_fwd_f32(const float* y, int64_t ldy, int n, int d, const float* gamma, const float* beta,
                                      float eps, float* mean, float* var, float* out, int64_t ldo, double* workspace,
                                      void* stream) {
  OCC4D_REQUIRE(y && gamma && beta && mean && var && out && workspace && n >= 2 && d >= 1,
                "occ4d_bn_train_fwd_f32: bad arguments (n = %d rows: batch statistics need at least 2)", n);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)occ4d::cdiv(n, BN_CHUNK);
  const dim3 grid(chunks, (unsigned)occ4d::cdiv(d, BN_COLS));
  bn_partial_kernel<0><<<grid, 256, 0, st>>>(y, ldy, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f, n, d, workspace);
  bn_final_kernel<0><<<occ4d::cdiv(d, 256), 256, 0, st>>>(workspace, chunks, n, d, mean, var);
  bn_relu_kernel<<<occ4d::cdiv((int64_t)n * d, 256), 256, 0, st>>>(y, ldy, n, d, mean, var, gamma, beta, eps, out, ldo);
  return occ4d::check_launch("occ4d_bn_train_fwd_f32");
}

extern "C" int occ4d_bn_train_bwd_f32(const float* y, int64_t ldy, const float* g, int64_t ldg, const float* out, int64_t ldo,
                                      int n, int d, const float* mean, const float* var, const float* gamma, float eps,
                                      float* dx, int64_t ldx, float* dgamma, float* dbeta, double* workspace, void* stream) {
  OCC4D_REQUIRE(y && g && out && mean && var && gamma && dx && dgamma && dbeta && workspace && n >= 2 && d >= 1,
                "occ4d_bn_train_bwd_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)occ4d::cdiv(n, BN_CHUNK);
  const dim3 grid(chunks, (unsigned)occ4d::cdiv(d, BN_COLS));
  bn_partial_kernel<1><<<grid, 256, 0, st>>>(y, ldy, g, ldg, out, ldo, mean, var, eps, n, d, workspace);
  bn_final_kernel<1><<<occ4d::cdiv(d, 256), 256, 0, st>>>(workspace, chunks, n, d, dbeta, dgamma);
  bn_relu_bwd_kernel<<<occ4d::cdiv((int64_t)n * d, 256), 256, 0, st>>>(y, ldy, g, ldg, out, ldo, mean, var, gamma, eps, dbeta,
                                                                        dgamma, n, d, dx, ldx);
  return occ4d::check_launch("occ4d_bn_train_bwd_f32");
}
