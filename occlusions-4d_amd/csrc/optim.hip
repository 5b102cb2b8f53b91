// Gradient clipping + AdamW over ALL parameters of the two networks as one multi-tensor launch sequence (train.py:107-109:
// torch.nn.utils.clip_grad_norm_(max_norm 0.2) + the optimiser step of train.py:287-293).  The reference runs ~150
// per-parameter / per-group ATen launches here with ~15 ms of host work per step for 0.5 ms of device work
// (profiles/r05_train_phases.txt); the parameters and both moment buffers live in flat arrays, the gradients stay where
// the backward pass left them (a table of their addresses is the only per-step upload), and three launches do the step
// with no host read: per-chunk sums of squares -> norm and clip coefficient -> update.
#include "common.hpp"

namespace {

constexpr int OPT_TPB = 256;
constexpr int OPT_CHUNK = 4096;           // elements per workgroup (occ4d_adamw_chunk())

struct OptTables {
  const int64_t* grad_ptr;                // [2 T]: [t] address of tensor t's gradient, 0 = no gradient this step (skipped: no decay,
                                          //     no moment update -- torch's behaviour for .grad is None); [T + t] = two floats:
                                          //     (1 - beta1^k, sqrt(1 - beta2^k)), k = the number of updates tensor t has had
                                          //     including this one (torch keeps the step count per parameter)
  const int64_t* offset;                  // [T] first element of tensor t in the flat arrays
  const int64_t* numel;                   // [T]
  const int32_t* chunk_tensor;            // [C]
  const int32_t* chunk_start;             // [C] first element of the chunk inside its tensor
};

__global__ __launch_bounds__(OPT_TPB) void grad_sumsq_kernel(const OptTables tb, float* __restrict__ partial) {
  __shared__ float s_sum[OPT_TPB / 64];
  const int t = tb.chunk_tensor[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(tb.grad_ptr[t]);
  float acc = 0.f;
  if (g) {
    const int64_t n = tb.numel[t], lo = tb.chunk_start[blockIdx.x];
    const int64_t hi = lo + OPT_CHUNK < n ? lo + OPT_CHUNK : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += OPT_TPB) acc = fmaf(g[i], g[i], acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

// total norm (fp64 sum of the chunk sums: order fixed, reproducible) and clip_coef = min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(1024) void grad_norm_kernel(const float* __restrict__ partial, int n_chunks, float max_norm,
                                                         float* __restrict__ out) {
  __shared__ double s[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_chunks; i += 1024) acc += (double)partial[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s[0]);
    out[0] = norm;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    out[1] = coef;
  }
}

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay;
  int n_tensors;
};

// torch.optim.AdamW (amsgrad off), op for op: p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g;
// p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps), with g already multiplied by the clip coefficient
__global__ __launch_bounds__(OPT_TPB) void adamw_kernel(const OptTables tb, const AdamArgs a, const float* __restrict__ coef_ptr,
                                                        float* __restrict__ p_flat, float* __restrict__ m_flat,
                                                        float* __restrict__ v_flat) {
  const int t = tb.chunk_tensor[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(tb.grad_ptr[t]);
  if (!g) return;
  const float coef = coef_ptr[0];
  const int64_t n = tb.numel[t], lo = tb.chunk_start[blockIdx.x], base = tb.offset[t];
  const int64_t hi = lo + OPT_CHUNK < n ? lo + OPT_CHUNK : n;
  const float2 bias = reinterpret_cast<const float2*>(tb.grad_ptr + a.n_tensors)[t];
  const float decay = 1.f - a.lr * a.weight_decay, step = a.lr / bias.x;
  for (int64_t i = lo + threadIdx.x; i < hi; i += OPT_TPB) {
    const float gi = g[i] * coef;
    float p = p_flat[base + i] * decay;
    float m = m_flat[base + i];
    m = m + (1.f - a.beta1) * (gi - m);
    const float v = a.beta2 * v_flat[base + i] + (1.f - a.beta2) * gi * gi;
    const float denom = sqrtf(v) / bias.y + a.eps;
    p = p - step * (m / denom);
    p_flat[base + i] = p;
    m_flat[base + i] = m;
    v_flat[base + i] = v;
  }
}

}  // namespace

extern "C" int occ4d_adamw_chunk(void) { return OPT_CHUNK; }

extern "C" int occ4d_adamw_clip_f32(float* params_flat, float* exp_avg, float* exp_avg_sq, const int64_t* grad_ptrs,
                                    const int64_t* offsets, const int64_t* numels, int n_tensors, const int32_t* chunk_tensor,
                                    const int32_t* chunk_start, int n_chunks, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float max_norm, float* workspace, void* stream) {
  const char* who = "occ4d_adamw_clip_f32";
  OCC4D_REQUIRE(params_flat && exp_avg && exp_avg_sq && grad_ptrs && offsets && numels && chunk_tensor && chunk_start &&
                    workspace, "%s: null pointer", who);
  OCC4D_REQUIRE(n_tensors >= 1 && n_chunks >= 1, "%s: n_tensors = %d, n_chunks = %d", who, n_tensors, n_chunks);
  OCC4D_REQUIRE(lr >= 0.f && eps >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "%s: bad hyper-parameters", who);
  hipStream_t st = (hipStream_t)stream;
  const OptTables tb{grad_ptrs, offsets, numels, chunk_tensor, chunk_start};
  grad_sumsq_kernel<<<n_chunks, OPT_TPB, 0, st>>>(tb, workspace);
  grad_norm_kernel<<<1, 1024, 0, st>>>(workspace, n_chunks, max_norm, workspace + n_chunks);
  const AdamArgs a{lr, beta1, beta2, eps, weight_decay, n_tensors};
  adamw_kernel<<<n_chunks, OPT_TPB, 0, st>>>(tb, a, workspace + n_chunks + 1, params_flat, exp_avg, exp_avg_sq);
  return occ4d::check_launch(who);
}
