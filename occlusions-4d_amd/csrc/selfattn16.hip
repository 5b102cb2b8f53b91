// Fused vector attention over K = 16 neighbours for the ENCODER widths (E3 of SURVEY.md 8(a),
// model/point_transformer_layer.py:168-179 with d = 36 / 72 / 144 / 288, the self-attention of every
// PointTransformerBlock of model/model.py:73-90): everything between the per-point projections and the aggregated
// output stays on chip -- the (n 16, 2d) hidden activations, the (n 16, d) logits and positional encodings of the
// reference (and of the unfused kernel chain this replaces: pos_hidden -> Linear -> Linear -> Linear -> softmax_agg,
// five launches and four pair tensors through HBM) are never written.
//
// The arithmetic is tiny (d = 36: 12 KFLOP per pair); what the chain paid for was HBM round trips of the pair tensors
// and launches.  So this kernel is written for simplicity, not for the last MFMA cycle:
//   wave      = ONE query point: its 16 neighbours are the 16 columns of v_mfma_f32_16x16x4_f32, lane (g, c) = (lane >> 4,
//               lane & 15) works on pair c; all d channels of the query live in NT = ceil(d / 16) accumulator tiles.
//   workgroup = 4 waves (4 queries) sharing the weight stream through LDS: hidden units in stages of 16
//               (W2[:, 16 s .. 16 s + 15] as NT tiles of 1 KB + the 16 rows of Wp), double buffered, filled from the
//               reference-layout matrices by the workgroup itself (zero padding to the tile grid happens here: d and 2 d
//               need only be multiples of 4); P2 (d x 32) stays resident.
//   GEMM1 (transposed)  a^T[hid][pair] = Wp[hid][:] . r[pair][:]  (K = 32, r = relu(P1 (p_i - p_j) + c1) computed in
//               registers), accumulator initialised with aq_i[hid] - kt_j[hid]; its C/D registers (lane: 4 hidden units
//               of pair c) ARE the B operand of
//   GEMM2       logits^T[ch][pair] += W2[ch][hid] . relu(a)^T[hid][pair]: MFMA step e of a stage consumes the hidden
//               units 16 s + 4 g + e, and the W2 fragment is read from LDS with the same map (one ds_read_b128 per tile).
//   epilogue    per channel tile: pe^T = P2 . r^T (8 MFMAs, same r registers), val = vt_j + pe + c2, softmax over the 16
//               pairs = the 16 lanes of a DPP row (quad_perm / row_half_mirror / row_mirror butterflies: no LDS), one
//               float4 store per 4 channels.  attn_mlp[2].bias is not added: constant over the neighbour axis of the
//               softmax, it cancels.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SelfAttnArgs {
  const float* aq; int64_t ld_aq;        // (n, 2d)  (W1 Wq) x + merged bias
  const float* kt; int64_t ld_kt;        // (m, 2d)  (W1 Wk) x2
  const float* vt; int64_t ld_vt;        // (m, d)   to_v(x2)
  const float* qpos; int64_t qs;         // (n, 3)
  const float* apos; int64_t as;         // (m, 3)
  const int32_t* idx;                    // (n, 16)
  const float* P1; const float* c1;      // (32, 3), (32)
  const float* wp;                       // (2d, 32)  W1 P2
  const float* w2;                       // (d, 2d)   attn_mlp[2].weight
  const float* p2; const float* c2;      // (d, 32), (d)
  float* agg; int64_t ld_agg;
  int n, d;
  float sc;                              // log2(e) / divisor
};

// all-reduce over the 16 lanes of a DPP row (the 16 neighbours of this wave's query)
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp<0xB1>(v));            // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp<0x4E>(v));            // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp<0x141>(v));           // row_half_mirror (quads hold one value each: 0 <-> 1, 2 <-> 3)
  v = fmaxf(v, dpp<0x140>(v));           // row_mirror      (halves hold one value each)
  return v;
}
__device__ __forceinline__ float row_sum(float v) {
  v += dpp<0xB1>(v);
  v += dpp<0x4E>(v);
  v += dpp<0x141>(v);
  v += dpp<0x140>(v);
  return v;
}

template <int NT>
__global__ __launch_bounds__(256) void self_attn16_kernel(const SelfAttnArgs a) {
  constexpr int W2S = NT * 256;                        // floats of a W2 stage: NT tiles [16 ch][16 hid]
  constexpr int WPS = 16 * 32;                         // floats of a Wp stage: [16 hid][32]
  constexpr int W2F4 = (NT * 64 + 255) / 256;          // float4 per thread of a W2 stage
  __shared__ __attribute__((aligned(16))) float s_w2[2][W2S];
  __shared__ __attribute__((aligned(16))) float s_wp[2][WPS];
  __shared__ __attribute__((aligned(16))) float s_p2[NT * 16 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int d = a.d, d2 = 2 * a.d;
  const int n_stage = (d2 + 15) / 16;
  const int q = min(blockIdx.x * 4 + wave, a.n - 1);    // (a workgroup's surplus waves recompute the last query)
  const bool live = blockIdx.x * 4 + wave < a.n;

  // ---- weight stage loads (global -> registers now, registers -> LDS after the stage's MFMAs)
  f32x4 rw2[W2F4], rwp;
  auto fetch = [&](int s) {
    const int hid0 = 16 * s;
#pragma unroll
    for (int i = 0; i < W2F4; ++i) {
      const int e = tid + 256 * i;                      // tile t = e / 64, channel row m = (e % 64) / 4, 4 hidden units
      const int ch = 16 * (e >> 6) + ((e & 63) >> 2), hid = hid0 + 4 * (e & 3);
      const bool ok = e < NT * 64 && ch < d && hid < d2;
      rw2[i] = ok ? *reinterpret_cast<const f32x4*>(a.w2 + (int64_t)ch * d2 + hid) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row = hid0 + (tid >> 3);
    rwp = (tid < 128 && row < d2) ? *reinterpret_cast<const f32x4*>(a.wp + (int64_t)row * 32 + 4 * (tid & 7))
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < W2F4; ++i) {
      const int e = tid + 256 * i;
      if (e < NT * 64) *reinterpret_cast<f32x4*>(&s_w2[buf][4 * e]) = rw2[i];
    }
    if (tid < 128) *reinterpret_cast<f32x4*>(&s_wp[buf][4 * tid]) = rwp;
  };
  fetch(0);
  for (int e = tid; e < NT * 16 * 8; e += 256) {         // P2, resident: [ch][32], rows past d zero
    const int ch = e >> 3;
    *reinterpret_cast<f32x4*>(&s_p2[4 * e]) =
        ch < d ? *reinterpret_cast<const f32x4*>(a.p2 + (int64_t)ch * 32 + 4 * (e & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  stash(0);

  // ---- this lane's pair: neighbour index, r[8 g .. 8 g + 7] = relu(P1 (p_i - p_j) + c1)  (operation order of
  // occ4d_pt_pos_hidden_f32)
  const int nb = a.idx[(int64_t)q * 16 + c];
  float r[8];
  {
    const float* pa = a.qpos + (int64_t)q * a.qs;
    const float* pb = a.apos + (int64_t)nb * a.as;
    const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* w = a.P1 + 3 * (8 * g + s);
      r[s] = fmaxf(fmaf(dz, w[2], fmaf(dy, w[1], dx * w[0])) + a.c1[8 * g + s], 0.f);
    }
  }
  const float* aq_row = a.aq + (int64_t)q * a.ld_aq + 4 * g;
  const float* kt_row = a.kt + (int64_t)nb * a.ld_kt + 4 * g;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int s = 0; s < n_stage; ++s) {
    const int buf = s & 1;
    if (s + 1 < n_stage) fetch(s + 1);
    // GEMM1: a^T chunk (16 hidden x 16 pairs), start = aq_i - kt_j
    f32x4 h = f32x4{0.f, 0.f, 0.f, 0.f};
    if (16 * s + 4 * g < d2) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(aq_row + 16 * s);
      const f32x4 v = *reinterpret_cast<const f32x4*>(kt_row + 16 * s);
      h = u - v;
    }
    {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&s_wp[buf][c * 32 + 8 * g]);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(&s_wp[buf][c * 32 + 8 * g + 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) h = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], r[e], h, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) h = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], r[4 + e], h, 0, 0, 0);
    }
    h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f);
    // GEMM2: logits^T += W2[:, stage] . relu(a)^T
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(&s_w2[buf][(t * 16 + c) * 16 + 4 * g]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], h[e], acc[t], 0, 0, 0);
    }
    if (s + 1 < n_stage) stash(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue, one channel tile at a time: lane (g, c) holds channels 16 t + 4 g + e of pair c
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = 16 * t + 4 * g;
    f32x4 pe = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(&s_p2[(t * 16 + c) * 32 + 8 * g]);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(&s_p2[(t * 16 + c) * 32 + 8 * g + 4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) pe = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], r[e], pe, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) pe = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], r[4 + e], pe, 0, 0, 0);
    f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ch < d) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(a.vt + (int64_t)nb * a.ld_vt + ch);
      const f32x4 cc = *reinterpret_cast<const f32x4*>(a.c2 + ch);
      val = vv + (pe + cc);
    }
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float l = acc[t][e];
      const float mx = row_max(l);
      const float p = __builtin_amdgcn_exp2f((l - mx) * a.sc);
      const float den = row_sum(p);
      out[e] = row_sum(p * val[e]) / den;
    }
    if (live && c == 0 && ch < d) *reinterpret_cast<f32x4*>(a.agg + (int64_t)q * a.ld_agg + ch) = out;
  }
}

}  // namespace

extern "C" int occ4d_pt_self_attn16_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs, const float* apos,
                                        int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                                        int64_t ld_vt, const float* P1, const float* c1, const float* wp, const float* w2,
                                        const float* p2, const float* c2, float* agg, int64_t ld_agg, int n, int m, int k,
                                        int d, float divisor, void* stream) {
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vt && P1 && c1 && wp && w2 && p2 && c2 && agg,
                "occ4d_pt_self_attn16_f32: null pointer");
  OCC4D_REQUIRE(k == 16 && d >= 4 && d <= 288 && d % 4 == 0 && n >= 0 && m >= 1 && divisor > 0.f,
                "occ4d_pt_self_attn16_f32: need k == 16 and d a multiple of 4 in [4, 288] (k = %d, d = %d)", k, d);
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ld_vt % 4 == 0 && ld_agg % 4 == 0 && ld_aq >= 2 * d && ld_kt >= 2 * d &&
                    ld_vt >= d && ld_agg >= d,
                "occ4d_pt_self_attn16_f32: row strides must be multiples of 4 floats and cover the rows");
  const void* ps[] = {aq, kt, vt, wp, w2, p2, c2, agg};
  for (const void* p : ps)
    OCC4D_REQUIRE(((uintptr_t)p % 16) == 0, "occ4d_pt_self_attn16_f32: tensors must be 16-byte aligned");
  if (n == 0) return OCC4D_OK;
  SelfAttnArgs a{aq, ld_aq, kt, ld_kt, vt, ld_vt, qpos, qs, apos, as, idx, P1, c1, wp, w2, p2, c2, agg, ld_agg, n, d,
                 1.44269504088896f / divisor};
  const int grid = occ4d::cdiv(n, 4);
  hipStream_t st = (hipStream_t)stream;
  const int nt = (d + 15) / 16;
  if (nt <= 3) self_attn16_kernel<3><<<grid, 256, 0, st>>>(a);
  else if (nt <= 5) self_attn16_kernel<5><<<grid, 256, 0, st>>>(a);
  else if (nt <= 9) self_attn16_kernel<9><<<grid, 256, 0, st>>>(a);
  else self_attn16_kernel<18><<<grid, 256, 0, st>>>(a);
  return occ4d::check_launch("occ4d_pt_self_attn16_f32");
}
