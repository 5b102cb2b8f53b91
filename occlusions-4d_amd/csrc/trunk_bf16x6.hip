// The decoder trunk's 416-input Linear layers (D6 / K11 of SURVEY.md: ResnetBlockFC.fc_0 / fc_1, model/implicit.py:92-101;
// PointTransformerBlock.layer3 and the merged query projection, model/modules.py:61-65) on SPLIT-PRECISION matrix
// instructions: y[:, 0 .. N) = [res +] W [relu](x) + b with K = 416 on v_mfma_f32_16x16x32_bf16, both operands split
// three ways into bf16 pieces (exact), six of the nine partial products accumulated in fp32 (csrc/bf16x6.hpp) --
// fp32-class results at 2.67 x less matrix time than v_mfma_f32_16x16x4_f32.  A residual block is two launches (the
// hidden activation makes one trip through HBM; the fp32 kernel of csrc/trunk.hip keeps it in registers).
//
// Decomposition (wave64, 8 waves, one workgroup per CU): a workgroup owns 256 rows x 208 output channels (13 tiles of
// 16); wave w = two row tiles of 16 rows x the 13 channel tiles (104 accumulators); stage ks = 32 input channels: the
// wave loads its rows' 8 consecutive inputs per lane (k = 32 ks + 8 g + j: the A operand's own order), applies the ReLU,
// splits them into the three pieces (VALU: the only per-element work) and runs 13 x 12 MFMAs against the stage's 39
// weight fragments (13 tiles x 3 pieces, 1 KB each, packed once per weight update) in LDS: a ring of three stage buffers
// filled by DMA one stage ahead, one barrier per stage, the two waves of a SIMD half a stage out of phase -- the
// protocol of csrc/crossattn_bf16x6.hip.  Channel blocks are separate workgroups (an XCD serves one block of a 416-wide
// layer, two of an 832-wide one: its L2 holds that share of the stream).
#include "bf16x6.hpp"

namespace {

constexpr int YK = 416;                   // input channels
constexpr int YKS = YK / 32;              // 13 stages
constexpr int YCB = 208;                  // output channels per workgroup
constexpr int YT = YCB / 16;              // 13 channel tiles
constexpr int YFW = 256;                  // u32 words per fragment image
constexpr int YWAVES = 8;
constexpr int YROWS = 32 * YWAVES;        // 256 rows per workgroup
// stage geometry of a split scheme S (csrc/bf16x6.hpp)
template <typename S> struct YG {
  static constexpr int SF = S::NP * YT;                       // fragments per stage: 39 (bf16 x 3) / 26 (fp16 x 2)
  static constexpr int STAGE = SF * YFW;                      // words: 39936 B / 26624 B
  static constexpr int PARTS = (SF + YWAVES - 1) / YWAVES;    // fragments a wave fetches per stage: 5 / 4
  static_assert(PARTS <= 5, "the DMA slots sit behind channel tiles 0-4 and 7-11");
};

struct RowlinX6Args {
  const float* x; int64_t ldx;
  float* y; int64_t ldy;
  const unsigned* wstream;                // [N / 208][13 stages][SF][64 lanes][4 words]
  const float* bias;
  const float* res; int64_t ldr;
  int n, nblk;                            // rows; channel blocks (N / 208)
  int rowgroups, per;                     // groups of 256 rows; groups per XCD share
  const float* mask; int64_t ldm;         // training data gradients: output zeroed where mask <= 0 (or null)
  int res_post;                           // with a mask: res is added AFTER the mask (a skip connection's gradient)
};

template <typename S, bool RELU_IN>
__global__ __launch_bounds__(512, 2) void rowlin_split_kernel(const RowlinX6Args a) {
  using Op = typename S::Op;
  constexpr int NP = S::NP, YSF = YG<S>::SF, YSTAGE = YG<S>::STAGE, PARTS = YG<S>::PARTS;
  __shared__ __attribute__((aligned(16))) unsigned buf0[YSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[YSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf2[YSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // workgroup b runs on XCD b % 8: the XCDs are dealt to the channel blocks evenly (8 / nblk each), and every XCD takes
  // one contiguous share of the row groups
  const int xcd = blockIdx.x & 7, share = 8 / a.nblk;
  const int cb = xcd / share, slab = xcd % share, in_slab = blockIdx.x >> 3;
  const int rg = slab * a.per + in_slab;
  if (in_slab >= a.per || rg >= a.rowgroups) return;
  const int row0 = rg * YROWS + wave * 32;
  const unsigned* const wst = a.wstream + (int64_t)cb * YKS * YSTAGE;
  const unsigned lane16 = lane * 16;
  const bool grp_b = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) != 0;   // (see crossattn_bf16x6.hip)

  // this wave's i-th fragment of a stage (PARTS per wave and stage; the tail repeats the last fragment)
  auto dma_part = [&](int stage_no, const unsigned* dst, int i) {
    const int f = min(wave + YWAVES * i, YSF - 1);
    dma_frag_x(wst + (int64_t)stage_no * YSTAGE + f * YFW, lds_addr_x(dst) + (unsigned)f * (YFW * 4), lane16);
  };
#pragma unroll
  for (int i = 0; i < PARTS; ++i) dma_part(0, buf0, i);

  // this lane's operand rows: row tile rt, row c; k = 32 ks + 8 g + j
  const float* xrow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) xrow[rt] = a.x + (int64_t)min(row0 + 16 * rt + c, a.n - 1) * a.ldx + 8 * g;
  f32x4 xa[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    xa[rt][0] = *reinterpret_cast<const f32x4*>(xrow[rt]);
    xa[rt][1] = *reinterpret_cast<const f32x4*>(xrow[rt] + 4);
  }
  f32x4 acc[2][YT];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < YT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  dma_wait_x();
  __builtin_amdgcn_s_barrier();                       // barrier 0: stage 0 is complete
  if (grp_b) {
#pragma unroll
    for (int i = 0; i < PARTS; ++i) dma_part(1, buf1, i);
  }

  auto stage = [&](const int s, const unsigned* __restrict__ cur, const unsigned* dA, const unsigned* dB) {
    const unsigned* f = cur + lane * 4;
    Op xs[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      xs[rt] = RELU_IN ? S::split8(relu4x(xa[rt][0]), relu4x(xa[rt][1])) : S::split8(xa[rt][0], xa[rt][1]);
    const int sn = s + 1 < YKS ? s + 1 : s;           // next stage's inputs (consumed at its top)
#ifndef OCC4D_X6T_ABL_NOX                              // (timing-only ablations: profiles/time_rowlin_x6.py)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      xa[rt][0] = *reinterpret_cast<const f32x4*>(xrow[rt] + 32 * sn);
      xa[rt][1] = *reinterpret_cast<const f32x4*>(xrow[rt] + 32 * sn + 4);
    }
#endif
    u32x4 bn[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) bn[p] = *reinterpret_cast<const u32x4*>(f + p * YFW);
#pragma unroll
    for (int t = 0; t < YT; ++t) {
      u32x4 bc[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bc[p] = bn[p];
      if (t + 1 < YT) {
#pragma unroll
        for (int p = 0; p < NP; ++p) bn[p] = *reinterpret_cast<const u32x4*>(f + (NP * (t + 1) + p) * YFW);
      }
      if (t < PARTS) {
        if (!grp_b && s + 1 < YKS) dma_part(s + 1, dA, t);
      } else if (t > 6 && t - 7 < PARTS) {
        if (grp_b && s + 2 < YKS) dma_part(s + 2, dB, t - 7);
      }
      S::mm_x2_b(bc, xs[0], xs[1], acc[0][t], acc[1][t]);           // (weights on the A side: TRANSPOSED tiles)
      if (t == 6 && grp_b) {
        dma_wait_x();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!grp_b) {
      dma_wait_x();
      __builtin_amdgcn_s_barrier();
    }
  };
  // 13 stages: buffers s % 3
#pragma clang loop unroll(disable)
  for (int s = 0; s < YKS - 1; s += 3) {
    stage(s, buf0, buf1, buf2);
    stage(s + 1, buf1, buf2, buf0);
    stage(s + 2, buf2, buf0, buf1);
  }
  stage(YKS - 1, buf0, buf1, buf2);

  // ---- epilogue.  The weight fragments were the A operand, so a tile is TRANSPOSED: lane (g, c) holds channels 16 t + 4 g
  // + (0 .. 3) of row c -- one float4 per tile and row (bias, residual, store).  The residual rows of a row tile are
  // fetched first, together (y may alias res: loads interleaved with the stores would be serialised by the compiler)
  const int ch0 = YCB * cb + 4 * g;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = row0 + 16 * rt + c;
    const int rowc = min(row, a.n - 1);
    f32x4 r[YT];
#ifdef OCC4D_X6T_ABL_NORES
    if (a.res && a.n < 0) {
#else
    if (a.res) {
#endif
#pragma unroll
      for (int t = 0; t < YT; ++t) r[t] = *reinterpret_cast<const f32x4*>(a.res + (int64_t)rowc * a.ldr + ch0 + 16 * t);
    }
    if (a.mask) {                         // (kernel-uniform; the masked epilogue is its own pass over the tiles)
      f32x4 mk[YT];
#pragma unroll
      for (int t = 0; t < YT; ++t) mk[t] = *reinterpret_cast<const f32x4*>(a.mask + (int64_t)rowc * a.ldm + ch0 + 16 * t);
#pragma unroll
      for (int t = 0; t < YT; ++t) {
        f32x4 v = S::WSCALE == 1.f ? acc[rt][t] : acc[rt][t] * S::INV_WSCALE;
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + ch0 + 16 * t);
        if (a.res && !a.res_post) v += r[t];
        v.x = mk[t].x > 0.f ? v.x : 0.f; v.y = mk[t].y > 0.f ? v.y : 0.f;
        v.z = mk[t].z > 0.f ? v.z : 0.f; v.w = mk[t].w > 0.f ? v.w : 0.f;
        if (a.res && a.res_post) v += r[t];
        if (row < a.n) *reinterpret_cast<f32x4*>(a.y + (int64_t)row * a.ldy + ch0 + 16 * t) = v;
      }
      continue;
    }
#pragma unroll
    for (int t = 0; t < YT; ++t) {
      f32x4 v = S::WSCALE == 1.f ? acc[rt][t] : acc[rt][t] * S::INV_WSCALE;
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + ch0 + 16 * t);
#ifndef OCC4D_X6T_ABL_NORES
      if (a.res) v += r[t];
#endif
#ifdef OCC4D_X6T_ABL_NOSTORE
      if (v.x == 123.456f)                           // (ablation: accumulators stay live, nothing is stored)
#endif
      if (row < a.n) *reinterpret_cast<f32x4*>(a.y + (int64_t)row * a.ldy + ch0 + 16 * t) = v;
    }
  }
}

template <typename S>
__global__ void pack_rowlin_split_kernel(const float* __restrict__ w, int64_t ldw, int n_out, unsigned* __restrict__ out) {
  constexpr int NP = S::NP, YSF = YG<S>::SF, YSTAGE = YG<S>::STAGE;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)(n_out / YCB) * YKS * YSTAGE;
  if (e >= total) return;
  const int word = (int)(e & 3), lane = (int)((e >> 2) & 63);
  const int frag = (int)((e / YFW) % YSF), stage = (int)((e / YSTAGE) % YKS), cb = (int)(e / ((int64_t)YKS * YSTAGE));
  const int c = lane & 15, g = lane >> 4, t = frag / NP, p = frag % NP;
  const float* row = w + (int64_t)(YCB * cb + 16 * t + c) * ldw + 32 * stage + 8 * g + 2 * word;
  out[e] = S::piece(row[0], p) | (S::piece(row[1], p) << 16);
}

}  // namespace

namespace {
template <typename S> int64_t packed_floats(int n_out) { return (int64_t)(n_out / YCB) * YKS * YG<S>::STAGE; }

template <typename S>
int pack_rowlin(const char* who, const float* w, int64_t ldw, int n_out, float* packed, void* stream) {
  OCC4D_REQUIRE(w && packed && n_out >= YCB && n_out % YCB == 0 && 8 % (n_out / YCB) == 0 && ldw >= YK,
                "%s: (%d, %d) weight with n_out in {208, 416, 832, 1664} expected", who, n_out, YK);
  const int64_t total = packed_floats<S>(n_out);
  pack_rowlin_split_kernel<S><<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, ldw, n_out,
                                                                                        reinterpret_cast<unsigned*>(packed));
  return occ4d::check_launch(who);
}

template <typename S>
int rowlin_split_launch(const char* who, const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                        const float* b, int n_out, int relu_in, const float* res, int64_t ldr, int res_post,
                        const float* mask, int64_t ldm, int n, void* stream) {
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(x && y && w_packed && n > 0, "%s: null pointer", who);
  OCC4D_REQUIRE(n_out >= YCB && n_out % YCB == 0 && 8 % (n_out / YCB) == 0 && ldy >= n_out && ldx >= YK && ldx % 4 == 0 &&
                    ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ldy % 4 == 0 &&
                    (!b || ((uintptr_t)b % 16) == 0) && (!res || (ldr >= n_out && ldr % 4 == 0 && ((uintptr_t)res % 16) == 0)),
                "%s: n_out = %d (208, 416, 832 or 1664), rows 16-byte aligned with ldx %% 4 == 0", who, n_out);
  OCC4D_REQUIRE(!mask || (ldm >= n_out && ldm % 4 == 0 && ((uintptr_t)mask % 16) == 0),
                "%s: mask rows must be 16-byte aligned with ldm %% 4 == 0 and ldm >= n_out", who);
  RowlinX6Args a{x, ldx, y, ldy, reinterpret_cast<const unsigned*>(w_packed), b, res, ldr, n, n_out / YCB, 0, 0,
                 mask, ldm, res_post};
  a.rowgroups = (int)occ4d::cdiv(n, YROWS);
  a.per = (int)occ4d::cdiv(a.rowgroups, 8 / a.nblk);
  hipStream_t st = (hipStream_t)stream;
  if (relu_in) rowlin_split_kernel<S, true><<<8 * a.per, 512, 0, st>>>(a);
  else rowlin_split_kernel<S, false><<<8 * a.per, 512, 0, st>>>(a);
  return occ4d::check_launch(who);
}
}  // namespace

extern "C" int64_t occ4d_rowlin_bf16x6_packed_floats(int n_out) { return packed_floats<SplitBf16x6>(n_out); }
extern "C" int64_t occ4d_rowlin_f16x3_packed_floats(int n_out) { return packed_floats<SplitF16x3>(n_out); }

extern "C" int occ4d_pack_rowlin_bf16x6_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream) {
  return pack_rowlin<SplitBf16x6>("occ4d_pack_rowlin_bf16x6_f32", w, ldw, n_out, packed, stream);
}
extern "C" int occ4d_pack_rowlin_f16x3_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream) {
  return pack_rowlin<SplitF16x3>("occ4d_pack_rowlin_f16x3_f32", w, ldw, n_out, packed, stream);
}

extern "C" int occ4d_rowlin_bf16x6_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                       const float* b, int n_out, int relu_in, const float* res, int64_t ldr, int n,
                                       void* stream) {
  return rowlin_split_launch<SplitBf16x6>("occ4d_rowlin_bf16x6_f32", x, ldx, y, ldy, w_packed, b, n_out, relu_in, res, ldr, 0,
                                          nullptr, 0, n, stream);
}
// The fp16 two-piece scheme (csrc/bf16x6.hpp: forward passes only; |w| < 255, |x| < 65504)
extern "C" int occ4d_rowlin_f16x3_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                      const float* b, int n_out, int relu_in, const float* res, int64_t ldr, int n,
                                      void* stream) {
  return rowlin_split_launch<SplitF16x3>("occ4d_rowlin_f16x3_f32", x, ldx, y, ldy, w_packed, b, n_out, relu_in, res, ldr, 0,
                                         nullptr, 0, n, stream);
}

// The same with the epilogue of a training data gradient: y = [mask > 0] ([relu](x) W^T + b [+ res]) [+ res], `res`
// before (res_after_mask = 0) or after the mask (the contracts of occ4d_rowlin4_masked_f32 / _masked_skip_f32).
extern "C" int occ4d_rowlin_bf16x6_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                              const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                              int res_after_mask, const float* mask, int64_t ldm, int n, void* stream) {
  OCC4D_REQUIRE(mask, "occ4d_rowlin_bf16x6_masked_f32: null mask");
  return rowlin_split_launch<SplitBf16x6>("occ4d_rowlin_bf16x6_masked_f32", x, ldx, y, ldy, w_packed, b, n_out, relu_in, res,
                                          ldr, res_after_mask != 0, mask, ldm, n, stream);
}
