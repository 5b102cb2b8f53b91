// The decoder trunk's residual block (D6 / K11 of SURVEY.md: ResnetBlockFC, model/implicit.py:92-101) as ONE launch in the
// fp16 two-piece split scheme (csrc/bf16x6.hpp: x = x1 + x2, three partial products, fp32 accumulate):
//     y = x + W1 relu(W0 relu(x) + b0) + b1,        width 416
// VERDICT r5 item 1b.  The two-launch form (csrc/trunk_bf16x6.hip) moves x, h, h, x (residual) and y through HBM: five passes
// over (n, 416) for 2 x 13.4 us of matrix time at the fp16 peak -- each launch is bound by its memory phases (44.6 us per
// layer at 32256 rows = 3.6-4.8 TB/s).  Here h never leaves the registers: x is read (twice: operand, residual; the second
// time from L2) and y written.
//
// Decomposition (wave64, 8 waves, one workgroup per CU): a wave owns ONE 16-row tile x all 416 channels, a workgroup 128 rows.
//   layer 1, TRANSPOSED (weights = A operand): h^T tile t (16 channels x 16 rows), 26 accumulators; stage ks = 32 input
//            channels: the lane's 8 consecutive inputs (k = 32 ks + 8 g + j) of its row, ReLU, split -> B operand; 26 tiles x
//            3 products against the stage's 52 weight fragments (26 tiles x 2 pieces, 1 KB each) in LDS.
//   hand-over: lane (g, c) holds channels 16 t + 4 g + (0 .. 3) of row c in accumulator t -- + b0, ReLU, split: tiles
//            (2 T, 2 T + 1) give the 8 k slots of k-step T of layer 2's B operand, k slot 8 g + j = channel
//            32 T + 16 (j >> 2) + 4 g + (j & 3) (W1 is packed in that order): no lane exchange, no LDS (the trick of the attention
//            kernels, csrc/crossattn_bf16x6.hip).
//   layer 2, TRANSPOSED as well (float4 epilogue), in two HALVES of 13 output tiles (52 accumulators beside the 104 registers
//            of relu(h)'s pieces): stage = two k-steps x 13 tiles = 52 fragments again (the 7th stage of a half holds one
//            k-step); epilogue of a half: + b1 + x, one float4 per tile and row; half 0's stores drain under half 1.
// Stream = 13 + 14 stages of 52 KB, double buffered (104 KB), DMA one stage ahead by all waves, one barrier per stage.
#include "bf16x6.hpp"
#include "common.hpp"

namespace {

using S = SplitF16x3;
constexpr int RD = 416;                   // width
constexpr int RT = RD / 16;               // 26 channel tiles
constexpr int RKS = RD / 32;              // 13 k-steps of 32
constexpr int RFW = 256;                  // u32 words per fragment image
constexpr int RSF = 2 * RT;               // 52 fragments per stage
constexpr int RSTAGE = RSF * RFW;         // 13312 words = 53248 B
constexpr int RS1 = RKS;                  // layer 1: 13 stages [k-step][tile 26][piece 2]
constexpr int RHS = (RKS + 1) / 2;        // layer 2: 7 stages per half [k-step pair][k-step in pair 2][tile 13][piece 2]
constexpr int RNSTAGE = RS1 + 2 * RHS;    // 27
constexpr int RWAVES = 8;
constexpr int RROWS = 16 * RWAVES;        // 128 rows per workgroup
constexpr int RPARTS = (RSF + RWAVES - 1) / RWAVES;     // 7 fragments per wave and stage

struct ResblockX3Args {
  const float* x; int64_t ldx;
  float* y; int64_t ldy;
  const unsigned* wstream;                // [RNSTAGE][RSF][64 lanes][4 words]
  const float* b0; const float* b1;
  int n;
};

__device__ __forceinline__ f32x4 mm3_t(const u32x4 (&w)[2], const S::Op& b, f32x4 c) {   // weights = A: transposed tile
  c = mmh(w[1], b.p[0], c);
  c = mmh(w[0], b.p[1], c);
  c = mmh(w[0], b.p[0], c);
  return c;
}
// two tiles, alternating accumulators (a dependent 16 x 16 x 32 instruction every other issue slot)
__device__ __forceinline__ void mm3_t2(const u32x4 (&w0)[2], const u32x4 (&w1)[2], const S::Op& b, f32x4& c0, f32x4& c1) {
  c0 = mmh(w0[1], b.p[0], c0); c1 = mmh(w1[1], b.p[0], c1);
  c0 = mmh(w0[0], b.p[1], c0); c1 = mmh(w1[0], b.p[1], c1);
  c0 = mmh(w0[0], b.p[0], c0); c1 = mmh(w1[0], b.p[0], c1);
}

__global__ __launch_bounds__(512, 2) void resblock_f16x3_kernel(const ResblockX3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned buf0[RSTAGE];
  __shared__ __attribute__((aligned(16))) unsigned buf1[RSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int row = (int)blockIdx.x * RROWS + 16 * wave + c;
  const int rowc = min(row, a.n - 1);
  const unsigned lane16 = lane * 16;
  // this wave's i-th fragment of a stage (the tail repeats the last fragment: same bytes, same place)
  auto dma_stage = [&](int stage_no, const unsigned* dst) {
#ifdef OCC4D_RB_ABL_NODMA                              // (timing-only ablations: profiles/time_resblock_f16x3.py -D...)
    if (stage_no > 0) return;
#endif
#pragma unroll
    for (int i = 0; i < RPARTS; ++i) {
      const int f = min(wave + RWAVES * i, RSF - 1);
      dma_frag_x(a.wstream + (int64_t)stage_no * RSTAGE + f * RFW, lds_addr_x(dst) + (unsigned)f * (RFW * 4), lane16);
    }
  };
  dma_stage(0, buf0);
  const float* xrow = a.x + (int64_t)rowc * a.ldx + 8 * g;           // operand order: k = 32 ks + 8 g + j
  f32x4 xa0 = *reinterpret_cast<const f32x4*>(xrow), xa1 = *reinterpret_cast<const f32x4*>(xrow + 4);
  auto frag = [&](const unsigned* buf, int i) { return *reinterpret_cast<const u32x4*>(buf + lane * 4 + i * RFW); };

  // ---------------------------------------------------------------- layer 1: h^T = W0 relu(x)^T, 26 tiles
  f32x4 h[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  dma_wait_x();
  __builtin_amdgcn_s_barrier();
  auto stage1 = [&](const int ks, const unsigned* __restrict__ cur, const unsigned* nxt) {
    dma_stage(ks + 1, nxt);                                           // (stage 13 = layer 2's first: the stream goes on)
    const S::Op xs = S::split8(relu4x(xa0), relu4x(xa1));
    const int kn = ks + 1 < RKS ? ks + 1 : ks;
#ifndef OCC4D_RB_ABL_NOX
    xa0 = *reinterpret_cast<const f32x4*>(xrow + 32 * kn);
    xa1 = *reinterpret_cast<const f32x4*>(xrow + 32 * kn + 4);
#endif
    u32x4 wn[2][2] = {{frag(cur, 0), frag(cur, 1)}, {frag(cur, 2), frag(cur, 3)}};
#pragma unroll
    for (int tp = 0; tp < RT / 2; ++tp) {                             // tile pairs (2 tp, 2 tp + 1)
      const u32x4 w0[2] = {wn[0][0], wn[0][1]}, w1[2] = {wn[1][0], wn[1][1]};
#ifndef OCC4D_RB_ABL_NOLDS
      if (tp + 1 < RT / 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) wn[q][p] = frag(cur, 2 * (2 * (tp + 1) + q) + p);
      }
#endif
      mm3_t2(w0, w1, xs, h[2 * tp], h[2 * tp + 1]);
    }
    dma_wait_x();
#ifndef OCC4D_RB_ABL_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
  };
#pragma clang loop unroll(disable)
  for (int ks = 0; ks < RKS - 1; ks += 2) {
    stage1(ks, buf0, buf1);
    stage1(ks + 1, buf1, buf0);
  }
  stage1(RKS - 1, buf0, buf1);                                        // (13 stages: layer 2 starts in buf1)

  // ---------------------------------------------------------------- hand-over: relu(h + b0), split: layer 2's B operand
  S::Op hs[RKS];
#pragma unroll
  for (int T = 0; T < RKS; ++T) {
    const f32x4 ba = *reinterpret_cast<const f32x4*>(a.b0 + 32 * T + 4 * g);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b0 + 32 * T + 16 + 4 * g);
    hs[T] = S::split8(relu4x(h[2 * T] * S::INV_WSCALE + ba), relu4x(h[2 * T + 1] * S::INV_WSCALE + bb));
  }

  // ---------------------------------------------------------------- layer 2: two halves of 13 output tiles
  const float* xres = a.x + (int64_t)rowc * a.ldx + 4 * g;           // transposed tiles: channels 16 t + 4 g + (0 .. 3) of row c
  float* yrow = a.y + (int64_t)rowc * a.ldy + 4 * g;
  int sidx = RS1;                                                     // stream stage index; buffers alternate from buf1
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 acc[RT / 2];
#pragma unroll
    for (int t = 0; t < RT / 2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < RHS; ++sp, ++sidx) {
      const unsigned* cur = (sidx & 1) ? buf1 : buf0;
      const unsigned* nxt = (sidx & 1) ? buf0 : buf1;
      if (sidx + 1 < RNSTAGE) dma_stage(sidx + 1, nxt);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int T = 2 * sp + kk;
        if (T >= RKS) break;
        const unsigned* fb = cur + kk * (RT / 2) * 2 * RFW;          // [k-step in pair][tile 13][piece 2]
        u32x4 wn[2][2] = {{frag(fb, 0), frag(fb, 1)}, {frag(fb, 2), frag(fb, 3)}};
#pragma unroll
        for (int tp = 0; tp < RT / 4; ++tp) {                         // 6 tile pairs + the 13th tile
          const u32x4 w0[2] = {wn[0][0], wn[0][1]}, w1[2] = {wn[1][0], wn[1][1]};
#ifndef OCC4D_RB_ABL_NOLDS
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              if (2 * (tp + 1) + q < RT / 2) wn[q][p] = frag(fb, 2 * (2 * (tp + 1) + q) + p);
#endif
          mm3_t2(w0, w1, hs[T], acc[2 * tp], acc[2 * tp + 1]);
        }
        const u32x4 wl[2] = {wn[0][0], wn[0][1]};                     // (tile 12: loaded by the last prefetch above)
        acc[RT / 2 - 1] = mm3_t(wl, hs[T], acc[RT / 2 - 1]);
      }
      dma_wait_x();
#ifndef OCC4D_RB_ABL_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
    }
    // epilogue of the half: y = x + acc / WSCALE + b1, float4 per tile and row
    const int ch0 = 16 * (RT / 2) * half;
#pragma unroll
    for (int t = 0; t < RT / 2; ++t) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(xres + ch0 + 16 * t);
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.b1 + ch0 + 16 * t + 4 * g);
      const f32x4 v = acc[t] * S::INV_WSCALE + b + r;
#ifdef OCC4D_RB_ABL_NOSTORE
      if (v.x == 123.456f)
#endif
      if (row < a.n) *reinterpret_cast<f32x4*>(yrow + ch0 + 16 * t) = v;
    }
  }
}

// ---- packer: W0, W1 (416, 416) row-major -> the kernel's fragment stream (two fp16 pieces per weight, * WSCALE)
__global__ void pack_resblock_f16x3_kernel(const float* __restrict__ w0, int64_t ld0, const float* __restrict__ w1, int64_t ld1,
                                           unsigned* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)RNSTAGE * RSTAGE) return;
  const int word = (int)(e & 3), lane = (int)((e >> 2) & 63);
  const int frag = (int)((e / RFW) % RSF), stage = (int)(e / RSTAGE);
  const int c = lane & 15, g = lane >> 4;
  unsigned res = 0u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = 2 * word + q;                         // k slot 8 g + j of the instruction
    float v = 0.f;
    int p;
    if (stage < RS1) {                                  // layer 1: A row c = output channel 16 t + c; k = input 32 ks + 8 g + j
      const int t = frag / 2;
      p = frag % 2;
      v = w0[(int64_t)(16 * t + c) * ld0 + 32 * stage + 8 * g + j];
    } else {                                            // layer 2: [half][k-step pair][k-step in pair][tile 13][piece]
      const int s2 = stage - RS1, half = s2 / RHS, sp = s2 % RHS;
      const int kk = frag / (2 * (RT / 2)), t = (frag / 2) % (RT / 2), T = 2 * sp + kk;
      p = frag % 2;
      if (T < RKS) v = w1[(int64_t)(16 * ((RT / 2) * half + t) + c) * ld1 + 32 * T + 16 * (j >> 2) + 4 * g + (j & 3)];
    }
    res |= S::piece(v, p) << (16 * q);
  }
  out[e] = res;
}

}  // namespace

extern "C" int64_t occ4d_resblock_f16x3_packed_floats(void) { return (int64_t)RNSTAGE * RSTAGE; }

extern "C" int occ4d_pack_resblock_f16x3_f32(const float* w0, int64_t ld0, const float* w1, int64_t ld1, float* packed,
                                             void* stream) {
  OCC4D_REQUIRE(w0 && w1 && packed && ld0 >= RD && ld1 >= RD && ((uintptr_t)packed % 16) == 0,
                "occ4d_pack_resblock_f16x3_f32: two (416, 416) weights and a 16-byte aligned stream expected");
  const int64_t total = (int64_t)RNSTAGE * RSTAGE;
  pack_resblock_f16x3_kernel<<<occ4d::cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w0, ld0, w1, ld1,
                                                                                       reinterpret_cast<unsigned*>(packed));
  return occ4d::check_launch("occ4d_pack_resblock_f16x3_f32");
}

extern "C" int occ4d_resblock_f16x3_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                        const float* b0, const float* b1, int n, void* stream) {
  const char* who = "occ4d_resblock_f16x3_f32";
  if (n == 0) return OCC4D_OK;
  OCC4D_REQUIRE(x && y && w_packed && b0 && b1 && n > 0, "%s: null pointer", who);
  OCC4D_REQUIRE(ldx >= RD && ldy >= RD && ldx % 4 == 0 && ldy % 4 == 0 &&
                    (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed | (uintptr_t)b0 | (uintptr_t)b1) % 16) == 0,
                "%s: rows, biases and the stream 16-byte aligned, ldx / ldy multiples of 4 and >= 416", who);
  ResblockX3Args a{x, ldx, y, ldy, reinterpret_cast<const unsigned*>(w_packed), b0, b1, n};
  resblock_f16x3_kernel<<<occ4d::cdiv(n, RROWS), 512, 0, (hipStream_t)stream>>>(a);
  return occ4d::check_launch(who);
}
