// Shared device helpers of the split-precision (bf16 x 3 pieces, 6 partial products, fp32 accumulate) kernels:
// csrc/crossattn_bf16x6.hip (vector attention) and csrc/trunk_bf16x6.hip (the decoder's 416-input Linear layers).
//
//     x = x1 + x2 + x3 exactly   (x1 = x truncated to bf16, x2 = (x - x1) truncated, x3 = x - x1 - x2: 3 x 8 = 24 bits)
//     a b ~ a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3, a3 b2, a3 b3 <= 2^-23 |a b|)
#pragma once
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned lds_addr_x(const unsigned* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned*)p;
}
// one 1 KB fragment, global (L2) -> LDS by DMA (see csrc/crossattn16p.hip: scalar base + lane offset, inline asm so that
// the compiler's LDS wait bookkeeping does not see it; ordering = dma_wait_x() + the stage barrier)
__device__ __forceinline__ void dma_frag_x(const unsigned* __restrict__ src_frag, unsigned lds_dst, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(src_frag) : "memory");
}
__device__ __forceinline__ void dma_wait_x() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ f32x4 mm(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Split { u32x4 h, m, l; };

// gfx950 lane-swap exchanges (16-lane rows r0..r3 of a wave), as in csrc/crossattn16p.hip:
//   swap16(x, y) -> lo = (x.r0, y.r0, x.r2, y.r2), hi = (x.r1, y.r1, x.r3, y.r3)
//   swap32(x, y) -> lo = (x.r0, x.r1, y.r0, y.r1), hi = (x.r2, x.r3, y.r2, y.r3)
struct PairX { float lo, hi; };
__device__ __forceinline__ PairX swap16x(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return PairX{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ PairX swap32x(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return PairX{__uint_as_float(r[0]), __uint_as_float(r[1])};
}

// two fp32 -> the packed bf16 pairs of their three truncation pieces (even element in the low half); the two
// subtractions are packed (v_pk_add_f32): on gfx950 every VALU instruction of a wave costs matrix-pipe time
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const f32x2 x = {x0, x1};
  const u32x2 u = __builtin_bit_cast(u32x2, x);
  h = __builtin_amdgcn_perm(u.y, u.x, 0x07060302u);
  const f32x2 r = x - __builtin_bit_cast(f32x2, u & 0xffff0000u);
  const u32x2 v = __builtin_bit_cast(u32x2, r);
  m = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
  const f32x2 t = r - __builtin_bit_cast(f32x2, v & 0xffff0000u);
  const u32x2 w = __builtin_bit_cast(u32x2, t);
  l = __builtin_amdgcn_perm(w.y, w.x, 0x07060302u);
}
// eight fp32 (elements j = 0..7 of an operand row) -> the three operand registers sets
__device__ __forceinline__ Split split8(const f32x4 a, const f32x4 b) {
  unsigned h[4], m[4], l[4];
  split2(a.x, a.y, h[0], m[0], l[0]);
  split2(a.z, a.w, h[1], m[1], l[1]);
  split2(b.x, b.y, h[2], m[2], l[2]);
  split2(b.z, b.w, h[3], m[3], l[3]);
  return Split{u32x4{h[0], h[1], h[2], h[3]}, u32x4{m[0], m[1], m[2], m[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}
__device__ __forceinline__ f32x4 relu4x(f32x4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}

// the six partial products of (A1 + A2 + A3)(B1 + B2 + B3), small terms first, on two accumulators alternately
// (row tiles 0 / 1 share the B pieces): consecutive MFMAs never depend on each other
__device__ __forceinline__ void mm6x2(const Split& a0, const Split& a1, const u32x4 bh, const u32x4 bm, const u32x4 bl,
                                      f32x4& c0, f32x4& c1) {
  c0 = mm(a0.l, bh, c0); c1 = mm(a1.l, bh, c1);
  c0 = mm(a0.h, bl, c0); c1 = mm(a1.h, bl, c1);
  c0 = mm(a0.m, bm, c0); c1 = mm(a1.m, bm, c1);
  c0 = mm(a0.m, bh, c0); c1 = mm(a1.m, bh, c1);
  c0 = mm(a0.h, bm, c0); c1 = mm(a1.h, bm, c1);
  c0 = mm(a0.h, bh, c0); c1 = mm(a1.h, bh, c1);
}
// the same with the SHARED operand on the A side (GEMM1: the Wp fragment), the per-row-tile operand on the B side
__device__ __forceinline__ void mm6x2_b(const u32x4 ah, const u32x4 am, const u32x4 al, const Split& b0, const Split& b1,
                                        f32x4& c0, f32x4& c1) {
  c0 = mm(al, b0.h, c0); c1 = mm(al, b1.h, c1);
  c0 = mm(ah, b0.l, c0); c1 = mm(ah, b1.l, c1);
  c0 = mm(am, b0.m, c0); c1 = mm(am, b1.m, c1);
  c0 = mm(am, b0.h, c0); c1 = mm(am, b1.h, c1);
  c0 = mm(ah, b0.m, c0); c1 = mm(ah, b1.m, c1);
  c0 = mm(ah, b0.h, c0); c1 = mm(ah, b1.h, c1);
}

__device__ __forceinline__ unsigned piece16(float x, int p) {
  unsigned u = __float_as_uint(x);
  if (p == 0) return u >> 16;
  const float r1 = x - __uint_as_float(u & 0xffff0000u);
  u = __float_as_uint(r1);
  if (p == 1) return u >> 16;
  const float r2 = r1 - __uint_as_float(u & 0xffff0000u);
  return __float_as_uint(r2) >> 16;
}
// ----------------------------------------------------------------------------------------------------------------
// The two split schemes as types (csrc/crossattn_bf16x6.hip and csrc/trunk_bf16x6.hip are templates over them).
//   NP            pieces per operand = weight fragments per (tile, stage)
//   Op            the NP operand register sets of one 16 x 32 activation tile (p[0] = leading piece)
//   split8        eight fp32 of a lane -> Op
//   mm_x2 / _b    all kept partial products of two row tiles against one shared weight tile, small terms first, the
//                 two accumulators alternately (shared operand on the B / on the A side)
//   mm_1          the same for one row tile
//   piece         packer side: piece p of x * WSCALE as 16 bits
//   WSCALE        power of two the packers multiply every weight with (exact); the kernels multiply their accumulators
//                 with INV_WSCALE where they leave the matrix pipe
// ----------------------------------------------------------------------------------------------------------------
struct SplitBf16x6 {
  static constexpr int NP = 3;
  static constexpr float WSCALE = 1.f, INV_WSCALE = 1.f, HSCALE = 1.f;
  struct Op { u32x4 p[3]; };
  static __device__ __forceinline__ Op split8(const f32x4 a, const f32x4 b) {
    const Split s = ::split8(a, b);
    return Op{{s.h, s.m, s.l}};
  }
  static __device__ __forceinline__ void mm_x2(const Op& a0, const Op& a1, const u32x4 (&b)[3], f32x4& c0, f32x4& c1) {
    mm6x2(Split{a0.p[0], a0.p[1], a0.p[2]}, Split{a1.p[0], a1.p[1], a1.p[2]}, b[0], b[1], b[2], c0, c1);
  }
  static __device__ __forceinline__ void mm_x2_b(const u32x4 (&a)[3], const Op& b0, const Op& b1, f32x4& c0, f32x4& c1) {
    mm6x2_b(a[0], a[1], a[2], Split{b0.p[0], b0.p[1], b0.p[2]}, Split{b1.p[0], b1.p[1], b1.p[2]}, c0, c1);
  }
  static __device__ __forceinline__ f32x4 mm_1(const Op& a, const u32x4 (&b)[3], f32x4 e) {
    e = mm(a.p[2], b[0], e);
    e = mm(a.p[0], b[2], e);
    e = mm(a.p[1], b[1], e);
    e = mm(a.p[1], b[0], e);
    e = mm(a.p[0], b[1], e);
    e = mm(a.p[0], b[0], e);
    return e;
  }
  static __device__ __forceinline__ unsigned piece(float x, int p) { return piece16(x, p); }
  static __device__ __forceinline__ unsigned piece_scaled(float x, int p, float) { return piece16(x, p); }
};

// fp16 x 2 pieces, 3 partial products (round 6):
//     x = x1 + x2 + e,  x1 = rn_f16(x),  x2 = rn_f16(x - x1),  |e| <= 2^-23 |x|   (x - x1 is exact in fp32; 11 + 11 bits + the
//         two round-to-nearest half bits; fp32 itself keeps 24)
//     a b ~ a1 b1 + a1 b2 + a2 b1                                      (dropped: a2 b2 <= 2^-22 |a b|, typically 2^-25)
// HALF the matrix instructions of the bf16 scheme (v_mfma_f32_16x16x32_f16 runs at the bf16 rate) and a third of its
// split arithmetic (4 VALU per element pair: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32).  The price is
// fp16's RANGE, which the bf16 pieces do not have to think about:
//   * weights are packed as the pieces of w * 2^8 (exact), so that the second piece of a weight of ordinary size
//     (>= 2^-10) is a normal fp16 number; smaller ones are kept to an absolute 2^-33; |w| must stay below 255.
//     (The attention kernel's first GEMM packs its merged 832 x 32 matrix * 2^4 instead and keeps the hidden activations
//     at 2^4 x their value up to the second GEMM -- no scaling instruction in the loop: |w| < 4094 there, entries below
//     2^-6 kept to an absolute 2^-29, hidden activations below 4094.)
//   * activations are split as they are: |x| must stay below 65504, and below |x| = 0.25 the second piece is an
//     fp16 subnormal (gfx950's matrix pipe does not flush them): absolute error <= 2^-25 instead of 2^-23 |x|.
// Forward passes only -- gradient magnitudes do not live in that window.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mmh(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& h, unsigned& l) {
  const f32x2 x = {x0, x1};
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));     // v_cvt_pk_f16_f32 (rn)
  float r0, r1;      // x - f32(x1) in ONE instruction per element (the compiler's own form is 2 cvt + 1 packed add)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(x1));
  const f32x2 r = {r0, r1};
  h = hu;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
struct SplitF16x3 {
  static constexpr int NP = 2;
  static constexpr float WSCALE = 256.f, INV_WSCALE = 1.f / 256.f;
  // the attention kernel keeps its hidden activations at HSCALE x their value between GEMM1 and GEMM2 (GEMM1's weights are
  // packed * HSCALE, its init term arrives * HSCALE, nothing is descaled in the loop): |hidden| < 65504 / 16 = 4094
  static constexpr float HSCALE = 16.f;
  struct Op { u32x4 p[2]; };
  static __device__ __forceinline__ Op split8(const f32x4 a, const f32x4 b) {
    unsigned h[4], l[4];
    split2h(a.x, a.y, h[0], l[0]);
    split2h(a.z, a.w, h[1], l[1]);
    split2h(b.x, b.y, h[2], l[2]);
    split2h(b.z, b.w, h[3], l[3]);
    return Op{{u32x4{h[0], h[1], h[2], h[3]}, u32x4{l[0], l[1], l[2], l[3]}}};
  }
  static __device__ __forceinline__ void mm_x2(const Op& a0, const Op& a1, const u32x4 (&b)[2], f32x4& c0, f32x4& c1) {
    c0 = mmh(a0.p[1], b[0], c0); c1 = mmh(a1.p[1], b[0], c1);
    c0 = mmh(a0.p[0], b[1], c0); c1 = mmh(a1.p[0], b[1], c1);
    c0 = mmh(a0.p[0], b[0], c0); c1 = mmh(a1.p[0], b[0], c1);
  }
  static __device__ __forceinline__ void mm_x2_b(const u32x4 (&a)[2], const Op& b0, const Op& b1, f32x4& c0, f32x4& c1) {
    c0 = mmh(a[1], b0.p[0], c0); c1 = mmh(a[1], b1.p[0], c1);
    c0 = mmh(a[0], b0.p[1], c0); c1 = mmh(a[0], b1.p[1], c1);
    c0 = mmh(a[0], b0.p[0], c0); c1 = mmh(a[0], b1.p[0], c1);
  }
  static __device__ __forceinline__ f32x4 mm_1(const Op& a, const u32x4 (&b)[2], f32x4 e) {
    e = mmh(a.p[1], b[0], e);
    e = mmh(a.p[0], b[1], e);
    e = mmh(a.p[0], b[0], e);
    return e;
  }
  static __device__ __forceinline__ unsigned piece(float x, int p) { return piece_scaled(x, p, WSCALE); }
  static __device__ __forceinline__ unsigned piece_scaled(float x, int p, float scale) {
    const float xs = x * scale;
    const _Float16 h = (_Float16)xs;
    if (p == 0) return (unsigned)__builtin_bit_cast(unsigned short, h);
    const _Float16 l = (_Float16)(xs - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, l);
  }
};
}  // namespace
