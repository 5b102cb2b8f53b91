// Fused vector attention over K <= 14 neighbours (E3 of SURVEY.md §8(a); K2-K4 of §2.1):
// for a tile of queries, everything between the per-point projections and the aggregated
// output stays on chip -- the (N*K, 2D) hidden activations, the (N*K, D) logits and the
// (N*K, D) positional encodings of the reference (1.5 GB + 0.76 GB + 0.76 GB per 32768
// queries at D = 416) are never written to HBM.
//
// Work decomposition (wave64, TWO waves per SIMD, 8 waves per workgroup, 9 queries per workgroup):
//   row tile = 32 "pair rows" of the 32x32 MFMA = 14 neighbours of query A + 14 of query B +
//           4 of the workgroup's 9th query, whose 14 (<= 16) neighbours are spread over the 4 row
//           tiles: 128 MFMA rows carry 126 live pairs (padding every query to 16 rows wasted
//           12.5 % of the MFMA work).  Row labels are free, so they are assigned such that BOTH
//           half-waves see the same register pattern in the C/D layout: registers 0-6 = query
//           A, 7-13 = query B, 14-15 = the 9th query (each half-wave holds 7 + 7 + 2 rows).
//   wave  = one row tile x one HALF of the D = 32*NT output channels (7 + 6 tiles of 32 at
//           D = 416) -> <= 112 fp32 accumulators per lane, so two waves share a SIMD and one
//           wave's LDS / barrier / global-load waits are covered by the other's MFMAs
//           (measured with one 13-tile wave per SIMD: MFMA pipe 73 % busy, 17 % of wave
//           cycles parked in s_waitcnt / s_barrier).  Price: both channel halves run GEMM1
//           for the same rows (+7 % MFMA work).
//   block = the weight stream (W2: D x 2D, Wp: 2D x 32) is shared by the 8 waves through LDS,
//           one 32-wide hidden block at a time, double buffered, one barrier per block.
// Chained MFMAs, no data movement between the two GEMMs of attn_mlp:
//   GEMM1 (transposed form)  Hpre^T[hid][pair] = Wp[hid][:] . r[pair][:]   (K = 32)
//          accumulator initialised with Aq[query][hid] - Kt[neighbour][hid];
//          its C/D registers (lane: column = pair, 16 hidden rows) ARE the A operand of
//   GEMM2  logits[pair][ch] += relu(Hpre)[pair][hid] * W2[ch][hid]         (K = 2D)
//          because the k order of a dot product is free: MFMA step s of a hidden block
//          consumes hid = (s&3) + 8*(s>>2) + 4*(lane>>5), and the W2 fragment is read
//          from LDS with the same map.
//   GEMM3  pe[pair][ch] = r[pair][:] . P2[ch][:]  reuses the r registers as A operand.
// Per-channel softmax over a query's neighbours: 7 live in a lane's registers, the other 7 in
// lane ^ 32 -> one cross-half exchange per reduction.  The 9th query's four per-row-tile partial
// softmaxes (max, sum, weighted sum) are merged through LDS after the main loop.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HB = 32;        // hidden block (k-tile of GEMM2)
constexpr int LDW = 36;       // padded LDS row (floats): stride 9 x 16 B -> conflict-free ds_read_b128
constexpr int QPB = 9;        // queries per block
constexpr int KMAX = 14;      // neighbours per query supported by the row packing

struct CrossAttnArgs {
  const float* aq; int64_t ld_aq;
  const float* qpos; int64_t qs;
  const float* apos; int64_t as;
  const int32_t* idx;
  const float* kt; int64_t ld_kt;
  const float* vt; int64_t ld_vt;
  const float* P1; const float* c1;
  const float* wp;
  const float* w2; const float* b2;
  const float* p2; const float* c2;
  float* agg; int64_t ld_agg;
  int N, M, K;
  float divisor;
};

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

// (local query, neighbour slot) carried by C/D register `reg` of half-wave `half` in row tile `w`
__device__ __forceinline__ void pair_of(int w, int half, int reg, int& ql, int& slot) {
  if (reg < 7) { ql = 2 * w; slot = 7 * half + reg; }
  else if (reg < 14) { ql = 2 * w + 1; slot = 7 * half + reg - 7; }
  else { ql = 8; slot = 4 * w + 2 * half + reg - 14; }
}

// Body for one channel group: tiles [CBEG, CBEG + NTW) of the NT channel tiles (compile-time, so
// the MFMA / ds_read stream of a hidden block is one straight-line basic block).
template <int NT, int CBEG, int NTW>
__device__ __forceinline__ void cross_attn_body(const CrossAttnArgs& a, float* smem, int* s_idx) {
  constexpr int D = 32 * NT;
  constexpr int H2 = 2 * D;
  constexpr int NHB = H2 / HB;
  constexpr int W2_F4 = D * 8;                     // float4 per hidden block of W2
  constexpr int W2_LOADS = (W2_F4 + HB * 8) / 512; // float4 per thread per hidden block (W2 rows + Wp rows)
  constexpr int BUF = (D + HB) * LDW;              // floats per LDS buffer: W2 block + Wp block
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3;                 // row tile (2 queries)
  constexpr int cbeg = CBEG;
  const int half = lane >> 5, prow = lane & 31;
  // XCD-aware group assignment: workgroup b is dispatched to XCD b % 8 (each XCD has its own 4 MB L2); give every
  // XCD one CONTIGUOUS range of query groups, so that its L2 only has to hold the Kt / Vt rows of the abstract
  // points near that slab of the query grid (CARLA: Kt + Vt = 10.6 MB do not fit one L2).  Bijective for any grid.
#ifndef OCC4D_CA_NO_XCD_MAP
  const int nwg = gridDim.x, xcd = blockIdx.x & 7;
  const int per = nwg >> 3, rem = nwg & 7;
  const int group = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
#else
  const int group = blockIdx.x;
#endif
  const int q0 = group * QPB;

  // ---- neighbour indices of the block's 9 queries -> LDS (invalid slots/queries repeat a valid one)
  if (tid < QPB * 16) {
    const int q = min(q0 + (tid >> 4), a.N - 1);
    const int s = min(tid & 15, a.K - 1);
    s_idx[tid] = a.idx[(int64_t)q * a.K + s];
  }
  // ---- this lane's pair (A-operand row prow <-> C/D (half, reg) of that row)
  int my_ql, my_slot;
  pair_of(wave, (prow >> 2) & 1, (prow & 3) + 4 * (prow >> 3), my_ql, my_slot);
  const int my_q = min(q0 + my_ql, a.N - 1);
  const bool my_valid = my_slot < a.K;
  const int my_j = a.idx[(int64_t)my_q * a.K + min(my_slot, a.K - 1)];
  float r[16];   // r = relu(P1 d + c1) for this lane's 16 k-slots
  {
    const float* qp = a.qpos + (int64_t)my_q * a.qs;
    const float* ap = a.apos + (int64_t)my_j * a.as;
    const float dx = qp[0] - ap[0], dy = qp[1] - ap[1], dz = qp[2] - ap[2];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = 8 * (s >> 2) + 4 * half + (s & 3);
      const float* w = a.P1 + 3 * m;
      const float v = fmaf(dz, w[2], fmaf(dy, w[1], dx * w[0])) + a.c1[m];
      r[s] = my_valid ? fmaxf(v, 0.f) : 0.f;
    }
  }
  const float* aq_row = a.aq + (int64_t)my_q * a.ld_aq + 4 * half;
  const float* kt_row = a.kt + (int64_t)my_j * a.ld_kt + 4 * half;

  // Weight staging, branch-free (the hidden-block body below must be ONE scheduling region): the block's D rows
  // of W2 and HB rows of Wp form (D + HB) * 8 float4 = W2_LOADS * 512 exactly; thread t moves float4 number
  // t + 512 i, whose LDS row (f >> 3) runs through the W2 rows and on into the Wp rows.
  static_assert((D + HB) * 8 == W2_LOADS * 512, "weight block must be a whole number of float4 per thread");
  f32x4 pw[W2_LOADS];
  int woff[W2_LOADS];     // source offset (floats) of this thread's i-th float4 inside W2 (or the Wp block)
#pragma unroll
  for (int i = 0; i < W2_LOADS; ++i) {
    const int f = tid + 512 * i;
    woff[i] = f < W2_F4 ? (f >> 3) * H2 + 4 * (f & 7) : ((f >> 3) - D) * 32 + 4 * (f & 7);
  }
  auto gload = [&](int hb) {
#pragma unroll
    for (int i = 0; i < W2_LOADS; ++i) {
      const int f = tid + 512 * i;
      const float* src = f < W2_F4 ? a.w2 + hb * HB + woff[i] : a.wp + hb * HB * 32 + woff[i];
      pw[i] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto sstore = [&](int buf) {
    float* W = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < W2_LOADS; ++i) {
      const int f = tid + 512 * i;
      *reinterpret_cast<f32x4*>(W + (f >> 3) * LDW + 4 * (f & 7)) = pw[i];
    }
  };

  f32x16 acc[NTW];
#pragma unroll
  for (int c = 0; c < NTW; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  // Software pipeline over the hidden blocks (one branch-free scheduling region per block, ordered by the
  // sched_group_barrier sequence at the end of the loop body):
  //   first MFMAs : the Aq/Kt slices and the weight block (W2/Wp) of hb+1 are loaded in their shadow
  //   body        : GEMM1 -> relu -> GEMM2 on LDS buffer hb&1
  //   last MFMAs  : registers -> LDS buffer (hb+1)&1 in their shadow, then one barrier
  f32x4 av[4], kv[4];
  auto iload = [&](int hb, f32x4* A, f32x4* Kk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      A[g] = *reinterpret_cast<const f32x4*>(aq_row + hb * HB + 8 * g);
      Kk[g] = *reinterpret_cast<const f32x4*>(kt_row + hb * HB + 8 * g);
    }
  };
  gload(0);
  iload(0, av, kv);
  sstore(0);
  __syncthreads();
  const int frag_off = prow * LDW + 4 * half;

  for (int hb = 0; hb < NHB; ++hb) {
    const int buf = hb & 1;
    const int nb = hb + 1 < NHB ? hb + 1 : 0;      // the last iteration re-loads block 0 (harmless, keeps the body branch-free)
    const float* W = smem + buf * BUF;
    // GEMM1 accumulator init: Aq[q][hid] - Kt[j][hid], hid = 32 hb + 8 g + 4 half + i  (reg = 4 g + i)
    f32x16 hacc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      hacc[4 * g + 0] = av[g].x - kv[g].x; hacc[4 * g + 1] = av[g].y - kv[g].y;
      hacc[4 * g + 2] = av[g].z - kv[g].z; hacc[4 * g + 3] = av[g].w - kv[g].w;
    }
#ifndef OCC4D_ABLATE_NOINIT
    iload(nb, av, kv);                             // same registers: consumed just above
#endif
#ifndef OCC4D_ABLATE_NOLOAD
    gload(nb);
#endif
    const float* Wp = W + D * LDW + frag_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(Wp + 8 * g);
      hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, r[4 * g + 0], hacc, 0, 0, 0);
      hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, r[4 * g + 1], hacc, 0, 0, 0);
      hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, r[4 * g + 2], hacc, 0, 0, 0);
      hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, r[4 * g + 3], hacc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[i] = fmaxf(hacc[i], 0.f);
    // GEMM2 over this wave's channel tiles
    const float* W2 = W + CBEG * 32 * LDW + frag_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int c = 0; c < NTW; ++c) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(W2 + c * 32 * LDW + 8 * g);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[4 * g + 0], bv.x, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[4 * g + 1], bv.y, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[4 * g + 2], bv.z, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(hacc[4 * g + 3], bv.w, acc[c], 0, 0, 0);
      }
    }
#ifndef OCC4D_ABLATE_NOLOAD
    sstore(buf ^ 1);
#endif
#ifndef OCC4D_CA_NO_PIPE
    // Pipeline description for the scheduler (the whole hidden block is one region): the 8 Aq / Kt loads and the
    // W2_LOADS weight loads of the next block ride in the shadow of the first MFMAs, the ds_write_b128s that publish
    // the weights in the shadow of the last ones; fragment ds_reads and the VALU work are left to the compiler.
    // (Pinned at the top / bottom of the block instead, every wave of the workgroup did its memory phase at the
    // same time right after the barrier and the MFMA pipes idled: 13 % of the kernel.)
    {
      constexpr int NMFMA = 16 + 16 * NTW;
      constexpr int NVM = 8 + W2_LOADS;
#ifndef OCC4D_CA_P1
#define OCC4D_CA_P1 2
#endif
#ifndef OCC4D_CA_P2
#define OCC4D_CA_P2 3
#endif
      constexpr int P1 = OCC4D_CA_P1, P2 = OCC4D_CA_P2;
#ifndef OCC4D_CA_TAIL
#define OCC4D_CA_TAIL 0
#endif
      constexpr int TAIL = OCC4D_CA_TAIL;          // MFMAs after the last ds_write (covers its latency before the barrier)
      static_assert(NVM * P1 + W2_LOADS * P2 + TAIL <= NMFMA, "pipeline needs enough MFMAs");
#pragma unroll
      for (int i = 0; i < NVM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, P1, 0);    // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NVM * P1 - W2_LOADS * P2 - TAIL, 0);
#pragma unroll
      for (int i = 0; i < W2_LOADS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, P2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
      }
      if (TAIL > 0) __builtin_amdgcn_sched_group_barrier(0x008, TAIL, 0);
    }
#endif
#ifndef OCC4D_ABLATE_NOBAR
    __syncthreads();
#endif
  }
#ifdef OCC4D_ABLATE_NOEPI
  {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < NTW; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[c][i];
    if (t == 123.456f) a.agg[0] = t;   // keep the accumulators live
    return;
  }
#endif

  // ---- epilogue: positional-encoding GEMM, per-channel softmax over the neighbours, weighted sum.
  // C/D registers of this lane: 0-6 query A, 7-13 query B (slot 7*half + i), 14-15 the 9th query.
  // The weight buffers are free now (last loop barrier passed): the first 3*4*D floats of smem hold
  // the 9th query's per-row-tile partial softmax (max, sum, weighted sum).
  float* s_part = smem;
  int jrow[16];
  bool vrow[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int ql, slot;
    pair_of(wave, half, i, ql, slot);
    jrow[i] = s_idx[ql * 16 + min(slot, 15)];
    vrow[i] = slot < a.K;
  }
  const int qa = q0 + 2 * wave, qb = qa + 1;
  const float inv_div = 1.0f / a.divisor;
  constexpr float LOG2E = 1.44269504088896f;
  // Software pipeline over the channel tiles: the positional-encoding MFMAs of tile c + 1 are issued in the same
  // region as the softmax VALU work of tile c (independent streams of one wave: the matrix pipe runs in the
  // shadow of the VALU), gathered V rows are fetched one tile ahead, the P2 fragment of tile c + 2 is loaded into
  // the registers the MFMAs of tile c + 1 have just consumed.
  f32x4 pv[4];
  float vv[16], nvv[16];
  auto loadP = [&](int c) {
    const int ch = 32 * (CBEG + c) + prow;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      pv[g] = *reinterpret_cast<const f32x4*>(a.p2 + (int64_t)ch * 32 + 8 * g + 4 * half);
  };
  auto loadV = [&](int c, float* V) {
    const int ch = 32 * (CBEG + c) + prow;
#ifndef OCC4D_ABLATE_EPI_NOV
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = a.vt[(int64_t)jrow[i] * a.ld_vt + ch];
#else
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = (float)jrow[i];
#endif
  };
  auto pos_gemm = [&]() {
    f32x16 pe;
#pragma unroll
    for (int i = 0; i < 16; ++i) pe[i] = 0.f;
#ifndef OCC4D_ABLATE_EPI_NOPE
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      pe = __builtin_amdgcn_mfma_f32_32x32x2f32(r[4 * g + 0], pv[g].x, pe, 0, 0, 0);
      pe = __builtin_amdgcn_mfma_f32_32x32x2f32(r[4 * g + 1], pv[g].y, pe, 0, 0, 0);
      pe = __builtin_amdgcn_mfma_f32_32x32x2f32(r[4 * g + 2], pv[g].z, pe, 0, 0, 0);
      pe = __builtin_amdgcn_mfma_f32_32x32x2f32(r[4 * g + 3], pv[g].w, pe, 0, 0, 0);
    }
#else
    pe[0] = pv[0].x + pv[1].y + pv[2].z + pv[3].w;
#endif
    return pe;
  };
  loadP(0);
  loadV(0, vv);
  f32x16 pe = pos_gemm();
  if (NTW > 1) loadP(1);
#pragma unroll
  for (int c = 0; c < NTW; ++c) {
    const int ch = 32 * (CBEG + c) + prow;
    if (c + 1 < NTW) loadV(c + 1, nvv);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 pe_next = pe;
    if (c + 1 < NTW) {
      pe_next = pos_gemm();
      if (c + 2 < NTW) loadP(c + 2);
    }
    const float b2c = a.b2[ch], c2c = a.c2[ch];
    float lg[16], val[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lg[i] = vrow[i] ? (acc[c][i] + b2c) * inv_div : -__builtin_inff();
      val[i] = (pe[i] + c2c) + vv[i];
    }
    // queries A (regs 0-6) and B (regs 7-13): complete softmax with one cross-half exchange each
    float out2[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      float mx = -__builtin_inff();
#pragma unroll
      for (int i = 0; i < 7; ++i) mx = fmaxf(mx, lg[7 * qq + i]);
      mx = fmaxf(mx, xhalf(mx));
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        // exp(x) = 2^(x log2 e) on the hardware v_exp_f32; arguments are <= 0, exp2(-inf) = 0
        const float e = __builtin_amdgcn_exp2f((lg[7 * qq + i] - mx) * LOG2E);
        den += e;
        num += e * val[7 * qq + i];
      }
      den += xhalf(den);
      num += xhalf(num);
      out2[qq] = num / den;
    }
    // half 0 stores query A, half 1 stores query B (128 B coalesced each)
    const int qs = half ? qb : qa;
    if (qs < a.N) a.agg[(int64_t)qs * a.ld_agg + ch] = half ? out2[1] : out2[0];
    // 9th query: this row tile's partial over its 4 slots (regs 14, 15 of both halves)
    {
      float mx = fmaxf(lg[14], lg[15]);
      mx = fmaxf(mx, xhalf(mx));
      float den = 0.f, num = 0.f;
      if (mx > -__builtin_inff()) {
#pragma unroll
        for (int i = 14; i < 16; ++i) {
          const float e = __builtin_amdgcn_exp2f((lg[i] - mx) * LOG2E);
          den += e;
          num += e * val[i];
        }
      }
      den += xhalf(den);
      num += xhalf(num);
      if (half == 0) {
        s_part[(0 * 4 + wave) * D + ch] = mx;
        s_part[(1 * 4 + wave) * D + ch] = den;
        s_part[(2 * 4 + wave) * D + ch] = num;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NTW) {
      pe = pe_next;
#pragma unroll
      for (int i = 0; i < 16; ++i) vv[i] = nvv[i];
    }
  }
  __syncthreads();
  // merge the four partials of the 9th query: row tile 0's wave of each channel group, one channel per lane
  const int q8 = q0 + 8;
  if (wave == 0 && q8 < a.N) {
    for (int ch = 32 * CBEG + lane; ch < 32 * (CBEG + NTW); ch += 64) {
      float m = -__builtin_inff();
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fmaxf(m, s_part[(0 * 4 + w) * D + ch]);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float sc = __builtin_amdgcn_exp2f((s_part[(0 * 4 + w) * D + ch] - m) * LOG2E);
        den += sc * s_part[(1 * 4 + w) * D + ch];
        num += sc * s_part[(2 * 4 + w) * D + ch];
      }
      a.agg[(int64_t)q8 * a.ld_agg + ch] = num / den;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(512, 2) void cross_attn_kernel(const CrossAttnArgs a) {
  constexpr int BUF = (32 * NT + HB) * LDW;
  static_assert(2 * BUF >= 12 * 32 * NT, "partial-softmax scratch must fit the weight buffers");
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
  __shared__ int s_idx[QPB * 16];
  constexpr int NTW0 = (NT + 1) / 2;               // channel group 0: tiles [0, NTW0), group 1: the rest
  if (threadIdx.x < 256) cross_attn_body<NT, 0, NTW0>(a, smem, s_idx);
  else cross_attn_body<NT, NTW0, NT - NTW0>(a, smem, s_idx);
}

static int cross_attn_launch(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs,
                             const float* apos, int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt,
                             const float* vt, int64_t ld_vt, const float* P1, const float* c1, const float* wp,
                             const float* w2, const float* b2, const float* p2, const float* c2, float* agg,
                             int64_t ld_agg, int n, int m, int k, int d, float divisor, void* stream) {
  OCC4D_REQUIRE(aq && qpos && apos && idx && kt && vt && P1 && c1 && wp && w2 && b2 && p2 && c2 && agg,
                "occ4d_pt_cross_attn: null pointer");
  OCC4D_REQUIRE(d == 416 || d == 288, "occ4d_pt_cross_attn: fused kernel is built for d in {288, 416}, got %d", d);
  OCC4D_REQUIRE(k >= 1 && k <= KMAX, "occ4d_pt_cross_attn: k=%d outside [1,%d]", k, KMAX);
  OCC4D_REQUIRE(m >= 1 && n >= 0, "occ4d_pt_cross_attn: bad n/m");
  OCC4D_REQUIRE(ld_aq >= 2 * d && ld_kt >= 2 * d && ld_vt >= d && ld_agg >= d && qs >= 3 && as >= 3,
                "occ4d_pt_cross_attn: leading dimension too small");
  OCC4D_REQUIRE(ld_aq % 4 == 0 && ld_kt % 4 == 0 && ((uintptr_t)aq % 16) == 0 && ((uintptr_t)kt % 16) == 0 &&
                    ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)wp % 16) == 0 && ((uintptr_t)p2 % 16) == 0,
                "occ4d_pt_cross_attn: aq/kt/w2/wp/p2 must be 16-byte aligned with ld %% 4 == 0");
  OCC4D_REQUIRE(divisor > 0.f, "occ4d_pt_cross_attn: divisor must be > 0");
  if (n == 0) return OCC4D_OK;
  CrossAttnArgs a{aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wp, w2, b2, p2, c2,
                  agg, ld_agg, n, m, k, divisor};
  dim3 grid(occ4d::cdiv(n, QPB)), block(512);
  hipStream_t st = (hipStream_t)stream;
  if (d == 416) cross_attn_kernel<13><<<grid, block, 0, st>>>(a);
  else cross_attn_kernel<9><<<grid, block, 0, st>>>(a);
  return occ4d::check_launch("occ4d_pt_cross_attn");
}

}  // namespace

extern "C" int occ4d_pt_cross_attn_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs,
                                       const float* apos, int64_t as, const int32_t* idx, const float* kt,
                                       int64_t ld_kt, const float* vt, int64_t ld_vt, const float* P1,
                                       const float* c1, const float* wp, const float* w2, const float* b2,
                                       const float* p2, const float* c2, float* agg, int64_t ld_agg, int n, int m,
                                       int k, int d, float divisor, void* stream) {
  return cross_attn_launch(aq, ld_aq, qpos, qs, apos, as, idx, kt, ld_kt, vt, ld_vt, P1, c1, wp, w2, b2, p2, c2,
                           agg, ld_agg, n, m, k, d, divisor, stream);
}
