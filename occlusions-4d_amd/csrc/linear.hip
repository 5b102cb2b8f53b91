// torch.nn.Linear on row tiles, exact fp32 on v_mfma_f32_32x32x2_f32 (K3/K11).
//
// Tiling (wave64, one wave per SIMD): a 256-thread workgroup owns BM = 128 rows x
// BN = 32*NT columns (NT = 13 covers the decoder's 416-wide layers with no padding
// waste: 416 = 13 * 32).  Wave w owns row tile w and ALL NT column tiles, so one A
// fragment feeds NT MFMAs and the 16*NT accumulators stay in the unified VGPR/AGPR
// file.  x and w tiles (BK = 32) are staged global -> registers -> LDS, double
// buffered, one barrier per k-tile; rows are padded to 36 floats so the
// ds_read_b128 fragment reads are bank-conflict free (stride 9 x 16 B is odd).  A k-tile is one
// branch-free scheduling region: a sched_group_barrier sequence puts the next tile's global loads
// between the first MFMAs and its ds_write_b128s between the last (one wave per SIMD has nobody
// else to cover a memory phase).
//
// Fragment / k mapping: lane l supplies row (l & 31); within an 8-wide k group the
// lower half-wave reads k = 0..3 and the upper k = 4..7 as one float4 each, and MFMA
// step i consumes element i of both, i.e. the k order inside a group is
// (0,4),(1,5),(2,6),(3,7).  A and B use the same map, so the product is exact; only
// the fp32 summation order differs from a serial dot product.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
#ifndef OCC4D_LINEAR_BK
#define OCC4D_LINEAR_BK 32
#endif
constexpr int BK = OCC4D_LINEAR_BK;   // k-tile depth
constexpr int LDT = BK + 4;           // padded LDS row (floats): stride 5 or 9 x 16 B is odd -> conflict-free ds_read_b128

// EPI: 0 = scalar epilogue (unaligned / N % 4 != 0), 1 = float4 epilogue with bias / relu only,
//      2 = float4 epilogue with residual and/or gathered row terms (their loads are batched).
// SWISH: the activation applied to x on load is x * sigmoid(x) (relu_in == 2: the reference's 'swish' option,
//      model/implicit.py:46-64) instead of the branch-free relu / identity floor.
__device__ inline float swish1(float v) { return v * (1.0f / (1.0f + __expf(-v))); }

template <int NT, int EPI, bool SWISH = false>
__global__ __launch_bounds__(256, 1) void linear_kernel(const occ4d_linear_args a) {
  constexpr int BN = 32 * NT;
  constexpr int F4K = BK / 4;                              // float4 per tile row
  constexpr int ALOADS = (BM * F4K) / 256;
  constexpr int WLOADS = (NT * 32 * F4K + 255) / 256;      // float4 per thread for the w tile
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
  float* const As0 = smem;
  float* const Ws0 = smem + 2 * BM * LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;
  const int M = a.M, K = a.K, N = a.N;

  f32x4 ra[ALOADS], rw[WLOADS];

  // Unconditional, clamped loads + select (no exec-mask branches): the whole k-tile body must be ONE scheduling
  // region so that the prefetch can be interleaved with the MFMA stream (sched_group_barrier below).
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < ALOADS; ++i) {
      const int f = tid + 256 * i;
      const int r = row0 + f / F4K, k = k0 + 4 * (f % F4K);
      const bool ok = r < M && k < K;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.x + (int64_t)min(r, M - 1) * a.ldx + min(k, K - 4));
      ra[i].x = ok ? v.x : 0.f; ra[i].y = ok ? v.y : 0.f; ra[i].z = ok ? v.z : 0.f; ra[i].w = ok ? v.w : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int f = tid + 256 * i;
      const int c = col0 + f / F4K, k = k0 + 4 * (f % F4K);
      const bool ok = f / F4K < BN && c < N && k < K;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.w + (int64_t)min(c, N - 1) * a.ldw + min(k, K - 4));
      rw[i].x = ok ? v.x : 0.f; rw[i].y = ok ? v.y : 0.f; rw[i].z = ok ? v.z : 0.f; rw[i].w = ok ? v.w : 0.f;
    }
  };
  const float relu_floor = a.relu_in == 1 ? 0.f : -__builtin_inff();     // branch-free relu-on-load
  auto sstore = [&](int buf) {
    float* As = As0 + buf * BM * LDT;
    float* Ws = Ws0 + buf * BN * LDT;
#pragma unroll
    for (int i = 0; i < ALOADS; ++i) {
      const int f = tid + 256 * i;
      f32x4 v = ra[i];
      if (SWISH) {
        v.x = swish1(v.x); v.y = swish1(v.y); v.z = swish1(v.z); v.w = swish1(v.w);
      } else {
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
      }
      *reinterpret_cast<f32x4*>(As + (f / F4K) * LDT + 4 * (f % F4K)) = v;
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int f = tid + 256 * i;
      if ((NT * 32 * F4K) % 256 == 0 || f / F4K < BN) *reinterpret_cast<f32x4*>(Ws + (f / F4K) * LDT + 4 * (f % F4K)) = rw[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int frag_off = (lane & 31) * LDT + 4 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
#ifndef OCC4D_ABLATE_NOLOAD
    gload(kt + 1 < nk ? (kt + 1) * BK : 0);          // always (the last one is a harmless re-load of tile 0)
#endif
    const float* Ab = As0 + buf * BM * LDT + wave * 32 * LDT + frag_off;
    const float* Wb = Ws0 + buf * BN * LDT + frag_off;
    constexpr int GH = NT > 7 ? (NT + 1) / 2 : NT;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Ab + 8 * j);
#pragma unroll
      for (int c0 = 0; c0 < NT; c0 += GH) {
        f32x4 bv[GH];
#pragma unroll
        for (int cc = 0; cc < GH; ++cc)
          if (c0 + cc < NT) bv[cc] = *reinterpret_cast<const f32x4*>(Wb + (c0 + cc) * 32 * LDT + 8 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int cc = 0; cc < GH; ++cc)
            if (c0 + cc < NT)
              acc[c0 + cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[cc][i], acc[c0 + cc], 0, 0, 0);
        }
      }
    }
#ifndef OCC4D_ABLATE_NOLOAD
    sstore(buf ^ 1);
#endif
#ifndef OCC4D_LINEAR_NO_PIPE
    // Pipeline description for the scheduler (one region = the whole k-tile): the ALOADS + WLOADS global loads ride
    // in the shadow of the first MFMAs, the matching ds_write_b128s in the shadow of the last ones (by then the
    // data has had > 100 MFMAs = 6000+ cycles to arrive); ds_read_b128 fragment reads are left to the compiler.
    {
      constexpr int NLD = ALOADS + WLOADS;
      constexpr int NMFMA = (BK / 2) * NT;
      constexpr int PER = 3;                                    // MFMAs between two memory instructions
      static_assert(2 * NLD * PER <= NMFMA || NT < 4, "pipeline needs enough MFMAs");
      if (NT >= 4) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 2 * NLD * PER, 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
        }
      }
    }
#endif
    __syncthreads();
  }
#ifdef OCC4D_ABLATE_NOEPI
  {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[c][r];
    if (t == 123.456f) a.y[0] = t;
    return;
  }
#endif

  // epilogue: C/D map of 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int half = lane >> 5;
  if (EPI != 0) {
    // Through LDS (free after the k-loop's last barrier) so that every global access is a
    // 16-byte, row-contiguous one: per-lane 4-byte stores at a row stride were latency/issue
    // bound (the K = 32 pair GEMM ran at 0.6 TB/s).  Each wave transposes its own 32 x (32*CT)
    // chunk in a private LDS region: write conflict-free ds_write_b32, read ds_read_b128.
    constexpr int REGION = (2 * (BM + BN) * LDT) / 4;                       // floats per wave
    constexpr int CT_FIT = (REGION / 32 - 4) / 32;
    constexpr int CT = CT_FIT < 1 ? 1 : (CT_FIT > NT ? NT : (CT_FIT > 4 ? 4 : CT_FIT));
    constexpr int LDE = 32 * CT + 4;
    static_assert(32 * LDE <= REGION, "epilogue staging does not fit the wave's LDS region");
    float* E = smem + wave * REGION;
    constexpr int F4_PER_ROW = 8 * CT;                  // float4 per staged row
    constexpr int ROWS_PER_PASS = 64 / F4_PER_ROW > 0 ? 64 / F4_PER_ROW : 1;
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += CT) {
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) {
        if (c0 + cc < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            E[((r & 3) + 8 * (r >> 2) + 4 * half) * LDE + 32 * cc + (lane & 31)] = acc[c0 + cc][r];
        }
      }
      // (same wave wrote and reads: program order + lgkmcnt is enough, no barrier)
      const int ncol4 = 8 * ((NT - c0) < CT ? (NT - c0) : CT);     // live float4 per row in this chunk
      const int c4 = lane % F4_PER_ROW, rsub = lane / F4_PER_ROW;
      const int col = col0 + 32 * c0 + 4 * c4;
      const bool col_ok = c4 < ncol4 && col < N && rsub < ROWS_PER_PASS;   // 64 % F4_PER_ROW lanes idle
      f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(a.bias + col);
      if (EPI == 1) {
        for (int rp = 0; rp < 32; rp += ROWS_PER_PASS) {
          const int rl = rp + rsub;
          const int row = row0 + wave * 32 + rl;
          if (col_ok && row < M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(E + rl * LDE + 4 * c4);
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (a.relu_out) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<f32x4*>(a.y + (int64_t)row * a.ldy + col) = v;
          }
        }
        continue;
      }
      // All global loads of the chunk (residual, gathered rows) are issued before the first store:
      // y may alias residual (in-place residual blocks), so the compiler keeps source order, and a
      // load-add-store per pass serialised 16 HBM round trips per chunk (measured: 29 us of a 147 us
      // launch).
      constexpr int PASSES = 32 / ROWS_PER_PASS;
      constexpr int PB = PASSES > 8 ? 8 : PASSES;          // passes per load batch (register budget)
#pragma unroll
      for (int p0 = 0; p0 < PASSES; p0 += PB) {
        f32x4 res[PB], addv[PB], subv[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
          const int row = row0 + wave * 32 + (p0 + p) * ROWS_PER_PASS + rsub;
          const bool ok = col_ok && row < M;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          res[p] = (a.residual && ok) ? *reinterpret_cast<const f32x4*>(a.residual + (int64_t)row * a.ldr + col) : z;
          addv[p] = (a.add_rows && ok)
                        ? *reinterpret_cast<const f32x4*>(a.add_rows + (int64_t)(row / a.add_div) * a.ld_add + col) : z;
          subv[p] = (a.sub_rows && ok)
                        ? *reinterpret_cast<const f32x4*>(a.sub_rows + (int64_t)a.sub_idx[row] * a.ld_sub + col) : z;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
          const int rl = (p0 + p) * ROWS_PER_PASS + rsub;
          const int row = row0 + wave * 32 + rl;
          if (col_ok && row < M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(E + rl * LDE + 4 * c4);
            v.x += bias4.x + addv[p].x - subv[p].x; v.y += bias4.y + addv[p].y - subv[p].y;
            v.z += bias4.z + addv[p].z - subv[p].z; v.w += bias4.w + addv[p].w - subv[p].w;
            if (a.relu_out) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            v.x += res[p].x; v.y += res[p].y; v.z += res[p].z; v.w += res[p].w;
            *reinterpret_cast<f32x4*>(a.y + (int64_t)row * a.ldy + col) = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row >= M) continue;
    const float* addp = a.add_rows ? a.add_rows + (int64_t)(row / a.add_div) * a.ld_add : nullptr;
    const float* subp = a.sub_rows ? a.sub_rows + (int64_t)a.sub_idx[row] * a.ld_sub : nullptr;
    const float* resp = a.residual ? a.residual + (int64_t)row * a.ldr : nullptr;
    float* yp = a.y + (int64_t)row * a.ldy;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int col = col0 + 32 * c + (lane & 31);
      if (col >= N) continue;
      float v = acc[c][r];
      if (a.bias) v += a.bias[col];
      if (addp) v += addp[col];
      if (subp) v -= subp[col];
      if (a.relu_out) v = fmaxf(v, 0.f);
      if (resp) v += resp[col];
      yp[col] = v;
    }
  }
}

template <int NT>
int launch(const occ4d_linear_args& a, hipStream_t st) {
  dim3 grid(occ4d::cdiv(a.M, BM), occ4d::cdiv(a.N, 32 * NT)), block(256);
  auto al16 = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
  const bool vec = a.N % 4 == 0 && a.ldy % 4 == 0 && al16(a.y) && (!a.bias || al16(a.bias)) &&
                   (!a.residual || (al16(a.residual) && a.ldr % 4 == 0)) &&
                   (!a.add_rows || (al16(a.add_rows) && a.ld_add % 4 == 0)) &&
                   (!a.sub_rows || (al16(a.sub_rows) && a.ld_sub % 4 == 0));
  const bool extras = a.residual || a.add_rows || a.sub_rows;
  if (a.relu_in == 2) {
    if (!vec) linear_kernel<NT, 0, true><<<grid, block, 0, st>>>(a);
    else if (!extras) linear_kernel<NT, 1, true><<<grid, block, 0, st>>>(a);
    else linear_kernel<NT, 2, true><<<grid, block, 0, st>>>(a);
  } else if (!vec) linear_kernel<NT, 0><<<grid, block, 0, st>>>(a);
  else if (!extras) linear_kernel<NT, 1><<<grid, block, 0, st>>>(a);
  else linear_kernel<NT, 2><<<grid, block, 0, st>>>(a);
  return occ4d::check_launch("occ4d_linear_f32");
}

}  // namespace

extern "C" int occ4d_linear_f32(const occ4d_linear_args* args, void* stream) {
  OCC4D_REQUIRE(args, "occ4d_linear_f32: null args");
  const occ4d_linear_args& a = *args;
  OCC4D_REQUIRE(a.x && a.w && a.y, "occ4d_linear_f32: null x/w/y");
  OCC4D_REQUIRE(a.M >= 0 && a.N >= 1 && a.K >= 4, "occ4d_linear_f32: bad M/N/K = %d/%d/%d", a.M, a.N, a.K);
  OCC4D_REQUIRE(a.K % 4 == 0 && a.ldx % 4 == 0 && a.ldw % 4 == 0,
                "occ4d_linear_f32: K, ldx, ldw must be multiples of 4 (K=%d ldx=%lld ldw=%lld)", a.K,
                (long long)a.ldx, (long long)a.ldw);
  OCC4D_REQUIRE(((uintptr_t)a.x % 16) == 0 && ((uintptr_t)a.w % 16) == 0,
                "occ4d_linear_f32: x and w must be 16-byte aligned");
  OCC4D_REQUIRE(a.ldx >= a.K && a.ldw >= a.K && a.ldy >= a.N, "occ4d_linear_f32: leading dimension too small");
  OCC4D_REQUIRE(!a.add_rows || a.add_div >= 1, "occ4d_linear_f32: add_div must be >= 1");
  OCC4D_REQUIRE(a.relu_in >= 0 && a.relu_in <= 2 && (a.relu_out == 0 || a.relu_out == 1),
                "occ4d_linear_f32: relu_in = %d (0 none, 1 relu, 2 swish), relu_out = %d (0 / 1)", a.relu_in, a.relu_out);
  OCC4D_REQUIRE(!a.sub_rows || a.sub_idx, "occ4d_linear_f32: sub_rows needs sub_idx");
  if (a.M == 0) return OCC4D_OK;
  hipStream_t st = (hipStream_t)stream;
  const int n = a.N;
  // Column tiles per workgroup: as many as cover N (one A fragment feeds NT MFMAs) -- unless the launch would leave
  // most of the 256 CUs idle: the encoder's last levels are (531 .. 1593) x 288 GEMMs, 5-13 row tiles; with all 288
  // columns in one workgroup they ran 60-75 us each on 5 CUs, on the critical path behind the last FPS.  Fewer columns
  // per workgroup = more, shorter workgroups; the k order of every output element is unchanged (bit-identical).
  static const int kNT[7] = {13, 9, 5, 4, 3, 2, 1};
  int pick = 6;
  for (int i = 6; i >= 0; --i)
    if (32 * kNT[i] >= n || i == 0) { pick = i; break; }          // smallest NT that covers N (or 13)
  const int row_tiles = occ4d::cdiv(a.M, BM);
  while (pick < 6 && row_tiles * occ4d::cdiv(n, 32 * kNT[pick]) < 128) ++pick;
  switch (kNT[pick]) {
    case 1: return launch<1>(a, st);
    case 2: return launch<2>(a, st);
    case 3: return launch<3>(a, st);
    case 4: return launch<4>(a, st);
    case 5: return launch<5>(a, st);
    case 9: return launch<9>(a, st);
    default: return launch<13>(a, st);
  }
}
