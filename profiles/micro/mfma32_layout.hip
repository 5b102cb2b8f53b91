// Which lane / register holds which element of v_mfma_f32_32x32x16_f16's operands and result (gfx950)?
// hipcc --offload-arch=gfx950 -O2 profiles/micro/mfma32_layout.hip -o /tmp/mfma32 && /tmp/mfma32
// A[i][k] = 1 only at (i0, k0), B[k][j] = 1 only at (k0, j0) under the ASSUMED operand layout
//   A: lane l -> row l % 32, k = 8 (l / 32) + e (e = 0 .. 7);  B: lane l -> column l % 32, k = 8 (l / 32) + e
// then D has a single 1 at (i0, j0): the program finds the (lane, register) that holds it and prints the C/D map,
// and checks the assumed operand layout by varying k0 (a mismatch between A's and B's k gives an all-zero D).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int i0, int j0, int ka, int kb, float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (l % 32 == i0 && 8 * (l / 32) + e == ka) ? (_Float16)1.0f : (_Float16)0.0f;
    b[e] = (l % 32 == j0 && 8 * (l / 32) + e == kb) ? (_Float16)1.0f : (_Float16)0.0f;
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
  float* d;
  hipMalloc(&d, 64 * 16 * 4);
  float h[64 * 16];
  int bad = 0;
  for (int t = 0; t < 200; ++t) {
    const int i0 = (t * 7) % 32, j0 = (t * 11 + 3) % 32, k0 = (t * 5) % 16, k1 = (t % 3 == 0) ? (k0 + 1) % 16 : k0;
    k<<<1, 64>>>(i0, j0, k0, k1, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int hits = 0, lane = -1, reg = -1;
    for (int i = 0; i < 64 * 16; ++i) if (h[i] != 0.f) { ++hits; lane = i / 16; reg = i % 16; }
    const int want_hits = k0 == k1 ? 1 : 0;
    // assumed C/D map: column j = lane % 32, row i = (reg % 4) + 4 (lane / 32) + 8 (reg / 4)
    const int row = hits == 1 ? (reg % 4) + 4 * (lane / 32) + 8 * (reg / 4) : -1, col = hits == 1 ? lane % 32 : -1;
    if (hits != want_hits || (hits == 1 && (row != i0 || col != j0))) {
      ++bad;
      printf("MISMATCH i0 %d j0 %d ka %d kb %d: hits %d at lane %d reg %d (assumed map gives row %d col %d)\n", i0, j0, k0, k1, hits, lane, reg, row, col);
    }
  }
  printf(bad ? "layout assumptions WRONG (%d cases)\n" : "layout assumptions hold: A lane l -> row l%%32, k 8(l/32)+e; B lane l -> col l%%32, k 8(l/32)+e; D lane l reg r -> col l%%32, row (r%%4)+4(l/32)+8(r/4)   (%d mismatches)\n", bad);
  return bad != 0;
}
