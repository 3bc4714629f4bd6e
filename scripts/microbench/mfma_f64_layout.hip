// Register layout of v_mfma_f64_16x16x4_f64 on gfx950 (development check): prints, for every lane and
// accumulator register, which D[i][j] it holds, assuming A lane l = A[l%16][l/16], B lane l = B[l/16][l%16].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int l = threadIdx.x;
  const int i = l & 15, kk = l >> 4;
  // A[i][k] = (i+1) if k==1 else 0 ; B[k][j] = 100*(j+1) if k==1 else 0  -> D[i][j] = 100 (i+1)(j+1)
  const double a = kk == 1 ? double(i + 1) : 0.0;
  const double b = kk == 1 ? 100.0 * (i + 1) : 0.0;
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[4 * l + r] = acc[r];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok_same = 1, ok_alt = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int v = int(h[4 * l + r] + 0.5);
      // decode i, j from v = 100 (i+1)(j+1) is ambiguous; test the two candidate layouts instead
      const int j = l & 15;
      const int i_same = (l >> 4) * 4 + r, i_alt = (l >> 4) + 4 * r;
      if (v != 100 * (i_same + 1) * (j + 1)) ok_same = 0;
      if (v != 100 * (i_alt + 1) * (j + 1)) ok_alt = 0;
    }
  printf("layout D[(l>>4)*4+r][l&15]: %s;  layout D[(l>>4)+4r][l&15]: %s\n", ok_same ? "YES" : "no", ok_alt ? "YES" : "no");
  for (int l = 0; l < 64; l += 16) printf("lane %2d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  printf("lane  1: %g %g %g %g\n", h[4], h[5], h[6], h[7]);
  return 0;
}
