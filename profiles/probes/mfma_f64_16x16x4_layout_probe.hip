#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* C1, double* C2) {
  int lane = threadIdx.x;
  double p = ldexp(1.0, lane % 50);  // unique-ish power (lanes 50..63 alias 0..13, disambiguate with second test)
  double4_t c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(lane + 1), 1.0, c, 0, 0, 0);   // sum over k of a-lane ids feeding row i
  for (int r = 0; r < 4; r++) C1[lane * 4 + r] = c[r];
  double4_t d = {0,0,0,0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, (double)(lane + 1), d, 0, 0, 0);
  for (int r = 0; r < 4; r++) C2[lane * 4 + r] = d[r];
  (void)p;
}
int main() {
  double hC1[256], hC2[256];
  double *d1, *d2;
  (void)hipMalloc(&d1, 2048); (void)hipMalloc(&d2, 2048);
  k<<<1, 64>>>(d1, d2);
  (void)hipMemcpy(hC1, d1, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(hC2, d2, 2048, hipMemcpyDeviceToHost);
  // hypothesis: A[i][k] lane = i + 16k -> row sum = sum_k (i+16k+1) = 4i + 4 + 96 = 4i+100 ; B[k][j] lane = j+16k -> col sum = 4j+100
  for (int lane = 0; lane < 64; lane += 1) {
    printf("lane %2d:", lane);
    for (int r = 0; r < 4; r++) printf("  (row %5.1f col %5.1f)", (hC1[lane*4+r]-100)/4, (hC2[lane*4+r]-100)/4);
    printf("\n");
  }
  return 0;
}
