#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[4] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1"};
  for (int j = 0; j < 4; j++) { printf("%s:", nm[j]); for (int i = 0; i < 64; i += 4) printf(" %u", h[j * 64 + i]); printf("\n"); }
  return 0;
}
