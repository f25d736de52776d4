// Prints what v_permlane16_swap does to lane ids (diagnostic for direct_epilogue's column assembly).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int v = 0; v < 2; ++v) { printf(v ? "src:  " : "vdst: "); for (int g = 0; g < 4; ++g) printf("row%d=[%u..%u] ", g, h[v * 64 + g * 16], h[v * 64 + g * 16 + 15]); printf("\n"); }
  return 0;
}
