// Probe of v_mfma_f32_16x16x128_f8f6f4 on gfx950: (1) numerics with the operand layout the 8-bit GEMM uses — lane l supplies row
// l % 16 and 32 consecutive bytes (k = 32 (l / 16) .. + 32) of a 128-k row, A and B alike (any k permutation is fine as long as both
// operands share it); e4m3 x e4m3 and e5m2 x e4m3; (2) issue rate against v_mfma_f32_16x16x32_bf16 (cycles per instruction with 16
// independent accumulators).  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f8_probe.hip -o /tmp/mfma_f8_probe && /tmp/mfma_f8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int FA>
__global__ void mm(const uint8_t* a, const uint8_t* b, float* d) {
  const int lane = threadIdx.x;
  const i32x8 fa = *(const i32x8*)(a + (lane & 15) * 128 + (lane >> 4) * 32);
  const i32x8 fb = *(const i32x8*)(b + (lane & 15) * 128 + (lane >> 4) * 32);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, c, FA, 0, 0, 0, 0, 0);
  // D[i][j]: lane holds column j = lane % 16 (row of B), rows i = 4 (lane / 16) + r (rows of A)
  for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

template <bool F8>
__global__ void rate(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  i32x8 a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {8, 7, 6, 5, 4, 3, 2, 1};
  bf16x8 a2, b2;
  for (int i = 0; i < 8; ++i) { a2[i] = (__bf16)(float)(threadIdx.x + i); b2[i] = (__bf16)(float)(i); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (F8) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float dec(uint8_t v, int e5m2) {
  const int s = v >> 7;
  float r;
  if (!e5m2) { const int e = (v >> 3) & 15, m = v & 7; r = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7); }
  else { const int e = (v >> 2) & 31, m = v & 3; r = e == 0 ? ldexpf((float)m, -16) : ldexpf(1.0f + m / 4.0f, e - 15); }
  return s ? -r : r;
}

int main() {
  uint8_t ha[16 * 128], hb[16 * 128];
  float hd[256];
  uint8_t *a, *b; float* d;
  hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd);
  int bad = 0;
  for (int fa = 0; fa < 2; ++fa) {
    srand(7 + fa);
    for (int i = 0; i < 16 * 128; ++i) {
      // finite, moderate magnitudes: e4m3 exponent 4..9, e5m2 exponent 12..17
      ha[i] = fa ? (uint8_t)(((rand() & 1) << 7) | ((12 + rand() % 6) << 2) | (rand() & 3)) : (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
      hb[i] = (uint8_t)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));
    }
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    if (fa) hipLaunchKernelGGL(mm<1>, 1, 64, 0, 0, a, b, d); else hipLaunchKernelGGL(mm<0>, 1, 64, 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)dec(ha[i * 128 + k], fa) * (double)dec(hb[j * 128 + k], 0);
        const double err = fabs(ref - hd[i * 16 + j]) / (fabs(ref) + 1.0);
        if (err > worst) worst = err;
      }
    printf("A %s x B e4m3: worst relative error %.3g %s\n", fa ? "e5m2" : "e4m3", worst, worst < 1e-5 ? "ok" : "MISMATCH");
    for (int j = 0; j < 4; ++j) {
      double ref = 0, ref32 = 0; float f = 0;
      for (int k = 0; k < 128; ++k) { const float pr = dec(ha[k], fa) * dec(hb[j * 128 + k], 0); ref += pr; f += pr; }
      printf("   D[0][%d] = %.6f   fp64 reference %.6f   fp32 sequential %.6f\n", j, hd[j], ref, (double)f);
    }
    bad |= !(worst < 1e-5);
  }
  {   // small integers: every partial sum is exact in fp32 whatever the order — separates layout from accumulation precision
    srand(11);
    for (int i = 0; i < 16 * 128; ++i) {
      ha[i] = (uint8_t)(((rand() & 1) << 7) | ((7 + rand() % 2) << 3) | ((rand() & 3) << 1));   // +-{1, 1.25, 1.5, 1.75, 2, 2.5, 3, 3.5}
      hb[i] = (uint8_t)(((rand() & 1) << 7) | ((7 + rand() % 2) << 3) | ((rand() & 3) << 1));
    }
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mm<0>, 1, 64, 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)dec(ha[i * 128 + k], 0) * (double)dec(hb[j * 128 + k], 0);
        worst = fmax(worst, fabs(ref - hd[i * 16 + j]));
      }
    printf("quarter-integer operands (exact sums): worst absolute error %.3g\n", worst);
  }
  float* out; unsigned long long* cyc; unsigned long long hc;
  hipMalloc(&out, 256 * 4 * 1024); hipMalloc(&cyc, 8);
  for (int f8 = 0; f8 < 2; ++f8)
    for (int rep = 0; rep < 2; ++rep) {
      if (f8) hipLaunchKernelGGL(rate<true>, 1, 256, 0, 0, out, cyc, 2000); else hipLaunchKernelGGL(rate<false>, 1, 256, 0, 0, out, cyc, 2000);
      hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
      // s_memtime ticks at the constant 100 MHz reference: report ticks per instruction and let the ratio speak
      if (rep) printf("%s: %.4f reference ticks per MFMA (one wave per SIMD, 16 accumulators)\n", f8 ? "f8 16x16x128" : "bf16 16x16x32", (double)hc / (2000.0 * 16));
    }
  return bad;
}
