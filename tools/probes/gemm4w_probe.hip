// Main-loop probe for a 4-wave form of the 256 x 256 bf16 GEMM tile (what the vendor's hand-written kernel uses):
// one wave per SIMD, 128 x 128 of output per wave (256 accumulator registers), operands staged through a ring of
// 32-KiB LDS stages by LDS-DMA, fragments software-pipelined in registers (the reads of step s + 1 are issued among the
// MFMAs of step s), one workgroup barrier per 32-k step.  No epilogue, operands L2-resident (every workgroup reads the
// same panels): the number this prints is the best case of the LOOP, to be compared with the production kernel's
// all-L2-hit, no-store figure (MDT_GEMM_DIAG=9: 2 330-2 800 shader cycles per 64-deep K-tile; 2 048 = pure MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gemm4w_probe tools/probes/gemm4w_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int STAGE = 2 * 256 * 64;     // A 256 rows x 32 k + B 256 cols x 32 k, 64-byte rows
constexpr int NB = 5, DIST = 4;

__device__ __forceinline__ int swz_h(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// VARIANT 0: loads first, then MFMAs, compiler's own interleaving; 1: sched_group_barrier pattern (2 MFMA : 1 ds_read, then 4 MFMA : 1 DMA);
// 2: source order = issue order (MFMAs as volatile asm with memory clobbers)
template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void loop_probe(const __bf16* A, const __bf16* B, int K, int tiles,
                                                                                            unsigned long long* cycles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t ld_b = (int64_t)K * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)(256 * ld_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)(256 * ld_b), 0x00020000);
  // the wave's 8 pieces of a step: pieces wave, wave + 4, wave + 8, wave + 12 of A and of B (a piece = 16 rows x 64 B)
  unsigned voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2);
    voff[i] = (unsigned)(row * ld_b + (((lane & 3) ^ swz_h(row)) * 16));
  }
  const int nhs = K / 32;
  auto issue = [&](int hs, int buf) {
    char* st = smem + buf * STAGE;
    const int ko = hs * 64;                       // byte offset of the step inside a row: rides in the scalar offset
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(st + (wave + 4 * i) * 1024), 16, voff[i], ko, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(st + 256 * 64 + (wave + 4 * i) * 1024), 16, voff[i], ko, 0, 0);
    }
  };
  auto frag = [&](const char* tile, int rc0) -> bf16x8 {
    const int row = rc0 + (lane & 15);
    return *(const bf16x8*)(tile + row * 64 + (((lane >> 4) ^ swz_h(row)) * 16));
  };
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][8], fb[2][8];

  // prologue: steps 0 .. DIST - 1 in flight, fragments of step 0 in registers
#pragma unroll
  for (int h = 0; h < DIST; ++h) issue(h, h);
  wait_vm<(DIST - 1) * 8>();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[0][i] = frag(smem, wr * 128 + i * 16); fb[0][i] = frag(smem + 256 * 64, wc * 128 + i * 16); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int b_next = 1, b_wr = DIST % NB, hs_global = 0;
  const int total = tiles * nhs;
  // one step: stage (s + 1) -> the other fragment set, stage (s + DIST) requested, 64 MFMAs on the current set
  auto step = [&](auto cur_c, int s) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
    wait_vm<(DIST - 2) * 8>();                    // own pieces of step s + 1 have landed (s + 2, s + 3 may be in flight)
    __builtin_amdgcn_s_barrier();                 // ... and everybody's; stage s - 1 is free to be overwritten
    const char* tn = smem + b_next * STAGE;
    if constexpr (VARIANT == 2) {
      // source order IS the issue order: every MFMA is a volatile asm statement with a memory clobber, so the fragment
      // reads and the LDS-DMA issues stay where they are written — 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a DMA, 2 MFMAs
      char* st = smem + b_wr * STAGE;
      const int ko = ((s + DIST) % nhs) * 64;
#define MF(i_, j_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fb[cur][j_]), "v"(fa[cur][i_]) : "memory")
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        MF(b, 0); MF(b, 1);
        fa[nxt][b] = frag(tn, wr * 128 + b * 16);
        MF(b, 2); MF(b, 3);
        fb[nxt][b] = frag(tn + 256 * 64, wc * 128 + b * 16);
        MF(b, 4); MF(b, 5);
        if (b < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(st + (wave + 4 * b) * 1024), 16, voff[b], ko, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(st + 256 * 64 + (wave + 4 * (b - 4)) * 1024), 16, voff[b - 4], ko, 0, 0);
        MF(b, 6); MF(b, 7);
      }
#undef MF
    } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[nxt][i] = frag(tn, wr * 128 + i * 16); fb[nxt][i] = frag(tn + 256 * 64, wc * 128 + i * 16); }
    issue((s + DIST) % nhs, b_wr);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
    }
    if constexpr (VARIANT == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
      for (int k = 0; k < 8; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    b_next = b_next + 1 == NB ? 0 : b_next + 1;
    b_wr = b_wr + 1 == NB ? 0 : b_wr + 1;
  };
  for (; hs_global + 1 < total; hs_global += 2) {
    step(std::integral_constant<int, 0>{}, hs_global % nhs);
    step(std::integral_constant<int, 1>{}, (hs_global + 1) % nhs);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  wait_vm<0>();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) sink[0] = s;
}

template <int VARIANT>
static void run(const char* name, const __bf16* A, const __bf16* B, int K, int tiles, unsigned long long* dcyc, float* sink) {
  const int ncu = 256;
  CHECK(hipFuncSetAttribute((const void*)loop_probe<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * STAGE));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(loop_probe<VARIANT>, ncu, 256, NB * STAGE, 0, A, B, K, tiles, dcyc, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(ncu);
  CHECK(hipMemcpy(h.data(), dcyc, ncu * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double ktiles = (double)tiles * (K / 64);
  const double fl = 2.0 * 256 * 256 * 64 * ktiles * ncu;
  printf("%-40s K=%d: cycles per 64-deep K-tile median %.0f (p10 %.0f p90 %.0f)   %.0f TFLOP/s over the launch\n", name, K,
         h[ncu / 2] / ktiles, h[ncu / 10] / ktiles, h[ncu * 9 / 10] / ktiles, fl / (ms * 1e-3) / 1e12);
}

int main() {
  const int K = 3072, rows = 256;
  std::vector<unsigned short> hA((size_t)rows * K), hB((size_t)rows * K);
  srand(1);
  for (auto& v : hA) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));     // random mantissas, both signs below
  for (auto& v : hB) v = (unsigned short)(((rand() & 1) << 15) | 0x3c00 | (rand() & 0x3ff));
  __bf16 *A, *B; unsigned long long* dcyc; float* sink;
  CHECK(hipMalloc(&A, hA.size() * 2)); CHECK(hipMalloc(&B, hB.size() * 2));
  CHECK(hipMalloc(&dcyc, 256 * 8)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  for (int k : {768, 3072}) {
    run<0>("4 waves, compiler's interleaving", A, B, k, 64 * 3072 / k, dcyc, sink);
    run<1>("4 waves, sched_group_barrier pattern", A, B, k, 64 * 3072 / k, dcyc, sink);
    run<2>("4 waves, hand-ordered (asm MFMAs)", A, B, k, 64 * 3072 / k, dcyc, sink);
  }
  return 0;
}
