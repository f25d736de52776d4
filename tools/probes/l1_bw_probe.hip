// Per-CU fill-rate probe: how many bytes per clock can ONE CU pull from its XCD's L2 (a) into LDS through LDS-DMA
// (buffer_load ... lds, 16 B per lane), (b) into VGPRs (buffer_load_dwordx4), (c) both at once, and (d) into VGPRs when
// two waves of the workgroup read the SAME lines a little apart (the second reader of a weight fragment).
// Decides whether the 256x256 GEMM gains from taking the weight operand global -> VGPR instead of through LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/l1_bw_probe tools/probes/l1_bw_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// mode bit 0: LDS-DMA stream, bit 1: VGPR stream, bit 2: VGPR stream reads the same addresses in waves w and w^4
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* buf, size_t span, int iters, unsigned long long* cycles, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (unsigned)span, 0x00020000);
  // every CU walks its own 1 KiB pieces; 32 CUs of an XCD cover the span repeatedly -> L2 hits after the first touch
  const unsigned piece0 = (blockIdx.x * 37u) % (unsigned)(span / 1024);
  const int wsel = (MODE & 4) ? (wave & 3) : wave;
  i32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned piece = (piece0 + (unsigned)(it * 4 + u) * 8u + (unsigned)wave) % (unsigned)(span / 1024);
      const unsigned piece_v = (piece0 + 4096u + (unsigned)(it * 4 + u) * 8u + (unsigned)wsel) % (unsigned)(span / 1024);
      if (MODE & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (wave * 4 + u) * 1024), 16, piece * 1024u + lane * 16u, 0, 0, 0);
      if (MODE & 2) {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, piece_v * 1024u + lane * 16u, 0, 0);
        acc += v;
      }
    }
    if (MODE & 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* buf, size_t span, int iters, unsigned long long* dcyc, int* sink) {
  const int ncu = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<MODE>, ncu, 512, 32 * 1024, 0, buf, span, iters / 4, dcyc, sink);   // warm L2
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<MODE>, ncu, 512, 32 * 1024, 0, buf, span, iters, dcyc, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(ncu);
  CHECK(hipMemcpy(h.data(), dcyc, ncu * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (auto c : h) avg += (double)c; avg /= ncu;
  const int streams = ((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0);
  const double bytes_cu = (double)iters * 4 * 8 * 1024 * streams;   // per CU: 8 waves x 4 x 1 KiB per iteration per stream
  printf("%-44s span %6.1f MB  %7.1f B/clk/CU  %6.1f GB/s/CU  chip %6.2f TB/s  (%.0f kcycles, %.3f ms)\n", name, span / 1048576.0,
         bytes_cu / avg, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * ncu / (ms * 1e-3) / 1e12, avg / 1e3, ms);
}

int main() {
  const size_t big = 512ull << 20;
  char* buf; unsigned long long* dcyc; int* sink;
  CHECK(hipMalloc(&buf, big)); CHECK(hipMemset(buf, 1, big));
  CHECK(hipMalloc(&dcyc, 256 * 8)); CHECK(hipMalloc(&sink, 4));
  for (size_t span : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20}) {
    const int iters = 2000;
    run<1>("LDS-DMA only", buf, span, iters, dcyc, sink);
    run<2>("VGPR only", buf, span, iters, dcyc, sink);
    run<3>("LDS-DMA + VGPR (distinct lines)", buf, span, iters, dcyc, sink);
    run<6>("VGPR only, waves w / w+4 same lines", buf, span, iters, dcyc, sink);
    run<7>("LDS-DMA + VGPR(shared by w / w+4)", buf, span, iters, dcyc, sink);
  }
  return 0;
}
