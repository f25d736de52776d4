// What does the GEMM's A-operand access pattern cost on the HBM side?  A 256-row x 64-k stage of a k-contiguous [M, K] bf16 matrix is
// 256 segments of 128 bytes at a stride of 2 K bytes (6 KiB at K = 3072): every 128 bytes in another DRAM page, and the same page is
// not asked again before the next k-step, microseconds later.  This probe streams a matrix through an LDS-DMA ring exactly like the
// 4-wave GEMM does (one workgroup of 4 waves per CU, 32-KiB stages, a counted vmcnt keeping DEPTH stages in flight, persistent tile
// walk), with no MFMA at all, in two layouts of the same bytes:
//   strided   row-major [M, K]: stage (tile, s) = rows tile*256 .. +255, bytes s*128 .. +127 of each
//   blocked   [M/256][K/64][256][64]: the same stage is one contiguous 32 KiB
// and prints the rate each sustains.  If "strided" cannot go much faster than what the GEMMs draw in situ (1-2 TB/s), the ring is
// starved by DRAM page misses and a tile-blocked layout of the GEMM-to-GEMM tensors (FFN intermediates) would pay; if it streams at
// the copy rate (5+ TB/s), the GEMMs' gap to their L2-resident rate lies elsewhere.
// hipcc --offload-arch=gfx950 -O2 tools/probes/dram_locality_probe.hip -o tools/probes/dram_locality_probe && ./dram_locality_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: strided, 1: blocked.  DEPTH stages of 32 KiB in flight per workgroup (8 LDS-DMA instructions per wave and stage).
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ A, long M, long K, int tiles, int nst, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long ld_b = K * 2;
  unsigned acc = 0;
  int issued = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const char* base = MODE == 0 ? A + (long)t * 256 * ld_b : A + (long)t * 256 * ld_b;      // both layouts: a tile's bytes start here
    const long bytes = 256 * ld_b;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(bytes > 0xFFFFFFF0l ? 0xFFFFFFF0l : bytes), 0x00020000);
    for (int s = 0; s < nst; ++s) {
      char* st = smem + (issued % DEPTH) * 32768;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = wave + 4 * i;                // 32 pieces of 1 KiB
        unsigned voff;
        if (MODE == 0) {
          const int row = piece * 8 + (lane >> 3);
          voff = (unsigned)(row * ld_b + s * 128 + (lane & 7) * 16);
        } else {
          voff = (unsigned)((long)s * 32768 + piece * 1024 + lane * 16);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(st + piece * 1024), 16, voff, 0, 0, 0);
      }
      ++issued;
      wait_vm<8 * (DEPTH - 1)>();                      // the oldest stage in flight has landed
      if ((issued & 15) == 0) acc += *(const unsigned*)(smem + ((issued + 1) % DEPTH) * 32768 + lane * 4);   // keep the LDS image alive
    }
  }
  wait_vm<0>();
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 100864, K = argc > 2 ? atol(argv[2]) : 3072;
  const long Mp = (M + 255) / 256 * 256;
  const int tiles = (int)(Mp / 256), nst = (int)(K / 64);
  const size_t bytes = (size_t)Mp * K * 2;
  char *A, *B;
  unsigned* sink;
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&sink, 64);
  hipMemset(A, 1, bytes); hipMemset(B, 2, bytes);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("matrix %ld x %ld bf16 = %.1f MB, %d tiles x %d stages of 32 KiB, %d CUs\n", Mp, K, bytes / 1e6, tiles, nst, cus);
#define RUN(MODE, DEPTH, WGS)                                                                                   \
  {                                                                                                             \
    const int grid = cus * WGS;                                                                                  \
    hipFuncSetAttribute((const void*)stream<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768); \
    float best = 1e9f;                                                                                          \
    for (int it = 0; it < 6; ++it) {                                                                            \
      const char* src = (it & 1) ? A : B; /* alternate: nothing comes from the 256-MB Infinity Cache */        \
      hipEventRecord(e0);                                                                                       \
      hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(grid), dim3(256), DEPTH * 32768, 0, src, Mp, K, tiles, nst, sink); \
      hipEventRecord(e1); hipEventSynchronize(e1);                                                              \
      float ms; hipEventElapsedTime(&ms, e0, e1);                                                               \
      if (it >= 2 && ms < best) best = ms;                                                                      \
    }                                                                                                           \
    printf("%-8s depth %d, %d workgroup(s) per CU: %7.1f us  %5.2f TB/s\n", MODE == 0 ? "strided" : "blocked", DEPTH, WGS, best * 1e3, bytes / (best * 1e-3) / 1e12); \
  }
  RUN(0, 4, 1); RUN(1, 4, 1);
  RUN(0, 2, 1); RUN(1, 2, 1);
  RUN(0, 2, 2); RUN(1, 2, 2);
  RUN(0, 1, 4); RUN(1, 1, 4);
  hipError_t e = hipDeviceSynchronize();
  printf("%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
