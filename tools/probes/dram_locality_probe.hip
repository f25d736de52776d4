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

// The output side: a 256 x 256 bf16 tile leaves as 16-row x 64-byte store instructions (the register epilogue's pattern: a lane
// holds 8 consecutive columns of one row) into a row-major [M, N] matrix (rows 2 N bytes apart) — or, blocked, into one contiguous
// 128 KiB per tile.  MODE 2: strided, 3: blocked.
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int MODE>
__global__ __launch_bounds__(256) void store_tiles(char* __restrict__ C, long M, long N, int tiles_m, int tiles_n) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c_lane = lane & 15, g_lane = lane >> 4;
  const long ld_b = N * 2;
  const i32x4 v = i32x4{lane, wave, 3, 4};
  for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    char* base = MODE == 2 ? C + (long)tm * 256 * ld_b + (long)tn * 512 : C + (long)t * 131072;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xFFFFFFF0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned voff;
        if (MODE == 2) voff = (unsigned)((wr * 128 + i * 16 + c_lane) * ld_b + (wc * 128 + j * 32 + 16 * (g_lane & 1) + 8 * (g_lane >> 1)) * 2);
        else voff = (unsigned)(((wave * 8 + i) * 4 + j) * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
      }
  }
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
#define RUNS(MODE)                                                                                              \
  {                                                                                                             \
    float best = 1e9f;                                                                                          \
    for (int it = 0; it < 6; ++it) {                                                                            \
      char* dst = (it & 1) ? A : B;                                                                             \
      hipEventRecord(e0);                                                                                       \
      hipLaunchKernelGGL((store_tiles<MODE>), dim3(cus), dim3(256), 0, 0, dst, Mp, K, tiles, (int)(K / 256));   \
      hipEventRecord(e1); hipEventSynchronize(e1);                                                              \
      float ms; hipEventElapsedTime(&ms, e0, e1);                                                               \
      if (it >= 2 && ms < best) best = ms;                                                                      \
    }                                                                                                           \
    printf("stores %-8s (256 x 256 tiles of a %ld-column matrix): %7.1f us  %5.2f TB/s\n", MODE == 2 ? "strided" : "blocked", K, best * 1e3, bytes / (best * 1e-3) / 1e12); \
  }
  RUNS(2); RUNS(3);
  hipError_t e = hipDeviceSynchronize();
  printf("%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  return e == hipSuccess ? 0 : 1;
}
