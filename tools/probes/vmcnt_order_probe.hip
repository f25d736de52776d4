// Does a counted s_waitcnt vmcnt(N) see vector-memory operations retire in issue order on gfx950?
// Each wave issues ONE cold load (a line nobody touched: HBM latency), then a younger operation of another kind, then waits with
// vmcnt(1) — "all but my youngest operation are done" — and looks at the load's destination register: if it still holds the
// sentinel, the YOUNGER operation left the count first.  Younger operations tried:
//   0  a real global store to a line that is hot in L2
//   1  a buffer store through a descriptor of zero records (every lane out of range: dropped)
//   2  a buffer load through a descriptor of zero records (returns 0)
//   3  a second cold load (control: loads among themselves)
// hipcc --offload-arch=gfx950 -O2 tools/probes/vmcnt_order_probe.hip -o tools/probes/vmcnt_order_probe && ./vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int KIND>
__global__ void probe(const int* __restrict__ cold, int* __restrict__ hot, int* __restrict__ seen, long stride) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int* src = cold + w * stride + lane;          // one cold 256-byte row per wave
  int* dst = hot + (w & 1023) * 64 + lane;
  const __amdgpu_buffer_rsrc_t nullrs = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 0, 0x00020000);
  int v = -12345, early, junk = 0;
  asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(src) : "memory");
  if (KIND == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(lane) : "memory");
  if (KIND == 1) asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(lane), "v"(lane * 4), "s"(nullrs) : "memory");
  if (KIND == 2) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(junk) : "v"(lane * 4), "s"(nullrs) : "memory");
  if (KIND == 3) asm volatile("global_load_dword %0, %1, off" : "=v"(junk) : "v"(src + stride / 2) : "memory");
  asm volatile("s_waitcnt vmcnt(1)\n\tv_mov_b32 %0, %1" : "=v"(early) : "v"(v) : "memory");      // what the register holds when "all but one" are done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  seen[w * 64 + lane] = (early == -12345 && v != -12345) ? 1 : 0;
  if (junk == 0x7fffffff) hot[0] = junk;
}

// The same question for the operation the GEMM rings wait for: an LDS-DMA load (buffer_load ... lds).  "Done" there means the bytes
// are in LDS; the wave reads the LDS word right behind s_waitcnt vmcnt(1).
template <int KIND>
__global__ void probe_lds(const int* __restrict__ cold, int* __restrict__ hot, float* __restrict__ fsum, int* __restrict__ seen, long stride, long cold_bytes) {
  __shared__ int buf[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + wv;
  int* dst = hot + (w & 1023) * 64 + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, (unsigned)cold_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t nullrs = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 0, 0x00020000);
  int* mine = buf + wv * 256;                              // 1 KiB per wave: one 16-byte LDS-DMA per lane
  for (int i = lane; i < 256; i += 64) mine[i] = -12345;
  __builtin_amdgcn_s_waitcnt(0);
  int junk = 0;
  const unsigned voff = (unsigned)((w * stride + lane * 4) * 4);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)mine, 16, voff, 0, 0, 0);
  if (KIND == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(lane) : "memory");
  if (KIND == 1) asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(lane), "v"(lane * 4), "s"(nullrs) : "memory");
  if (KIND == 2) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(junk) : "v"(lane * 4), "s"(nullrs) : "memory");
  if (KIND == 3) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(fsum + (lane & 7)), "v"(1.0f) : "memory");
  if (KIND == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(nullrs, (__attribute__((address_space(3))) void*)(buf + 1024 - 256 + 0), 16, (unsigned)(lane * 16), 0, 0, 0);   // LDS-DMA through a zero-record descriptor
  int early;
  asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  asm volatile("ds_read_b32 %0, %1" : "=v"(early) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) int*)(mine + lane * 4)) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  seen[w * 64 + lane] = (early == -12345) ? 1 : 0;
  if (junk == 0x7fffffff) hot[0] = junk;
}

int main() {
  const long waves = 256 * 16 * 4, stride = 4096;      // 16 KB between the rows of different waves: every row a cold line
  int *cold, *hot, *seen;
  hipMalloc(&cold, waves * stride * 4 * 2);
  hipMalloc(&hot, 1024 * 64 * 4);
  hipMalloc(&seen, waves * 64 * 4);
  int* h = (int*)malloc(waves * 64 * 4);
  const char* names[4] = {"real global store (hot L2 line)", "buffer store, zero-record descriptor (dropped)", "buffer load, zero-record descriptor", "second cold load (control)"};
  for (int kind = 0; kind < 4; ++kind) {
    long total = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(cold, 1, waves * stride * 4 * 2);      // also evicts: 512 MB written > every cache
      hipMemset(hot, 0, 1024 * 64 * 4);
      hipMemset(seen, 0, waves * 64 * 4);
      hipDeviceSynchronize();
      if (kind == 0) probe<0><<<waves / 4, 256>>>(cold + (rep & 1) * waves * stride, hot, seen, stride);
      if (kind == 1) probe<1><<<waves / 4, 256>>>(cold + (rep & 1) * waves * stride, hot, seen, stride);
      if (kind == 2) probe<2><<<waves / 4, 256>>>(cold + (rep & 1) * waves * stride, hot, seen, stride);
      if (kind == 3) probe<3><<<waves / 4, 256>>>(cold + (rep & 1) * waves * stride, hot, seen, stride);
      hipDeviceSynchronize();
      hipMemcpy(h, seen, waves * 64 * 4, hipMemcpyDeviceToHost);
      for (long i = 0; i < waves * 64; ++i) total += h[i];
    }
    printf("younger op = %-48s: the OLDER cold load was still in flight after s_waitcnt vmcnt(1) in %ld of %ld lanes\n", names[kind], total, 5 * waves * 64);
  }
  float* fsum;
  hipMalloc(&fsum, 64);
  hipMemset(fsum, 0, 64);
  const char* lnames[5] = {"real global store (hot L2 line)", "buffer store, zero-record descriptor (dropped)", "buffer load, zero-record descriptor", "global_atomic_add_f32 (no return)", "LDS-DMA load, zero-record descriptor"};
  for (int kind = 0; kind < 5; ++kind) {
    long total = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(cold, 1, waves * stride * 4 * 2);
      hipMemset(seen, 0, waves * 64 * 4);
      hipDeviceSynchronize();
      const int* c = cold + (rep & 1) * waves * stride;
      const long cb = waves * stride * 4;
      if (kind == 0) probe_lds<0><<<waves / 4, 256>>>(c, hot, fsum, seen, stride, cb);
      if (kind == 1) probe_lds<1><<<waves / 4, 256>>>(c, hot, fsum, seen, stride, cb);
      if (kind == 2) probe_lds<2><<<waves / 4, 256>>>(c, hot, fsum, seen, stride, cb);
      if (kind == 3) probe_lds<3><<<waves / 4, 256>>>(c, hot, fsum, seen, stride, cb);
      if (kind == 4) probe_lds<4><<<waves / 4, 256>>>(c, hot, fsum, seen, stride, cb);
      hipDeviceSynchronize();
      hipMemcpy(h, seen, waves * 64 * 4, hipMemcpyDeviceToHost);
      for (long i = 0; i < waves * 64; ++i) total += h[i];
    }
    printf("OLDER = cold LDS-DMA load, younger = %-48s: LDS still held the old bytes after s_waitcnt vmcnt(1) in %ld of %ld lanes\n", lnames[kind], total, 5 * waves * 64);
  }
  return 0;
}
