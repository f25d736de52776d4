// GEMM kernels for the mDT hot path on gfx950.
//
//  * gemm_bf16_tile128: the workhorse.  bf16 operands, fp32 accumulate on
//    v_mfma_f32_16x16x32_bf16; 128x128x64 tiles, 4 waves (2x2, 64x64 each), LDS
//    double-buffered and filled by buffer_load ... lds (16 B / lane, no VGPR staging,
//    out-of-range rows read as zero through the buffer descriptor).  Each operand may be
//    k-contiguous (nn.Linear weights, activations in forward / dgrad) or k-major
//    (both operands of the weight gradient dW = dY^T X): k-major tiles keep their
//    memory layout in LDS and fragments are fetched with ds_read_b64_tr_b16.
//    LDS images are XOR-swizzled on the *source* address (LDS-DMA writes lane-linear).
//  * gemm_generic: any shape / stride / dtype mix on v_mfma_f32_16x16x4_f32 (exact
//    fp32 FMA chains) — the fp32 parity path and the fallback for odd shapes.
//
// Replaces the nn.Linear calls of the reference (see include/mdt_hip.h, mdt_gemm).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

namespace mdt {

// ------------------------------------------------------------------ generic kernel
// 64x64x16 tiles, 4 waves (2x2 of 32x32), operands converted to fp32 in LDS.
template <typename TIn, typename TOut, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmParams p) {
  constexpr int BM = 64, BN = 64, BK = 16, LD = 68;
  __shared__ float As[BK][LD];
  __shared__ float Bs[BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_chunk;
  const int64_t kend = (kbeg + p.k_chunk < p.K) ? kbeg + p.k_chunk : p.K;
  const TIn* A = (const TIn*)p.A;
  const TIn* B = (const TIn*)p.B;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int r, k;
      if constexpr (!TA) { r = e >> 4; k = e & 15; } else { k = e >> 6; r = e & 63; }
      const int64_t gr = m0 + r, gk = k0 + k;
      float v = 0.f;
      if (gr < p.M && gk < kend) v = to_f32(TA ? A[gk * p.lda + gr] : A[gr * p.lda + gk]);
      As[k][r] = v;
      int c, kb;
      if constexpr (!TB) { c = e >> 4; kb = e & 15; } else { kb = e >> 6; c = e & 63; }
      const int64_t gc = n0 + c, gkb = k0 + kb;
      float w = 0.f;
      if (gc < p.N && gkb < kend) w = to_f32(TB ? B[gkb * p.ldb + gc] : B[gc * p.ldb + gkb]);
      Bs[kb][c] = w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk + (lane >> 4)][wr * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (lane >> 4)][wc * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        epilogue_store<TIn, TOut>(p, m0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r,
                                  n0 + wc * 32 + j * 16 + (lane & 15), acc[i][j][r]);
}

template <typename TOut, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_bf16_tile128(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (A 16K + B 16K)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware tile order: the blocks sharing an XCD (blockIdx % 8) walk one contiguous
  // chunk of the row-major tile grid, so an A row panel is reused from that XCD's L2.
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * T_BM, n0 = (int64_t)tn * T_BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_chunk;
  const int64_t kend = (kbeg + p.k_chunk < p.K) ? kbeg + p.k_chunk : p.K;
  const int nk = (int)((kend - kbeg + T_BK - 1) / T_BK);

  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  // descriptors based at the first row this block touches (32-bit offsets stay small)
  const char* a_base;
  const char* b_base;
  int64_t a_bytes, b_bytes;
  if constexpr (!A_KM) { a_base = (const char*)p.A + m0 * lda_b; a_bytes = (p.M - m0) * lda_b; }
  else { a_base = (const char*)p.A + kbeg * lda_b; a_bytes = (kend - kbeg) * lda_b; }
  if constexpr (!B_KM) { b_base = (const char*)p.B + n0 * ldb_b; b_bytes = (p.N - n0) * ldb_b; }
  else { b_base = (const char*)p.B + kbeg * ldb_b; b_bytes = (kend - kbeg) * ldb_b; }
  const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
  const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);
  // k offsets are relative to the descriptor base: k-contiguous operands start at
  // absolute k (base = row m0, column 0), k-major ones at kbeg.
  const int64_t a_k0 = A_KM ? 0 : kbeg, b_k0 = B_KM ? 0 : kbeg;
  const int a_col0 = A_KM ? (int)m0 : 0, b_col0 = B_KM ? (int)n0 : 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    stage_tile<A_KM, 128, 4>(rsA, lda_b, a_k0, a_col0, smem, wave, lane);
    stage_tile<B_KM, 128, 4>(rsB, ldb_b, b_k0, b_col0, smem + T_TILE_BYTES, wave, lane);
  }
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * T_TILE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * T_TILE_BYTES;
    // own loads of tile kt landed (vmcnt(0) inserted by the compiler for the LDS-DMA)
    // + everyone finished reading the buffer tile kt+1 is about to overwrite
    __syncthreads();
    if (kt + 1 < nk) {
      stage_tile<A_KM, 128, 4>(rsA, lda_b, a_k0 + (int64_t)(kt + 1) * T_BK, a_col0, nxt, wave, lane);
      stage_tile<B_KM, 128, 4>(rsB, ldb_b, b_k0 + (int64_t)(kt + 1) * T_BK, b_col0, nxt + T_TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = load_frag<A_KM, 128>(cur, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag<B_KM, 128>(cur + T_TILE_BYTES, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
  }
  tile_epilogue<TOut>(p, acc, smem, wave, lane, m0 + wr * 64, n0 + wc * 64);
}

// ------------------------------------------------------------------ bf16 256-row tiles, 8 waves
// One block (512 threads = 8 waves, two per SIMD) per CU.
//   <BN=128, 4x2 waves, 3 stages>: 64x64 per wave; the LDS-DMA loads of K-step kt+2 are issued
//       while kt computes, each wave waits with a COUNTED vmcnt (its newest loads — step kt+1 —
//       stay in flight) and the block meets at a raw s_barrier, so loads never drain to zero in
//       the loop (cdna_hip_programming.md "Pipelining across barriers").
//   <BN=256, 2x4 waves, 2 stages>: 128x64 per wave — 131 flops per byte pulled from L2 into LDS
//       (128x128 tiles: 65) and 0.375 ds_read_b128 per MFMA (0.5): the variant for the big
//       token-count GEMMs, whose limiter is the L2 -> LDS stream, not HBM.
template <typename TOut, bool A_KM, bool B_KM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(512) void gemm_bf16_tile256(GemmParams p) {
  constexpr int BM = 256;
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;          // 16x16 accumulator tiles per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int LOADS = (BM + BN) / 64;                          // LDS-DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_chunk;
  const int64_t kend = (kbeg + p.k_chunk < p.K) ? kbeg + p.k_chunk : p.K;
  const int nk = (int)((kend - kbeg + T_BK - 1) / T_BK);

  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const char* a_base;
  const char* b_base;
  int64_t a_bytes, b_bytes;
  if constexpr (!A_KM) { a_base = (const char*)p.A + m0 * lda_b; a_bytes = (p.M - m0) * lda_b; }
  else { a_base = (const char*)p.A + kbeg * lda_b; a_bytes = (kend - kbeg) * lda_b; }
  if constexpr (!B_KM) { b_base = (const char*)p.B + n0 * ldb_b; b_bytes = (p.N - n0) * ldb_b; }
  else { b_base = (const char*)p.B + kbeg * ldb_b; b_bytes = (kend - kbeg) * ldb_b; }
  const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
  const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);
  const int64_t a_k0 = A_KM ? 0 : kbeg, b_k0 = B_KM ? 0 : kbeg;
  const int a_col0 = A_KM ? (int)m0 : 0, b_col0 = B_KM ? (int)n0 : 0;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kt) {
    char* st = smem + (kt % NSTAGE) * STAGE;
    stage_tile<A_KM, BM, 8>(rsA, lda_b, a_k0 + (int64_t)kt * T_BK, a_col0, st, wave, lane);
    stage_tile<B_KM, BN, 8>(rsB, ldb_b, b_k0 + (int64_t)kt * T_BK, b_col0, st + A_BYTES, wave, lane);
  };
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (nk > s) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's loads of step kt have landed once only the loads of later steps are pending
    if (NSTAGE == 3 && kt + 1 < nk) {
      if constexpr (LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // every wave's part of step kt is in LDS; step kt-1 is fully consumed
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);   // overwrites the stage step kt-1 used
    const char* cur = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = load_frag<B_KM, BN>(cur + A_BYTES, wc * (NI * 16) + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = load_frag<A_KM, BM>(cur, wr * (MI * 16) + i * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
  }
  // 64x64 blocks through the wave's 16 KiB LDS slot, one (MI = 4) or two (MI = 8) of them
#pragma unroll
  for (int h = 0; h < MI / 4; ++h) {
    f32x4 blk[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) blk[i][j] = acc[h * 4 + i][j];
    tile_epilogue<TOut>(p, blk, smem, wave, lane, m0 + wr * (MI * 16) + h * 64, n0 + wc * 64);
  }
}

// ------------------------------------------------------------------ bf16 256x256, ping-pong over a ring of 32-k stages
// 8 waves, each a 128x64 block of the 256x256 tile (32 accumulator tiles).  K advances in steps of 32: one
// step = a READ segment (12 fragment reads of the step's stage, LDS-DMA issue of the stage PP_DIST steps
// ahead) and an MFMA segment (32 x v_mfma_f32_16x16x32_bf16), separated by s_barrier.  Waves 4-7 run one
// barrier behind waves 0-3, so on every SIMD one of its two resident waves is in an MFMA segment while the
// other fetches operands (MI355X_MICROARCH.md "Two waves per SIMD").
//   Stages: PP_NB buffers of 32 KiB (A 256 rows x 32 k, B 256 cols x 32 k) used as a ring; step s reads buffer
//   s % PP_NB and issues step s + PP_DIST into the buffer step s - 1 used (PP_NB = PP_DIST + 1).  The L2 -> LDS fill
//   is what bounds this kernel (64 KiB per 64 k per CU; a CU fills ~70 GB/s from L2 and ~30 GB/s from the
//   Infinity Cache), so the ring keeps 3-4 stages = 96-128 KiB in flight per CU.
//   barrier pairing: the early group's barrier n is the late group's barrier n-1.
//   RAW: a wave drains its own pieces of step s+1 (counted vmcnt) before the barrier that closes its READ
//        segment of step s; anyone's first read of step s+1 comes after both groups passed that barrier.
//   WAR: the buffer of step s-1 is overwritten from READ segment s on; the late group's reads of step s-1
//        completed (lgkmcnt(0)) before the barrier the early group passes on entering READ segment s.
constexpr int pp_nb(int dist) { return dist + 1 < 4 ? 4 : dist + 1; }   // 4 buffers = 128 KiB, 5 = all 160 KiB of LDS

template <typename TOut, bool A_KM, bool B_KM, int PP_DIST>
__global__ __launch_bounds__(512) void gemm_bf16_pp256(GemmParams p) {
  constexpr int PP_NB = pp_nb(PP_DIST);
  constexpr int BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 64;
  constexpr bool SWAP = sizeof(TOut) == 2;   // bf16 output: transposed accumulators + direct_epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const bool late = wave >= 4;

  // Work items = (K slab, tile), slab-major.  Workgroups are dispatched x-fastest and dealt round-robin to the 8 XCDs
  // by their linear id, so XCD x = id % 8 takes the x-th contiguous run of the work list: with split-K (the weight
  // gradients: 9-36 tiles per slab, 7-28 slabs) the tiles that share a slab's dY / X panels sit in ONE XCD's L2
  // instead of being re-fetched through the fabric by all eight (3 x fewer fabric bytes on these launches).
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = tiles * (int)gridDim.z;
  const int bid = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int slab = work / tiles;
  const int tile = work - slab * tiles;
  // Within a slab each XCD owns a contiguous run of this order: `group_n` column tiles wide, all the way down the rows,
  // then the next group (group_n = tiles_n is plain row-major).
  int tm, tn;
  {
    const int G = p.group_n, per_group = p.tiles_m * G;
    const int gi = tile / per_group;
    const int full = p.tiles_n / G;
    if (gi < full) {
      const int r = tile - gi * per_group;
      tm = r / G;
      tn = gi * G + (r - tm * G);
    } else {
      const int gsz = p.tiles_n - full * G;
      const int r = tile - full * per_group;
      tm = r / gsz;
      tn = full * G + (r - tm * gsz);
    }
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kbeg = (int64_t)slab * p.k_chunk;
  const int64_t kend = (kbeg + p.k_chunk < p.K) ? kbeg + p.k_chunk : p.K;
  const int nk = (int)((kend - kbeg + T_BK - 1) / T_BK);
  const int nhs = 2 * nk;                      // 32-deep steps

  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const char* a_base;
  const char* b_base;
  int64_t a_bytes, b_bytes;
  if constexpr (!A_KM) { a_base = (const char*)p.A + m0 * lda_b; a_bytes = (p.M - m0) * lda_b; }
  else { a_base = (const char*)p.A + kbeg * lda_b; a_bytes = (kend - kbeg) * lda_b; }
  if constexpr (!B_KM) { b_base = (const char*)p.B + n0 * ldb_b; b_bytes = (p.N - n0) * ldb_b; }
  else { b_base = (const char*)p.B + kbeg * ldb_b; b_bytes = (kend - kbeg) * ldb_b; }
  const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
  const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);
  const int64_t a_k0 = A_KM ? 0 : kbeg, b_k0 = B_KM ? 0 : kbeg;
  const int a_col0 = A_KM ? (int)m0 : 0, b_col0 = B_KM ? (int)n0 : 0;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // MDT_EPI_ASUM (weight gradients): the workgroups of tile column 0 also sum op(A) over k — each of the four waves
  // that hold the same A fragments takes two of the eight row tiles against a B fragment of ones
  const bool asum = !SWAP && (p.epilogue & MDT_EPI_ASUM) && tn == 0;
  f32x4 accb0 = f32x4{0.f, 0.f, 0.f, 0.f}, accb1 = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = bf16x8{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

  // step hs (32 k) -> ring buffer `buf`: 4 LDS-DMA pieces per wave (2 of A, 2 of B)
  auto issue_step = [&](int hs, int buf) {
    char* st = smem + buf * PP_STAGE;
    stage_step<A_KM, BM, 8>(rsA, lda_b, a_k0 + (int64_t)hs * 32, a_col0, st, wave, lane);
    stage_step<B_KM, BN, 8>(rsB, ldb_b, b_k0 + (int64_t)hs * 32, b_col0, st + A_BYTES, wave, lane);
  };
  auto wait_pieces = [&](int halves) {           // all but the youngest `halves` steps of this wave have landed
    if (halves >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (halves == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (halves == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(PP_DIST >= 2 && PP_DIST <= 4, "wait_pieces assumes at most 3 steps left in flight");
#pragma unroll
  for (int h = 0; h < PP_DIST; ++h)
    if (h < nhs) issue_step(h, h);            // PP_DIST <= PP_NB - 1
  wait_pieces((nhs < PP_DIST ? nhs : PP_DIST) - 1);     // step 0 has landed
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();      // the stagger: waves 4-7 run one segment behind
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t_cyc = 0, t_real = 0;
  if (p.stamps) { t_cyc = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); }

  int b_rd = 0, b_wr = PP_DIST % PP_NB;        // b_wr == (hs + PP_DIST) % PP_NB: a buffer last read at step hs - 1 or earlier
  for (int hs = 0; hs < nhs; ++hs) {
    const char* cur = smem + b_rd * PP_STAGE;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = load_frag_h<B_KM, BN>(cur + A_BYTES, wc * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = load_frag_h<A_KM, BM>(cur, wr * 128 + i * 16, lane);
    if (hs + PP_DIST < nhs) issue_step(hs + PP_DIST, b_wr);
    {
      // steps issued so far reach min(hs + PP_DIST, nhs - 1); step hs + 1 must have landed when this segment closes
      const int last = hs + PP_DIST < nhs - 1 ? hs + PP_DIST : nhs - 1;
      const int fly = last - (hs + 1);
      wait_pieces(fly > 0 ? fly : 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = SWAP ? mfma_bf16(b[j], a[i], acc[i][j]) : mfma_bf16(a[i], b[j], acc[i][j]);
    if (asum) {
      if (wc == 0) { accb0 = mfma_bf16(a[0], ones, accb0); accb1 = mfma_bf16(a[1], ones, accb1); }
      else if (wc == 1) { accb0 = mfma_bf16(a[2], ones, accb0); accb1 = mfma_bf16(a[3], ones, accb1); }
      else if (wc == 2) { accb0 = mfma_bf16(a[4], ones, accb0); accb1 = mfma_bf16(a[5], ones, accb1); }
      else { accb0 = mfma_bf16(a[6], ones, accb0); accb1 = mfma_bf16(a[7], ones, accb1); }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    b_wr = b_wr + 1 == PP_NB ? 0 : b_wr + 1;
    b_rd = b_rd + 1 == PP_NB ? 0 : b_rd + 1;
  }
  if (!late) __builtin_amdgcn_s_barrier();     // equalise the barrier count of the two groups
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + 4 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x);
    o[0] = __builtin_amdgcn_s_memtime() - t_cyc;
    o[1] = __builtin_amdgcn_s_memrealtime() - t_real;
    o[2] = (unsigned long long)nk;
    o[3] = t_real;
  }
  if constexpr (SWAP) {
    direct_epilogue<2>(p, acc, lane, m0 + wr * 128, n0 + wc * 64);
    return;
  }
  if (asum && (lane & 15) == 0) {      // every column of the ones product holds the row sum: column 0 reports it
    const int64_t mr = m0 + wr * 128 + wc * 32 + 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (mr + r < p.M) atomicAdd(p.colsum + mr + r, p.alpha * accb0[r]);
      if (mr + 16 + r < p.M) atomicAdd(p.colsum + mr + 16 + r, p.alpha * accb1[r]);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 blk[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) blk[i][j] = acc[h * 4 + i][j];
    tile_epilogue<TOut>(p, blk, smem, wave, lane, m0 + wr * 128 + h * 64, n0 + wc * 64);
  }
}

// ------------------------------------------------------------------ persistent form of the ping-pong kernel
// One workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ... (the XCD-aware order above holds because
// gridDim.x is a multiple of 8).  The stage ring simply keeps turning across tile boundaries: the last PP_DIST
// steps of a tile issue the first PP_DIST steps of the NEXT tile, so its operands land while this tile's
// register-resident epilogue (no LDS, no barrier) runs — the ~2 us first-fetch latency and the workgroup
// launch / drain that a 12-step (K = 768) tile pays per tile otherwise are gone.  At a boundary the early
// group takes one extra barrier so both groups run their epilogues at the same time, and the stagger is
// re-created on entry to the next tile.  bf16 output, split_k == 1, at least PP_DIST steps per tile.
// F8 != 0: 8-bit operands (OCP e4m3 / e5m2), both k-contiguous.  A ring stage is the same 64-byte-row image — now 64 k deep —
// and a 16-byte fragment read holds TWO K = 32 operands (bytes 0-7 and 8-15 of the lane's chunk: the k order inside a stage
// is permuted the same way for both operands, which a contraction does not see), so a step issues 64
// v_mfma_f32_16x16x32_{fp8,bf8}_fp8 per wave on half the bytes per flop.  F8 = 1: activations e4m3, 2: e5m2 (gradients);
// weights e4m3.  The output is scaled by *alpha_dev = 1 / (scale_a * scale_b).
typedef __attribute__((ext_vector_type(2))) long i64x2;
template <int F8>
__device__ __forceinline__ f32x4 mfma_f8(long w, long x, f32x4 c) {      // w: weight operand (e4m3), x: activation operand
  if constexpr (F8 == 2) return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(w, x, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w, x, c, 0, 0, 0);
}

// JIT (stress builds of the same source, MDT_GEMM_DIAG=16, tools/gemm_stress.py): every wave sleeps a pseudo-random 0-900 cycles at
// each point of the protocol where another wave may be waiting for it or racing it — before its fragment reads, between reads
// and LDS-DMA issue, before its counted wait, on either side of both barriers of a step, around the epilogue and in front of a
// tile's first barrier.  The ring's RAW / WAR argument (above gemm_bf16_pp256) must hold under ANY interleaving: a hole shows
// as a bit difference against the 128 x 128 kernel within a few hundred launches instead of on one box in fifteen.
template <bool A_KM, bool B_KM, int PP_DIST, int EPK = -1, int F8 = 0, bool JIT = false>
__global__ __launch_bounds__(512) void gemm_bf16_pp256p(GemmParams p) {
  static_assert(F8 == 0 || (!A_KM && !B_KM), "8-bit operands are k-contiguous");
  constexpr int PP_NB = pp_nb(PP_DIST);
  constexpr int BM = 256, BN = 256;
  constexpr int A_BYTES = BM * 64;
  constexpr int ESZ = F8 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const bool late = wave >= 4;
  const int nvt = p.tiles_m * p.tiles_n;
  const int nhs = (int)((p.K * ESZ + 63) / 64);           // 64-byte steps
  const int64_t lda_b = p.lda * ESZ, ldb_b = p.ldb * ESZ;

  struct Desc { __amdgpu_buffer_rsrc_t rsA, rsB; int a_col0, b_col0; int64_t m0, n0; };
  auto make_desc = [&](int v) {
    const int q8 = nvt >> 3, r8 = nvt & 7, xcd = v & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
    int tm, tn;
    {
      const int G = p.group_n, per_group = p.tiles_m * G;
      const int gi = tile / per_group;
      const int full = p.tiles_n / G;
      if (gi < full) {
        const int r = tile - gi * per_group;
        tm = r / G;
        tn = gi * G + (r - tm * G);
      } else {
        const int gsz = p.tiles_n - full * G;
        const int r = tile - full * per_group;
        tm = r / gsz;
        tn = full * G + (r - tm * gsz);
      }
    }
    Desc d;
    d.m0 = (int64_t)tm * BM;
    d.n0 = (int64_t)tn * BN;
    // diagnostic (MDT_GEMM_DIAG=8): every tile LOADS the operand panels of tile (0, 0) — all fills hit L2 — while
    // stores still go to the tile's own place: separates the fill's miss path from the loop's own cost
    const int64_t lm0 = (p.epilogue & (1 << 23)) ? 0 : d.m0, ln0 = (p.epilogue & (1 << 23)) ? 0 : d.n0;
    const char* a_base;
    const char* b_base;
    int64_t a_bytes, b_bytes;
    if constexpr (!A_KM) { a_base = (const char*)p.A + lm0 * lda_b; a_bytes = (p.M - lm0) * lda_b; }
    else { a_base = (const char*)p.A; a_bytes = p.K * lda_b; }
    if constexpr (!B_KM) { b_base = (const char*)p.B + ln0 * ldb_b; b_bytes = (p.N - ln0) * ldb_b; }
    else { b_base = (const char*)p.B; b_bytes = p.K * ldb_b; }
    const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
    const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);
    d.a_col0 = A_KM ? (int)lm0 : 0;
    d.b_col0 = B_KM ? (int)ln0 : 0;
    return d;
  };
  auto issue_step = [&](const Desc& d, int hs, int buf) {
    char* st = smem + buf * PP_STAGE;
    stage_step<A_KM, BM, 8, MDT_GEMM_A_AUX>(d.rsA, lda_b, (int64_t)hs * 32, d.a_col0, st, wave, lane);
    stage_step<B_KM, BN, 8>(d.rsB, ldb_b, (int64_t)hs * 32, d.b_col0, st + A_BYTES, wave, lane);
  };
  auto wait_pieces = [&](int halves) {
    if (halves >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (halves == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (halves == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(PP_DIST >= 2 && PP_DIST <= 4, "wait_pieces assumes at most 3 steps left in flight");
  unsigned jit_state = JIT ? ((unsigned)blockIdx.x * 2654435761u) ^ ((unsigned)wave * 0x9E3779B9u) ^ (unsigned)__builtin_amdgcn_s_memtime() : 0u;
  auto jitter = [&]() __attribute__((always_inline)) {
    if constexpr (JIT) {
      jit_state = jit_state * 1664525u + 1013904223u;
      const int n = __builtin_amdgcn_readfirstlane((int)(jit_state >> 24) & 7);
      __builtin_amdgcn_sched_barrier(0);
      for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (p.epilogue & (1 << 22)) {   // diagnostic (MDT_GEMM_DIAG=4): skew the workgroups of an XCD so their epilogues do not coincide
    const int steps = ((blockIdx.x >> 3) & 7) * (nhs / 8);
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(28);   // ~ one 32-k step each
  }
  if (p.epilogue & (1 << 26)) {   // diagnostic (MDT_GEMM_DIAG=64): skew whole XCDs against each other (their workgroups stay in phase)
    const int steps = (blockIdx.x & 7) * ((nhs + 12) / 8);          // a tile period ~ nhs steps + an epilogue worth ~12
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(28);
  }
  // Tile walk.  Static (tile_queue == NULL): v = blockIdx.x, + gridDim.x, ...  Dynamic: the first tile is blockIdx.x,
  // every later one comes from a queue — one head per XCD (virtual ids v = x + 8 j keep the XCD-aware order), other
  // XCDs' queues are raided when the own one is empty — so a CU that starts late or shares its time with another
  // kernel (RCCL during backward) simply takes fewer tiles instead of holding the whole launch back.  Wave 0 pops the
  // id one tile ahead (the atomic's latency hides under the epilogue) and hands it to the other waves through the one
  // ring buffer that is idle at a tile boundary (the stage to be rewritten next), between two workgroup barriers.
  const bool dyn = p.tile_queue != nullptr;
  const int my_xcd = blockIdx.x & 7;
  auto pop_tile = [&]() -> int {                 // wave 0, lane 0 only; -1 = no tile left
    for (int k = 0; k < 8; ++k) {
      const int x = (my_xcd + k) & 7;
      const int cnt = (nvt - x + 7) >> 3;        // ids x, x+8, ... below nvt
      const int j = (int)(gridDim.x >> 3) + atomicAdd(p.tile_queue + x, 1);   // the first gridDim.x ids were dealt statically
      if (j < cnt) return x + 8 * j;
    }
    return -1;
  };
  int* slot = (int*)(smem + (PP_DIST % PP_NB) * PP_STAGE);      // idle until step 0 of the first tile issues step PP_DIST
  int v = blockIdx.x;
  int v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
  if (dyn) {
    if (tid == 0) *slot = pop_tile();
    __syncthreads();
    v_next = *slot;
    __syncthreads();
  }
  Desc cur = make_desc(v);
  bool has_next = v_next >= 0;
  Desc nxt = make_desc(has_next ? v_next : v);
#pragma unroll
  for (int h = 0; h < PP_DIST; ++h) issue_step(cur, h, h);       // host guarantees nhs >= PP_DIST
  wait_pieces(PP_DIST - 1);
  int b_rd = 0, b_wr = PP_DIST % PP_NB;
  unsigned long long t_cyc = 0, t_real = 0, s_cyc = 0, s_real = 0, s_nk = 0, t_first = 0;
  int tile_no = 0;

  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    jitter();
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();      // the stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);
    if (p.stamps) { t_cyc = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); if (!t_first) t_first = t_real; }
    ++tile_no;
    const bool step_stamp = p.stamps && tid == 0 && blockIdx.x == 9 && (tile_no == 4 || tile_no == 5) && nhs <= 100;
    for (int hs = 0; hs < nhs; ++hs) {
      if (step_stamp) p.stamps[4 * (size_t)gridDim.x + (tile_no - 4) * (nhs + 2) + hs] = __builtin_amdgcn_s_memtime();
      jitter();
      const char* rd = smem + b_rd * PP_STAGE;
      bf16x8 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag_h<B_KM, BN>(rd + A_BYTES, wc * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = load_frag_h<A_KM, BM>(rd, wr * 128 + i * 16, lane);
      jitter();
      const int tgt = hs + PP_DIST;
      if (tgt < nhs) {
        issue_step(cur, tgt, b_wr);
        wait_pieces(PP_DIST - 1);
      } else if (has_next) {
        issue_step(nxt, tgt - nhs, b_wr);
        wait_pieces(PP_DIST - 1);
      } else {
        wait_pieces(nhs - 2 - hs > 0 ? nhs - 2 - hs : 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      jitter();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      jitter();
      __builtin_amdgcn_s_setprio(1);
      if constexpr (F8 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(b[j], a[i], acc[i][j]);
      } else {
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = mfma_f8<F8>(__builtin_bit_cast(i64x2, b[j])[hk], __builtin_bit_cast(i64x2, a[i])[hk], acc[i][j]);
      }
      __builtin_amdgcn_s_setprio(0);
      jitter();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      b_wr = b_wr + 1 == PP_NB ? 0 : b_wr + 1;
      b_rd = b_rd + 1 == PP_NB ? 0 : b_rd + 1;
    }
    jitter();
    if (!late) __builtin_amdgcn_s_barrier();     // both groups leave the tile together
    jitter();
    if (p.stamps) { s_cyc += __builtin_amdgcn_s_memtime() - t_cyc; s_real += __builtin_amdgcn_s_memrealtime() - t_real; s_nk += nhs / 2; }
    if (step_stamp) p.stamps[4 * (size_t)gridDim.x + (tile_no - 4) * (nhs + 2) + nhs] = __builtin_amdgcn_s_memtime();
    int popped = -1;
    if (dyn && has_next && tid == 0) popped = pop_tile();       // the tile after next; returns while the epilogue runs
    if constexpr (F8 != 0) {
      GemmParams pe = p;
      pe.alpha = *p.alpha_dev * *p.alpha_dev2;
      direct_epilogue<2, EPK>(pe, acc, lane, cur.m0 + wr * 128, cur.n0 + wc * 64);
    } else {
      direct_epilogue<2, EPK>(p, acc, lane, cur.m0 + wr * 128, cur.n0 + wc * 64);
    }
    jitter();
    if (step_stamp) p.stamps[4 * (size_t)gridDim.x + (tile_no - 4) * (nhs + 2) + nhs + 1] = __builtin_amdgcn_s_memtime();
    if (!has_next) break;
    cur = nxt;
    if (dyn) {
      int* bslot = (int*)(smem + b_wr * PP_STAGE);               // the stage buffer step 0 of the next tile will rewrite
      if (tid == 0) *bslot = popped;
      __syncthreads();
      v_next = *bslot;
      __syncthreads();                                           // every wave has read it before any LDS-DMA lands there
    } else {
      v += gridDim.x;
      v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
    }
    has_next = v_next >= 0;
    if (has_next) nxt = make_desc(v_next);
  }
  if (dyn && tid == 0) {          // the last workgroup to leave resets the queue for the next launch
    __threadfence();
    if (atomicAdd(p.tile_queue + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int k = 0; k < 9; ++k) atomicExch(p.tile_queue + k, 0);
    }
  }
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + 4 * (size_t)blockIdx.x;
    o[0] = s_cyc; o[1] = s_real; o[2] = s_nk; o[3] = t_first;
  }
}

constexpr int W4_PEND_ROWS = MDT_W4_PEND_ROWS;          // row tiles (of a wave's eight) whose outputs wait in registers
constexpr int W4_NPEND = 4 * W4_PEND_ROWS;              // ... = pending 16-byte vectors per lane, one leaves per step
constexpr int W4_NEXPL = W4_NPEND + 2;                  // explicit steps of a tile (compile-time vmcnt budget and pending index)
// operations a wave issues in step s of a tile (8 LDS-DMA pieces + the pending store); steps before the tile: 8
constexpr int w4_ops(int s) { return 8 + (s >= 0 && s < W4_NPEND ? 1 : 0); }
// what may be in flight at the start of step s while the pieces of step s + 1 (requested in step s - 3) must have landed:
// the operations of steps s - 2 and s - 1, and in a tile's first three steps the epilogue's direct stores issued in between
constexpr int w4_budget(int s) { return w4_ops(s - 2) + w4_ops(s - 1) + (s < 3 ? 32 - W4_NPEND : 0); }
template <class F, int... S>
__device__ __forceinline__ void w4_unroll(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

template <bool A_KM, bool B_KM, int EPK = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4p(GemmParams p) {
  constexpr int PP_DIST = 4, PP_NB = 5;
  constexpr int BM = 256, BN = 256, A_BYTES = BM * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nvt = p.tiles_m * p.tiles_n;
  const int nhs = (int)((p.K + 31) / 32);
  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;

  struct Desc { __amdgpu_buffer_rsrc_t rsA, rsB; int a_col0, b_col0; int64_t m0, n0; };
  auto make_desc = [&](int v) {
    const int q8 = nvt >> 3, r8 = nvt & 7, xcd = v & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
    int tm, tn;
    {
      const int G = p.group_n, per_group = p.tiles_m * G;
      const int gi = tile / per_group;
      const int full = p.tiles_n / G;
      if (gi < full) {
        const int r = tile - gi * per_group;
        tm = r / G;
        tn = gi * G + (r - tm * G);
      } else {
        const int gsz = p.tiles_n - full * G;
        const int r = tile - full * per_group;
        tm = r / gsz;
        tn = full * G + (r - tm * gsz);
      }
    }
    Desc d;
    d.m0 = (int64_t)tm * BM;
    d.n0 = (int64_t)tn * BN;
    const char* a_base;
    const char* b_base;
    int64_t a_bytes, b_bytes;
    if constexpr (!A_KM) { a_base = (const char*)p.A + d.m0 * lda_b; a_bytes = (p.M - d.m0) * lda_b; }
    else { a_base = (const char*)p.A; a_bytes = p.K * lda_b; }
    if constexpr (!B_KM) { b_base = (const char*)p.B + d.n0 * ldb_b; b_bytes = (p.N - d.n0) * ldb_b; }
    else { b_base = (const char*)p.B; b_bytes = p.K * ldb_b; }
    const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
    const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);
    d.a_col0 = A_KM ? (int)d.m0 : 0;
    d.b_col0 = B_KM ? (int)d.n0 : 0;
    return d;
  };
  // The wave's 8 LDS-DMA pieces of a step: pieces 0-3 of A, 4-7 of B (piece index inside the operand = wave + 4 i).
  // Their per-lane offsets do not depend on the step or the tile: the step (and, for a k-major operand, the tile's
  // column origin) rides in the instruction's SCALAR offset, so a piece costs no vector arithmetic.
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave + 4 * i;
    if constexpr (!A_KM) {
      const int row = piece * 16 + (lane >> 2);
      voffA[i] = (unsigned)(row * lda_b + (((lane & 3) ^ swz_h(row)) * 16));
    } else {
      const int k = piece * 2 + (lane >> 5), c16 = lane & 31;
      voffA[i] = (unsigned)(k * lda_b + ((((c16 >> 1) ^ swz_km(k)) * 16 + (c16 & 1) * 8) * 2));
    }
    if constexpr (!B_KM) {
      const int row = piece * 16 + (lane >> 2);
      voffB[i] = (unsigned)(row * ldb_b + (((lane & 3) ^ swz_h(row)) * 16));
    } else {
      const int k = piece * 2 + (lane >> 5), c16 = lane & 31;
      voffB[i] = (unsigned)(k * ldb_b + ((((c16 >> 1) ^ swz_km(k)) * 16 + (c16 & 1) * 8) * 2));
    }
  }
  auto soff_a = [&](const Desc& d, int hs) -> int { return A_KM ? (int)((int64_t)hs * 32 * lda_b) + d.a_col0 * 2 : hs * 64; };
  auto soff_b = [&](const Desc& d, int hs) -> int { return B_KM ? (int)((int64_t)hs * 32 * ldb_b) + d.b_col0 * 2 : hs * 64; };
  auto issue_piece = [&](const Desc& d, int sa, int sb, int buf, int q) __attribute__((always_inline)) {
    char* st = smem + buf * PP_STAGE;
    const int piece = wave + 4 * (q & 3);
    if (q < 4) w4_dma<MDT_GEMM_A_AUX>(d.rsA, st + piece * 1024, voffA[q & 3], sa);
    else w4_dma(d.rsB, st + A_BYTES + piece * 1024, voffB[q & 3], sb);
  };
  auto issue_step = [&](const Desc& d, int hs, int buf) {
    const int sa = soff_a(d, hs), sb = soff_b(d, hs);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(d, sa, sb, buf, q);
  };

  if (p.epilogue & (1 << 26)) {   // diagnostic (MDT_GEMM_DIAG=64): skew whole XCDs against each other (their workgroups stay in phase)
    const int steps = (blockIdx.x & 7) * ((nhs + 12) / 8);
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(28);
  }
  if (p.epilogue & (1 << 28)) {   // diagnostic (MDT_GEMM_DIAG=256): inside an XCD, skew the groups of workgroups that share an A row panel
    const int groups = 32 / p.group_n;                              // meant for group_n = tiles_n dividing 32 (N = 1024: 8 groups of 4)
    const int steps = (((int)blockIdx.x >> 3) / p.group_n) * ((nhs + 12) / (groups > 0 ? groups : 1));
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(28);
  }
  int v = blockIdx.x;
  int v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
  Desc cur = make_desc(v);
  bool has_next = v_next >= 0;
#ifdef MDT_W4_STAMPS
  // diagnostic build (-DMDT_W4_STAMPS, MDT_GEMM_STAMP=1): shader-clock stamps of every step of tiles 4 and 5 of workgroup 9,
  // of the loop's end, the epilogue's end and the end of the fragment prime — same report as the 8-wave kernel's
  const unsigned long long w4_t0c = __builtin_amdgcn_s_memtime(), w4_t0r = __builtin_amdgcn_s_memrealtime();
  int tile_no = 0, n_tiles_done = 0;
#define W4_STAMP(slot_)                                                                                         \
  if (p.stamps && tid == 0 && blockIdx.x == 9 && (tile_no == 4 || tile_no == 5) && nhs <= 100)                \
    p.stamps[4 * (size_t)gridDim.x + (tile_no - 4) * (nhs + 3) + (slot_)] = __builtin_amdgcn_s_memtime()
#else
#define W4_STAMP(slot_) (void)0
#endif
  // after the last tile the ring keeps turning on a descriptor of zero records (every load reads as 0, nobody reads the
  // stage): the loop needs no "nothing left to request" case and the vmcnt arithmetic is the same in every step
  auto null_desc = [&](Desc d) {
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0, 0x00020000);
    return d;
  };
  Desc nxt = has_next ? make_desc(v_next) : null_desc(cur);
#pragma unroll
  for (int h = 0; h < PP_DIST; ++h) issue_step(cur, h, h);        // host guarantees nhs >= PP_DIST
  wait_vm<(PP_DIST - 1) * 8>();
  __builtin_amdgcn_s_barrier();
  bf16x8 fa[2][8], fb[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[0][i] = w4_frag<A_KM, BM>(smem, wr * 128 + i * 16, lane);
    fb[0][i] = w4_frag<B_KM, BN>(smem + A_BYTES, wc * 128 + i * 16, lane);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int b_next = 1, b_wr = PP_DIST % PP_NB;
  f32x4 acc[8][8];
  constexpr int W4_DS = 32 - W4_NPEND;             // direct stores of an epilogue (the other row tiles x 4 column pairs): a constant, see EpiBuf

  // Pending outputs: row tiles 4-7 of a finished tile stay packed in registers (16 vectors of 16 bytes per lane) and
  // leave one per step during steps 0-15 of the next tile, through a buffer descriptor based at the tile's origin (rows past
  // M fall outside it and are dropped).  The other half is stored at the tile's end as before: the burst is half as long.
  constexpr bool PEND = EPK >= 0;                  // the specialised epilogues (the runtime-flag kernel keeps the plain form)
  bf16x8 pend[W4_NPEND];
#pragma unroll
  for (int i = 0; i < W4_NPEND; ++i) pend[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0, 0x00020000);
  const int c_lane = lane & 15, g_lane = lane >> 4;
  const unsigned voffP = (unsigned)((wr * 128 + c_lane) * (p.ldc * 2) + (wc * 128 + 16 * (g_lane & 1) + 8 * (g_lane >> 1)) * 2);
  const int ldc16 = (int)(p.ldc * 2 * 16);         // bytes between row tiles
  if constexpr (PEND) {
    // Steps 0-2 of a tile count the epilogue's W4_DS direct stores among the operations that may still be in flight (they are
    // younger than the LDS-DMA pieces those steps wait for; without counting them the steps would wait for the stores'
    // acknowledgements instead).  A workgroup's first tile has no epilogue before it: W4_DS dropped stores stand in.
    // They MUST be asm statements.  As builtin stores (rounds 2-3) the sixteen identical stores — same data, same address, same
    // descriptor — were merged into ONE by the compiler, so steps 0-2 of every workgroup's FIRST tile waited with a budget that
    // was 15 too large: vmcnt(32) against 25 operations in flight is no wait at all, and the first fragments were read from a
    // stage that need not have landed.  On an idle card the cold loads are in LDS long before; with a second tenant on the card
    // (or a slow box) they are not: one garbage 256 x 256 tile in ~1000 launches — round 3's "box-dependent" failures, round 4's
    // failing exchange self-check (tools/finite_hunt.py found the tile, tools/probes/vmcnt_order_probe.hip cleared the counter
    // itself: vmcnt retires strictly in issue order, dropped requests included).  tests/test_isa_hazards_cpu.py counts them.
#if defined(__HIP_DEVICE_COMPILE__)
    const i32x4 zero4 = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < W4_DS; ++i) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(zero4), "v"(voffP), "s"(rsP) : "memory");
#endif
  }
  auto desc_c = [&](const Desc& d) {
    const int64_t bytes = (p.M - d.m0) * p.ldc * 2 - d.n0 * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.C + d.m0 * p.ldc + d.n0), 0,
                                             (unsigned)(bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : bytes), 0x00020000);
  };

  // one 32-k step on fragment set `cur`: FIRST starts the accumulators from zero; NW = vector-memory operations that may
  // still be in flight when the step begins; ST >= 0: pending vectors ST and ST + 1 leave in this step
  // (the host pass of hipcc parses kernel bodies too and knows no "a" / "v" register classes: it gets an empty statement)
#if defined(__HIP_DEVICE_COMPILE__)
#define W4_MF(i_, j_)                                                                                             \
  if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i_][j_]) : "v"(fb[cs][j_]), "v"(fa[cs][i_]) : "memory"); \
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fb[cs][j_]), "v"(fa[cs][i_]) : "memory")
#else
#define W4_MF(i_, j_) (void)first
#endif
  auto step = [&](auto cs_c, auto first_c, auto nw_c, auto st_c, auto ld_c, const Desc& d_issue, int hs_issue) __attribute__((always_inline)) {
    constexpr int cs = decltype(cs_c)::value, ns = cs ^ 1;
    constexpr bool first = decltype(first_c)::value;
    constexpr int NW = decltype(nw_c)::value, ST = decltype(st_c)::value;
    constexpr bool LD = decltype(ld_c)::value;       // false in a tile's last step: the next tile's first fragments are read after the epilogue
    const int sa = soff_a(d_issue, hs_issue), sb = soff_b(d_issue, hs_issue);
    wait_vm<NW>();                                // own pieces of the next step have landed
    __builtin_amdgcn_s_barrier();                 // ... and everybody's; the stage of the previous step is free
    const char* tn = smem + b_next * PP_STAGE;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      W4_MF(b, 0); W4_MF(b, 1);
      if constexpr (LD) fa[ns][b] = w4_frag<A_KM, BM>(tn, wr * 128 + b * 16, lane);
      W4_MF(b, 2); W4_MF(b, 3);
      if constexpr (LD) fb[ns][b] = w4_frag<B_KM, BN>(tn + A_BYTES, wc * 128 + b * 16, lane);
      W4_MF(b, 4); W4_MF(b, 5);
      issue_piece(d_issue, sa, sb, b_wr, b);
      if constexpr (ST >= 0) {
        if (b == 4) {                                // pending vector ST -> row tile (8 - W4_PEND_ROWS) + ST / 4, column pair ST % 4
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pend[ST]), rsP, voffP,
                                                 (8 - W4_PEND_ROWS + (ST >> 2)) * ldc16 + (ST & 3) * 64, 0);
        }
      }
      W4_MF(b, 6); W4_MF(b, 7);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    b_next = b_next + 1 == PP_NB ? 0 : b_next + 1;
    b_wr = b_wr + 1 == PP_NB ? 0 : b_wr + 1;
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  // which stage a step requests: the same tile's step hs + 4, or — in the last four steps — the next tile's first steps
  auto run_step = [&](auto cs_c, auto first_c, auto nw_c, auto st_c, auto ld_c, int hs) __attribute__((always_inline)) {
    const int tgt = hs + PP_DIST;
    const bool same = tgt < nhs;
    Desc d = nxt;
    if (same) d = cur;
    W4_STAMP(hs);
    step(cs_c, first_c, nw_c, st_c, ld_c, d, same ? tgt : tgt - nhs);
  };
#define W4_N(n_) std::integral_constant<int, n_> {}
  for (;;) {
    // K is a multiple of 64: a tile has an even number of 32-k steps, so every tile starts on fragment set 0.
    // vmcnt budgets: a step issues 8 pieces, and 1 store while pending vectors leave (steps 0-15 after a tile of this
    // workgroup; in-call A/B: one vector per step over sixteen steps beats two per step over eight by 1-3 % on the K = 768
    // launches): pieces of the next step were requested three steps ago, so what may be in flight is what the two steps in
    // between issued — 16, 17 or 18 operations (+ the epilogue's W4_DS direct stores in a tile's first three steps).
    // The first tile of a workgroup walks the same eighteen steps: its sixteen "pending" vectors are zeros sent through a descriptor
    // of zero records (dropped by the bounds check, counted by vmcnt like any store).  A separate first-tile path cost more than
    // its code: the compiler parked registers in scratch around it and, where the two paths met, drained the whole prefetch ring
    // (s_waitcnt vmcnt(0)) once per TILE.
    if constexpr (PEND) {
      w4_unroll([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        run_step(std::integral_constant<int, S & 1>{}, std::integral_constant<bool, S == 0>{}, W4_N(w4_budget(S)),
                 W4_N(S < W4_NPEND ? S : -1), T_{}, S);
      }, std::make_integer_sequence<int, W4_NEXPL>{});
    } else {
      for (int hs = 0; hs < W4_NEXPL; hs += 2) {
        if (hs == 0) run_step(C0{}, T_{}, W4_N(16), W4_N(-1), T_{}, 0);
        else run_step(C0{}, F_{}, W4_N(16), W4_N(-1), T_{}, hs);
        run_step(C1{}, F_{}, W4_N(16), W4_N(-1), T_{}, hs + 1);
      }
    }
    for (int hs = W4_NEXPL; hs < nhs - 2; hs += 2) {    // host: nhs >= W4_NEXPL + 2
      run_step(C0{}, F_{}, W4_N(16), W4_N(-1), T_{}, hs);
      run_step(C1{}, F_{}, W4_N(16), W4_N(-1), T_{}, hs + 1);
    }
    run_step(C0{}, F_{}, W4_N(16), W4_N(-1), T_{}, nhs - 2);
    const int b_prime = b_next;                   // the stage holding the next tile's first step
    EpiPre<4> epf;
    EpiBuf eb;
    if constexpr (PEND) {
      auto desc_of = [&](const void* base, int64_t ld) {
        const int64_t bytes = (p.M - cur.m0) * ld * 2 - cur.n0 * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)base + cur.m0 * ld + cur.n0), 0,
                                                 (unsigned)(bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : bytes), 0x00020000);
      };
      auto voff_of = [&](int64_t ld) { return (unsigned)((wr * 128 + c_lane) * (ld * 2) + (wc * 128 + 16 * (g_lane & 1) + 8 * (g_lane >> 1)) * 2); };
      eb.rsC = desc_of(p.C, p.ldc); eb.voffC = voffP; eb.ldc16 = ldc16;
      eb.rsX = eb.rsC; eb.voffX = voffP; eb.ldx16 = ldc16;
      eb.rsR = eb.rsC; eb.voffR = voffP; eb.ldr16 = ldc16;
      if constexpr ((EPK & MDT_EPI_AUX_GRAD) != 0) { eb.rsX = desc_of(p.aux, p.ldaux); eb.voffX = voff_of(p.ldaux); eb.ldx16 = (int)(p.ldaux * 32); }
      constexpr int PK = epi_pre_kind<EPK>();
      if constexpr (PK == 1) { eb.rsR = desc_of(p.aux, p.ldaux); eb.voffR = voff_of(p.ldaux); eb.ldr16 = (int)(p.ldaux * 32); }
      if constexpr (PK == 2) { eb.rsR = desc_of(p.residual, p.ldr); eb.voffR = voff_of(p.ldr); eb.ldr16 = (int)(p.ldr * 32); }
      epi_prefetch<4, EPK, true>(p, lane, cur.m0 + wr * 128, cur.n0 + wc * 128, epf, &eb);
    }
    run_step(C1{}, F_{}, W4_N(16), W4_N(-1), F_{}, nhs - 1);
    W4_STAMP(nhs);
#if defined(__HIP_DEVICE_COMPILE__)
    // The MFMAs are asm statements: the compiler does not know that their results need wait states before a VALU may read
    // them, and it is free to hoist an accumulator read of the epilogue up to right behind the last MFMA that wrote it
    // (seen: v_accvgpr_read three instructions after the MFMA — a stale value in one element per lane of one row group).
    // So: the wait states by hand, and every accumulator re-defined behind them, which pins all epilogue reads below.
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
#endif
    direct_epilogue<4, EPK, PEND, PEND>(p, acc, lane, cur.m0 + wr * 128, cur.n0 + wc * 128, pend, &epf, &eb);
    W4_STAMP(nhs + 1);
    {                                             // fragments of the next tile's first step (its stage landed a step ago)
      const char* t0 = smem + b_prime * PP_STAGE;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        fa[0][i] = w4_frag<A_KM, BM>(t0, wr * 128 + i * 16, lane);
        fb[0][i] = w4_frag<B_KM, BN>(t0 + A_BYTES, wc * 128 + i * 16, lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    W4_STAMP(nhs + 2);
#ifdef MDT_W4_STAMPS
    ++tile_no; ++n_tiles_done;
#endif
    if constexpr (PEND) {
      if (has_next) rsP = eb.rsC;
      else {                                      // nothing follows: the pending half leaves now
        const __amdgpu_buffer_rsrc_t rl = eb.rsC;
#pragma unroll
        for (int idx = 0; idx < W4_NPEND; ++idx)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pend[idx]), rl, voffP,
                                                 (8 - W4_PEND_ROWS + (idx >> 2)) * ldc16 + (idx & 3) * 64, 0);
      }
    }
    if (!has_next) break;
    cur = nxt;
    v += gridDim.x;
    v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
    has_next = v_next >= 0;
    nxt = has_next ? make_desc(v_next) : null_desc(cur);
  }
#ifdef MDT_W4_STAMPS
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + 4 * (size_t)blockIdx.x;
    o[0] = __builtin_amdgcn_s_memtime() - w4_t0c;
    o[1] = __builtin_amdgcn_s_memrealtime() - w4_t0r;
    o[2] = (unsigned long long)n_tiles_done * (nhs / 2);
    o[3] = w4_t0r;
  }
#endif
#undef W4_STAMP
#undef W4_N
#undef W4_MF
}

// ------------------------------------------------------------------ helpers
template <typename TIn, int BLOCK>
__global__ __launch_bounds__(BLOCK) void colsum_kernel(int64_t M, int64_t N, const TIn* X, int64_t ldx, float* out,
                                                        int rows_per_block, const int32_t* row_weight) {
  // block = 256 threads: 64 columns x 4 row-lanes; grid.x = column groups, grid.y = row chunks
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float s = 0.f;
  if (c < N)
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float w = row_weight ? (float)row_weight[r] : 1.f;
      s += w * to_f32(X[r * ldx + c]);
    }
  __shared__ float red[4][64];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// 16-byte-vectorised column sum: lane = VN consecutive columns, the 4 waves of a block interleave rows
template <typename TIn, int VN>
__global__ __launch_bounds__(256) void colsum_vec_kernel(int64_t M, int64_t N, const TIn* X, int64_t ldx, float* out,
                                                         int rows_per_block, const int32_t* row_weight) {
  typedef __attribute__((ext_vector_type(VN))) TIn vec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * 64 + lane) * VN;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float acc[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) acc[e] = 0.f;
  if (c < N) {
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      const vec v = *(const vec*)(X + r * ldx + c);
      const float w = row_weight ? (float)row_weight[r] : 1.f;
#pragma unroll
      for (int e = 0; e < VN; ++e) acc[e] += w * to_f32((TIn)v[e]);
    }
  }
  __shared__ float red[4][64 * VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) red[wave][lane * VN + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * VN; i += 256) {
    const int64_t col = (int64_t)blockIdx.x * 64 * VN + i;
    if (col < N) atomicAdd(out + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(int64_t rows, int D, const T* x, int64_t ldx, T* y, int64_t ldy, DropCfg d) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4)
    for (int c = lane; c < D; c += 64)
      y[r * ldy + c] = from_f32<T>(to_f32(x[r * ldx + c]) * drop_scale(d, (uint64_t)r * D + c));
}
// 16-byte vector variant (D, strides multiples of VN; D even so every vector starts on an even counter)
template <typename T, int VN>
__global__ __launch_bounds__(256) void dropout_vec_kernel(int64_t rows, int D, const T* x, int64_t ldx, T* y, int64_t ldy, DropCfg d) {
  typedef __attribute__((ext_vector_type(VN))) T vec;
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4)
    for (int c = lane * VN; c < D; c += 64 * VN) {
      const vec v = *(const vec*)(x + r * ldx + c);
      vec o;
#pragma unroll
      for (int e = 0; e < VN; e += 2) {
        float s0, s1;
        drop_scale2(d, (uint64_t)r * D + c + e, s0, s1);
        o[e] = from_f32<T>(to_f32((T)v[e]) * s0);
        o[e + 1] = from_f32<T>(to_f32((T)v[e + 1]) * s1);
      }
      *(vec*)(y + r * ldy + c) = o;
    }
}
__global__ void dropout_mask_kernel(int64_t n, DropCfg d, uint8_t* mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    mask[i] = drop_scale(d, (uint64_t)i) != 0.f;
}

template <typename TS, typename TD>
__global__ void cast_kernel(int64_t n, const TS* s, TD* d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = from_f32<TD>(to_f32(s[i]));
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(int64_t rows, int64_t cols, const TS* s, int64_t lds_, TD* d,
                                                        int64_t ldd) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = to_f32(s[(r0 + i) * lds_ + c0 + tx]);
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) d[(c0 + i) * ldd + r0 + tx] = from_f32<TD>(tile[tx][i]);
}

template <typename TIn, typename TOut>
static int launch_generic(hipStream_t st, const GemmParams& p, int ta, int tb) {
  dim3 grid((unsigned)((p.N + 63) / 64), (unsigned)((p.M + 63) / 64), (unsigned)p.split_k);
  if (!ta && !tb) hipLaunchKernelGGL((gemm_generic_kernel<TIn, TOut, false, false>), grid, 256, 0, st, p);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_generic_kernel<TIn, TOut, false, true>), grid, 256, 0, st, p);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_generic_kernel<TIn, TOut, true, false>), grid, 256, 0, st, p);
  else hipLaunchKernelGGL((gemm_generic_kernel<TIn, TOut, true, true>), grid, 256, 0, st, p);
  return check_launch("gemm_generic");
}

template <typename TOut>
static int launch_tile128(hipStream_t st, const GemmParams& p, int ta, int tb) {
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)p.split_k);
  const size_t lds = 4 * T_TILE_BYTES;
  if (!ta && !tb) hipLaunchKernelGGL((gemm_bf16_tile128<TOut, false, false>), grid, 256, lds, st, p);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_bf16_tile128<TOut, false, true>), grid, 256, lds, st, p);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_bf16_tile128<TOut, true, false>), grid, 256, lds, st, p);
  else hipLaunchKernelGGL((gemm_bf16_tile128<TOut, true, true>), grid, 256, lds, st, p);
  return check_launch("gemm_bf16_tile128");
}

template <typename TOut, int BN, int WM, int WN, int NSTAGE>
static int launch_tile256(hipStream_t st, const GemmParams& p, int ta, int tb) {
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)p.split_k);
  const size_t lds = (size_t)NSTAGE * (256 + BN) * 128;
#define L256(A_, B_)                                                                                         \
  {                                                                                                          \
    auto kern = gemm_bf16_tile256<TOut, A_, B_, BN, WM, WN, NSTAGE>;                                         \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_bf16_tile256: cannot reserve %zu bytes of LDS", lds);                                \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 512, lds, st, p);                                                         \
  }
  if (!ta && !tb) L256(false, false)
  else if (!ta && tb) L256(false, true)
  else if (ta && !tb) L256(true, false)
  else L256(true, true)
#undef L256
  return check_launch("gemm_bf16_tile256");
}

// MDT_GEMM_STAMP=1: every ping-pong launch is followed by a device sync and one stderr line with the median
// in-kernel clock and shader cycles per 64-deep K-tile of its main loop (tools/kbench.py; never in production).
static void report_stamps(unsigned long long* dev, size_t nwg, const GemmParams& p) {
  std::vector<unsigned long long> h(nwg * 4);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), dev, nwg * 32, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
  std::vector<double> cyc, clk;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < nwg; ++i) {
    if (h[4 * i + 2] == 0 || h[4 * i + 1] == 0) continue;
    cyc.push_back((double)h[4 * i] / (double)h[4 * i + 2]);
    clk.push_back((double)h[4 * i] / (double)h[4 * i + 1] * 100.0);
    t0 = std::min(t0, h[4 * i + 3]);
    t1 = std::max(t1, h[4 * i + 3] + h[4 * i + 1]);
  }
  if (cyc.empty()) return;
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  fprintf(stderr, "[mdt gemm stamp] M=%lld N=%lld K=%lld wgs=%zu  cycles/k-tile median %.0f (p10 %.0f p90 %.0f)  clock median %.0f MHz  span %.1f us\n",
          (long long)p.M, (long long)p.N, (long long)p.K, nwg, cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10],
          clk[clk.size() / 2], (double)(t1 - t0) / 100.0);
  // step stamps of one tile of one workgroup (persistent kernel): shader cycles between consecutive 32-k steps, then
  // loop end -> epilogue end -> first step of the next tile
  std::vector<unsigned long long> st(256);
  if (hipMemcpy(st.data(), dev + nwg * 4, 256 * 8, hipMemcpyDeviceToHost) == hipSuccess && st[0]) {
    fprintf(stderr, "[mdt gemm steps]");
    for (int i = 1; i < 256 && st[i]; ++i) fprintf(stderr, " %llu", st[i] - st[i - 1]);
    fprintf(stderr, "\n");
  } else (void)hipGetLastError();
}

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    n &= ~7;                      // the XCD-aware tile order needs a grid that is a multiple of the 8 XCDs
    if (n < 8) n = 8;
  }
  return n;
}

// dynamic tile queue: caller-owned device memory (mdt_gemm_set_tile_queue), 16 ints per set
static int* g_tile_queues = nullptr;
static int g_tile_queue_sets = 0;

template <typename TOut>
static int launch_pp256(hipStream_t st, const GemmParams& p_in, int ta, int tb) {
  GemmParams p = p_in;
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)p.split_k);
  const Switches& sw = switches();
  const int dist = sw.gemm_pp_dist;
  const int nhs_total = 2 * (int)((p.K + T_BK - 1) / T_BK);
  // The persistent walk pays where a tile is short against its launch / first-fetch / drain (K = 768: 24 steps) and needs at
  // least PP_DIST steps per tile.  Launches with fewer than 16 steps per tile (K < 512) — none in a training step — take one
  // workgroup per tile: a tile boundary then never falls inside the prefetch window of the previous one (round 3 saw an
  // intermittent wrong result on ONE box at K = 256 with two tiles per workgroup that no other device and no jittered stress
  // run reproduces, DESIGN.md "pp256p"; MDT_GEMM_PERSIST=2 — tools/gemm_stress.py — keeps the short-K walk reachable).
  const int min_steps = sw.gemm_persist >= 2 ? 4 : 16;
  const bool persist = sw.gemm_persist != 0 && sizeof(TOut) == 2 && p.split_k == 1 && nhs_total >= min_steps && (int)grid.x > num_cus();
  if (persist) grid.x = (unsigned)num_cus();
  p.tile_queue = nullptr;
  if (persist && nhs_total >= W4_NEXPL + 2) {
    // MDT_GEMM_W4: 0 off; 1 every persistent launch; 2 the launches it is measured faster on (k-contiguous operands, light
    // epilogues: plain, bias, residual, bias + dropout + residual, saved derivative + column sums — not the GELU form, not k-major operands)
    const int w4 = sw.gemm_w4;
    const int e_ = p.epilogue & ((1 << 22) - 1);   // start-skew diagnostics (bits 22+) keep the compile-time epilogues
    const bool light = e_ == 0 || e_ == MDT_EPI_BIAS || e_ == MDT_EPI_RESIDUAL || e_ == (MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT) ||
                       e_ == (MDT_EPI_MULAUX | MDT_EPI_COLSUM);
    const bool use_w4 = (w4 == 1 || (w4 == 2 && !ta && light && !sw.gemm_no_spec)) && !(sw.gemm_dynamic && g_tile_queues) &&
                        !(p.epilogue & (1 << 24));      // the jittered stress build exists for the 8-wave kernel
    if (use_w4) {
#ifdef MDT_W4_STAMPS
      if (sw.gemm_stamp) {
        if (hipMalloc(&p.stamps, (size_t)grid.x * 32 + 256 * 8) != hipSuccess) { (void)hipGetLastError(); p.stamps = nullptr; }
        else (void)hipMemsetAsync(p.stamps, 0, (size_t)grid.x * 32 + 256 * 8, st);
      }
#endif
      p.group_n = p.tiles_n;
      {
        const double b_panel = 256.0 * (double)p.k_chunk * 2.0;
        if (p.tiles_n % 2 == 0 && b_panel * p.tiles_n > 3.5e6 && b_panel * (p.tiles_n / 2) <= 2.5e6 && (double)p.tiles_m * p.tiles_n >= 4.0 * num_cus())
          p.group_n = p.tiles_n / 2;
      }
      if (sw.gemm_group >= 1) p.group_n = sw.gemm_group < p.tiles_n ? sw.gemm_group : p.tiles_n;   // MDT_GEMM_GROUP: A/B runs
      const size_t lds = (size_t)5 * PP_STAGE;
#define LW4(A_, B_, E_)                                                                                      \
  {                                                                                                          \
    auto kern = gemm_bf16_w4p<A_, B_, E_>;                                                                   \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_bf16_w4p: cannot reserve %zu bytes of LDS", lds);                                    \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 256, lds, st, p);                                                         \
  }
      constexpr int E_BIAS = MDT_EPI_BIAS, E_DENSE = MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT,
                    E_FC1 = MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_AUX_GRAD, E_RES = MDT_EPI_RESIDUAL,
                    E_DFC2 = MDT_EPI_MULAUX | MDT_EPI_COLSUM;
      const int e = sw.gemm_no_spec ? -2 : (p.epilogue & ((1 << 22) - 1));
      if (!ta && !tb) {
        if (e == E_BIAS) LW4(false, false, E_BIAS)
        else if (e == E_DENSE) LW4(false, false, E_DENSE)
        else if (e == E_FC1 && p.aux) LW4(false, false, E_FC1)
        else if (e == 0) LW4(false, false, 0)
        else if (e == E_RES) LW4(false, false, E_RES)
        else if (e == E_DFC2) LW4(false, false, E_DFC2)
        else LW4(false, false, -1)
      } else if (!ta && tb) {
        if (e == 0) LW4(false, true, 0)
        else if (e == E_RES) LW4(false, true, E_RES)
        else if (e == E_DFC2) LW4(false, true, E_DFC2)
        else LW4(false, true, -1)
      } else if (ta && !tb) LW4(true, false, -1)
      else LW4(true, true, -1)
#undef LW4
#ifdef MDT_W4_STAMPS
      if (p.stamps) { report_stamps(p.stamps, grid.x, p); (void)hipFree(p.stamps); }
#endif
      return check_launch("gemm_bf16_w4p");
    }
  }
  if (persist) {
    // MDT_GEMM_DYNAMIC=1: dynamic tile queue instead of the static round-robin walk (in-call A/B on an otherwise idle
    // chip: static is 1.5 % faster — two more barriers per tile, and raided tiles leave their XCD's L2; the queue is
    // there for a node where other kernels — RCCL — hold compute units for long, see ddp.py).  The queue sets live in a
    // CALLER-OWNED, zero-initialised device buffer (mdt_gemm_set_tile_queue; the library allocates nothing): they are
    // used in turn, each put back to zero by the last workgroup of the launch that used it (launches of the two branch
    // streams run concurrently, at most a few dozen launches apart in issue order).  Without a registered buffer the
    // static walk is used.
    static unsigned turn = 0;
    if (sw.gemm_dynamic && g_tile_queues) p.tile_queue = g_tile_queues + 16 * (turn++ % (unsigned)g_tile_queue_sets);
  }
  {
    // Tile order.  Row-major (group_n = tiles_n) unless B is too wide for an XCD's 4-MiB L2 and splits evenly in
    // two halves that do fit: then XCDs 0-3 sweep the left half of the columns and XCDs 4-7 the right half, over the
    // same rows at the same time — each XCD keeps its half of B resident and the second reader of an A panel finds
    // it in the Infinity Cache.  In-call A/B at N = 3072, K = 768: +3-4 % (fabric reads 2.7 -> ~1.3 GB).  Finer
    // groups lose (A then streams from HBM once per group); odd splits (N = 2304) lose.  MDT_GEMM_GROUP=n forces a
    // width, MDT_GEMM_GROUP=0 asks the fabric-read model below.
    p.group_n = p.tiles_n;
    {
      const double b_panel = 256.0 * (double)p.k_chunk * 2.0;
      if (p.split_k == 1 && p.tiles_n % 2 == 0 && b_panel * p.tiles_n > 3.5e6 && b_panel * (p.tiles_n / 2) <= 2.5e6 &&
          (double)p.tiles_m * p.tiles_n >= 4.0 * num_cus())
        p.group_n = p.tiles_n / 2;
      // split-K (weight gradients): an XCD takes a contiguous run of a slab's tiles, so the operand whose panels are
      // the run's slow index is fetched once and the other once per XCD touching the slab — let the larger operand
      // (more panels) be the slow index: column-major when dW is wider than tall (fc2: 1.66 -> 1.25 x operand bytes)
      if (p.split_k > 1 && p.tiles_n > p.tiles_m) p.group_n = 1;
    }
    if (sw.gemm_group >= 0) {
      int v = sw.gemm_group;
      if (v == 0) {
        const double a_bytes = (double)p.M * (double)p.k_chunk * 2.0, b_bytes = (double)p.N * (double)p.k_chunk * 2.0;
        const double rounds = (double)p.tiles_m * p.tiles_n / 256.0;
        double best_cost = 1e300;
        for (int G = 1; G <= p.tiles_n; ++G) {
          const int ngroups = (p.tiles_n + G - 1) / G;
          const bool fits = (double)G * 256.0 * (double)p.k_chunk * 2.0 <= 1.6e6;
          const double cost = a_bytes * ngroups + (fits ? b_bytes * 8.0 : b_bytes * 8.0 * (rounds > 1.0 ? rounds : 1.0));
          if (cost < best_cost * 0.999) { best_cost = cost; v = G; }
        }
      }
      if (v >= 1) p.group_n = v < p.tiles_n ? v : p.tiles_n;
    }
  }
  // fp32-accumulating launches (the split-K weight gradients) with at least four 64-deep K-tiles per slab: the 4-wave kernel
  // of gemm_wgrad.hip (MDT_GEMM_W4=0 keeps the 8-wave ping-pong kernel below)
  if (!persist && sizeof(TOut) == 4 && sw.gemm_w4 != 0 && !sw.gemm_stamp && (p.epilogue & MDT_EPI_ATOMIC) &&
      !(p.epilogue & ((1 << 20) - 1) & ~(MDT_EPI_ATOMIC | MDT_EPI_ASUM)) && p.k_chunk >= 256 && (p.K - (int64_t)(p.split_k - 1) * p.k_chunk) >= 256)
    return launch_w4s(st, p, ta, tb);
  const bool stamp = sw.gemm_stamp;
  const size_t nwg = (size_t)grid.x * grid.z;
  if (stamp) {
    if (hipMalloc(&p.stamps, nwg * 32 + 256 * 8) != hipSuccess) { (void)hipGetLastError(); p.stamps = nullptr; }
    else (void)hipMemsetAsync(p.stamps, 0, nwg * 32 + 256 * 8, st);
  }
  // prefetch distance in 32-k steps.  In-call A/B at the C2 shapes: 4 (five stages = all of LDS) beats 2 and 3 by
  // 1-3 % per GEMM, 0.6 % on the whole step — the fill is throughput- rather than latency-bound
  if (persist) {
    const size_t lds = (size_t)pp_nb(4) * PP_STAGE;
#define LPS(A_, B_, E_)                                                                                      \
  {                                                                                                          \
    auto kern = gemm_bf16_pp256p<A_, B_, 4, E_>;                                                             \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_bf16_pp256p: cannot reserve %zu bytes of LDS", lds);                                 \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 512, lds, st, p);                                                         \
  }
    // the epilogue flag sets of a training step get instantiations with the flags folded in (see direct_epilogue);
    // anything else — eval-mode combinations, diagnostics — runs the runtime-flag kernel
    constexpr int E_BIAS = MDT_EPI_BIAS, E_DENSE = MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT,
                  E_FC1 = MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_AUX_GRAD /* HF blocks have no activation dropout */, E_RES = MDT_EPI_RESIDUAL,
                  E_DFC2 = MDT_EPI_MULAUX | MDT_EPI_COLSUM;
    const int e = sw.gemm_no_spec ? -2 : (p.epilogue & ((1 << 22) - 1));
    if (p.epilogue & (1 << 24)) {          // MDT_GEMM_DIAG=16: the jittered stress build of the same source (tools/gemm_stress.py)
#define LPJ(A_, B_, E_)                                                                                      \
  {                                                                                                          \
    auto kern = gemm_bf16_pp256p<A_, B_, 4, E_, 0, true>;                                                    \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                               \
      set_error("gemm_bf16_pp256p (jitter): cannot reserve %zu bytes of LDS", lds);                          \
      return MDT_ERR_LAUNCH;                                                                                 \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 512, lds, st, p);                                                         \
  }
      if (!ta && !tb && e == E_FC1 && p.aux) LPJ(false, false, E_FC1)
      else if (!ta && !tb && e == E_DENSE) LPJ(false, false, E_DENSE)
      else if (!ta && !tb) LPJ(false, false, -1)
      else if (!ta && tb && e == E_DFC2) LPJ(false, true, E_DFC2)
      else if (!ta && tb) LPJ(false, true, -1)
      else if (ta && !tb) LPJ(true, false, -1)
      else LPJ(true, true, -1)
#undef LPJ
      return check_launch("gemm_bf16_pp256p (jitter)");
    }
    if (!ta && !tb) {
      if (e == E_BIAS) LPS(false, false, E_BIAS)
      else if (e == E_DENSE) LPS(false, false, E_DENSE)
      else if (e == E_FC1 && p.aux) LPS(false, false, E_FC1)
      else if (e == 0) LPS(false, false, 0)                 // input gradients against a transposed weight copy
      else if (e == E_RES) LPS(false, false, E_RES)
      else if (e == E_DFC2) LPS(false, false, E_DFC2)
      else LPS(false, false, -1)
    } else if (!ta && tb) {
      if (e == 0) LPS(false, true, 0)
      else if (e == E_RES) LPS(false, true, E_RES)
      else if (e == E_DFC2) LPS(false, true, E_DFC2)
      else LPS(false, true, -1)
    } else if (ta && !tb) LPS(true, false, -1)
    else LPS(true, true, -1)
#undef LPS
    if (p.stamps) {
      report_stamps(p.stamps, nwg, p);
      (void)hipFree(p.stamps);
    }
    return check_launch("gemm_bf16_pp256p");
  }
#define LPP(A_, B_, D_)                                                                                      \
  {                                                                                                          \
    const size_t lds = (size_t)pp_nb(D_) * PP_STAGE;                                                         \
    auto kern = gemm_bf16_pp256<TOut, A_, B_, D_>;                                                           \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_bf16_pp256: cannot reserve %zu bytes of LDS", lds);                                  \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 512, lds, st, p);                                                         \
  }
#define LPPD(D_)                          \
  {                                       \
    if (!ta && !tb) LPP(false, false, D_) \
    else if (!ta && tb) LPP(false, true, D_) \
    else if (ta && !tb) LPP(true, false, D_) \
    else LPP(true, true, D_)              \
  }
  if (dist >= 4) LPPD(4)
  else if (dist == 3) LPPD(3)
  else LPPD(2)
#undef LPPD
#undef LPP
  if (p.stamps) {
    report_stamps(p.stamps, nwg, p);
    (void)hipFree(p.stamps);
  }
  return check_launch("gemm_bf16_pp256");
}

template int launch_pp256<float>(hipStream_t, const GemmParams&, int, int);
template int launch_pp256<bf16_t>(hipStream_t, const GemmParams&, int, int);

// 8-bit operands: persistent kernel only (large problems), NN layout only.
template <int F8>
static int launch_pp256p_f8(hipStream_t st, const GemmParams& p_in) {
  GemmParams p = p_in;
  // one workgroup per CU walks the tiles; a problem with fewer tiles than CUs gets one workgroup per tile (the tile
  // order stays a bijection for any tile count, the walk just ends after the first tile)
  const int nvt = p.tiles_m * p.tiles_n;
  dim3 grid((unsigned)(nvt < num_cus() ? nvt : num_cus()), 1, 1);
  p.tile_queue = nullptr;
  p.stamps = nullptr;
  p.group_n = p.tiles_n;
  {
    const double b_panel = 256.0 * (double)p.K;      // bytes: one byte per element
    if (p.tiles_n % 2 == 0 && b_panel * p.tiles_n > 3.5e6 && b_panel * (p.tiles_n / 2) <= 2.5e6 &&
        (double)p.tiles_m * p.tiles_n >= 4.0 * num_cus())
      p.group_n = p.tiles_n / 2;
  }
  const size_t lds = (size_t)pp_nb(4) * PP_STAGE;
#define LF8(E_)                                                                                              \
  {                                                                                                          \
    auto kern = gemm_bf16_pp256p<false, false, 4, E_, F8>;                                                   \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_fp8: cannot reserve %zu bytes of LDS", lds);                                         \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 512, lds, st, p);                                                         \
  }
  constexpr int E_BIAS = MDT_EPI_BIAS, E_DENSE = MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT,
                E_FC1 = MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_AUX_GRAD, E_RES = MDT_EPI_RESIDUAL,
                E_DFC2 = MDT_EPI_MULAUX | MDT_EPI_COLSUM;
  const int e = p.epilogue;
  if constexpr (F8 == 1) {          // forward: activations e4m3
    if (e == E_BIAS) LF8(E_BIAS)
    else if (e == E_DENSE) LF8(E_DENSE)
    else if (e == E_FC1 && p.aux) LF8(E_FC1)
    else LF8(-1)
  } else {                          // input gradients: dY e5m2 against the transposed e4m3 weight copy
    if (e == 0) LF8(0)
    else if (e == E_RES) LF8(E_RES)
    else if (e == E_DFC2) LF8(E_DFC2)
    else LF8(-1)
  }
#undef LF8
  return check_launch("gemm_fp8_pp256p");
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_gemm(void* stream, int dtype, int out_dtype, int trans_a, int trans_b, int64_t M, int64_t N,
                        int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                        int epilogue, float alpha, const void* bias, const void* residual, int64_t ldr, void* aux,
                        int64_t ldaux, int split_k, float drop_p, uint64_t drop_seed, float* colsum) {
  MDT_CHECK_ARG(dtype == MDT_F32 || dtype == MDT_BF16, "mdt_gemm: bad dtype %d", dtype);
  MDT_CHECK_ARG(out_dtype == MDT_F32 || out_dtype == MDT_BF16, "mdt_gemm: bad out_dtype %d", out_dtype);
  MDT_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "mdt_gemm: negative shape");
  if (M == 0 || N == 0) return MDT_OK;
  MDT_CHECK_ARG(A && B && C, "mdt_gemm: null operand");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_BIAS) || bias, "mdt_gemm: MDT_EPI_BIAS without bias");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_RESIDUAL) || residual, "mdt_gemm: MDT_EPI_RESIDUAL without residual");
  MDT_CHECK_ARG(!(epilogue & (MDT_EPI_DGELU | MDT_EPI_MULAUX)) || aux, "mdt_gemm: MDT_EPI_DGELU / MDT_EPI_MULAUX without aux");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_AUX_GRAD) || (epilogue & MDT_EPI_GELU), "mdt_gemm: MDT_EPI_AUX_GRAD needs MDT_EPI_GELU");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_ATOMIC) || out_dtype == MDT_F32, "mdt_gemm: MDT_EPI_ATOMIC needs fp32 C");
  MDT_CHECK_ARG(dtype == MDT_BF16 || out_dtype == MDT_F32, "mdt_gemm: fp32 inputs need fp32 output");
  if (split_k < 1) split_k = 1;
  MDT_CHECK_ARG(split_k == 1 || (epilogue & MDT_EPI_ATOMIC), "mdt_gemm: split_k > 1 needs MDT_EPI_ATOMIC");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_DROPOUT) || (drop_p >= 0.f && drop_p < 1.f), "mdt_gemm: dropout p=%f out of [0,1)", drop_p);
  MDT_CHECK_ARG(split_k == 1 || !(epilogue & (MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_RESIDUAL | MDT_EPI_DGELU | MDT_EPI_DROPOUT | MDT_EPI_MULAUX)),
                "mdt_gemm: split_k > 1 supports only the plain accumulate epilogue");
  hipStream_t st = (hipStream_t)stream;
  GemmParams p;
  p.M = M; p.N = N; p.K = K; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.epilogue = epilogue; p.alpha = alpha; p.bias = bias; p.residual = residual; p.ldr = ldr;
  p.aux = aux; p.ldaux = ldaux; p.split_k = split_k;
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_DROPOUT) || M * N < DROP_MAX_ELEMS, "mdt_gemm: dropout site of %lld elements (limit 2^33)", (long long)(M * N));
  p.drop = make_drop((epilogue & MDT_EPI_DROPOUT) ? drop_p : 0.f, drop_seed);
  p.colsum = colsum;
  p.stamps = nullptr;
  p.tile_queue = nullptr;
  p.alpha_dev = p.alpha_dev2 = nullptr;
  p.group_n = 1 << 30;   // row-major unless launch_pp256 decides otherwise
  p.epilogue |= switches().gemm_diag << 20;   // diagnostics: 1 skip stores, 2 sc1 stores, 16 jittered stress build of the 8-wave persistent kernel, 4 skewed starts within an XCD, 8 every tile loads tile (0,0)'s panels, 64 XCDs skewed against each other, 256 row-panel groups skewed inside an XCD (4-wave kernel)
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_COLSUM) || (colsum && split_k == 1), "mdt_gemm: MDT_EPI_COLSUM needs a colsum buffer and split_k == 1");
  // (MDT_EPI_ATOMIC already implies an fp32 C; said again because the kernels compute the sums in their fp32-output form only)
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_ASUM) || (colsum && trans_a && (epilogue & MDT_EPI_ATOMIC) && !(epilogue & MDT_EPI_COLSUM) && dtype == MDT_BF16 && out_dtype == MDT_F32),
                "mdt_gemm: MDT_EPI_ASUM needs bf16 operands, an fp32 C, trans_a = 1, MDT_EPI_ATOMIC, a colsum buffer and no MDT_EPI_COLSUM");
  // tile128 contract: bf16, output dims that are tiled along a contiguous axis must be
  // whole tiles, 16-B aligned rows.
  // MDT_EPI_ASUM rides on the 256 x 256 ping-pong kernel only; every other path sums the stored A ([K, M]) with the
  // column-sum kernel first and runs without the flag
  auto strip_asum = [&]() -> int {
    if (!(p.epilogue & MDT_EPI_ASUM)) return MDT_OK;
    p.epilogue &= ~MDT_EPI_ASUM;
    return mdt_colsum(stream, dtype, K, M, A, lda, colsum, nullptr);
  };
  bool fast = dtype == MDT_BF16 && K > 0;
  if (fast) {
    const bool a_ok = trans_a ? (M % T_BM == 0) : (K % T_BK == 0);
    const bool b_ok = trans_b ? (N % T_BN == 0) : (K % T_BK == 0);
    const bool al = (lda % 8 == 0) && (ldb % 8 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    // 32-bit buffer offsets: one 128-row panel (k-contiguous) or one k-chunk (k-major) must stay < 4 GiB
    fast = a_ok && b_ok && al && (N % T_BN == 0 || !trans_b) && lda * 2 * 128 < (1ll << 31) && ldb * 2 * 128 < (1ll << 31);
    if (!trans_b && N % T_BN != 0) fast = false;  // B rows beyond N would alias the next tensor
    // vector epilogue: 16-byte accesses on C / bias / residual / aux
    const int ov = out_dtype == MDT_F32 ? 4 : 8;
    if (!(epilogue & MDT_EPI_ATOMIC) && (ldc % ov || ((uintptr_t)C & 15))) fast = false;   // atomics are scalar
    if ((epilogue & MDT_EPI_BIAS) && ((uintptr_t)bias & 15)) fast = false;
    if ((epilogue & MDT_EPI_RESIDUAL) && (ldr % 8 || ((uintptr_t)residual & 15))) fast = false;
    if (aux && (ldaux % 8 || ((uintptr_t)aux & 15))) fast = false;
  }
  if (fast) {
    int64_t chunk = ((K + split_k - 1) / split_k + T_BK - 1) / T_BK * T_BK;
    p.k_chunk = chunk;
    p.split_k = (int)((K + chunk - 1) / chunk);
    p.tiles_m = (int)((M + T_BM - 1) / T_BM);
    p.tiles_n = (int)(N / T_BN);
    if ((trans_a || trans_b) && chunk * (trans_a ? lda : ldb) * 2 >= (1ll << 32)) fast = false;
  }
  if (fast) {
    // big problems: 256-row tiles, one 8-wave block per CU; small ones keep 128x128 (2 blocks / CU)
    const char* force = switches().gemm_tile[0] ? switches().gemm_tile : nullptr;      // "128" | "256x128" | "256x256" | "pp" (tuning / A-B runs)
    const bool m256 = trans_a ? (M % 256 == 0) : true;
    const int64_t t256 = ((M + 255) / 256) * (N / 128) * p.split_k;
    const bool n256 = N % 256 == 0;
    // 256x256 whenever it still fills most of the chip (1 block per CU): its L2 -> LDS traffic per
    // flop is half that of 128x128 and measured 10-40 % faster at every C2 shape (profiles/)
    bool use256x256 = m256 && n256 && t256 / 2 >= 200;
    bool use256x128 = m256 && !use256x256 && t256 >= 256;
    if (force) {
      use256x256 = m256 && n256 && !strcmp(force, "256x256");
      use256x128 = m256 && !strcmp(force, "256x128");
    }
    if (use256x256 || (force && m256 && n256 && !strcmp(force, "pp"))) {
      p.tiles_m = (int)((M + 255) / 256);
      p.tiles_n = (int)(N / 256);
      const bool pp = force ? !strcmp(force, "pp") : !switches().gemm_no_pp;
      if (pp) return out_dtype == MDT_F32 ? launch_pp256<float>(st, p, trans_a, trans_b)
                                          : launch_pp256<bf16_t>(st, p, trans_a, trans_b);
      if (int e = strip_asum()) return e;
      return out_dtype == MDT_F32 ? launch_tile256<float, 256, 2, 4, 2>(st, p, trans_a, trans_b)
                                  : launch_tile256<bf16_t, 256, 2, 4, 2>(st, p, trans_a, trans_b);
    }
    if (int e = strip_asum()) return e;
    if (use256x128) {
      p.tiles_m = (int)((M + 255) / 256);
      return out_dtype == MDT_F32 ? launch_tile256<float, 128, 4, 2, 3>(st, p, trans_a, trans_b)
                                  : launch_tile256<bf16_t, 128, 4, 2, 3>(st, p, trans_a, trans_b);
    }
    return out_dtype == MDT_F32 ? launch_tile128<float>(st, p, trans_a, trans_b)
                                : launch_tile128<bf16_t>(st, p, trans_a, trans_b);
  }
  if (int e = strip_asum()) return e;
  int64_t chunk = ((K + split_k - 1) / split_k + 15) / 16 * 16;
  if (chunk < 16) chunk = 16;
  p.k_chunk = chunk;
  p.split_k = K > 0 ? (int)((K + chunk - 1) / chunk) : 1;
  if (dtype == MDT_F32) return launch_generic<float, float>(st, p, trans_a, trans_b);
  return out_dtype == MDT_F32 ? launch_generic<bf16_t, float>(st, p, trans_a, trans_b)
                              : launch_generic<bf16_t, bf16_t>(st, p, trans_a, trans_b);
}

extern "C" size_t mdt_gemm_tile_queue_bytes(void) { return (size_t)512 * 16 * sizeof(int); }

extern "C" int mdt_gemm_set_tile_queue(void* zeroed_device_buffer, size_t bytes) {
  if (zeroed_device_buffer == nullptr) { g_tile_queues = nullptr; g_tile_queue_sets = 0; return MDT_OK; }
  MDT_CHECK_ARG(bytes >= 16 * sizeof(int) && ((uintptr_t)zeroed_device_buffer & 3) == 0, "mdt_gemm_set_tile_queue: buffer too small / unaligned");
  g_tile_queues = (int*)zeroed_device_buffer;
  g_tile_queue_sets = (int)(bytes / (16 * sizeof(int)));
  return MDT_OK;
}

extern "C" int mdt_gemm_fp8(void* stream, int a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                            const void* B, int64_t ldb, void* C, int64_t ldc, int epilogue, const float* inv_scale_a,
                            const float* inv_scale_b, const void* bias, const void* residual, int64_t ldr, void* aux,
                            int64_t ldaux, float drop_p, uint64_t drop_seed, float* colsum) {
  return mdt_gemm_fp8_q8(stream, a_format, M, N, K, A, lda, B, ldb, C, ldc, epilogue, inv_scale_a, inv_scale_b, bias, residual, ldr,
                         aux, ldaux, drop_p, drop_seed, colsum, nullptr, 0, 0, nullptr, nullptr);
}

extern "C" int mdt_gemm_fp8_q8(void* stream, int a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                               const void* B, int64_t ldb, void* C, int64_t ldc, int epilogue, const float* inv_scale_a,
                               const float* inv_scale_b, const void* bias, const void* residual, int64_t ldr, void* aux,
                               int64_t ldaux, float drop_p, uint64_t drop_seed, float* colsum, void* q8_out, int64_t ld_q8,
                               int q8_format, const float* q8_scale, float* q8_amax) {
  MDT_CHECK_ARG(a_format == 0 || a_format == 1, "mdt_gemm_fp8: a_format %d (0 = e4m3, 1 = e5m2)", a_format);
  MDT_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "mdt_gemm_fp8: negative shape");
  if (M == 0 || N == 0) return MDT_OK;
  MDT_CHECK_ARG(A && B && C && inv_scale_a && inv_scale_b, "mdt_gemm_fp8: null operand / scale");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_BIAS) || bias, "mdt_gemm_fp8: MDT_EPI_BIAS without bias");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_RESIDUAL) || residual, "mdt_gemm_fp8: MDT_EPI_RESIDUAL without residual");
  MDT_CHECK_ARG(!(epilogue & (MDT_EPI_DGELU | MDT_EPI_MULAUX)) || aux, "mdt_gemm_fp8: MDT_EPI_DGELU / MDT_EPI_MULAUX without aux");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_AUX_GRAD) || (epilogue & MDT_EPI_GELU), "mdt_gemm_fp8: MDT_EPI_AUX_GRAD needs MDT_EPI_GELU");
  MDT_CHECK_ARG(!(epilogue & (MDT_EPI_ATOMIC | MDT_EPI_ACCUM)), "mdt_gemm_fp8: bf16 store epilogues only");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_COLSUM) || colsum, "mdt_gemm_fp8: MDT_EPI_COLSUM needs a colsum buffer");
  MDT_CHECK_ARG(!(epilogue & MDT_EPI_DROPOUT) || (drop_p >= 0.f && drop_p < 1.f && M * N < DROP_MAX_ELEMS), "mdt_gemm_fp8: bad dropout site");
  // the 8-bit instantiation exists for the big k-contiguous GEMMs only: anything else is the caller's cue to stay in bf16
  const bool ok = N % 256 == 0 && K % 64 == 0 && K >= 256 && lda % 16 == 0 && ldb % 16 == 0 && (((uintptr_t)A | (uintptr_t)B) % 16 == 0) &&
                  ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 && (!(epilogue & MDT_EPI_BIAS) || ((uintptr_t)bias & 15) == 0) &&
                  (!(epilogue & MDT_EPI_RESIDUAL) || (ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                  (!aux || (ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0)) && lda * 256 < (1ll << 31) && ldb * 256 < (1ll << 31);
  if (!ok) MDT_UNSUPPORTED("mdt_gemm_fp8: shape M=%lld N=%lld K=%lld (needs N %% 256 == 0, K %% 64 == 0, K >= 256, 16-byte rows)",
                           (long long)M, (long long)N, (long long)K);
  GemmParams p;
  p.M = M; p.N = N; p.K = K; p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.epilogue = epilogue; p.alpha = 1.0f; p.alpha_dev = inv_scale_a; p.alpha_dev2 = inv_scale_b;
  p.bias = bias; p.residual = residual; p.ldr = ldr; p.aux = aux; p.ldaux = ldaux; p.split_k = 1; p.k_chunk = K;
  p.drop = make_drop((epilogue & MDT_EPI_DROPOUT) ? drop_p : 0.f, drop_seed);
  p.colsum = colsum;
  p.tiles_m = (int)((M + 255) / 256);
  p.tiles_n = (int)(N / 256);
  if (q8_out) {
    MDT_CHECK_ARG(q8_format == 0 || q8_format == 1, "mdt_gemm_fp8_q8: q8_format %d (0 = e4m3, 1 = e5m2)", q8_format);
    MDT_CHECK_ARG(q8_scale && q8_amax && ld_q8 >= N, "mdt_gemm_fp8_q8: the fp8 output needs its scale, its maximum slot and rows of at least N bytes");
    p.q8_out = q8_out; p.ld_q8 = ld_q8; p.q8_fmt = q8_format; p.q8_scale = q8_scale; p.q8_amax = q8_amax;
  }
  if (switches().gemm_f8w) {
    GemmParams q = p;
    q.tile_queue = nullptr;
    q.stamps = nullptr;
    q.group_n = q.tiles_n;
    const double b_panel = 256.0 * (double)K;        // bytes: one byte per element
    if (q.tiles_n % 2 == 0 && b_panel * q.tiles_n > 3.5e6 && b_panel * (q.tiles_n / 2) <= 2.5e6 && (double)q.tiles_m * q.tiles_n >= 4.0 * num_cus())
      q.group_n = q.tiles_n / 2;
    const int r = launch_f8_w4((hipStream_t)stream, q, a_format, num_cus());
    if (r != -1) return r;
  }
  // only the block-MFMA kernel writes the fp8 copy (and only for the epilogues a training step asks it for): the caller quantises the bf16 output itself
  if (q8_out) MDT_UNSUPPORTED("mdt_gemm_fp8_q8: no kernel writes an fp8 copy for epilogue %d, K = %lld (MDT_GEMM_F8W=%d)", epilogue, (long long)K, switches().gemm_f8w);
  return a_format == 0 ? launch_pp256p_f8<1>((hipStream_t)stream, p) : launch_pp256p_f8<2>((hipStream_t)stream, p);
}

extern "C" int mdt_colsum(void* stream, int dtype, int64_t M, int64_t N, const void* X, int64_t ldx, float* out,
                          const int32_t* row_weight) {
  MDT_CHECK_ARG(dtype == MDT_F32 || dtype == MDT_BF16, "mdt_colsum: dtype %d", dtype);
  if (M == 0 || N == 0) return MDT_OK;           // nothing to add (an empty X has no address)
  MDT_CHECK_ARG(X && out, "mdt_colsum: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int vn = dtype == MDT_BF16 ? 8 : 4;
  if (N % vn == 0 && ldx % vn == 0 && ((uintptr_t)X & 15) == 0) {
    const unsigned gx = (unsigned)((N + 64 * vn - 1) / (64 * vn));
    int64_t chunks = 2048 / gx;
    if (chunks < 1) chunks = 1;
    int rows_per_block = (int)((M + chunks - 1) / chunks);
    if (rows_per_block < 32) rows_per_block = 32;
    dim3 grid(gx, (unsigned)((M + rows_per_block - 1) / rows_per_block));
    if (dtype == MDT_F32) hipLaunchKernelGGL((colsum_vec_kernel<float, 4>), grid, 256, 0, st, M, N, (const float*)X, ldx, out, rows_per_block, row_weight);
    else hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 8>), grid, 256, 0, st, M, N, (const bf16_t*)X, ldx, out, rows_per_block, row_weight);
    return check_launch("colsum_vec");
  }
  const int rows_per_block = 512;
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + rows_per_block - 1) / rows_per_block));
  if (dtype == MDT_F32) hipLaunchKernelGGL((colsum_kernel<float, 256>), grid, 256, 0, st, M, N, (const float*)X, ldx, out, rows_per_block, row_weight);
  else hipLaunchKernelGGL((colsum_kernel<bf16_t, 256>), grid, 256, 0, st, M, N, (const bf16_t*)X, ldx, out, rows_per_block, row_weight);
  return check_launch("colsum");
}

extern "C" int mdt_dropout(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx, void* y, int64_t ldy,
                           float p_, uint64_t seed) {
  if (rows == 0 || D == 0) return MDT_OK;
  MDT_CHECK_ARG(x && y && p_ >= 0.f && p_ < 1.f, "mdt_dropout: bad arguments (p=%f)", p_);
  hipStream_t st = (hipStream_t)stream;
  MDT_CHECK_ARG(rows * D < DROP_MAX_ELEMS, "mdt_dropout: site of %lld elements (limit 2^33)", (long long)(rows * D));
  const DropCfg d = make_drop(p_, seed);
  const unsigned grid = (unsigned)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4);
  const int vn = dtype == MDT_BF16 ? 8 : 4;
  if (D % vn == 0 && ldx % vn == 0 && ldy % vn == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    if (dtype == MDT_F32) hipLaunchKernelGGL((dropout_vec_kernel<float, 4>), grid, 256, 0, st, rows, D, (const float*)x, ldx, (float*)y, ldy, d);
    else if (dtype == MDT_BF16) hipLaunchKernelGGL((dropout_vec_kernel<bf16_t, 8>), grid, 256, 0, st, rows, D, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, d);
    else MDT_UNSUPPORTED("mdt_dropout: dtype %d", dtype);
    return check_launch("dropout_vec");
  }
  if (dtype == MDT_F32) hipLaunchKernelGGL((dropout_kernel<float>), grid, 256, 0, st, rows, D, (const float*)x, ldx, (float*)y, ldy, d);
  else if (dtype == MDT_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), grid, 256, 0, st, rows, D, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, d);
  else MDT_UNSUPPORTED("mdt_dropout: dtype %d", dtype);
  return check_launch("dropout");
}

extern "C" int mdt_dropout_mask(void* stream, int64_t n, float p_, uint64_t seed, uint8_t* mask) {
  if (n == 0) return MDT_OK;
  MDT_CHECK_ARG(mask && p_ >= 0.f && p_ < 1.f && n < DROP_MAX_ELEMS, "mdt_dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), 256, 0,
                     (hipStream_t)stream, n, make_drop(p_, seed), mask);
  return check_launch("dropout_mask");
}

extern "C" int mdt_cast(void* stream, int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst) {
  if (n == 0) return MDT_OK;
  MDT_CHECK_ARG(src && dst, "mdt_cast: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (src_dtype == MDT_F32 && dst_dtype == MDT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, 256, 0, st, n, (const float*)src, (bf16_t*)dst);
  else if (src_dtype == MDT_BF16 && dst_dtype == MDT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, 256, 0, st, n, (const bf16_t*)src, (float*)dst);
  else if (src_dtype == MDT_F32 && dst_dtype == MDT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, 256, 0, st, n, (const float*)src, (float*)dst);
  else if (src_dtype == MDT_BF16 && dst_dtype == MDT_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, 256, 0, st, n, (const bf16_t*)src, (bf16_t*)dst);
  else MDT_UNSUPPORTED("mdt_cast: dtypes %d -> %d", src_dtype, dst_dtype);
  return check_launch("cast");
}

extern "C" int mdt_transpose2d(void* stream, int src_dtype, int dst_dtype, int64_t rows, int64_t cols, const void* src,
                               int64_t lds_, void* dst, int64_t ldd) {
  if (rows == 0 || cols == 0) return MDT_OK;
  MDT_CHECK_ARG(src && dst, "mdt_transpose2d: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  if (src_dtype == MDT_F32 && dst_dtype == MDT_BF16) hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, 256, 0, st, rows, cols, (const float*)src, lds_, (bf16_t*)dst, ldd);
  else if (src_dtype == MDT_BF16 && dst_dtype == MDT_BF16) hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, 256, 0, st, rows, cols, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd);
  else if (src_dtype == MDT_F32 && dst_dtype == MDT_F32) hipLaunchKernelGGL((transpose_kernel<float, float>), grid, 256, 0, st, rows, cols, (const float*)src, lds_, (float*)dst, ldd);
  else MDT_UNSUPPORTED("mdt_transpose2d: dtypes %d -> %d", src_dtype, dst_dtype);
  return check_launch("transpose2d");
}
