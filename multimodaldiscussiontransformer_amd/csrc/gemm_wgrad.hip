// gemm_bf16_w4s — the split-K, fp32-accumulating GEMM of the weight gradients (dW = dY^T X: both operands k-major, K = the
// token count, 7-28 slabs of 200-240 K-tiles) in the 4-wave form of gemm_bf16_w4p: one wave per SIMD with the whole
// register file (256 accumulators in AGPRs), 128 x 128 of the 256 x 256 tile per wave, MFMAs written as volatile asm so
// that source order is issue order — per 32-k step eight blocks of
//     2 MFMA, fragment read, 2 MFMA, fragment read, 2 MFMA, LDS-DMA piece, 2 MFMA
// on one fragment set while the reads fill the other.  Against the 8-wave ping-pong kernel (gemm_bf16_pp256<float>), whose
// READ segment (24 transposing fragment reads + 4 LDS-DMA issues + two waits) outlasts the partner group's 32-MFMA segment
// on these launches (MFMA pipe busy 0.515, profiles/round2_r2v8_sq_counters.json): a third less LDS read traffic and no
// segment hand-over.  One work item (K slab, tile) per workgroup, slab-major and XCD-aware exactly as in gemm_bf16_pp256
// (the tiles that share a slab's dY / X panels sit in one XCD's L2); no persistence — a slab's 200+ K-tiles amortise the
// prologue.  Epilogue: the four 64 x 64 blocks of a wave go through its 16-KiB LDS slot (the ring is idle by then) and leave
// as 256-byte contiguous fp32 atomics (tile_epilogue<float>); MDT_EPI_ASUM (the bias gradient riding on the weight
// gradient) as in the 8-wave kernel: in the workgroups of tile column 0 the two waves that hold the same A fragments take
// four of the eight row tiles each against a B fragment of ones, and the tiles_n workgroups of a row panel share the K steps.
// Replaces the autograd of every nn.Linear weight (reference: modules/multi_graphormer_fusion_layer.py:94-96,138-146 and the
// HF BertLayer / ViTLayer linears behind them).
#include <utility>

#include "gemm_tiles.hpp"

namespace mdt {

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4s(GemmParams p) {
  constexpr int PP_DIST = 4, PP_NB = 5;
  constexpr int BM = 256, BN = 256, A_BYTES = BM * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // work items = (K slab, tile), slab-major; XCD x = id % 8 takes the x-th contiguous run of the list (gemm_bf16_pp256)
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = tiles * (int)gridDim.z;
  const int bid = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int slab = work / tiles;
  const int tile = work - slab * tiles;
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
  const int nhs = 2 * (int)((kend - kbeg + 63) / 64);          // 32-deep steps (even; the host guarantees >= 8)

  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  // diagnostic (MDT_GEMM_DIAG=8): every work item LOADS the panels of item (slab 0, tile (0, 0)) — all fills hit L2 — while
  // the atomics still go to the item's own place: separates the fill's miss path from the loop's own cost
  const bool diag_l2 = (p.epilogue & (1 << 23)) != 0;
  const int64_t lm0 = diag_l2 ? 0 : m0, ln0 = diag_l2 ? 0 : n0, lk0 = diag_l2 ? 0 : kbeg;
  const char* a_base;
  const char* b_base;
  int64_t a_bytes, b_bytes;
  if constexpr (!A_KM) { a_base = (const char*)p.A + lm0 * lda_b + lk0 * 2; a_bytes = (p.M - lm0) * lda_b - lk0 * 2; }
  else { a_base = (const char*)p.A + lk0 * lda_b + lm0 * 2; a_bytes = (kend - kbeg) * lda_b - lm0 * 2; }
  if constexpr (!B_KM) { b_base = (const char*)p.B + ln0 * ldb_b + lk0 * 2; b_bytes = (p.N - ln0) * ldb_b - lk0 * 2; }
  else { b_base = (const char*)p.B + lk0 * ldb_b + ln0 * 2; b_bytes = (kend - kbeg) * ldb_b - ln0 * 2; }
  const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
  const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_rec, 0x00020000);

  // The wave's 8 LDS-DMA pieces of a step: pieces 0-3 of A, 4-7 of B (piece index inside the operand = wave + 4 i); the
  // per-lane offsets do not depend on the step, which rides in the instruction's scalar offset (gemm_bf16_w4p).
  // A k-major operand past its last k-row (a slab's tail) and a k-contiguous one past row M read as zeros through the
  // descriptor; a k-major row's columns past the tile (m0 + 256 > M) belong to the next k-row and are masked by the atomics'
  // row test in the epilogue.
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
  auto soff_a = [&](int hs) -> int { return A_KM ? (int)((int64_t)hs * 32 * lda_b) : hs * 64; };
  auto soff_b = [&](int hs) -> int { return B_KM ? (int)((int64_t)hs * 32 * ldb_b) : hs * 64; };
  auto issue_piece = [&](int sa, int sb, int buf, int q) __attribute__((always_inline)) {
    char* st = smem + buf * PP_STAGE;
    const int piece = wave + 4 * (q & 3);
    if (q < 4) w4_dma(rsA, st + piece * 1024, voffA[q & 3], sa);
    else w4_dma(rsB, st + A_BYTES + piece * 1024, voffB[q & 3], sb);
  };

#pragma unroll
  for (int h = 0; h < PP_DIST; ++h) {
    const int sa = soff_a(h), sb = soff_b(h);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(sa, sb, h, q);
  }
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
  // MDT_EPI_ASUM: column sums of op(A) over k.  The workgroups (tm, 0 .. tiles_n - 1) of a slab read the same A fragments, so they
  // SHARE the work: workgroup tn takes every tiles_n-th 32-k step (its counter starts at tn and fires at 0) — four extra MFMAs in
  // a tiles_n-th of the steps of every workgroup instead of in every step of one tile column, whose workgroups then were the
  // slowest of the launch (measured before the split, one call: +4 % qkv, +11 % o, +7 % fc1, +5 % fc2 over the same launch
  // without the riding sums)
  const bool asum_on = (p.epilogue & MDT_EPI_ASUM) != 0;
  int asum_ctr = tn;
  f32x4 accb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) accb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = bf16x8{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

#if defined(__HIP_DEVICE_COMPILE__)
#define W4S_MF(i_, j_)                                                                                            \
  if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i_][j_]) : "v"(fa[cs][i_]), "v"(fb[cs][j_]) : "memory"); \
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(fa[cs][i_]), "v"(fb[cs][j_]) : "memory")
#else
#define W4S_MF(i_, j_) (void)first
#endif
  // The column-sum MFMAs are BUILTINS, not asm: their accumulators live in VGPRs beside 240 other live registers, the
  // allocator moves them around the wave-uniform branch below, and only for an instruction it knows to be an MFMA does it
  // keep the wait states between the MFMA's write and those moves (as asm statements the moves read stale registers:
  // sums off by whole slabs).  What the builtin does not know is that its A fragment was loaded by an asm ds_read whose
  // wait is another asm statement: W4S_PIN re-defines the fragment behind that wait, so the MFMA cannot be scheduled above it.
#define W4S_PIN(i_) asm volatile("" : "+v"(fa[cs][i_]))
#define W4S_MB(q_, i_) accb[q_] = mfma_bf16(fa[cs][i_], ones, accb[q_])
  // one 32-k step on fragment set cs: FIRST starts the accumulators from zero; NW = vector-memory operations that may still be
  // in flight when the step begins; LD: read the next step's fragments; IS: request step hs_issue into the ring
  auto step = [&](auto cs_c, auto first_c, auto nw_c, auto ld_c, auto is_c, int hs_issue) __attribute__((always_inline)) {
    constexpr int cs = decltype(cs_c)::value, ns = cs ^ 1;
    constexpr bool first = decltype(first_c)::value;
    constexpr int NW = decltype(nw_c)::value;
    constexpr bool LD = decltype(ld_c)::value, IS = decltype(is_c)::value;
    const int sa = soff_a(hs_issue), sb = soff_b(hs_issue);
    wait_vm<NW>();                                // own pieces of the next step have landed
    __builtin_amdgcn_s_barrier();                 // ... and everybody's; the stage of the previous step is free
    const char* tnx = smem + b_next * PP_STAGE;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      W4S_MF(b, 0); W4S_MF(b, 1);
      if constexpr (LD) fa[ns][b] = w4_frag<A_KM, BM>(tnx, wr * 128 + b * 16, lane);
      W4S_MF(b, 2); W4S_MF(b, 3);
      if constexpr (LD) fb[ns][b] = w4_frag<B_KM, BN>(tnx + A_BYTES, wc * 128 + b * 16, lane);
      W4S_MF(b, 4); W4S_MF(b, 5);
      if constexpr (IS) issue_piece(sa, sb, b_wr, b);
      W4S_MF(b, 6); W4S_MF(b, 7);
    }
    const bool asum = asum_on && asum_ctr == 0;
    asum_ctr = asum_ctr + 1 == p.tiles_n ? 0 : asum_ctr + 1;
    if (asum) {                                   // wave-uniform: the two waves of a row half split its eight row tiles
      if (wc == 0) { W4S_PIN(0); W4S_PIN(1); W4S_PIN(2); W4S_PIN(3); W4S_MB(0, 0); W4S_MB(1, 1); W4S_MB(2, 2); W4S_MB(3, 3); }
      else { W4S_PIN(4); W4S_PIN(5); W4S_PIN(6); W4S_PIN(7); W4S_MB(0, 4); W4S_MB(1, 5); W4S_MB(2, 6); W4S_MB(3, 7); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    b_next = b_next + 1 == PP_NB ? 0 : b_next + 1;
    b_wr = b_wr + 1 == PP_NB ? 0 : b_wr + 1;
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
#define W4S_N(n_) std::integral_constant<int, n_> {}
  // pieces of step s + 1 were requested in step s - 3: what may still be in flight when step s begins is what steps s - 2 and
  // s - 1 requested (16 pieces); the last four steps request nothing, so the count runs down 16, 8, 0, 0
  step(C0{}, T_{}, W4S_N(16), T_{}, T_{}, 4);
  step(C1{}, F_{}, W4S_N(16), T_{}, T_{}, 5);
  for (int hs = 2; hs < nhs - 4; hs += 2) {
    step(C0{}, F_{}, W4S_N(16), T_{}, T_{}, hs + 4);
    step(C1{}, F_{}, W4S_N(16), T_{}, T_{}, hs + 5);
  }
  step(C0{}, F_{}, W4S_N(16), T_{}, F_{}, 0);
  step(C1{}, F_{}, W4S_N(8), T_{}, F_{}, 0);
  step(C0{}, F_{}, W4S_N(0), T_{}, F_{}, 0);
  step(C1{}, F_{}, W4S_N(0), F_{}, F_{}, 0);
#if defined(__HIP_DEVICE_COMPILE__)
  // asm MFMAs: the compiler does not know their results need wait states before a VALU may read them (gemm_bf16_w4p)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
#endif
#undef W4S_N
#undef W4S_MF
#undef W4S_MB
#undef W4S_PIN
  if (asum_on && (lane & 15) == 0) {   // every column of the ones product holds the row sum: column 0 reports it
    const int64_t mr = m0 + wr * 128 + wc * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (mr + 16 * q + r < p.M) atomicAdd(p.colsum + mr + 16 * q + r, p.alpha * accb[q][r]);
  }
  // four 64 x 64 blocks per wave through its 16-KiB LDS slot (the first 64 KiB of the ring; every request has landed)
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    f32x4 blk[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) blk[i][j] = acc[(h >> 1) * 4 + i][(h & 1) * 4 + j];
    tile_epilogue<float>(p, blk, smem, wave, lane, m0 + wr * 128 + (h >> 1) * 64, n0 + wc * 128 + (h & 1) * 64);
  }
}

int launch_w4s(hipStream_t st, const GemmParams& p, int ta, int tb) {
  dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)p.split_k);
  const size_t lds = (size_t)5 * PP_STAGE;
#define LW4S(A_, B_)                                                                                         \
  {                                                                                                          \
    auto kern = gemm_bf16_w4s<A_, B_>;                                                                       \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_bf16_w4s: cannot reserve %zu bytes of LDS", lds);                                    \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 256, lds, st, p);                                                         \
  }
  if (ta && tb) LW4S(true, true)
  else if (!ta && tb) LW4S(false, true)
  else if (ta && !tb) LW4S(true, false)
  else LW4S(false, false)
#undef LW4S
  return check_launch("gemm_bf16_w4s");
}

}  // namespace mdt
