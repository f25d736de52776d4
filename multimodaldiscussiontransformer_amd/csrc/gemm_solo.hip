// gemm_bf16_solo: the 8-wave ping-pong kernel's HALF as a workgroup of its own — 4 waves on a 128 x 256 tile (each wave 128 x 64,
// 128 accumulator registers), 32-k ring stages of 24 KiB (A 128 rows, B 256 columns), three stage buffers (72 KiB), TWO workgroups
// per CU.  Why: the heavy epilogues (fc1 forward: bias + erf-GELU + its derivative, ~20 VALU issue slots per element, two outputs)
// cost the 8-wave kernel 0.4 of its MFMA time, and there all eight waves of a CU reach the epilogue together — the matrix pipe
// idles while the VALU works.  Two independent workgroups per CU drift apart: one is in its K loop while the other converts and
// stores, and the SIMD's scheduler interleaves the MFMA stream of one wave with the VALU stream of the other.  The price: B is
// staged twice per CU (48 instead of 32 KiB of LDS-DMA per 32-k step and CU); LDS fragment reads are the same.
//
// Ring protocol (one barrier per step, unlike the two of the ping-pong pair): at step s a wave reads the fragments of stage s,
// requests stage s + 2 into the buffer stage s - 1 lived in, waits for its own pieces of stage s + 1 (counted vmcnt: the six newest
// operations — stage s + 2 — stay in flight) and for its fragment reads, and meets the others at the barrier.  WAR: everybody's
// reads of stage s - 1 precede the barrier of step s - 1, which precedes every request of step s.  RAW: everybody's pieces of stage
// s + 1 have landed before the barrier of step s, which precedes every read of step s + 1.  The ring keeps turning across tile
// boundaries (the last two steps of a tile request the first two stages of the workgroup's next tile).
// bf16 in / out, both operands k-contiguous, split_k = 1, K a multiple of 32 with at least 4 steps; same accumulation order as
// every other bf16 kernel of the library (bit-identical outputs, tests/test_kernels_gpu.py).
#include "common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

namespace mdt {

constexpr int SOLO_BM = 128, SOLO_BN = 256, SOLO_A_BYTES = SOLO_BM * 64, SOLO_STAGE = SOLO_A_BYTES + SOLO_BN * 64, SOLO_NB = 3, SOLO_DIST = 2;

template <int EPK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_solo(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave's 64-column block
  const int nvt = p.tiles_m * p.tiles_n;                        // tiles_m counts 128-row tiles here
  const int nhs = (int)(p.K / 32);
  const int64_t lda_b = p.lda * 2, ldb_b = p.ldb * 2;

  struct Desc { __amdgpu_buffer_rsrc_t rsA, rsB; int64_t m0, n0; };
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
    d.m0 = (int64_t)tm * SOLO_BM;
    d.n0 = (int64_t)tn * SOLO_BN;
    const int64_t a_bytes = (p.M - d.m0) * lda_b, b_bytes = (p.N - d.n0) * ldb_b;
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + d.m0 * lda_b), 0, (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes), 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + d.n0 * ldb_b), 0, (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes), 0x00020000);
    return d;
  };
  auto issue_step = [&](const Desc& d, int hs, int buf) {          // 2 + 4 LDS-DMA instructions per wave
    char* st = smem + buf * SOLO_STAGE;
    stage_step<false, SOLO_BM, 4>(d.rsA, lda_b, (int64_t)hs * 32, 0, st, wc, lane);
    stage_step<false, SOLO_BN, 4>(d.rsB, ldb_b, (int64_t)hs * 32, 0, st + SOLO_A_BYTES, wc, lane);
  };

  // workgroups are dealt round-robin to the XCDs and, inside an XCD, to its CUs: the second half of an XCD's workgroups are the
  // second tenants of their CUs.  They start half a tile late, so that a CU's two workgroups alternate K loop and epilogue.
  if (p.solo_skew > 0 && (int)(blockIdx.x >> 3) >= (int)(gridDim.x >> 4))
    for (int i = 0; i < p.solo_skew; ++i) __builtin_amdgcn_s_sleep(1);
  int v = blockIdx.x;
  int v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
  Desc cur = make_desc(v);
  bool has_next = v_next >= 0;
  Desc nxt = make_desc(has_next ? v_next : v);
  issue_step(cur, 0, 0);
  issue_step(cur, 1, 1);                                           // host guarantees nhs >= 4
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                 // own pieces of stage 0
  int b_rd = 0, b_wr = SOLO_DIST;

  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();                                  // everybody's pieces of this tile's stage 0
    __builtin_amdgcn_sched_barrier(0);
    for (int hs = 0; hs < nhs; ++hs) {
      const char* rd = smem + b_rd * SOLO_STAGE;
      bf16x8 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag_h<false, SOLO_BN>(rd + SOLO_A_BYTES, wc * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = load_frag_h<false, SOLO_BM>(rd, i * 16, lane);
      const int tgt = hs + SOLO_DIST;
      if (tgt < nhs) {
        issue_step(cur, tgt, b_wr);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else if (has_next) {
        issue_step(nxt, tgt - nhs, b_wr);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(b[j], a[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      b_wr = b_wr + 1 == SOLO_NB ? 0 : b_wr + 1;
      b_rd = b_rd + 1 == SOLO_NB ? 0 : b_rd + 1;
    }
    direct_epilogue<2, EPK>(p, acc, lane, cur.m0, cur.n0 + wc * 64);
    if (!has_next) break;
    cur = nxt;
    v += gridDim.x;
    v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
    has_next = v_next >= 0;
    if (has_next) nxt = make_desc(v_next);
  }
}

// → MDT_OK after a launch, -1 when this kernel has no instantiation for the launch (the caller goes on to the 8-wave kernel)
int launch_solo(hipStream_t st, const GemmParams& p_in, int ta, int tb, int n_cus) {
  constexpr int E_FC1 = MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_AUX_GRAD, E_BIAS = MDT_EPI_BIAS,
                E_DENSE = MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT;
  const int e = p_in.epilogue & ((1 << 22) - 1);
  if (ta || tb || p_in.split_k != 1 || p_in.K % 32 != 0 || p_in.K < 128 || p_in.N % SOLO_BN != 0) return -1;
  if (!((e == E_FC1 && p_in.aux) || e == E_BIAS || e == E_DENSE)) return -1;
  GemmParams p = p_in;
  p.tiles_m = (int)((p.M + SOLO_BM - 1) / SOLO_BM);
  p.tiles_n = (int)(p.N / SOLO_BN);
  // tile order as for the 8-wave kernel (gemm.hip): row-major, or the two column halves swept by XCDs 0-3 / 4-7 when B is too wide
  // for an XCD's L2 and its halves fit
  p.group_n = p.tiles_n;
  {
    const double b_panel = 256.0 * (double)p.K * 2.0;
    if (p.tiles_n % 2 == 0 && b_panel * p.tiles_n > 3.5e6 && b_panel * (p.tiles_n / 2) <= 2.5e6 && (double)p.tiles_m * p.tiles_n >= 8.0 * n_cus)
      p.group_n = p.tiles_n / 2;
    if (switches().gemm_group >= 1) p.group_n = switches().gemm_group < p.tiles_n ? switches().gemm_group : p.tiles_n;
  }
  const int64_t nvt = (int64_t)p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(nvt < 2 * n_cus ? ((nvt + 7) & ~7ll) : 2 * n_cus);
  if (grid > nvt) return -1;                                       // fewer tiles than one round of workgroups: not this kernel's case
  // half a tile: K loop ~ 1.1 k cycles per 32-k step, epilogue ~ as long as 12-20 steps; MDT_GEMM_SOLO_SKEW overrides (0 = none)
  p.solo_skew = switches().gemm_solo_skew >= 0 ? switches().gemm_solo_skew : (int)((p.K / 32 + 16) * 1100 / 2 / 64);
  const size_t lds = (size_t)SOLO_NB * SOLO_STAGE;
#define LS(E_)                                                                                                \
  {                                                                                                           \
    auto kern = gemm_bf16_solo<E_>;                                                                           \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                              \
        set_error("gemm_bf16_solo: cannot reserve %zu bytes of LDS", lds);                                    \
        return MDT_ERR_LAUNCH;                                                                                \
      }                                                                                                       \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid), 256, lds, st, p);                                                    \
  }
  if (e == E_FC1) LS(E_FC1)
  else if (e == E_BIAS) LS(E_BIAS)
  else LS(E_DENSE)
#undef LS
  return check_launch("gemm_bf16_solo");
}

}  // namespace mdt
