// 8-bit GEMM on the block MFMA v_mfma_f32_16x16x128_f8f6f4 (unit scales): 4 waves, 256 x 256 tiles, persistent.
//
// The 16x16x32 fp8 MFMA of gemm_bf16_pp256p<.., F8> runs at the bf16 rate: that kernel only saves operand BYTES.  The f8f6f4 form
// takes 128 k per instruction in twice the cycles of the bf16 16x16x32 — 2.15 x the flops per clock measured
// (tools/probes/mfma_f8_probe.hip) — at the price of 32-byte fragments: a wave's 8 + 8 fragments of a 128-k step are 128 VGPRs,
// as many as BOTH fragment sets of the bf16 4-wave kernel.  So nothing is double-buffered here; every fragment register is
// refilled as soon as its last MFMA of the step has been issued (the rolling order tried for bf16 in
// tools/experiments/gemm_w4r_rolling_fragments.patch, where it lost 4 % to double buffering; at twice the MFMA rate it is the
// only form that fits):
//
//     a 128-k step = two passes over the wave's 8 row blocks b:  pass 0 = column blocks 0-3, pass 1 = column blocks 4-7
//       fb[4..7]  of THIS step are read during pass 0 (pass 1 of the step before was their last use)
//       fb[0..3]  of the NEXT step are read during pass 1, blocks 0-3 (pass 0 was their last use)
//       fa[b]     of the NEXT step is read right behind pass 1's MFMAs of row block b
//
// LDS: the ring of gemm_bf16_w4p — five 32-KiB stages of 64-byte rows (64 k of A and of B each); a step consumes a PAIR of
// stages, a lane's 32-byte fragment being the 16-byte chunk (lane >> 4) of its row in either stage (any assignment of k to
// lanes and bytes is as good as another as long as A and B share it).  Pair P is read from the middle of step P - 1 to the
// middle of step P; the barrier between a step's passes releases its two buffers, which stage 2P + 5 (requested during pass 1,
// needed at the NEXT barrier: the ring holds 2.5 pairs, so half of every pair has one step to arrive, the other half two) and
// stage 2P + 6 (pass 0 of step P + 1) take over.  Fragment reads are plain LDS loads (the compiler counts lgkmcnt), LDS-DMA
// arrival is a hand-counted vmcnt in front of the barrier, MFMAs are asm statements in source order (see gemm_bf16_w4p).
// Epilogue: direct_epilogue with buffer-descriptor stores, the last 4 row tiles of a finished tile pending in registers and
// leaving during the next tile's first steps, as in gemm_bf16_w4p.
//
// Operands: both k-contiguous, A e4m3 (FA = 0, activations) or e5m2 (FA = 1, gradients), B e4m3 (weights); fp32 accumulation,
// one product of two device-resident dequantisation factors on the way out (GemmParams::alpha_dev / alpha_dev2).
// Accumulation inside the instruction aligns the 128 products of a row to their largest exponent and keeps ~13 bits below it
// (probe: quarter-integer operands exact, random e4m3 operands 1e-4 of the largest product) — three orders of magnitude under
// the quantisation noise of the operands.
#include <type_traits>
#include <utility>

#include "common.hpp"
#include "gemm_tiles.hpp"
#include "gemm_epilogue.hpp"

namespace mdt {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int F8W_MAX_PEND_ROWS = 4;                   // row tiles whose outputs wait in registers (PR, per instantiation: at most)
constexpr int F8W_SPS = 4;                             // pending vectors leaving per step = one row tile: PR steps carry pending stores

template <class F, int... S>
__device__ __forceinline__ void f8w_unroll(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }

template <int FA, int EPK = -1, int AUXDS = 0, int Q8 = 0, int PR = F8W_MAX_PEND_ROWS>   // PR: pending row tiles; AUXDS: further direct stores of an epilogue (the GELU forward's second output, the fp8 copy); Q8: 1 / 2 = the output also leaves as e4m3 / e5m2
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_f8_w4(GemmParams p_in) {
  constexpr int PP_NB = 5;
  constexpr int BM = 256, BN = 256, A_BYTES = BM * 64;
  constexpr int F8W_PEND_ROWS = PR, F8W_NPEND = 4 * PR, F8W_NST = PR, F8W_DS = 32 - F8W_NPEND;   // pending vectors, steps that carry them, direct stores of an epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmParams p = p_in;
  p.alpha = *p_in.alpha_dev * *p_in.alpha_dev2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nvt = p.tiles_m * p.tiles_n;
  const int nst = (int)(p.K / 64);                    // 64-k stages of a tile (host: K % 128 == 0)
  const int nsp = nst >> 1;                           // 128-k steps
  const int64_t lda_b = p.lda, ldb_b = p.ldb;         // bytes: one per element

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
    d.m0 = (int64_t)tm * BM;
    d.n0 = (int64_t)tn * BN;
    const int64_t a_bytes = (p.M - d.m0) * lda_b, b_bytes = (p.N - d.n0) * ldb_b;
    const unsigned a_rec = (unsigned)(a_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : a_bytes);
    const unsigned b_rec = (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes);
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + d.m0 * lda_b), 0, a_rec, 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + d.n0 * ldb_b), 0, b_rec, 0x00020000);
    return d;
  };
  // the wave's 8 LDS-DMA pieces of a stage: 4 of A, 4 of B, 16 rows of 64 bytes each; the stage rides in the scalar offset
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2);
    voffA[i] = (unsigned)(row * lda_b + (((lane & 3) ^ swz_h(row)) * 16));
    voffB[i] = (unsigned)(row * ldb_b + (((lane & 3) ^ swz_h(row)) * 16));
  }
  auto issue_piece = [&](const Desc& d, int soff, int buf, int q) __attribute__((always_inline)) {
    char* st = smem + buf * PP_STAGE;
    const int piece = wave + 4 * (q & 3);
    if (q < 4) w4_dma(d.rsA, st + piece * 1024, voffA[q & 3], soff);
    else w4_dma(d.rsB, st + A_BYTES + piece * 1024, voffB[q & 3], soff);
  };

  int v = blockIdx.x;
  int v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
  Desc cur = make_desc(v);
  bool has_next = v_next >= 0;
  auto null_desc = [&](Desc d) {        // after the last tile the ring keeps turning on descriptors of zero records
    d.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    d.rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0, 0x00020000);
    return d;
  };
  Desc nxt = has_next ? make_desc(v_next) : null_desc(cur);

  // fragment reads: row (lane & 15) of a 16-row block, chunk (lane >> 4) swizzled by the row, in both stages of the pair
  const int laneA = (wr * 128 + (lane & 15)) * 64 + (((lane >> 4) ^ swz_h(lane & 15)) * 16);
  const int laneB = A_BYTES + (wc * 128 + (lane & 15)) * 64 + (((lane >> 4) ^ swz_h(lane & 15)) * 16);
  i32x8 fa[8], fb[8];
  auto frag = [&](int lane_off, int blk, int buf0, int buf1) __attribute__((always_inline)) -> i32x8 {
    const i32x4 lo = *(const i32x4*)(smem + buf0 * PP_STAGE + lane_off + blk * 1024);
    const i32x4 hi = *(const i32x4*)(smem + buf1 * PP_STAGE + lane_off + blk * 1024);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto ring = [](int b, int k) { const int r = b + k; return r >= PP_NB ? r - PP_NB : r; };     // k < PP_NB

  // prologue: stages 0-3 (pairs 0 and 1) requested, pair 0 landed, the first step's fragments read
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(cur, h * 64, h, q);            // host guarantees nst >= 4
  wait_vm<16>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) fa[i] = frag(laneA, i, 0, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) fb[i] = frag(laneB, i, 0, 1);
  int b0 = 0;                                       // ring buffer of the current pair's first stage
  f32x4 acc[8][8];
  float q_amax = 0.f;                               // Q8: this lane's running maximum of |output| over all its tiles
  const float q_scale = Q8 != 0 ? *p.q8_scale : 1.0f;

  constexpr bool PEND = EPK >= 0;
  constexpr bool EPF_LATE = EPK == (MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT);
  bf16x8 pend[F8W_NPEND];
#pragma unroll
  for (int i = 0; i < F8W_NPEND; ++i) pend[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0, 0x00020000);
  const int c_lane = lane & 15, g_lane = lane >> 4;
  const unsigned voffP = (unsigned)((wr * 128 + c_lane) * (p.ldc * 2) + (wc * 128 + 16 * (g_lane & 1) + 8 * (g_lane >> 1)) * 2);
  const int ldc16 = (int)(p.ldc * 2 * 16);
  if constexpr (PEND) {
    // a tile's first barriers count the epilogue's direct stores among what may be in flight; a workgroup's first tile has no
    // epilogue before it: dropped stores (descriptor of zero records) stand in, so the arithmetic is the same for every tile
    // asm statements: as builtins the identical stores were merged into one by the compiler and the first tile's budgets were
    // too large by all the others (gemm.hip, gemm_bf16_w4p)
#if defined(__HIP_DEVICE_COMPILE__)
    const i32x4 zero4 = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < F8W_DS + AUXDS; ++i) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(zero4), "v"(voffP), "s"(rsP) : "memory");
#endif
  }

#if defined(__HIP_DEVICE_COMPILE__)
#define F8W_MF(i_, j_)                                                                                            \
  if (first) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0 blgp:%3" : "=a"(acc[i_][j_]) : "v"(fb[j_]), "v"(fa[i_]), "n"(FA) : "memory"); \
  else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 blgp:%3" : "+a"(acc[i_][j_]) : "v"(fb[j_]), "v"(fa[i_]), "n"(FA) : "memory")
#else
#define F8W_MF(i_, j_) (void)first
#endif
#define F8W_I(n_) std::integral_constant<int, n_> {}
  // One 128-k step P of the current tile.  FIRST: accumulators start from zero; NW: vmcnt budget at the barrier; ST >= 0:
  // pending vectors ST .. ST + 3 leave; LD: pass 1 refills the fragments (false in a tile's last step: the next tile's first
  // fragments are read after the epilogue).  d0 / s0: descriptor and byte offset of the stage requested in pass 0 (stage
  // 2P + 4), d1 / s1: in pass 1 (stage 2P + 5).
  auto step = [&](auto first_c, auto nw_c, auto st_c, auto ld_c, const Desc& d0, int s0, const Desc& d1, int s1) __attribute__((always_inline)) {
    constexpr bool first = decltype(first_c)::value;
    constexpr int NW = decltype(nw_c)::value, ST = decltype(st_c)::value;
    constexpr bool LD = decltype(ld_c)::value;
    const int b1 = ring(b0, 1), b2 = ring(b0, 2), b3 = ring(b0, 3), b4 = ring(b0, 4);
    // ---- pass 0: column blocks 0-3; column blocks 4-7 of this pair arrive; stage 2P + 4 is requested into the buffer of stage 2P - 1
    f8w_unroll([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      F8W_MF(b, 0); F8W_MF(b, 1);
      if constexpr (b < 4) fb[4 + b] = frag(laneB, 4 + b, b0, b1);
      F8W_MF(b, 2); F8W_MF(b, 3);
      issue_piece(d0, s0, b4, b);
      if constexpr (ST >= 0 && (b & 1) == 1) {       // pending vector idx -> row tile (8 - F8W_PEND_ROWS) + idx / 4, column pair idx % 4
        constexpr int idx = ST + (b >> 1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pend[idx]), rsP, voffP,
                                               (8 - F8W_PEND_ROWS + (idx >> 2)) * ldc16 + (idx & 3) * 64, 0);
      }
    }, std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of this pair are done
    wait_vm<NW>();                                       // own pieces of stage 2P + 3 (requested in pass 1 of the step before) have landed
    __builtin_amdgcn_s_barrier();                        // ... and everybody's; this pair's two buffers are free
    // ---- pass 1: column blocks 4-7; the next pair's A and column blocks 0-3 arrive; stage 2P + 5 is requested into the buffer of stage 2P
    f8w_unroll([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      F8W_MF(b, 4); F8W_MF(b, 5);
      if constexpr (LD && b < 4) fb[b] = frag(laneB, b, b2, b3);
      issue_piece(d1, s1, b0, b);
      F8W_MF(b, 6); F8W_MF(b, 7);
      if constexpr (LD) fa[b] = frag(laneA, b, b2, b3);
    }, std::make_integer_sequence<int, 8>{});
    b0 = b2;
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto run_step = [&](auto first_c, auto nw_c, auto st_c, auto ld_c, int P) __attribute__((always_inline)) {
    const int t0 = 2 * P + 4, t1 = 2 * P + 5;          // stages requested by this step: of this tile, or the next tile's first ones
    const bool same0 = t0 < nst, same1 = t1 < nst;
    Desc d0 = nxt, d1 = nxt;
    if (same0) d0 = cur;
    if (same1) d1 = cur;
    step(first_c, nw_c, st_c, ld_c, d0, (same0 ? t0 : t0 - nst) * 64, d1, (same1 ? t1 : t1 - nst) * 64);
  };
  for (;;) {
    if constexpr (PEND) {
      // steps 0 .. F8W_NST - 1 carry the pending stores; the first barrier after an epilogue also counts its direct stores
      // (younger than the pieces it waits for).  vmcnt is a 6-bit counter: a budget beyond 63 is 63 (waits a little early).
      f8w_unroll([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        // in flight at the barrier of step S, where the stage requested in pass 1 of step S - 1 must have landed: the 8 pieces and the
        // stores of pass 0 of step S (+ the epilogue's direct stores at a tile's first barrier)
        constexpr int nw_ = 8 + F8W_SPS + (S == 0 ? F8W_DS + AUXDS : 0);
        constexpr int nw = nw_ < 63 ? nw_ : 63;
        run_step(std::integral_constant<bool, S == 0>{}, F8W_I(nw), F8W_I(S * F8W_SPS), T_{}, S);
      }, std::make_integer_sequence<int, F8W_NST>{});
    } else {
      run_step(T_{}, F8W_I(8), F8W_I(-1), T_{}, 0);
      for (int P = 1; P < F8W_NST; ++P) run_step(F_{}, F8W_I(8), F8W_I(-1), T_{}, P);
    }
    for (int P = F8W_NST; P < nsp - 1; ++P) run_step(F_{}, F8W_I(8), F8W_I(-1), T_{}, P);       // host: nsp >= F8W_NST + 1
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
      if constexpr (Q8 != 0) {                      // one byte per element: a lane's 8 columns are 8 bytes, a column pair 32
        const int64_t bytes = (p.M - cur.m0) * p.ld_q8 - cur.n0;
        eb.rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.q8_out + cur.m0 * p.ld_q8 + cur.n0), 0,
                                                   (unsigned)(bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : bytes), 0x00020000);
        eb.voffQ = (unsigned)((wr * 128 + c_lane) * p.ld_q8 + (wc * 128 + 16 * (g_lane & 1) + 8 * (g_lane >> 1)));
        eb.ldq16 = (int)(p.ld_q8 * 16);
        eb.q_scale = q_scale;
        eb.q_amax = &q_amax;
      }
      // what the epilogue reads first is requested before the last step (older than its LDS-DMA pieces) — except where those 48
      // registers on top of the step's 128 fragment registers end up in scratch (bias + dropout + residual: the reload's
      // s_waitcnt vmcnt(0) inside the epilogue then waits for every store issued so far): there it is requested behind the step
      if constexpr (!EPF_LATE) epi_prefetch<4, EPK, true>(p, lane, cur.m0 + wr * 128, cur.n0 + wc * 128, epf, &eb);
    }
    run_step(F_{}, F8W_I(8), F8W_I(-1), F_{}, nsp - 1);
    if constexpr (PEND && EPF_LATE) epi_prefetch<4, EPK, true>(p, lane, cur.m0 + wr * 128, cur.n0 + wc * 128, epf, &eb);
#if defined(__HIP_DEVICE_COMPILE__)
    // asm MFMAs: the wait states before a VALU may read their results by hand, every accumulator re-defined behind them
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
#endif
    direct_epilogue<4, EPK, PEND, PEND, F8W_PEND_ROWS, Q8>(p, acc, lane, cur.m0 + wr * 128, cur.n0 + wc * 128, pend, &epf, &eb);
    {                                             // the next tile's first pair landed before the last step's barrier / during its pass 1
      const int b1 = ring(b0, 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = frag(laneA, i, b0, b1);
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = frag(laneB, i, b0, b1);
    }
    if constexpr (PEND) {
      if (has_next) rsP = eb.rsC;
      else {                                      // nothing follows: the pending half leaves now
        const __amdgpu_buffer_rsrc_t rl = eb.rsC;
#pragma unroll
        for (int idx = 0; idx < F8W_NPEND; ++idx)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, pend[idx]), rl, voffP,
                                                 (8 - F8W_PEND_ROWS + (idx >> 2)) * ldc16 + (idx & 3) * 64, 0);
      }
    }
    if (!has_next) break;
    cur = nxt;
    v += gridDim.x;
    v_next = v + (int)gridDim.x < nvt ? v + (int)gridDim.x : -1;
    has_next = v_next >= 0;
    nxt = has_next ? make_desc(v_next) : null_desc(cur);
  }
  if constexpr (Q8 != 0) {          // one atomic per wave and launch (non-negative floats order like their bit patterns)
    const float m = wave_max(q_amax);
    if (lane == 0 && m > 0.f) atomicMax((int*)p.q8_amax, __float_as_int(m));
  }
#undef F8W_MF
#undef F8W_I
}

// -1: no instantiation for this (format, epilogue) or shape — the caller keeps the 8-wave kernel
int launch_f8_w4(hipStream_t st, const GemmParams& p_in, int a_format, int n_cus) {
  GemmParams p = p_in;
  const int nsp = (int)(p.K / 128);
  if (p.K % 128 != 0 || nsp < F8W_MAX_PEND_ROWS + 1) return -1;
  const int nvt = p.tiles_m * p.tiles_n;
  dim3 grid((unsigned)(nvt < n_cus ? nvt : n_cus), 1, 1);
  const size_t lds = (size_t)5 * PP_STAGE;
#define LF8W(FA_, E_, X_, Q_, PR_)                                                                           \
  {                                                                                                          \
    auto kern = gemm_f8_w4<FA_, E_, X_, Q_, PR_>;                                                                                       \
    static bool attr_set = false;                                                                            \
    if (!attr_set) {                                                                                         \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                             \
        set_error("gemm_f8_w4: cannot reserve %zu bytes of LDS", lds);                                       \
        return MDT_ERR_LAUNCH;                                                                               \
      }                                                                                                      \
      attr_set = true;                                                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(kern, grid, 256, lds, st, p);                                                         \
    return check_launch("gemm_f8_w4");                                                                       \
  }
  constexpr int E_BIAS = MDT_EPI_BIAS, E_FC1 = MDT_EPI_BIAS | MDT_EPI_GELU | MDT_EPI_AUX_GRAD, E_DFC2 = MDT_EPI_MULAUX | MDT_EPI_COLSUM,
                E_DENSE = MDT_EPI_BIAS | MDT_EPI_RESIDUAL | MDT_EPI_DROPOUT, E_RES = MDT_EPI_RESIDUAL;
  const int e = p.epilogue;
  const bool q8 = p.q8_out != nullptr;
  if (q8 && !(p.q8_scale && p.q8_amax && p.ld_q8 % 8 == 0 && ((uintptr_t)p.q8_out & 7) == 0)) return -1;
  if (a_format == 0) {
    if (e == E_BIAS && !q8) LF8W(0, E_BIAS, 0, 0, 4)
    if (e == E_DENSE && !q8) LF8W(0, E_DENSE, 0, 0, 4)
    if (e == E_FC1 && p.aux && !q8) LF8W(0, E_FC1, 32, 0, 4)
    if (e == E_FC1 && p.aux && q8 && p.q8_fmt == 0) LF8W(0, E_FC1, 64, 1, 4)
  } else {
    if (e == 0 && !q8) LF8W(1, 0, 0, 0, 4)
    if (e == E_RES && !q8) LF8W(1, E_RES, 0, 0, 4)
    if (e == E_DFC2 && !q8) LF8W(1, E_DFC2, 0, 0, 4)
    if (e == E_DFC2 && q8 && p.q8_fmt == 1) LF8W(1, E_DFC2, 32, 2, 4)
  }
#undef LF8W
  return -1;
}

}  // namespace mdt
