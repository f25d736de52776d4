// Register-resident epilogue of the 256 x 256 kernels (gemm.hip: the 8-wave ping-pong and 4-wave forms; gemm_w4r.hip).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "gemm_tiles.hpp"

namespace mdt {

// Register-resident epilogue of the bf16-output ping-pong kernel.  The main loop issues the MFMAs with the
// operands swapped (weights as A, activations as B), so a lane's accumulator acc[i][j] holds ONE output row
// (m = 16 i + (lane & 15)) and four consecutive columns (n = 16 j + 4 (lane >> 4) + r).  One
// v_permlane16_swap per register between the column tiles j and j + 1 turns that into eight consecutive
// columns per lane (odd 16-lane rows of tile j trade places with even rows of tile j + 1), so bias, saved
// pre-activation, residual and output all move as 16-byte vectors covering 64 contiguous bytes of 16 rows
// per wave instruction — no LDS parking, no workgroup barrier, and no wave waits for another one.
//
// EPK >= 0: the epilogue flag set is a compile-time constant (the hot combinations of a training step get their own
// kernel instantiation, see launch_pp256).  The code is then straight-line, which is what lets the vectors the
// epilogue READS — saved GELU derivative, residual — be requested ahead: 8 row groups of the first column pair up
// front, and a row group that has consumed its vector requests the one of the next column pair into the same
// registers (32 VGPRs live).  With runtime flags (EPK = -1) every such load sits in its own branch and the compiler
// follows it with s_waitcnt vmcnt(0), which on this in-order counter also waits for the store of the row group
// before: 16 serialised memory round trips per tile (0.53-0.90 PFLOP/s in situ on the residual / saved-derivative
// GEMMs against 1.05-1.13 on the same shapes without).
// PEND: the outputs of row tiles 4-7 are not stored but handed back packed (pend[(i - 4) * NJP + jp], 16 bytes per lane each):
// the 4-wave kernel keeps them in registers and lets them leave during the next tile's first steps.
#ifndef MDT_W4_PEND_ROWS
#define MDT_W4_PEND_ROWS 4
#endif
constexpr int EPI_PEND_ROWS = MDT_W4_PEND_ROWS;          // PEND epilogues: the last EPI_PEND_ROWS row tiles are handed back, not stored
// What a compile-time epilogue reads before it can start — the bias vectors of the wave's columns and the first column pair's
// residual / saved-derivative vectors — requested ahead (EPF: the 4-wave kernel asks for them before its last 32-k step, whose
// free fragment registers hold them: at the epilogue they have long arrived, and the wait in front of their first use
// covers loads older than the last step's LDS-DMA pieces instead of every operation in flight).
template <int EPK>
constexpr int epi_pre_kind() { return EPK < 0 ? 0 : (EPK & (MDT_EPI_MULAUX | MDT_EPI_DGELU)) ? 1 : (EPK & MDT_EPI_RESIDUAL) ? 2 : (EPK & MDT_EPI_ACCUM) ? 3 : 0; }
template <int NJP>
struct EpiPre { bf16x8 bias[NJP]; bf16x8 pre[8]; };
// C (stores), X (saved-derivative stores), R (the tensor the epilogue reads: saved derivative / residual / C): descriptors based
// at the tile's origin, one 32-bit per-lane offset each, row tile and column pair as immediate offsets
struct EpiBuf { __amdgpu_buffer_rsrc_t rsC, rsX, rsR; unsigned voffC, voffX, voffR; int ldc16, ldx16, ldr16;
                __amdgpu_buffer_rsrc_t rsQ; unsigned voffQ; int ldq16; float q_scale; float* q_amax; };   // Q8 epilogues: the fp8 copy of the output (1 byte per element), its scale, the lane's running |out| maximum
// A 16-byte buffer store whose data registers may be rewritten right behind it.  hipcc (ROCm 7.2) takes a MUBUF store with
// an SGPR soffset to be free of the "store of more than 8 bytes, then VALU write of its data registers" hazard and puts the
// next VALU write directly behind it; on gfx950 that store then wrote the NEW contents of its second dword for the last lanes
// (tests: one 16 x 32 block per tile wrong in two elements of four rows).  The asm holds the data registers for three more
// wait states, whatever the scheduler does.
#ifndef MDT_EPI_STORE_AUX
#define MDT_EPI_STORE_AUX 0
#endif
// cache policy of the output stores (buffer instruction aux bits: 1 = sc0, 2 = nt, 16 = sc1)
constexpr int EPI_STORE_AUX = MDT_EPI_STORE_AUX;
#ifndef MDT_EPI_SAVED_AUX
#define MDT_EPI_SAVED_AUX 0
#endif
constexpr int EPI_SAVED_AUX = MDT_EPI_SAVED_AUX;      // ... of what is saved for backward only (the GELU forward's derivative tensor)
template <int POLICY = EPI_STORE_AUX>
__device__ __forceinline__ void epi_store_vec(bf16x8* ptr, bf16x8 v) {
  if constexpr ((POLICY & 2) != 0) __builtin_nontemporal_store(v, ptr);
  else *ptr = v;
}
template <int POLICY = EPI_STORE_AUX>
__device__ __forceinline__ void epi_store16(i32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, POLICY);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 2" ::"v"(v) : "memory");
#endif
}
template <int NJP, int EPK, bool PEND = false>
__device__ __forceinline__ void epi_prefetch(const GemmParams& p, int lane, int64_t m0w, int64_t n0w, EpiPre<NJP>& f, const EpiBuf* eb = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (PEND) {       // lane-derived address parts are recomputed per tile, not carried through the K loop (see direct_epilogue)
    int z = 0;
    asm volatile("" : "+v"(z));
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
  }
#endif
  const int c = lane & 15, g = lane >> 4;
  constexpr int PRE_KIND = epi_pre_kind<EPK>();
  const int64_t gc0 = n0w + 16 * (g & 1) + 8 * (g >> 1);
  if constexpr (EPK >= 0 && (EPK & MDT_EPI_BIAS) != 0) {
    const bf16_t* bb = (const bf16_t*)p.bias + n0w;                  // uniform base + 32-bit lane offset (no 64-bit lane address to hoist)
    const int lc = 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
    for (int jp = 0; jp < NJP; ++jp) f.bias[jp] = *(const bf16x8*)(bb + (lc + 32 * jp));
  }
  if constexpr (PRE_KIND != 0) {
    const bf16_t* base = PRE_KIND == 1 ? (const bf16_t*)p.aux : PRE_KIND == 2 ? (const bf16_t*)p.residual : (const bf16_t*)p.C;
    const int64_t ld = PRE_KIND == 1 ? p.ldaux : PRE_KIND == 2 ? p.ldr : p.ldc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (PEND) {
        f.pre[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(eb->rsR, eb->voffR, i * eb->ldr16, 0));
      } else {
        int64_t gr = m0w + 16 * i + c;
        gr = gr < p.M ? gr : p.M - 1;
        f.pre[i] = *(const bf16x8*)(base + gr * ld + gc0);
      }
    }
  }
}

// PEND kernels store through buffer descriptors based at the tile's origin (EpiBuf): rows past M fall outside the
// descriptor and are dropped by its bounds check, so there is no per-row-group branch and — what the 4-wave kernel's
// vmcnt arithmetic needs — the NUMBER of stores a wave issues per tile is a constant.
template <int NJP, int EPK = -1, bool PEND = false, bool EPF = false, int PROWS = EPI_PEND_ROWS, int Q8 = 0>   // Q8: 1 / 2 = the output also leaves as e4m3 / e5m2 (EpiBuf::rsQ);   // PROWS: row tiles handed back by a PEND epilogue; NJP pairs of 16-column tiles per wave: 2 (128x64 blocks) or 4 (128x128 blocks)
__device__ __forceinline__ void direct_epilogue(const GemmParams& p, f32x4 (&acc)[8][2 * NJP], int lane, int64_t m0w, int64_t n0w,
                                                bf16x8* pend = nullptr, const EpiPre<NJP>* pf = nullptr, const EpiBuf* eb = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  // everything derived from the lane number is recomputed here, per tile: left alone the compiler hoists the per-lane address
  // parts (64-bit) out of the persistent tile loop, keeps them alive through the K loop and ends up parking them in scratch
  // (the lane number itself too: two mbcnt on an opaque zero instead of one more register carried through the loop)
  if constexpr (PEND) {
    int z = 0;
    asm volatile("" : "+v"(z));
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
  }
#endif
  const int c = lane & 15, g = lane >> 4;
  const int ep = EPK >= 0 ? EPK : p.epilogue;
  constexpr int PRE_KIND = epi_pre_kind<EPK>();
  int64_t gcs[NJP];
  float bias[NJP][8];
#pragma unroll
  for (int jp = 0; jp < NJP; ++jp) {
    gcs[jp] = n0w + 32 * jp + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[jp][e] = 0.f;
    if (ep & MDT_EPI_BIAS) {
      bf16x8 b;
      if constexpr (EPF) b = pf->bias[jp];
      else b = *(const bf16x8*)((const bf16_t*)p.bias + gcs[jp]);
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[jp][e] = (float)b[e];
    }
  }
  float cs[NJP][8];
#pragma unroll
  for (int jp = 0; jp < NJP; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[jp][e] = 0.f;
  auto pre_load = [&](int i, int jp) -> bf16x8 {
    const bf16_t* base = PRE_KIND == 1 ? (const bf16_t*)p.aux : PRE_KIND == 2 ? (const bf16_t*)p.residual : (const bf16_t*)p.C;
    const int64_t ld = PRE_KIND == 1 ? p.ldaux : PRE_KIND == 2 ? p.ldr : p.ldc;
    if constexpr (PEND) {                          // rows past the end read as zeros
      return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(eb->rsR, eb->voffR, i * eb->ldr16 + jp * 64, 0));
    } else {
      int64_t gr = m0w + 16 * i + c;
      gr = gr < p.M ? gr : p.M - 1;                // rows past the end: any valid address, the value is never used
      return *(const bf16x8*)(base + gr * ld + gcs[jp]);
    }
  };
  const int64_t rows_left64 = p.M - m0w;           // uniform over the wave
  const int rows_left = rows_left64 > 0x40000000 ? 0x40000000 : (int)rows_left64;
  bf16x8 pre[8];      // (two column pairs in flight instead of one — 32 more registers, there is room — gain nothing: measured)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if constexpr (EPF && PRE_KIND != 0) pre[i] = pf->pre[i];
    else pre[i] = PRE_KIND ? pre_load(i, 0) : bf16x8{};
  }
  // The row groups are expanded by hand (generic lambda over compile-time indices): hipcc does not unroll a
  // loop around the convergent swap, and a rolled loop indexes the accumulators dynamically = 512 B of scratch per lane.
  auto row_group = [&](auto ic, auto jc) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int jp = decltype(jc)::value;
    const int64_t gr = m0w + 16 * i + c;
    const bool live = 16 * i + c < rows_left;      // 32-bit on purpose: eight hoisted 64-bit row numbers per lane ended up in scratch
    const bf16x8 pv = pre[i];
    if (PRE_KIND && jp + 1 < NJP) pre[i] = pre_load(i, jp + 1);
    do {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // inline asm on purpose: with __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) folds the four swaps of a
        // tile pair into one and broadcasts its result (every 4-column group came out as copies of its first column).
        // s_nop 1 = the two wait states a VALU write of either operand needs before the swap reads it (the compiler
        // may copy the accumulator into the asm operand right before; it cannot see into the string).
        float lo = acc[i][2 * jp][r], hi = acc[i][2 * jp + 1][r];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        v[r] = lo;
        v[4 + r] = hi;
      }
      if constexpr (!PEND) {
        if (!live) break;
      }
      const int64_t gc = gcs[jp];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], p.alpha, bias[jp][e]);
      if ((ep & MDT_EPI_GELU) && (ep & MDT_EPI_AUX_GRAD)) {
        // value and derivative from the same exponential; the dropout scale goes into both, so backward is one multiply
        float sc8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sc8[e] = 1.0f;
        if (ep & MDT_EPI_DROPOUT) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) drop_scale2(p.drop, (uint64_t)gr * p.N + gc + e, sc8[e], sc8[e + 1]);
        }
        bf16x8 dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float cdf, pdf;
          gelu_fast_parts(v[e], cdf, pdf);
          dg[e] = (bf16_t)(__builtin_fmaf(v[e], pdf, cdf) * sc8[e]);
          v[e] = v[e] * cdf * sc8[e];
        }
        if constexpr (PEND) {
          if (p.aux) epi_store16<EPI_SAVED_AUX>(__builtin_bit_cast(i32x4, dg), eb->rsX, eb->voffX, i * eb->ldx16 + jp * 64);
        } else {
          if (p.aux) epi_store_vec<EPI_SAVED_AUX>((bf16x8*)((bf16_t*)p.aux + gr * p.ldaux + gc), dg);
        }
      } else {
      if (ep & MDT_EPI_GELU) {
        if (p.aux) {
          bf16x8 u;
#pragma unroll
          for (int e = 0; e < 8; ++e) { u[e] = (bf16_t)v[e]; v[e] = (float)u[e]; }  // backward differentiates at the stored value
          epi_store_vec((bf16x8*)((bf16_t*)p.aux + gr * p.ldaux + gc), u);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
      }
      if (ep & MDT_EPI_DROPOUT) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float s0, s1;
          drop_scale2(p.drop, (uint64_t)gr * p.N + gc + e, s0, s1);
          v[e] *= s0;
          v[e + 1] *= s1;
        }
      }
      }
      if (ep & MDT_EPI_MULAUX) {
        const bf16x8 u = PRE_KIND == 1 ? pv : *(const bf16x8*)((const bf16_t*)p.aux + gr * p.ldaux + gc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= (float)u[e];
      }
      if (ep & MDT_EPI_DGELU) {
        const bf16x8 u = PRE_KIND == 1 ? pv : *(const bf16x8*)((const bf16_t*)p.aux + gr * p.ldaux + gc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_fast_grad((float)u[e]);
      }
      if (ep & MDT_EPI_RESIDUAL) {
        const bf16x8 r = PRE_KIND == 2 ? pv : *(const bf16x8*)((const bf16_t*)p.residual + gr * p.ldr + gc);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
      }
      if (ep & MDT_EPI_COLSUM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[jp][e] += (!PEND || live) ? v[e] : 0.f;
      }
      bf16_t* cptr = (bf16_t*)p.C + gr * p.ldc + gc;
      if (ep & MDT_EPI_ACCUM) {
        const bf16x8 o = PRE_KIND == 3 ? pv : *(const bf16x8*)cptr;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)o[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      if constexpr (Q8 != 0) {
        // the fp8 copy of the ROUNDED output with the stand-alone quantiser's arithmetic (fp8.hip): what the consumer would
        // have made of the bf16 tensor itself, bit for bit on finite values.  Kept to four VALU operations per element — in a
        // kernel with one wave per SIMD nothing hides the epilogue — so a NaN is not singled out here: it stays visible in
        // the bf16 output (and the loss), the running maximum ignores it, the fp8 copy clamps it.
        constexpr float FMAX = Q8 == 1 ? 448.0f : 57344.0f;
        float q[8];
        float m8 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x0 = (float)o[e];
          m8 = fmaxf(m8, fabsf(x0));
          q[e] = __builtin_amdgcn_fmed3f(x0 * eb->q_scale, -FMAX, FMAX);
        }
        *eb->q_amax = live ? fmaxf(*eb->q_amax, m8) : *eb->q_amax;
        int w0 = 0, w1 = 0;
        if constexpr (Q8 == 1) {
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
        } else {
          w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
        }
        typedef __attribute__((ext_vector_type(2))) int i32x2_;
        __builtin_amdgcn_raw_buffer_store_b64(i32x2_{w0, w1}, eb->rsQ, eb->voffQ, i * eb->ldq16 + jp * 32, 0);
      }
      if constexpr (PEND) {
        if constexpr (i >= 8 - PROWS) pend[(i - (8 - PROWS)) * NJP + jp] = o;
        else epi_store16(__builtin_bit_cast(i32x4, o), eb->rsC, eb->voffC, i * eb->ldc16 + jp * 64);
        break;
      }
      if (ep & (1 << 20)) break;                          // diagnostic: no output store
      if (ep & (1 << 21)) {                               // diagnostic: write-through, do not keep the line in L2
        const i32x4 raw = __builtin_bit_cast(i32x4, o);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(cptr), "v"(raw) : "memory");
      } else {
        epi_store_vec((bf16x8*)cptr, o);
      }
    } while (false);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
  if constexpr (PRE_KIND != 0) {     // column pair by column pair, so the next pair's vectors are a whole pass ahead
    auto column_pair = [&](auto jc) __attribute__((always_inline)) {
      row_group(I0{}, jc); row_group(I1{}, jc); row_group(I2{}, jc); row_group(I3{}, jc);
      row_group(I4{}, jc); row_group(I5{}, jc); row_group(I6{}, jc); row_group(I7{}, jc);
    };
    column_pair(I0{}); column_pair(I1{});
    if constexpr (NJP > 2) { column_pair(I2{}); column_pair(I3{}); }
  } else {
    auto rows = [&](auto ic) __attribute__((always_inline)) {
      row_group(ic, I0{}); row_group(ic, I1{});
      if constexpr (NJP > 2) { row_group(ic, I2{}); row_group(ic, I3{}); }
    };
    rows(I0{}); rows(I1{}); rows(I2{}); rows(I3{}); rows(I4{}); rows(I5{}); rows(I6{}); rows(I7{});
  }
  if (ep & MDT_EPI_COLSUM) {   // the 16 lanes of a row group hold 16 rows of the same columns
    int c2 = c, g2 = g;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PEND && Q8 != 0) {   // the lane number once more: with the fp8 copy's registers on top it is carried across the row groups in
      int z = 0;                       // scratch, and the reload's s_waitcnt vmcnt(0) then waits for every store the epilogue has issued
      asm volatile("" : "+v"(z));
      const int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
      c2 = l2 & 15; g2 = l2 >> 4;
    }
#endif
#pragma unroll
    for (int jp = 0; jp < NJP; ++jp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s_ = row16_sum_dpp(cs[jp][e]);
        // uniform 64-bit base + 32-bit lane offset: the per-lane 64-bit column numbers otherwise live through the whole K loop
        if (c2 == 0) atomicAdd(p.colsum + n0w + (32 * jp + 16 * (g2 & 1) + 8 * (g2 >> 1) + e), s_);
      }
  }
}

}  // namespace mdt
