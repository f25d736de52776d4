// Pieces shared by the GEMM kernels of gemm.hip and gemm_wgrad.hip: launch parameters, LDS stage images (swizzles, LDS-DMA
// issue, fragment reads), the LDS-parked fp32 / vector epilogue, and the 4-wave kernels' fragment / DMA helpers.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace mdt {

struct GemmParams {
  int64_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  int epilogue; float alpha;
  const float* alpha_dev;       // fp8 path: the dequantisation factors 1 / scale_a and 1 / scale_b live on the device
  const float* alpha_dev2;      //           (delayed scaling: no host sync); the output is scaled by their product
  const void* bias; const void* residual; int64_t ldr;
  void* aux; int64_t ldaux;
  int split_k; int64_t k_chunk;
  int tiles_m, tiles_n;
  int group_n;                  // column tiles swept together before moving down the rows (ping-pong kernel's tile order)
  DropCfg drop;
  float* colsum;
  int* tile_queue;              // persistent kernel, dynamic mode: 9 device ints (one head per XCD + finished-workgroup count), all 0 between launches
  unsigned long long* stamps;   // diagnostic (MDT_GEMM_STAMP=1): per workgroup {shader cycles, 100-MHz ticks, k-tiles} of the main loop
  // 8-bit kernel only (gemm_f8.hip): the output ALSO leaves as fp8 for the next 8-bit GEMM — q8_out u8[M, N] = saturate(fmt,
  // bf16(out) * *q8_scale), *q8_amax = max(*q8_amax, max |bf16(out)|): exactly what mdt_fp8_quantize would make of the bf16 output
  void* q8_out = nullptr; int64_t ld_q8 = 0; const float* q8_scale = nullptr; float* q8_amax = nullptr; int q8_fmt = 0;
};

template <typename TIn, typename TOut>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int64_t row, int64_t col, float v) {
  if (row >= p.M || col >= p.N) return;
  v *= p.alpha;
  if (p.epilogue & MDT_EPI_BIAS) v += to_f32(((const TIn*)p.bias)[col]);
  if ((p.epilogue & MDT_EPI_GELU) && (p.epilogue & MDT_EPI_AUX_GRAD)) {
    const float m = (p.epilogue & MDT_EPI_DROPOUT) ? drop_scale(p.drop, (uint64_t)row * p.N + col) : 1.0f;
    if (p.aux) ((TIn*)p.aux)[row * p.ldaux + col] = from_f32<TIn>(gelu_erf_grad(v) * m);
    v = gelu_erf(v) * m;
  } else {
    if (p.epilogue & MDT_EPI_GELU) {
      if (p.aux) ((TIn*)p.aux)[row * p.ldaux + col] = from_f32<TIn>(v);
      // the saved pre-activation is what backward differentiates at: use the rounded value
      if (p.aux) v = to_f32(from_f32<TIn>(v));
      v = gelu_erf(v);
    }
    if (p.epilogue & MDT_EPI_DROPOUT) v *= drop_scale(p.drop, (uint64_t)row * p.N + col);
  }
  if (p.epilogue & MDT_EPI_MULAUX) v *= to_f32(((const TIn*)p.aux)[row * p.ldaux + col]);
  if (p.epilogue & MDT_EPI_DGELU) v *= gelu_erf_grad(to_f32(((const TIn*)p.aux)[row * p.ldaux + col]));
  if (p.epilogue & MDT_EPI_RESIDUAL) v += to_f32(((const TIn*)p.residual)[row * p.ldr + col]);
  if (p.epilogue & MDT_EPI_COLSUM) atomicAdd(p.colsum + col, v);
  TOut* c = (TOut*)p.C + row * p.ldc + col;
  if constexpr (sizeof(TOut) == 4) {
    if (p.epilogue & MDT_EPI_ATOMIC) { atomicAdd((float*)c, v); return; }
  }
  if (p.epilogue & MDT_EPI_ACCUM) v += to_f32(*c);
  *c = from_f32<TOut>(v);
}


// ------------------------------------------------------------------ bf16 128x128x64 kernel
constexpr int T_BM = 128, T_BN = 128, T_BK = 64;
constexpr int T_TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand per stage

// XOR swizzles (see DESIGN.md "LDS images"):
//  k-contiguous tile [128 rows][64 k] (128-B rows, eight 16-B chunks): chunk ^= (row>>1)&7
//    → every ds_read_b128 lane group covers 16 distinct 16-B slots of the 256-B bank row.
//  k-major tile [64 k][128 cols] (256-B rows, eight 32-B blocks): block ^= (k&3)|((k>>3)&1)<<2
//    → the 8 rows a 32-lane half touches in one ds_read_b64_tr_b16 fall on distinct banks.
__device__ __forceinline__ int swz_kc(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_km(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue the LDS-DMA loads of one operand tile of ROWS rows|columns x 64 k (ROWS = 128 or 256):
// ROWS/8 pieces of 1 KiB, dealt round-robin to the NW waves of the block.
//   k-contiguous: piece = 8 rows x 128 B;  k-major: piece = (1024 / (2*ROWS)) k-rows x 2*ROWS B.
template <bool KM, int ROWS, int NW>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, int64_t ld_bytes, int64_t k0, int col0,
                                           char* lds_tile, int wave, int lane) {
  constexpr int NPIECE = ROWS / 8;
#pragma unroll
  for (int i = 0; i < NPIECE / NW; ++i) {
    const int piece = wave + i * NW;
    unsigned voff;
    if constexpr (!KM) {
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ swz_kc(row);
      voff = (unsigned)(row * ld_bytes + (k0 + chunk * 8) * 2);
    } else {
      constexpr int C16 = ROWS / 8;        // 16-B chunks per k-row
      constexpr int KPP = 64 / C16;        // k-rows per piece
      const int k = piece * KPP + lane / C16;
      const int c16 = lane % C16;
      const int blk = (c16 >> 1) ^ swz_km(k);
      voff = (unsigned)((k0 + k) * ld_bytes + (col0 + blk * 16 + (c16 & 1) * 8) * 2);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_tile + piece * 1024), 16, voff, 0, 0, 0);
  }
}

// Fragment of a 16(row|col) x 32(k) block for k-step ks (0/1) of the staged tile.
template <bool KM, int ROWS>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int rc_base, int ks, int lane) {
  if constexpr (!KM) {
    const int row = rc_base + (lane & 15);
    const int chunk = (ks * 4 + (lane >> 4)) ^ swz_kc(row);
    return *(const bf16x8*)(lds_tile + row * 128 + chunk * 16);
  } else {
    constexpr int RB = ROWS * 2;           // bytes per k-row
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int blk = rc_base >> 4;
    const int k_lo = ks * 32 + g * 8 + q;
    const int k_hi = k_lo + 4;
    const bf16x4 lo = lds_read_tr16((const bf16_t*)(lds_tile + k_lo * RB + ((blk ^ swz_km(k_lo)) * 32) + pp * 8));
    const bf16x4 hi = lds_read_tr16((const bf16_t*)(lds_tile + k_hi * RB + ((blk ^ swz_km(k_hi)) * 32) + pp * 8));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

// ---- 32-k stage images of the ping-pong kernel ----------------------------------------------------------
// k-contiguous operands: [ROWS][32 k] with 64-byte rows, a 1-KiB LDS-DMA piece = 16 rows, 16-byte chunk
// XOR-swizzled by {0,2,3,1}[(row >> 2) & 3] (conflict-free ds_read_b128, checked for all four lane groups).
// k-major operands: [32 k][ROWS] with 2*ROWS-byte rows, a piece = 1024 / (2*ROWS) k-rows, 32-byte blocks
// swizzled by swz_km(k) as in the 64-k image (conflict-free ds_read_b64_tr_b16).
__device__ __forceinline__ int swz_h(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// AUX: cache-policy bits of the LDS-DMA request (1 = sc0, 2 = nt, 16 = sc1); MDT_GEMM_A_AUX (build-time, experiments) sets it for
// the A operand of the persistent forward / input-gradient kernels — the streamed activation panels, read once per column group
#ifndef MDT_GEMM_A_AUX
#define MDT_GEMM_A_AUX 0
#endif
template <bool KM, int ROWS, int NW, int AUX = 0>
__device__ __forceinline__ void stage_step(__amdgpu_buffer_rsrc_t rs, int64_t ld_bytes, int64_t k0, int col0,
                                           char* lds_tile, int wave, int lane) {
  constexpr int NPIECE = ROWS / 16;    // 1-KiB pieces per 32-k stage of one operand
#pragma unroll
  for (int i = 0; i < NPIECE / NW; ++i) {
    const int piece = wave + i * NW;
    unsigned voff;
    if constexpr (!KM) {
      const int row = piece * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ swz_h(row);
      voff = (unsigned)(row * ld_bytes + (k0 + chunk * 8) * 2);
    } else {
      constexpr int C16 = ROWS / 8;        // 16-B chunks per k-row
      constexpr int KPP = 64 / C16;        // k-rows per piece
      const int k = piece * KPP + lane / C16;
      const int c16 = lane % C16;
      const int blk = (c16 >> 1) ^ swz_km(k);
      voff = (unsigned)((k0 + k) * ld_bytes + (col0 + blk * 16 + (c16 & 1) * 8) * 2);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_tile + piece * 1024), 16, voff, 0, 0, AUX);
  }
}

// fragment of a 16(row|col) x 32(k) block of a 32-k stage
template <bool KM, int ROWS>
__device__ __forceinline__ bf16x8 load_frag_h(const char* lds_tile, int rc_base, int lane) {
  if constexpr (!KM) {
    const int row = rc_base + (lane & 15);
    return *(const bf16x8*)(lds_tile + row * 64 + (((lane >> 4) ^ swz_h(row)) * 16));
  } else {
    return load_frag<true, ROWS>(lds_tile, rc_base, 0, lane);
  }
}

// Epilogue of the tile kernel: each wave parks its 64x64 fp32 accumulator block in LDS
// (the staging buffers are free once the K loop is done), then walks it row-wise so that
// every global access is a full 16-byte vector and a wave instruction covers whole 128-B
// (bf16) row segments — bias / GELU / residual / pre-activation traffic is coalesced too.
// Split-K weight gradients leave with one 256-byte contiguous atomic instruction per row.
template <typename TOut>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], char* smem, int wave, int lane,
                                              int64_t m0w, int64_t n0w) {
  __syncthreads();  // every wave is done reading the last staged tile
  float* ws = (float*)(smem + wave * 16384);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ws[(i * 16 + (lane >> 4) * 4 + r) * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int ep = p.epilogue;
  if constexpr (sizeof(TOut) == 4) {
    if (ep & MDT_EPI_ATOMIC) {
      if (ep & (1 << 20)) return;                 // diagnostic (MDT_GEMM_DIAG=1): no output — what the atomics' tail costs a launch
      float* c = (float*)p.C + n0w + lane;
      for (int row = 0; row < 64; ++row) {
        const int64_t gr = m0w + row;
        if (gr < p.M) atomicAdd(c + gr * p.ldc, p.alpha * ws[row * 64 + lane]);
      }
      return;
    }
  }
  const int c8 = (lane & 7) * 8;
  const int64_t gc = n0w + c8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = 0.f;
  if (ep & MDT_EPI_BIAS) {
    const bf16x8 b = *(const bf16x8*)((const bf16_t*)p.bias + gc);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (float)b[e];
  }
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 8 + (lane >> 3);
    const int64_t gr = m0w + row;
    if (gr >= p.M) continue;
    const f32x4 lo = *(const f32x4*)(ws + row * 64 + c8);
    const f32x4 hi = *(const f32x4*)(ws + row * 64 + c8 + 4);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], p.alpha, bias[e]);
    if ((ep & MDT_EPI_GELU) && (ep & MDT_EPI_AUX_GRAD)) {
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
      if (p.aux) *(bf16x8*)((bf16_t*)p.aux + gr * p.ldaux + gc) = dg;
    } else {
    if (ep & MDT_EPI_GELU) {
      if (p.aux) {
        bf16x8 u;
#pragma unroll
        for (int e = 0; e < 8; ++e) { u[e] = (bf16_t)v[e]; v[e] = (float)u[e]; }  // backward differentiates at the stored value
        *(bf16x8*)((bf16_t*)p.aux + gr * p.ldaux + gc) = u;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
    }
    if (ep & MDT_EPI_DROPOUT) {   // N is a multiple of 128 and gc of 8: the 8 counters start even
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
      const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + gr * p.ldaux + gc);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= (float)u[e];
    }
    if (ep & MDT_EPI_DGELU) {
      const bf16x8 u = *(const bf16x8*)((const bf16_t*)p.aux + gr * p.ldaux + gc);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gelu_fast_grad((float)u[e]);
    }
    if (ep & MDT_EPI_RESIDUAL) {
      const bf16x8 r = *(const bf16x8*)((const bf16_t*)p.residual + gr * p.ldr + gc);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    if (ep & MDT_EPI_COLSUM) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += v[e];
    }
    if constexpr (sizeof(TOut) == 4) {
      float* c = (float*)p.C + gr * p.ldc + gc;
      if (ep & MDT_EPI_ACCUM) {
        const f32x4 o0 = *(const f32x4*)c, o1 = *(const f32x4*)(c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
      }
      *(f32x4*)c = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(c + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      bf16_t* c = (bf16_t*)p.C + gr * p.ldc + gc;
      if (ep & MDT_EPI_ACCUM) {
        const bf16x8 o = *(const bf16x8*)c;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)o[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *(bf16x8*)c = o;
    }
  }
  if (ep & MDT_EPI_COLSUM) {   // 8 row-lanes per column group -> one atomic per column per wave
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s_ = cs[e];
      s_ += __shfl_xor(s_, 8, 64);
      s_ += __shfl_xor(s_, 16, 64);
      s_ += __shfl_xor(s_, 32, 64);
      if (lane < 8) atomicAdd(p.colsum + gc + e, s_);
    }
  }
}


constexpr int PP_STAGE = 2 * 256 * 64;   // 32 KiB: one 32-k ring stage of the 256 x 256 kernels (A 256 rows x 32 k, B 256 cols x 32 k)

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ------------------------------------------------------------------ 4-wave form of the persistent kernel
// One wave per SIMD with the whole register file (256 accumulators in AGPRs + 256 VGPRs), 128 x 128 of the 256 x 256 tile
// per wave: a third less LDS read traffic than the 8-wave form (64 KiB instead of 96 per 32-k step) and room to keep a
// finished tile in registers.  One instruction stream has to carry MFMAs, fragment reads and LDS-DMA issue together; left to
// the compiler that loop takes 4 900 cycles per K-tile (tools/probes/gemm4w_probe.hip), so the MFMAs are volatile asm
// statements with memory clobbers and SOURCE ORDER IS ISSUE ORDER: per 32-k step eight blocks of
//     2 MFMA, fragment read, 2 MFMA, fragment read, 2 MFMA, LDS-DMA piece, 2 MFMA
// on one fragment set while the reads fill the other (2 400 cycles per K-tile in that probe, L2-resident operands, no
// epilogue).  Ring, prefetch distance, tile walk and epilogue are those of gemm_bf16_pp256p; one barrier per step.
// Fragment of a 32-k stage for the 4-wave kernel.  k-contiguous operands: the plain 16-byte read.  k-major operands: the two
// ds_read_b64_tr_b16 halves as ASM — in front of the builtin the compiler puts s_waitcnt vmcnt(0) (an LDS read that might
// alias what an LDS-DMA is still writing), which drains the whole prefetch ring once per fragment; the kernel orders its
// reads behind the stage's arrival itself (counted vmcnt + barrier at the top of every step) and waits for them with its
// own lgkmcnt(0) before the step that consumes them.
typedef __attribute__((ext_vector_type(2))) int i32x2;
template <bool KM, int ROWS>
__device__ __forceinline__ bf16x8 w4_frag(const char* lds_tile, int rc_base, int lane) {
  if constexpr (!KM) {
    return load_frag_h<false, ROWS>(lds_tile, rc_base, lane);
  } else {
    constexpr int RB = ROWS * 2;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int blk = rc_base >> 4, k_lo = g * 8 + q;
    const unsigned addr = (unsigned)(size_t)LDS_PTR(lds_tile + k_lo * RB + ((blk ^ swz_km(k_lo)) * 32) + pp * 8);
    i32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(4 * RB) : "memory");
    return __builtin_bit_cast(bf16x8, i32x4{lo[0], lo[1], hi[0], hi[1]});
  }
}

template <int AUX = 0>
__device__ __forceinline__ void w4_dma(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, soff, 0, AUX);
}


// 4-wave split-K kernel for fp32-accumulating launches (weight gradients), gemm_wgrad.hip
int launch_w4s(hipStream_t st, const GemmParams& p, int ta, int tb);
// 4-wave 8-bit kernel on the 16x16x128 block MFMA, gemm_f8.hip; -1: no instantiation for this (format, epilogue, shape)
int launch_f8_w4(hipStream_t st, const GemmParams& p, int a_format, int n_cus);

}  // namespace mdt
