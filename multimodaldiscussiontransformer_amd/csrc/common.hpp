// Shared device/host helpers for the gfx950 (CDNA4) kernels of the mDT hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mdt_hip.h"

namespace mdt {

constexpr int WAVE = 64;

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// ---------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);
#define MDT_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ::mdt::set_error(__VA_ARGS__);      \
      return MDT_ERR_ARG;                 \
    }                                     \
  } while (0)
#define MDT_UNSUPPORTED(...)              \
  do {                                    \
    ::mdt::set_error(__VA_ARGS__);        \
    return MDT_ERR_UNSUPPORTED;           \
  } while (0)
int check_launch(const char* what);

// Environment switches (tuning / diagnostics; production runs set none of them).  Read ONCE, at the first launch that
// asks — the hot path makes no getenv calls — and again only when the host calls mdt_reload_env() (tests, A/B tools).
struct Switches {
  int gemm_pp_dist;          // MDT_GEMM_PP_DIST   (default 4)
  int gemm_persist;          // MDT_GEMM_PERSIST   (default 1: persistent tile walk for K loops of at least 16 steps; 0: never; 2: for every K >= 128 — stress runs)
  bool gemm_dynamic;         // MDT_GEMM_DYNAMIC   (default 0): dynamic tile queue, needs mdt_gemm_set_tile_queue
  int gemm_group;            // MDT_GEMM_GROUP     (-1: unset)
  bool gemm_stamp;           // MDT_GEMM_STAMP
  bool gemm_no_spec;         // MDT_GEMM_NO_SPEC
  int gemm_diag;             // MDT_GEMM_DIAG      (0: none)
  char gemm_tile[16];        // MDT_GEMM_TILE      ("" unset)
  bool gemm_no_pp;           // MDT_GEMM_NO_PP
  int gemm_f8w;              // MDT_GEMM_F8W       (default 1: 8-bit GEMMs on the 16x16x128 block-MFMA kernel where it has an instantiation; 0: the 8-wave kernel)
  int gemm_w4;               // MDT_GEMM_W4        (default 2: the 4-wave persistent kernel where it is measured faster; 0: never; 1: every persistent launch)
  bool attn_v1;              // MDT_ATTN_V1
  char attn_bwd[8];          // MDT_ATTN_BWD       ("" unset)
  bool attn_no_occ4;         // MDT_ATTN_NO_OCC4
  bool attn_no_w8;           // MDT_ATTN_NO_W8
  bool attn_exact_delta;     // MDT_ATTN_EXACT_DELTA (default 1: one-pass backward of rows <= 96 tokens sums delta = sum P o dP itself; 0: rowsum(dO o O) from the bf16 output)
  int attn_onepass;          // MDT_ATTN_ONEPASS   (0: two-pass backward kernels only; default: one-pass wherever its dS image fits LDS)
  bool ln_generic;           // MDT_LN_GENERIC     (default 0; 1: bf16 rows of 768 take the generic LayerNorm backward kernel — tests, A/B runs)
  int ln_bwd_wgs;            // MDT_LN_BWD_WGS     (workgroups a LayerNorm backward launch aims for; tuning)
};
const Switches& switches();

// ---------------------------------------------------------------- scalar conversions
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// ---------------------------------------------------------------- wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce across the 4 lanes l, l+16, l+32, l+48 (rows of a 16x16 MFMA accumulator column group)
__device__ __forceinline__ float quad16_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// reduce across the 16 lanes that share lane>>4 (one accumulator row across its 16 columns)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the same sum through DPP row rotations (a DPP row = 16 lanes): no LDS permutes, no lane numbers to keep alive — every lane
// of the row ends up with the total (summation order differs from row16_sum's butterfly in the last bit)
__device__ __forceinline__ float row16_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));   // row_ror:1
  return v;
}
// Sum over the 64 lanes, every lane gets the total, without a trip through the LDS crossbar (__shfl_xor = ds_bpermute_b32: six
// dependent LDS round trips per sum): four DPP row rotations leave each 16-lane row with its own total, row_bcast:15 / row_bcast:31
// carry the row totals upwards (lane 63 ends up with all four), v_readlane hands it to everybody as a scalar operand.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum_dpp(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16 epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution) on the raw
// v_rcp_f32 / v_exp_f32 instructions — ~15 VALU issues instead of erff's ~34, and GELU' reuses the same
// exponential (exp(-z^2) with z = |x|/sqrt(2) is the Gaussian pdf factor).  The fp32 parity path keeps erff.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& pdf) {
  // explicit FMAs: the build runs with -ffp-contract=off (the fp32 parity path wants separately rounded mul / add),
  // which would otherwise turn this Horner chain into 2 instructions per term
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  poly = __builtin_fmaf(t, poly, 1.421413741f);
  poly = __builtin_fmaf(t, poly, -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.254829592f);
  const float erf_abs = __builtin_fmaf(-(poly * t), e, 1.0f);
  cdf = __builtin_fmaf(0.5f, copysignf(erf_abs, x), 0.5f);
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float cdf, pdf;
  gelu_fast_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
  float cdf, pdf;
  gelu_fast_parts(x, cdf, pdf);
  return cdf + x * pdf;
}

// ---------------------------------------------------------------- dropout RNG
// Counter-based: the keep/drop decision of element `idx` of dropout site `seed` is a pure
// function of (seed, idx), so backward regenerates the forward mask without storing it.
// One 32-bit word serves the two counters 2i and 2i+1 (16 random bits each); p is realised as
// round(p * 65536) / 65536 (relative bias of the 1/(1-p) scale < 2e-5).
// The mixer avoids v_mul_lo_u32 (quarter rate on CDNA): three rounds of
//   x = (x & 0xFFFFFF) * C + rotl(x, r);  x ^= x >> s          (v_alignbit + v_mad_u32_u24 + shift + xor)
// = 12 full-rate VALU instructions per pair of elements.  Avalanche bias <= 0.02 on random and on
// sequential inputs, lag-1/2/768/3072 autocorrelation of the keep masks < 1e-3 (tools/hash_eval.py).
__device__ __forceinline__ uint32_t drop_mix(uint32_t x) {
  x = __umul24(x, 0x95F24Du) + __builtin_rotateleft32(x, 15); x ^= x >> 15;
  x = __umul24(x, 0xC2B2AEu) + __builtin_rotateleft32(x, 13); x ^= x >> 13;
  x = __umul24(x, 0x85EBCBu) + __builtin_rotateleft32(x, 17); x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t key;      // 64-bit site seed folded to 32 bits on the host
  uint32_t thresh;   // drop when the counter's 16 bits < thresh
  float inv_keep;    // 1 / (1 - p)
};
// sites hold fewer than 2^33 elements (checked by the host entry points), so idx >> 1 fits 32 bits
__device__ __forceinline__ float drop_scale(const DropCfg& d, uint64_t idx) {
  const uint32_t h = drop_mix((uint32_t)(idx >> 1) ^ d.key);
  const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
  return bits >= d.thresh ? d.inv_keep : 0.f;
}
// counters idx_even and idx_even + 1 (idx_even must be even)
__device__ __forceinline__ void drop_scale2(const DropCfg& d, uint64_t idx_even, float& s0, float& s1) {
  const uint32_t h = drop_mix((uint32_t)(idx_even >> 1) ^ d.key);
  s0 = (h & 0xFFFFu) >= d.thresh ? d.inv_keep : 0.f;
  s1 = (h >> 16) >= d.thresh ? d.inv_keep : 0.f;
}
constexpr int64_t DROP_MAX_ELEMS = (int64_t)1 << 33;
static inline DropCfg make_drop(float p, uint64_t seed) {
  DropCfg d;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;      // splitmix64 finaliser: nearby seeds -> unrelated keys
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  d.key = (uint32_t)(z ^ (z >> 32));
  d.thresh = (uint32_t)(p * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}

// ---------------------------------------------------------------- MFMA wrappers (16x16 tiles)
// C/D layout (all dtypes): col = lane & 15, row = (lane >> 4) * 4 + reg.
// f32:  A[row = lane&15][k = lane>>4],            B[k = lane>>4][col = lane&15]        (K = 4)
// bf16: A[row = lane&15][k = 8*(lane>>4) + j],    B[k = 8*(lane>>4) + j][col = lane&15] (K = 32)
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS address-space casts
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is
// delivered column-major — lane i of the group receives column i of the 4 rows
// (cdna_hip_programming.md T10).  Lane 4q+p supplies the address of row q, cols 4p..4p+3.
__device__ __forceinline__ bf16x4 lds_read_tr16(const bf16_t* lds_addr) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_addr));
  return __builtin_bit_cast(bf16x4, r);
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T> struct DType;
template <> struct DType<float> { static constexpr int id = MDT_F32; };
template <> struct DType<bf16_t> { static constexpr int id = MDT_BF16; };

static inline size_t dtype_size(int dt) { return dt == MDT_BF16 ? 2 : 4; }

}  // namespace mdt
