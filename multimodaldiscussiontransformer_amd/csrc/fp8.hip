// Per-tensor-scaled fp8 operands for the forward / input-gradient GEMMs (BASELINE.json configs[4]): quantisation of
// bf16 / fp32 tensors to OCP e4m3 (activations, weights) or e5m2 (gradients) with a device-resident scale, the running
// |x| maximum that the NEXT step's scale is derived from (delayed scaling: no host synchronisation anywhere), and the
// scale bookkeeping.  The GEMM itself is the 8-bit instantiation of the persistent 256x256 kernel (gemm.hip).
// HBM-bound: 2 (or 4) bytes read + 1 byte written per element, 16-byte loads, 8-byte stores.
#include "common.hpp"

namespace mdt {

template <typename T, int FMT>     // FMT 0: e4m3 (|x| <= 448), 1: e5m2 (|x| <= 57344)
__global__ __launch_bounds__(256) void fp8_quantize_kernel(int64_t rows, int64_t cols, const T* src, int64_t ld_src, uint8_t* dst,
                                                           int64_t ld_dst, const float* scale_dev, float* amax_dev) {
  constexpr float FMAX = FMT == 0 ? 448.0f : 57344.0f;
  const float scale = scale_dev ? *scale_dev : 1.0f;
  const int64_t c8 = cols >> 3;                        // 8 elements per thread step
  float amax = 0.f;
  const bool flat = ld_src == cols && ld_dst == cols;      // contiguous tensors: no row / column split (a 64-bit division per 8 elements)
  const unsigned c8u = (unsigned)c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * c8; i += (int64_t)gridDim.x * 256) {
    int64_t r, c;
    if (flat) { r = 0; c = i << 3; }
    else if (rows * c8 < (1ll << 32)) { const unsigned q = (unsigned)i / c8u; r = q; c = (int64_t)((unsigned)i - q * c8u) << 3; }
    else { r = i / c8; c = (i - r * c8) << 3; }
    float v[8];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 x = *(const bf16x8*)(src + r * ld_src + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
    } else {
      const f32x4 x0 = *(const f32x4*)(src + r * ld_src + c), x1 = *(const f32x4*)(src + r * ld_src + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
    }
    int w0 = 0, w1 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // fmaxf / fminf drop a NaN operand: a diverged activation must stay visible — it goes through as NaN (the fp8 NaN
      // code, which the MFMA propagates into the product and on to the loss, as the bf16 path would) and it marks the
      // running maximum as +inf, which fp8_scale_update leaves the scale alone for
      const float x = v[e] * scale;
      const bool bad = !(fabsf(v[e]) <= 3.0e38f);      // NaN or infinity
      amax = bad ? __builtin_inff() : fmaxf(amax, fabsf(v[e]));
      v[e] = (x != x) ? x : fminf(fmaxf(x, -FMAX), FMAX);   // saturate: the formats have no room above FMAX (e4m3fn: NaN)
    }
    if constexpr (FMT == 0) {
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
    } else {
      w0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], w1, true);
    }
    *(int2*)(dst + r * ld_dst + c) = make_int2(w0, w1);
  }
  if (amax_dev) {
    // one atomic per WORKGROUP (same-address atomics serialise: 32 k of them cost more than the whole pass)
    __shared__ float s_max[4];
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      // non-negative floats order like their bit patterns
      if (amax > 0.f) atomicMax((int*)amax_dev, __float_as_int(amax));
    }
  }
}

// scale[i] = fmax[i] / (amax[i] * margin) (1 when the tensor was all zero), inv_scale[i] = 1 / scale[i]; amax[i] is reset
__global__ void fp8_scale_update_kernel(int n, float* amax, float* scale, float* inv_scale, const float* fmt_max, float margin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = amax[i];
  if (a > 0.f && isfinite(a)) {
    const float s = fmt_max[i] / (a * margin);
    scale[i] = s;
    inv_scale[i] = 1.0f / s;
  }
  amax[i] = 0.f;
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_fp8_quantize(void* stream, int src_dtype, int fmt, int64_t rows, int64_t cols, const void* src, int64_t ld_src,
                                void* dst, int64_t ld_dst, const float* scale_dev, float* amax_dev) {
  if (rows == 0 || cols == 0) return MDT_OK;
  MDT_CHECK_ARG(src && dst, "fp8_quantize: null pointer");
  MDT_CHECK_ARG(src_dtype == MDT_F32 || src_dtype == MDT_BF16, "fp8_quantize: bad source dtype %d", src_dtype);
  MDT_CHECK_ARG(fmt == 0 || fmt == 1, "fp8_quantize: format %d (0 = e4m3, 1 = e5m2)", fmt);
  MDT_CHECK_ARG(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0,
                "fp8_quantize: rows must be 16-byte (source) / 8-byte (destination) vectorisable");
  hipStream_t st = (hipStream_t)stream;
  const int64_t work = rows * (cols >> 3);
  const unsigned grid = (unsigned)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);     // 8 workgroups per CU, grid-stride
#define Q_(T, F) hipLaunchKernelGGL((fp8_quantize_kernel<T, F>), grid, 256, 0, st, rows, cols, (const T*)src, ld_src, (uint8_t*)dst, ld_dst, scale_dev, amax_dev)
  if (src_dtype == MDT_BF16) { if (fmt == 0) Q_(bf16_t, 0); else Q_(bf16_t, 1); }
  else { if (fmt == 0) Q_(float, 0); else Q_(float, 1); }
#undef Q_
  return check_launch("fp8_quantize");
}

extern "C" int mdt_fp8_scale_update(void* stream, int n, float* amax, float* scale, float* inv_scale, const float* fmt_max, float margin) {
  if (n == 0) return MDT_OK;
  MDT_CHECK_ARG(amax && scale && inv_scale && fmt_max && margin > 0.f, "fp8_scale_update: bad arguments");
  hipLaunchKernelGGL(fp8_scale_update_kernel, (n + 255) / 256, 256, 0, (hipStream_t)stream, n, amax, scale, inv_scale, fmt_max, margin);
  return check_launch("fp8_scale_update");
}
