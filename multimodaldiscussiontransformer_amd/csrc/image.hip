// Image front end on the device: decoded RGB bytes in, ViT pixel tensors out.
// Replaces the per-image ViTImageProcessor call of the reference's dataset builder
// (experiments/hateful_discussions/datasets/hateful_discussions.py:168-184: PIL bilinear resize to 224, x 1/255,
// (x - 0.5) / 0.5) for a whole batch: images cross PCIe as uint8 at their own sizes (a C2 batch: tens of MB instead
// of 308 MB of fp32 pixels) and two integer passes reproduce PIL's resampling byte for byte —
//   horizontal:  tmp[y][xo][c] = clip8((2^21 + sum_k kh[xo][k] * src[y][x0(xo) + k][c]) >> 22)        (H_in x out x 3 bytes)
//   vertical:    out[c][yo][xo] = lut[c][clip8((2^21 + sum_k kv[yo][k] * tmp[y0(yo) + k][xo][c]) >> 22)]
// with the 22-bit coefficients of mdt_resize_plan (host.cpp) and the 3 x 256 rescale / normalise table of
// mdt_image_norm_lut.  HBM-bound byte work: every source byte is read once per pass (neighbouring outputs share
// taps through L1 / L2), one thread per output pixel (three channels), coalesced along x.
#include "common.hpp"

namespace mdt {

// desc[i] = {pixel offset (bytes), tmp offset (bytes), H, W, horizontal plan offset, kh, vertical plan offset, kv}
// plan (int32): at an offset, 2 * out_size bounds (first tap, tap count) followed by out_size * k coefficients
__global__ __launch_bounds__(256) void image_resize_h_kernel(const uint8_t* pix, const int64_t* desc, const int32_t* plan, uint8_t* tmp, int out_size) {
  const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
  const int H = (int)d[2], W = (int)d[3], kh = (int)d[5];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)H * out_size) return;
  const int y = (int)(t / out_size), xo = (int)(t - (int64_t)y * out_size);
  const int32_t* bounds = plan + d[4];
  const int32_t* kk = bounds + 2 * out_size + (int64_t)xo * kh;
  const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
  const uint8_t* row = pix + d[0] + ((int64_t)y * W + x0) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int k = 0; k < n; ++k) {
    const int c = kk[k];
    s0 += c * (int)row[3 * k];
    s1 += c * (int)row[3 * k + 1];
    s2 += c * (int)row[3 * k + 2];
  }
  uint8_t* o = tmp + d[1] + ((int64_t)y * out_size + xo) * 3;
  o[0] = (uint8_t)min(255, max(0, s0 >> 22));
  o[1] = (uint8_t)min(255, max(0, s1 >> 22));
  o[2] = (uint8_t)min(255, max(0, s2 >> 22));
}

template <typename TOut>
__global__ __launch_bounds__(256) void image_resize_v_kernel(const int64_t* desc, const int32_t* plan, const uint8_t* tmp, const float* lut,
                                                              TOut* out, uint8_t* out_u8, int out_size) {
  const int64_t* d = desc + 8 * (int64_t)blockIdx.y;
  const int kv = (int)d[7];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= out_size * out_size) return;
  const int yo = t / out_size, xo = t - yo * out_size;
  const int32_t* bounds = plan + d[6];
  const int32_t* kk = bounds + 2 * out_size + (int64_t)yo * kv;
  const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
  const uint8_t* col = tmp + d[1] + ((int64_t)y0 * out_size + xo) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int k = 0; k < n; ++k) {
    const int c = kk[k];
    const uint8_t* p = col + (int64_t)k * out_size * 3;
    s0 += c * (int)p[0];
    s1 += c * (int)p[1];
    s2 += c * (int)p[2];
  }
  const int u0 = min(255, max(0, s0 >> 22)), u1 = min(255, max(0, s1 >> 22)), u2 = min(255, max(0, s2 >> 22));
  const int64_t plane = (int64_t)out_size * out_size;
  if (out) {
    TOut* o = out + (int64_t)blockIdx.y * 3 * plane + t;
    o[0] = from_f32<TOut>(lut[u0]);
    o[plane] = from_f32<TOut>(lut[256 + u1]);
    o[2 * plane] = from_f32<TOut>(lut[512 + u2]);
  }
  if (out_u8) {      // the resized bytes themselves (HWC), for tests against PIL
    uint8_t* o = out_u8 + ((int64_t)blockIdx.y * plane + t) * 3;
    o[0] = (uint8_t)u0; o[1] = (uint8_t)u1; o[2] = (uint8_t)u2;
  }
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_image_preprocess(void* stream, int n_images, int max_h, const uint8_t* pixels, const int64_t* desc, const int32_t* plan,
                                    uint8_t* tmp, const float* lut, int out_dtype, void* out, uint8_t* out_u8, int out_size) {
  if (n_images == 0) return MDT_OK;
  MDT_CHECK_ARG(n_images > 0 && max_h > 0 && out_size > 0 && out_size <= 4096, "mdt_image_preprocess: bad sizes");
  MDT_CHECK_ARG(pixels && desc && plan && tmp && lut && (out || out_u8), "mdt_image_preprocess: null pointer");
  MDT_CHECK_ARG(out_dtype == MDT_F32 || out_dtype == MDT_BF16, "mdt_image_preprocess: out dtype %d", out_dtype);
  hipStream_t st = (hipStream_t)stream;
  const int64_t hwork = (int64_t)max_h * out_size;
  dim3 gh((unsigned)((hwork + 255) / 256), (unsigned)n_images);
  hipLaunchKernelGGL(image_resize_h_kernel, gh, 256, 0, st, pixels, desc, plan, tmp, out_size);
  dim3 gv((unsigned)((out_size * out_size + 255) / 256), (unsigned)n_images);
  if (out_dtype == MDT_F32) hipLaunchKernelGGL((image_resize_v_kernel<float>), gv, 256, 0, st, desc, plan, (const uint8_t*)tmp, lut, (float*)out, out_u8, out_size);
  else hipLaunchKernelGGL((image_resize_v_kernel<bf16_t>), gv, 256, 0, st, desc, plan, (const uint8_t*)tmp, lut, (bf16_t*)out, out_u8, out_size);
  return check_launch("image_preprocess");
}
