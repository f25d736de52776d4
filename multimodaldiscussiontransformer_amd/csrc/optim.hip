// Fused Adam step (FairSeq `adam` semantics: decoupled weight decay applied as p -= wd*lr*p, bias-corrected
// step size lr*sqrt(1-b2^t)/(1-b1^t), denominator sqrt(v)+eps) over one parameter tensor: reads the fp32
// gradient accumulated by the backward kernels, updates fp32 moments and the fp32 master copy, and writes
// the working-precision parameter — one HBM pass (28-30 B per element) instead of ~10 eager elementwise ops.
// Launch flags of the reference: --optimizer adam --adam-betas '(0.9, 0.999)' --adam-eps 1e-8
// --weight-decay 0.01 (mDT/experiments/hateful_discussions/run_train.sh:38).
#include "common.hpp"

namespace mdt {

template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(int64_t n, T* __restrict__ param, float* __restrict__ master,
                                                   const float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, float lr, float beta1, float beta2, float eps,
                                                   float wd, float step_size, const float* __restrict__ grad_scale) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = grad[i] * gs;
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    float p = master ? master[i] : to_f32(param[i]);
    p -= wd * lr * p;
    p -= step_size * mi / (sqrtf(vi) + eps);
    m[i] = mi;
    v[i] = vi;
    if (master) master[i] = p;
    param[i] = from_f32<T>(p);
  }
}

// Multi-tensor form: one launch walks a device table of tensors (the ~400 parameter tensors of mDT would otherwise
// cost ~400 launches of 10-20 us each).  Block b works on chunk b of 4096 elements; `chunk_first[t]` is the first
// chunk of tensor t (monotone, chunk_first[n] = total), found by binary search.
template <typename T>
__global__ __launch_bounds__(256) void adam_multi_kernel(int n_tensors, const mdt_adam_tensor* __restrict__ tab,
                                                         const int64_t* __restrict__ chunk_first, float lr, float beta1,
                                                         float beta2, float eps, float wd, float step_size,
                                                         const float* __restrict__ grad_scale) {
  const float gs = grad_scale ? grad_scale[0] : 1.0f;
  const int64_t total = chunk_first[n_tensors];
  for (int64_t chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (chunk_first[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    const mdt_adam_tensor t = tab[lo];
    const int64_t base = (chunk - chunk_first[lo]) * 4096;
    T* param = (T*)t.param;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      if (i >= t.numel) break;
      const float g = t.grad[i] * gs;
      const float mi = beta1 * t.m[i] + (1.0f - beta1) * g;
      const float vi = beta2 * t.v[i] + (1.0f - beta2) * g * g;
      float p = t.master ? t.master[i] : to_f32(param[i]);
      p -= wd * lr * p;
      p -= step_size * mi / (sqrtf(vi) + eps);
      t.m[i] = mi;
      t.v[i] = vi;
      if (t.master) t.master[i] = p;
      param[i] = from_f32<T>(p);
    }
  }
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_adam_step_multi(void* stream, int dtype, int n_tensors, const mdt_adam_tensor* table_dev,
                                   const int64_t* chunk_first_dev, int64_t total_chunks, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int step, const float* grad_scale) {
  if (n_tensors == 0 || total_chunks == 0) return MDT_OK;
  MDT_CHECK_ARG(table_dev && chunk_first_dev && step >= 1, "adam_step_multi: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr * sqrt(bc2) / bc1);
  const unsigned grid = (unsigned)(total_chunks > 65536 ? 65536 : total_chunks);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDT_F32) hipLaunchKernelGGL((adam_multi_kernel<float>), grid, 256, 0, st, n_tensors, table_dev, chunk_first_dev, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale);
  else if (dtype == MDT_BF16) hipLaunchKernelGGL((adam_multi_kernel<bf16_t>), grid, 256, 0, st, n_tensors, table_dev, chunk_first_dev, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale);
  else MDT_UNSUPPORTED("adam_step_multi: dtype %d", dtype);
  return check_launch("adam_step_multi");
}

extern "C" int mdt_adam_step(void* stream, int dtype, int64_t n, void* param, float* master, const float* grad, float* m,
                             float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             const float* grad_scale) {
  if (n == 0) return MDT_OK;
  MDT_CHECK_ARG(param && grad && m && v && step >= 1, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr * sqrt(bc2) / bc1);
  const unsigned grid = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDT_F32) hipLaunchKernelGGL((adam_kernel<float>), grid, 256, 0, st, n, (float*)param, master, grad, m, v, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale);
  else if (dtype == MDT_BF16) hipLaunchKernelGGL((adam_kernel<bf16_t>), grid, 256, 0, st, n, (bf16_t*)param, master, grad, m, v, lr, beta1, beta2, eps, weight_decay, step_size, grad_scale);
  else MDT_UNSUPPORTED("adam_step: dtype %d", dtype);
  return check_launch("adam_step");
}
