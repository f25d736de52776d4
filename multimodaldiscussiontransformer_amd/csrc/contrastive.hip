// Community-contrastive loss on the global discussion embeddings — the objective of task ``contrastive_learning``
// (mDT/src/criterions/contrastive_loss.py:76-180), the only consumer of the encoder's third output and therefore the
// only thing that gives the final graph stack a gradient.
//
//   n_i      = e_i / max(||e_i||, 1e-12)                                   F.normalize          (:111-112)
//   sim_ij   = scale * <n_i, n_j>                                           fp32                 (:113-116)
//   T_ij     = [y_i == y_j],  H_ij = [hard_y_i == y_j],  soft_ij = !T_ij && !H_ij                (:120-131)
//   extra_k  = 2 * #{j: T_kj || H_kj} / #{j: soft_kj}   (adaptive)   |  soft_negative_weight     (:132-139)
//   W_ij     = i == j ? 0 : (soft_ij ? extra_j : 1)      — the reference broadcasts the per-ROW vector ``extra``
//              along the LAST axis (torch.where(soft [B,B], extra [B], 1)), i.e. it is indexed by the COLUMN (:143-150)
//   loss     = sum_ij W_ij * BCEWithLogits(sim_ij, T_ij)                                        (:164-169)
//   counters : pred_ij = round(sigmoid(sim_ij)); ``pred == targets`` broadcasts y along the last axis as well:
//              ncorrect = #{pred_ij == y_j}, positive_correct = #{pred_ij == y_j == 1}, total_positive = #{y_j == 1}
//              (over B, not B^2), pred_positive = #{pred_ij == 1}                                (:153-161)
//   backward : G_ij = W_ij (sigmoid(sim_ij) - T_ij)   (fp32);  dn_i = scale * sum_j (G_ij + G_ji) n_j;
//              de_i = (dn_i - n_i <n_i, dn_i>) / ||e_i||
//
// B is the number of trees on this GPU (tens to a few hundred), D = 768 / 1024: a few MFLOP.  Three small launches —
// row normalisation (one wave per row), the B x B pair pass (ONE workgroup: thread-local partial sums folded by a
// fixed-order LDS tree, so the loss is deterministic), the gradient rows — all HBM/latency-trivial next to the encoder.
#include "common.hpp"

namespace mdt {

__device__ __forceinline__ float round_f16(float v) { return (float)(_Float16)v; }

template <typename T>
__global__ __launch_bounds__(64) void cl_normalize_kernel(int B, int D, const T* emb, int64_t ld, float* n, float* inv) {
  const int i = blockIdx.x, lane = threadIdx.x;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float v = to_f32(emb[(int64_t)i * ld + c]);
    ss += v * v;
  }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float r = 1.0f / fmaxf(nrm, 1e-12f);
  for (int c = lane; c < D; c += 64) n[(int64_t)i * D + c] = to_f32(emb[(int64_t)i * ld + c]) * r;
  if (lane == 0) inv[i] = nrm > 1e-12f ? r : -1e12f;      // negative: the clamp was active (no projection in backward)
}

__global__ __launch_bounds__(256) void cl_rowstats_kernel(int B, const float* y, const float* hard_y, int adaptive,
                                                          float soft_w, float* extra) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < B; k += gridDim.x * 256) {
    if (!adaptive) { extra[k] = soft_w; continue; }
    int hard = 0, soft = 0;
    for (int j = 0; j < B; ++j) {
      const bool t = y[k] == y[j], h = hard_y[k] == y[j];
      hard += (t || h);
      soft += (!t && !h);
    }
    extra[k] = ((float)hard / (float)soft) * 2.0f;        // 0 soft pairs -> inf, as in the reference
  }
}

__global__ __launch_bounds__(1024) void cl_pair_kernel(int B, int D, const float* n, const float* y, const float* hard_y,
                                                       const float* extra, float scale, float* G, float* out_loss,
                                                       int32_t* counters) {
  __shared__ float s_loss[1024];
  __shared__ int s_cnt[3][1024];
  float loss = 0.f;
  int c_ok = 0, c_pos_ok = 0, c_pred = 0;
  const int64_t npair = (int64_t)B * B;
  for (int64_t p = threadIdx.x; p < npair; p += 1024) {
    const int i = (int)(p / B), j = (int)(p - (int64_t)i * B);
    const float* a = n + (int64_t)i * D;
    const float* b = n + (int64_t)j * D;
    float dot = 0.f;
    for (int c = 0; c < D; ++c) dot += a[c] * b[c];
    const float x = dot * scale;
    const bool t = y[i] == y[j], h = hard_y[i] == y[j];
    const float tf = t ? 1.f : 0.f;
    float w = (!t && !h) ? extra[j] : 1.0f;
    if (i == j) w = 0.f;
    // binary_cross_entropy_with_logits with the reference's HALF target matrix (.half(), :122): PyTorch evaluates
    // (1 - target).mul_(input).sub_(log_sigmoid(input)).mul_(weight) in place on the half tensor, i.e. every step
    // rounds to half, the sum is taken in fp32 and stored as half; the op's own backward formula is
    // (sigmoid(x) - t) * weight in fp32 (checked against torch CPU with weights that half cannot represent).
    const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));           // log_sigmoid(x), fp32
    float e = round_f16((1.f - tf) * x);
    e = round_f16(e - ls);
    e = round_f16(e * w);
    if (w != 0.f) loss += e;
    const float sg = 1.0f / (1.0f + expf(-x));
    G[p] = (w != 0.f) ? w * (sg - tf) : 0.f;     // the backward formula applies the weight in fp32
    const float pred = rintf(sg);                           // round half to even, like torch.round
    const bool ok = pred == y[j];
    c_ok += ok;
    c_pos_ok += (ok && pred == 1.f);
    c_pred += (pred == 1.f);
  }
  s_loss[threadIdx.x] = loss;
  s_cnt[0][threadIdx.x] = c_ok; s_cnt[1][threadIdx.x] = c_pos_ok; s_cnt[2][threadIdx.x] = c_pred;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      s_loss[threadIdx.x] += s_loss[threadIdx.x + s];
      for (int k = 0; k < 3; ++k) s_cnt[k][threadIdx.x] += s_cnt[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_loss[0] = round_f16(s_loss[0]);
    int tp = 0;
    for (int j = 0; j < B; ++j) tp += (y[j] == 1.f);
    counters[0] = s_cnt[0][0]; counters[1] = s_cnt[1][0]; counters[2] = tp; counters[3] = s_cnt[2][0];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cl_grad_kernel(int B, int D, const float* n, const float* inv, const float* G,
                                                      float scale, float grad_scale, T* d_emb, int64_t ldd) {
  extern __shared__ float s_acc[];        // D floats + 256 reduction slots
  float* s_red = s_acc + D;
  const int i = blockIdx.x;
  float part = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < B; ++j) acc += (G[(int64_t)i * B + j] + G[(int64_t)j * B + i]) * n[(int64_t)j * D + c];
    acc *= scale;
    s_acc[c] = acc;
    part += acc * n[(int64_t)i * D + c];
  }
  s_red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
    __syncthreads();
  }
  const float dot = s_red[0];
  const float r = inv[i];
  for (int c = threadIdx.x; c < D; c += 256) {
    const float g = r > 0.f ? (s_acc[c] - n[(int64_t)i * D + c] * dot) * r : s_acc[c] * 1e12f;
    d_emb[(int64_t)i * ldd + c] = from_f32<T>(g * grad_scale);
  }
}

}  // namespace mdt

using namespace mdt;

extern "C" size_t mdt_contrastive_loss_workspace_bytes(int B, int D) {
  return ((size_t)B * D + (size_t)B * B + 2 * (size_t)B) * sizeof(float);
}

extern "C" int mdt_contrastive_loss(void* stream, int dtype, int B, int D, const void* emb, int64_t ld, const float* y,
                                    const float* hard_y, float scale, float soft_negative_weight, int adaptive,
                                    void* workspace, float grad_scale, float* out_loss, int32_t* counters, void* d_emb,
                                    int64_t ldd) {
  MDT_CHECK_ARG(dtype == MDT_F32 || dtype == MDT_BF16, "contrastive_loss: bad dtype %d", dtype);
  MDT_CHECK_ARG(B > 0 && D > 0, "contrastive_loss: bad shape B=%d D=%d", B, D);
  MDT_CHECK_ARG(emb && y && hard_y && workspace && out_loss && counters, "contrastive_loss: null pointer");
  MDT_CHECK_ARG(ld >= D && (!d_emb || ldd >= D), "contrastive_loss: row stride smaller than D");
  MDT_CHECK_ARG(!(adaptive && soft_negative_weight != 0.f),
                "contrastive_loss: adaptive_soft_negative_weight and soft_negative_weight are mutually exclusive");
  MDT_CHECK_ARG((size_t)D * 4 + 1024 <= 160 * 1024, "contrastive_loss: D=%d does not fit one workgroup's LDS", D);
  hipStream_t st = (hipStream_t)stream;
  float* n = (float*)workspace;
  float* G = n + (size_t)B * D;
  float* inv = G + (size_t)B * B;
  float* extra = inv + B;
  if (dtype == MDT_F32) hipLaunchKernelGGL((cl_normalize_kernel<float>), B, 64, 0, st, B, D, (const float*)emb, ld, n, inv);
  else hipLaunchKernelGGL((cl_normalize_kernel<bf16_t>), B, 64, 0, st, B, D, (const bf16_t*)emb, ld, n, inv);
  hipLaunchKernelGGL(cl_rowstats_kernel, (B + 255) / 256, 256, 0, st, B, y, hard_y, adaptive, soft_negative_weight, extra);
  hipLaunchKernelGGL(cl_pair_kernel, 1, 1024, 0, st, B, D, n, y, hard_y, extra, scale, G, out_loss, counters);
  if (d_emb) {
    const size_t lds = (size_t)(D + 256) * sizeof(float);
    if (dtype == MDT_F32) hipLaunchKernelGGL((cl_grad_kernel<float>), B, 256, lds, st, B, D, n, inv, G, scale, grad_scale, (float*)d_emb, ldd);
    else hipLaunchKernelGGL((cl_grad_kernel<bf16_t>), B, 256, lds, st, B, D, n, inv, G, scale, grad_scale, (bf16_t*)d_emb, ldd);
  }
  return check_launch("contrastive_loss");
}
