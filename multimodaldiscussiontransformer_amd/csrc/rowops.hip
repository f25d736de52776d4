// HBM-bound glue kernels of the mDT hot path: the bottleneck-token exchange between the
// text / image / graph token spaces, embedding gathers and their scatter-add gradients,
// ViT patch gather, graph node features, the pooler tanh and the fp16 weighted
// cross-entropy with F1 counters.  All are row movers: one wavefront per row, 16-byte
// vector accesses, indices precomputed by the packer (no nonzero / host sync).
#include "common.hpp"

namespace mdt {

template <typename T> struct V16;
template <> struct V16<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct V16<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };

__device__ __forceinline__ int64_t pick_row(const int32_t* idx, int64_t inner, int64_t stride, int64_t off, int64_t r) {
  if (idx) return (int64_t)idx[r];
  if (inner <= 1) return r * stride + off;
  const int64_t q = r / inner;
  return q * stride + (r - q * inner) + off;
}

// scalar variant for rows that are not 16-byte vectorisable (e.g. the [M, 2] logits)
template <typename T>
__global__ __launch_bounds__(256) void row_axpby_scalar_kernel(int64_t nrows, int D, T* dst, int64_t ldd, const int32_t* di,
                                                               int64_t d_inner, int64_t d_stride, int64_t d_off, const T* a,
                                                               int64_t lda, const int32_t* ai, int64_t a_inner,
                                                               int64_t a_stride, int64_t a_off, float alpha, const T* b,
                                                               int64_t ldb, const int32_t* bi, int64_t b_inner,
                                                               int64_t b_stride, int64_t b_off, float beta, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int64_t dr = pick_row(di, d_inner, d_stride, d_off, r);
  if (dr < 0) return;
  const int64_t ar = a ? pick_row(ai, a_inner, a_stride, a_off, r) : -1;
  const int64_t br = b ? pick_row(bi, b_inner, b_stride, b_off, r) : -1;
  for (int c = lane; c < D; c += 64) {
    float v = 0.f;
    if (ar >= 0) v += alpha * to_f32(a[ar * lda + c]);
    if (br >= 0) v += beta * to_f32(b[br * ldb + c]);
    if (accumulate) v += to_f32(dst[dr * ldd + c]);
    dst[dr * ldd + c] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_axpby_kernel(int64_t nrows, int D, T* dst, int64_t ldd, const int32_t* di,
                                                        int64_t d_inner, int64_t d_stride, int64_t d_off, const T* a, int64_t lda,
                                                        const int32_t* ai, int64_t a_inner, int64_t a_stride, int64_t a_off, float alpha,
                                                        const T* b, int64_t ldb, const int32_t* bi, int64_t b_inner, int64_t b_stride,
                                                        int64_t b_off, float beta, int accumulate) {
  constexpr int VN = V16<T>::N;
  typedef typename V16<T>::type vec;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int64_t dr = pick_row(di, d_inner, d_stride, d_off, r);
  if (dr < 0) return;
  const int64_t ar = a ? pick_row(ai, a_inner, a_stride, a_off, r) : -1;
  const int64_t br = b ? pick_row(bi, b_inner, b_stride, b_off, r) : -1;
  for (int c = lane * VN; c < D; c += 64 * VN) {
    vec va, vb, vd, o;
    if (ar >= 0) va = *(const vec*)(a + ar * lda + c);
    if (br >= 0) vb = *(const vec*)(b + br * ldb + c);
    if (accumulate) vd = *(const vec*)(dst + dr * ldd + c);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float v = 0.f;
      if (ar >= 0) v += alpha * to_f32((T)va[j]);
      if (br >= 0) v += beta * to_f32((T)vb[j]);
      if (accumulate) v += to_f32((T)vd[j]);
      o[j] = from_f32<T>(v);
    }
    *(vec*)(dst + dr * ldd + c) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_scatter_add_kernel(int64_t nrows, int D, float* table, int64_t ldt,
                                                              const int32_t* idx, const T* src, int64_t lds_,
                                                              int64_t s_stride, int64_t s_off) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int64_t t = idx[r];
  if (t < 0) return;
  const T* s = src + (r * s_stride + s_off) * lds_;
  for (int c = lane; c < D; c += 64) atomicAdd(table + t * ldt + c, to_f32(s[c]));
}

template <typename T>
__global__ __launch_bounds__(256) void bert_embed_sum_kernel(int64_t M, int L, const int32_t* ids, const int32_t* types,
                                                             const T* word, const T* pos, const T* type, int D, T* out,
                                                             int64_t ldo, int64_t out_seq_stride, int64_t out_off) {
  constexpr int VN = V16<T>::N;
  typedef typename V16<T>::type vec;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M * L) return;
  const int64_t m = r / L;
  const int l = (int)(r - m * L);
  const int64_t w = ids[r], t = types[r];
  T* o = out + (m * out_seq_stride + out_off + l) * ldo;
  for (int c = lane * VN; c < D; c += 64 * VN) {
    const vec vw = *(const vec*)(word + w * D + c);
    const vec vp = *(const vec*)(pos + (int64_t)l * D + c);
    const vec vt = *(const vec*)(type + t * D + c);
    vec ov;
#pragma unroll
    for (int j = 0; j < VN; ++j) ov[j] = from_f32<T>(to_f32((T)vw[j]) + to_f32((T)vt[j]) + to_f32((T)vp[j]));
    *(vec*)(o + c) = ov;
  }
}

// ragged form: one packed row per VALID token, explicit position ids
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_rows_kernel(int64_t rows, const int32_t* ids, const int32_t* types,
                                                              const int32_t* pos_ids, const T* word, const T* pos, const T* type,
                                                              int D, T* out, int64_t ldo) {
  constexpr int VN = V16<T>::N;
  typedef typename V16<T>::type vec;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t w = ids[r], t = types[r], l = pos_ids[r];
  T* o = out + r * ldo;
  for (int c = lane * VN; c < D; c += 64 * VN) {
    const vec vw = *(const vec*)(word + w * D + c);
    const vec vp = *(const vec*)(pos + l * D + c);
    const vec vt = *(const vec*)(type + t * D + c);
    vec ov;
#pragma unroll
    for (int j = 0; j < VN; ++j) ov[j] = from_f32<T>(to_f32((T)vw[j]) + to_f32((T)vt[j]) + to_f32((T)vp[j]));
    *(vec*)(o + c) = ov;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_patchify_kernel(int I, int C, int HW, int p, const float* img, T* cols,
                                                           int64_t ldc) {
  // one workgroup per (image, patch row): reads p full image rows per channel (coalesced)
  const int gw = HW / p;
  const int i = blockIdx.y, py = blockIdx.x;
  const int K = C * p * p;
  for (int e = threadIdx.x; e < gw * K; e += 256) {
    // e enumerates (c, dy, x) with x = px*p + dx fastest so that global reads are contiguous
    const int x = e % HW;
    const int dy = (e / HW) % p;
    const int c = e / (HW * p);
    const int px = x / p, dx = x - px * p;
    const float v = img[(((int64_t)i * C + c) * HW + (py * p + dy)) * HW + x];
    cols[((int64_t)i * gw * gw + py * gw + px) * ldc + c * p * p + dy * p + dx] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(int I, int np, int D, const T* patches, int64_t ldp,
                                                           const T* cls, const T* pos, T* tokens, int64_t ldt,
                                                           int64_t seq_stride, int64_t off) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)I * (np + 1)) return;
  const int64_t i = r / (np + 1);
  const int j = (int)(r - i * (np + 1));
  T* o = tokens + (i * seq_stride + off + j) * ldt;
  const T* s = (j == 0) ? cls : patches + (i * np + (j - 1)) * ldp;
  for (int c = lane; c < D; c += 64) o[c] = from_f32<T>(to_f32(s[c]) + to_f32(pos[(int64_t)j * D + c]));
}

template <typename T>
__global__ __launch_bounds__(256) void graph_node_feature_kernel(int B, int Tn, int D, const T* src, int64_t lds_,
                                                                 const int32_t* node_row, const int32_t* in_degree,
                                                                 const int32_t* out_degree,
                                                                 const T* in_emb, const T* out_emb, const T* graph_token,
                                                                 T* x, int64_t ldx) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * Tn) return;
  const int64_t b = r / Tn;
  const int t = (int)(r - b * Tn);
  T* o = x + r * ldx;
  if (t == 0) {
    for (int c = lane; c < D; c += 64) o[c] = graph_token[c];
    return;
  }
  const int64_t n = b * (Tn - 1) + (t - 1);
  const int64_t sr = node_row[n];
  const int64_t dgi = in_degree[n], dgo = out_degree[n];
  for (int c = lane; c < D; c += 64) {
    float v = to_f32(in_emb[dgi * D + c]) + to_f32(out_emb[dgo * D + c]);
    if (sr >= 0) v += to_f32(src[sr * lds_ + c]);
    o[c] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void tanh_fwd_kernel(int64_t n, const T* x, T* y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(tanhf(to_f32(x[i])));
}
template <typename T>
__global__ void tanh_bwd_kernel(int64_t n, const T* y, const T* dy, T* dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = to_f32(y[i]);
    dx[i] = from_f32<T>(to_f32(dy[i]) * (1.f - t * t));
  }
}

// fp16 round-trip (round to nearest even) — the reference casts logits with
// .type(torch.HalfTensor) (criterions/hatespeech_loss.py:95), so the loss and dL/dlogits
// see fp16-rounded logits.
__device__ __forceinline__ float round_f16(float v) { return (float)(_Float16)v; }

// Arithmetic of the reference's loss on its CPU path, step by step (PyTorch's Half kernels round where they store):
//   log_softmax (vec_log_softmax_lastdim, scalar_t = Half): the sum of exp(x - max) is stored as Half, so is its log,
//     then out = half((x - max) - logsum);
//   softmax for the predictions (vec_softmax_lastdim): exp and the sum stay in float, out = half(e * (1 / sum)) —
//     two close logits can tie after the rounding, and argmax then returns index 0;
//   nll_loss (nll_loss_out_frame, reduction "sum"): every term half(logp * w) is subtracted from Half partial sums
//     cascaded in blocks of 16 (level_power = max(4, ceil(log2 n) / 8)), the 8 levels are added up in Half;
//   backward: d logits = half(w * (exp(logp) - onehot)) (one rounding), cast back by the .type(HalfTensor) adjoint.
// Checked against torch 2.10 CPU on random logits: bit-equal but for 1-ulp differences between expf / logf here and
// sleef's there on knife-edge roundings (tests/test_kernels_gpu.py::test_node_ce_matches_torch_cpu_half).
template <typename T>
__global__ __launch_bounds__(256) void node_ce_kernel(int64_t M, int nlab, const T* logits, const int32_t* rows,
                                                      const int32_t* targets, float w_neg, float w_pos, int fp16_loss,
                                                      float grad_scale, float* out_loss, int32_t* counters, T* dlogits) {
  // single workgroup: nlab is the number of labelled comments in the batch (tens to a few thousand)
  __shared__ float s_loss[256];
  __shared__ int s_cnt[4][256];
  __shared__ float s_part[8];
  float loss = 0.f;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  int level_power = 4;
  if (fp16_loss) {
    int lg = 0;
    while ((1ll << lg) < (long long)nlab) ++lg;          // ceil(log2(nlab))
    if (lg / 8 > level_power) level_power = lg / 8;
    if (threadIdx.x < 8) s_part[threadIdx.x] = 0.f;
  }
  for (int base = 0; base < nlab; base += 256) {
    const int i = base + threadIdx.x;
    float term = 0.f;
    if (i < nlab) {
      const int64_t r = rows[i];
      const int y = targets[i];
      float l0 = to_f32(logits[r * 2]), l1 = to_f32(logits[r * 2 + 1]);
      float wn = w_neg, wp = w_pos;
      if (fp16_loss) { l0 = round_f16(l0); l1 = round_f16(l1); wn = round_f16(wn); wp = round_f16(wp); }
      const float m = fmaxf(l0, l1);
      const float e0 = expf(l0 - m), e1 = expf(l1 - m);
      float lp0, lp1;
      int pred;
      if (fp16_loss) {
        const float ls = round_f16(logf(round_f16(e0 + e1)));
        lp0 = round_f16((l0 - m) - ls);
        lp1 = round_f16((l1 - m) - ls);
        const float inv = 1.0f / (e0 + e1);
        pred = (round_f16(e1 * inv) > round_f16(e0 * inv)) ? 1 : 0;      // ties -> index 0 like torch.argmax
      } else {
        const float ls = logf(e0 + e1);
        lp0 = (l0 - m) - ls;
        lp1 = (l1 - m) - ls;
        pred = (l1 > l0) ? 1 : 0;
      }
      const float w = y ? wp : wn;
      term = w * (y ? lp1 : lp0);
      if (fp16_loss) term = round_f16(term);
      loss -= term;
      c0 += (pred == y);
      c1 += (pred == y && pred == 1);
      c2 += (y == 1);
      c3 += (pred == 1);
      if (dlogits) {
        // autograd differentiates through the (fp16-rounded) log-softmax output
        const float p0 = expf(lp0), p1 = expf(lp1);
        float g0 = w * (p0 - (y == 0 ? 1.f : 0.f)), g1 = w * (p1 - (y == 1 ? 1.f : 0.f));
        if (fp16_loss) { g0 = round_f16(g0); g1 = round_f16(g1); }
        dlogits[r * 2] = from_f32<T>(g0 * grad_scale);
        dlogits[r * 2 + 1] = from_f32<T>(g1 * grad_scale);
      }
    }
    if (fp16_loss) {
      // the Half cascade is order-dependent: one thread walks this chunk's terms in label order
      __syncthreads();
      s_loss[threadIdx.x] = term;
      __syncthreads();
      if (threadIdx.x == 0) {
        const int n = nlab - base < 256 ? nlab - base : 256;
        const int lmask = (1 << level_power) - 1;
        for (int k = 0; k < n; ++k) {
          const int idx = base + k;
          s_part[0] = round_f16(s_part[0] - s_loss[k]);
          for (int j = 0; j + 1 < 8; ++j) {
            if ((idx & (lmask << (j * level_power))) != 0) break;
            s_part[j + 1] = round_f16(s_part[j + 1] + s_part[j]);
            s_part[j] = 0.f;
          }
        }
      }
    }
  }
  __syncthreads();
  s_loss[threadIdx.x] = loss;
  s_cnt[0][threadIdx.x] = c0; s_cnt[1][threadIdx.x] = c1; s_cnt[2][threadIdx.x] = c2; s_cnt[3][threadIdx.x] = c3;
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f;
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < 256; ++i) { L += s_loss[i]; a0 += s_cnt[0][i]; a1 += s_cnt[1][i]; a2 += s_cnt[2][i]; a3 += s_cnt[3][i]; }
    if (fp16_loss) {
      L = 0.f;
      for (int j = 0; j < 8; ++j) L = round_f16(L + s_part[j]);
    }
    out_loss[0] = L;
    counters[0] = a0; counters[1] = a1; counters[2] = a2; counters[3] = a3;
  }
}

}  // namespace mdt

using namespace mdt;

#define DISPATCH_T(dtype, NAME, ...)                                  \
  if ((dtype) == MDT_F32) { NAME(float, __VA_ARGS__); }               \
  else if ((dtype) == MDT_BF16) { NAME(bf16_t, __VA_ARGS__); }        \
  else MDT_UNSUPPORTED("dtype %d", (dtype))

static int vec_ok(int dtype, int D, int64_t ld, const void* p) {
  const int vn = dtype == MDT_BF16 ? 8 : 4;
  MDT_CHECK_ARG(D % vn == 0 && ld % vn == 0 && ((uintptr_t)p & 15) == 0,
                "row op: D=%d / ld=%lld / pointer not 16-byte vectorisable", D, (long long)ld);
  return MDT_OK;
}

extern "C" int mdt_row_axpby(void* stream, int dtype, int64_t nrows, int D, void* dst, int64_t ldd, const int32_t* di,
                             int64_t d_inner, int64_t d_stride, int64_t d_off, const void* a, int64_t lda, const int32_t* ai,
                             int64_t a_inner, int64_t a_stride, int64_t a_off, float alpha, const void* b, int64_t ldb,
                             const int32_t* bi, int64_t b_inner, int64_t b_stride, int64_t b_off, float beta, int accumulate) {
  if (nrows == 0) return MDT_OK;
  MDT_CHECK_ARG(dst, "row_axpby: null dst");
  MDT_CHECK_ARG(dtype == MDT_F32 || dtype == MDT_BF16, "row_axpby: bad dtype %d", dtype);
  const int vn = dtype == MDT_BF16 ? 8 : 4;
  bool vec = D % vn == 0 && ldd % vn == 0 && ((uintptr_t)dst & 15) == 0;
  if (a) vec = vec && lda % vn == 0 && ((uintptr_t)a & 15) == 0;
  if (b) vec = vec && ldb % vn == 0 && ((uintptr_t)b & 15) == 0;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((nrows + 3) / 4);
#define ARGS_(T) nrows, D, (T*)dst, ldd, di, d_inner, d_stride, d_off, (const T*)a, lda, ai, a_inner, a_stride, a_off, alpha, (const T*)b, ldb, bi, b_inner, b_stride, b_off, beta, accumulate
  if (vec) {
    if (dtype == MDT_F32) hipLaunchKernelGGL((row_axpby_kernel<float>), grid, 256, 0, st, ARGS_(float));
    else hipLaunchKernelGGL((row_axpby_kernel<bf16_t>), grid, 256, 0, st, ARGS_(bf16_t));
  } else {
    if (dtype == MDT_F32) hipLaunchKernelGGL((row_axpby_scalar_kernel<float>), grid, 256, 0, st, ARGS_(float));
    else hipLaunchKernelGGL((row_axpby_scalar_kernel<bf16_t>), grid, 256, 0, st, ARGS_(bf16_t));
  }
#undef ARGS_
  return check_launch("row_axpby");
}

extern "C" int mdt_row_scatter_add_f32(void* stream, int dtype, int64_t nrows, int D, float* table, int64_t ldt,
                                       const int32_t* idx, const void* src, int64_t lds_, int64_t s_stride, int64_t s_off) {
  if (nrows == 0) return MDT_OK;
  MDT_CHECK_ARG(table && idx && src, "row_scatter_add: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((nrows + 3) / 4);
#define K_(T, ...) hipLaunchKernelGGL((row_scatter_add_kernel<T>), grid, 256, 0, st, nrows, D, table, ldt, idx, (const T*)src, lds_, s_stride, s_off)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("row_scatter_add");
}

extern "C" int mdt_bert_embed_sum(void* stream, int dtype, int64_t M, int L, const int32_t* ids, const int32_t* types,
                                  const void* word, const void* pos, const void* type, int D, void* out, int64_t ldo,
                                  int64_t out_seq_stride, int64_t out_off) {
  if (M == 0) return MDT_OK;
  MDT_CHECK_ARG(ids && types && word && pos && type && out, "bert_embed_sum: null pointer");
  if (int e = vec_ok(dtype, D, ldo, out)) return e;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M * L + 3) / 4);
#define K_(T, ...) hipLaunchKernelGGL((bert_embed_sum_kernel<T>), grid, 256, 0, st, M, L, ids, types, (const T*)word, (const T*)pos, (const T*)type, D, (T*)out, ldo, out_seq_stride, out_off)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("bert_embed_sum");
}

extern "C" int mdt_bert_embed_rows(void* stream, int dtype, int64_t rows, const int32_t* ids, const int32_t* types,
                                   const int32_t* pos_ids, const void* word, const void* pos, const void* type, int D,
                                   void* out, int64_t ldo) {
  if (rows == 0) return MDT_OK;
  MDT_CHECK_ARG(ids && types && pos_ids && word && pos && type && out, "bert_embed_rows: null pointer");
  if (int e = vec_ok(dtype, D, ldo, out)) return e;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 3) / 4);
#define K_(T, ...) hipLaunchKernelGGL((bert_embed_rows_kernel<T>), grid, 256, 0, st, rows, ids, types, pos_ids, (const T*)word, (const T*)pos, (const T*)type, D, (T*)out, ldo)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("bert_embed_rows");
}

extern "C" int mdt_vit_patchify(void* stream, int dtype, int I, int C, int HW, int p, const float* img, void* cols, int64_t ldc) {
  if (I == 0) return MDT_OK;
  MDT_CHECK_ARG(img && cols && p > 0 && HW % p == 0, "vit_patchify: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(HW / p, I);
#define K_(T, ...) hipLaunchKernelGGL((vit_patchify_kernel<T>), grid, 256, 0, st, I, C, HW, p, img, (T*)cols, ldc)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("vit_patchify");
}

extern "C" int mdt_vit_assemble(void* stream, int dtype, int I, int np, int D, const void* patches, int64_t ldp,
                                const void* cls, const void* pos, void* tokens, int64_t ldt, int64_t seq_stride, int64_t off) {
  if (I == 0) return MDT_OK;
  MDT_CHECK_ARG(patches && cls && pos && tokens, "vit_assemble: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(((int64_t)I * (np + 1) + 3) / 4);
#define K_(T, ...) hipLaunchKernelGGL((vit_assemble_kernel<T>), grid, 256, 0, st, I, np, D, (const T*)patches, ldp, (const T*)cls, (const T*)pos, (T*)tokens, ldt, seq_stride, off)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("vit_assemble");
}

extern "C" int mdt_graph_node_feature(void* stream, int dtype, int B, int T, int D, const void* src, int64_t lds_,
                                      const int32_t* node_row, const int32_t* in_degree, const int32_t* out_degree,
                                      const void* in_emb, const void* out_emb, const void* graph_token, void* x,
                                      int64_t ldx) {
  if (B == 0) return MDT_OK;
  MDT_CHECK_ARG(node_row && in_degree && out_degree && in_emb && out_emb && graph_token && x, "graph_node_feature: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(((int64_t)B * T + 3) / 4);
#define K_(T_, ...) hipLaunchKernelGGL((graph_node_feature_kernel<T_>), grid, 256, 0, st, B, T, D, (const T_*)src, lds_, node_row, in_degree, out_degree, (const T_*)in_emb, (const T_*)out_emb, (const T_*)graph_token, (T_*)x, ldx)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("graph_node_feature");
}

extern "C" int mdt_tanh_fwd(void* stream, int dtype, int64_t n, const void* x, void* y) {
  if (n == 0) return MDT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
#define K_(T, ...) hipLaunchKernelGGL((tanh_fwd_kernel<T>), grid, 256, 0, st, n, (const T*)x, (T*)y)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("tanh_fwd");
}

extern "C" int mdt_tanh_bwd(void* stream, int dtype, int64_t n, const void* y, const void* dy, void* dx) {
  if (n == 0) return MDT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
#define K_(T, ...) hipLaunchKernelGGL((tanh_bwd_kernel<T>), grid, 256, 0, st, n, (const T*)y, (const T*)dy, (T*)dx)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("tanh_bwd");
}

extern "C" int mdt_node_ce(void* stream, int dtype, int64_t M, int nlab, const void* logits, const int32_t* rows,
                           const int32_t* targets, float w_neg, float w_pos, int fp16_loss, float grad_scale,
                           float* out_loss, int32_t* counters, void* dlogits) {
  MDT_CHECK_ARG(logits && out_loss && counters, "node_ce: null pointer");
  MDT_CHECK_ARG(nlab == 0 || (rows && targets), "node_ce: null rows / targets");
  hipStream_t st = (hipStream_t)stream;
  if (dlogits) {
    if (hipMemsetAsync(dlogits, 0, (size_t)M * 2 * dtype_size(dtype), st) != hipSuccess) {
      (void)hipGetLastError();
      set_error("node_ce: memset failed");
      return MDT_ERR_LAUNCH;
    }
  }
#define K_(T, ...) hipLaunchKernelGGL((node_ce_kernel<T>), 1, 256, 0, st, M, nlab, (const T*)logits, rows, targets, w_neg, w_pos, fp16_loss, grad_scale, out_loss, counters, (T*)dlogits)
  DISPATCH_T(dtype, K_, 0);
#undef K_
  return check_launch("node_ce");
}
