// Attention over sequences longer than one workgroup holds (S > 272): discussion trees with more than 271 comments.
// The reference runs any tree size as dense O(T^2) attention (--max-nodes 10000 is declared and never enforced,
// mDT/src/tasks/task.py:41-44; modules/multihead_attention.py:139-202); the single-pass MFMA kernels of attention*.hip
// keep a whole (sequence, head) on chip and stop at 272 tokens.  This file is the key-chunked path behind the same C ABI:
// flash-style online softmax in the forward, log-sum-exp based recomputation in the backward, the same masks /
// structural bias / dropout counters — so a long tree gives the same numbers the short kernels would.  Plain fp32 FMA
// (one query or key per lane, the other operand broadcast from LDS): such trees are rare in the pruned dataset
// (Pre-Processing/3-prune-trees.py), this path is about not refusing them, not about MFMA utilisation.
#include <type_traits>

#include "attention_common.hpp"

namespace mdt {

constexpr int LC = 64;      // chunk of keys (forward, dQ pass) or queries (dK / dV pass) staged in LDS

template <typename T, int HD>
__device__ __forceinline__ void long_stage(float* dst, const T* src, int64_t ld, int row0, int rows, int lane) {
  // dst[LC][HD] fp32 <- rows row0 .. row0+LC of src (zero past `rows`)
  for (int e = lane; e < LC * HD; e += 64) {
    const int r = e / HD, c = e - r * HD;
    dst[e] = (row0 + r < rows) ? to_f32(src[(int64_t)(row0 + r) * ld + c]) : 0.f;
  }
}

template <typename T, int HD, bool STRUCT>
__global__ __launch_bounds__(64) void attn_long_fwd_kernel(AttnParams P) {
  __shared__ float sK[LC * HD], sV[LC * HD];
  const mdt_attn_fwd_args& a = P.f;
  const int lane = threadIdx.x, h = blockIdx.y, seq = blockIdx.z;
  const int S = a.S, D = a.H * HD;
  const int q = blockIdx.x * 64 + lane;
  const bool qok = q < S;
  const int qc = qok ? q : S - 1;
  const int64_t row0 = (int64_t)seq * a.seq_stride, tld = a.pos_stride * a.ld_qkv;
  const T* qkv = (const T*)a.qkv + row0 * a.ld_qkv + h * HD;
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  float qv[HD], acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { qv[d] = to_f32(qkv[(int64_t)qc * tld + d]) * a.scale; acc[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int bh = seq * a.H + h;
  const bool drop = a.drop_p > 0.f;
  for (int k0 = 0; k0 < S; k0 += LC) {
    __syncthreads();
    long_stage<T, HD>(sK, qkv + D, tld, k0, S, lane);
    long_stage<T, HD>(sV, qkv + 2 * D, tld, k0, S, lane);
    __syncthreads();
    const int kn = S - k0 < LC ? S - k0 : LC;
    for (int j = 0; j < kn; ++j) {
      const int key = k0 + j;
      float s = key_only_bias<T>(bc, key);
      if (s == 0.f) {
#pragma unroll
        for (int d = 0; d < HD; ++d) s = __builtin_fmaf(qv[d], sK[j * HD + d], s);
        s += pair_bias<T, STRUCT>(bc, qc, key);
      }
      if (s == -INFINITY) continue;
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);      // m = -inf on the first live key: corr = 0
      l = l * corr + p;
      const float pd = drop ? p * attn_drop_scale(P.drop, bh, S, qc, key) : p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = __builtin_fmaf(pd, sV[j * HD + d], acc[d] * corr);
      m = mn;
    }
  }
  if (qok) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    T* orow = (T*)a.out + (row0 + (int64_t)q * a.pos_stride) * a.ld_out + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = from_f32<T>(acc[d] * inv);
    a.lse[((int64_t)seq * a.H + h) * S + q] = l > 0.f ? m + __logf(l) : -INFINITY;
  }
}

// dQ (queries on lanes, keys streamed through LDS) + the bias gradients
template <typename T, int HD, bool STRUCT>
__global__ __launch_bounds__(64) void attn_long_dq_kernel(AttnParams P) {
  __shared__ float sK[LC * HD], sV[LC * HD];
  const mdt_attn_fwd_args& a = P.f;
  const int lane = threadIdx.x, h = blockIdx.y, seq = blockIdx.z;
  const int S = a.S, D = a.H * HD;
  const int q = blockIdx.x * 64 + lane;
  const bool qok = q < S;
  const int qc = qok ? q : S - 1;
  const int64_t row0 = (int64_t)seq * a.seq_stride, tld = a.pos_stride * a.ld_qkv;
  const T* qkv = (const T*)a.qkv + row0 * a.ld_qkv + h * HD;
  const T* dorow = (const T*)P.dout + (row0 + (int64_t)qc * a.pos_stride) * P.ld_dout + h * HD;
  const T* orow = (const T*)a.out + (row0 + (int64_t)qc * a.pos_stride) * a.ld_out + h * HD;
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  float qv[HD], dov[HD], dq[HD];
  float delta = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    qv[d] = to_f32(qkv[(int64_t)qc * tld + d]) * a.scale;
    dov[d] = to_f32(dorow[d]);
    delta = __builtin_fmaf(dov[d], to_f32(orow[d]), delta);
    dq[d] = 0.f;
  }
  const float lse = a.lse[((int64_t)seq * a.H + h) * S + qc];
  const int bh = seq * a.H + h;
  const bool drop = a.drop_p > 0.f;
  for (int k0 = 0; k0 < S; k0 += LC) {
    __syncthreads();
    long_stage<T, HD>(sK, qkv + D, tld, k0, S, lane);
    long_stage<T, HD>(sV, qkv + 2 * D, tld, k0, S, lane);
    __syncthreads();
    const int kn = S - k0 < LC ? S - k0 : LC;
    for (int j = 0; j < kn; ++j) {
      const int key = k0 + j;
      float s = key_only_bias<T>(bc, key);
      if (s == 0.f) {
#pragma unroll
        for (int d = 0; d < HD; ++d) s = __builtin_fmaf(qv[d], sK[j * HD + d], s);
        s += pair_bias<T, STRUCT>(bc, qc, key);
      }
      if (s == -INFINITY || lse == -INFINITY || !qok) continue;
      const float p = __expf(s - lse);
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) dp = __builtin_fmaf(dov[d], sV[j * HD + d], dp);
      if (drop) dp *= attn_drop_scale(P.drop, bh, S, q, key);
      const float ds = p * (dp - delta);
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = __builtin_fmaf(ds, sK[j * HD + d], dq[d]);
      if (P.d_dense_bias) P.d_dense_bias[(((int64_t)seq * a.H + h) * S + q) * S + key] = ds;
      if constexpr (STRUCT) {
        if (P.d_sp_table && ds != 0.f) {
          if (q >= 1 && key >= 1) {
            const int idx = a.spatial_pos[((int64_t)seq * (S - 1) + (q - 1)) * (S - 1) + (key - 1)];
            if (idx != 0) atomicAdd(P.d_sp_table + (int64_t)idx * a.H + h, ds);      // padding_idx row 0: no gradient
          } else if (P.d_virt) {
            atomicAdd(P.d_virt + h, ds);
          }
        }
      }
    }
  }
  if (qok) {
    T* drow = (T*)P.dqkv + (row0 + (int64_t)q * a.pos_stride) * P.ld_dqkv + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) drow[d] = from_f32<T>(dq[d] * a.scale);
  }
}

// dK, dV (keys on lanes, queries streamed through LDS)
template <typename T, int HD, bool STRUCT>
__global__ __launch_bounds__(64) void attn_long_dkv_kernel(AttnParams P) {
  __shared__ float sQ[LC * HD], sO[LC * HD], sL[LC], sD[LC];
  const mdt_attn_fwd_args& a = P.f;
  const int lane = threadIdx.x, h = blockIdx.y, seq = blockIdx.z;
  const int S = a.S, D = a.H * HD;
  const int key = blockIdx.x * 64 + lane;
  const bool kok = key < S;
  const int kc = kok ? key : S - 1;
  const int64_t row0 = (int64_t)seq * a.seq_stride, tld = a.pos_stride * a.ld_qkv;
  const T* qkv = (const T*)a.qkv + row0 * a.ld_qkv + h * HD;
  const T* dout = (const T*)P.dout + row0 * P.ld_dout + h * HD;
  const T* outp = (const T*)a.out + row0 * a.ld_out + h * HD;
  const int64_t dld = a.pos_stride * P.ld_dout, old_ = a.pos_stride * a.ld_out;
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  float kv[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    kv[d] = to_f32(qkv[(int64_t)kc * tld + D + d]);
    vv[d] = to_f32(qkv[(int64_t)kc * tld + 2 * D + d]);
    dk[d] = dv[d] = 0.f;
  }
  const float kb = key_only_bias<T>(bc, kc);
  const int bh = seq * a.H + h;
  const bool drop = a.drop_p > 0.f;
  for (int q0 = 0; q0 < S; q0 += LC) {
    __syncthreads();
    long_stage<T, HD>(sQ, qkv, tld, q0, S, lane);
    long_stage<T, HD>(sO, dout, dld, q0, S, lane);
    {
      const int qi = q0 + lane;
      float de = 0.f, l = -INFINITY;
      if (qi < S) {
        l = a.lse[((int64_t)seq * a.H + h) * S + qi];
        for (int d = 0; d < HD; ++d) de = __builtin_fmaf(to_f32(dout[(int64_t)qi * dld + d]), to_f32(outp[(int64_t)qi * old_ + d]), de);
      }
      sL[lane] = l;
      sD[lane] = de;
    }
    __syncthreads();
    const int qn = S - q0 < LC ? S - q0 : LC;
    if (!kok || kb == -INFINITY) continue;
    for (int j = 0; j < qn; ++j) {
      const int q = q0 + j;
      const float l = sL[j];
      if (l == -INFINITY) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        s = __builtin_fmaf(sQ[j * HD + d], kv[d], s);
        dp = __builtin_fmaf(sO[j * HD + d], vv[d], dp);
      }
      s = s * a.scale + pair_bias<T, STRUCT>(bc, q, key);
      if (s == -INFINITY) continue;
      const float p = __expf(s - l);
      const float mk = drop ? attn_drop_scale(P.drop, bh, S, q, key) : 1.0f;
      const float pd = p * mk;
      const float ds = p * (dp * mk - sD[j]);
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dv[d] = __builtin_fmaf(pd, sO[j * HD + d], dv[d]);
        dk[d] = __builtin_fmaf(ds, sQ[j * HD + d], dk[d]);
      }
    }
  }
  if (kok) {
    T* krow = (T*)P.dqkv + (row0 + (int64_t)key * a.pos_stride) * P.ld_dqkv + h * HD + D;
    T* vrow = krow + D;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      krow[d] = from_f32<T>(dk[d] * a.scale);
      vrow[d] = from_f32<T>(dv[d]);
    }
  }
}

template <typename T, int HD>
static int long_launch(hipStream_t st, const AttnParams& p, bool bwd) {
  const mdt_attn_fwd_args& a = p.f;
  const bool st_bias = a.attn_bias != nullptr;
  dim3 grid((unsigned)((a.S + 63) / 64), (unsigned)a.H, (unsigned)a.nseq);
  if (!bwd) {
    if (st_bias) hipLaunchKernelGGL((attn_long_fwd_kernel<T, HD, true>), grid, 64, 0, st, p);
    else hipLaunchKernelGGL((attn_long_fwd_kernel<T, HD, false>), grid, 64, 0, st, p);
    return check_launch("attention_long_fwd");
  }
  if (st_bias) {
    hipLaunchKernelGGL((attn_long_dq_kernel<T, HD, true>), grid, 64, 0, st, p);
    hipLaunchKernelGGL((attn_long_dkv_kernel<T, HD, true>), grid, 64, 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_long_dq_kernel<T, HD, false>), grid, 64, 0, st, p);
    hipLaunchKernelGGL((attn_long_dkv_kernel<T, HD, false>), grid, 64, 0, st, p);
  }
  return check_launch("attention_long_bwd");
}

int attention_long_dispatch(hipStream_t st, const AttnParams& p, bool bwd) {
  const mdt_attn_fwd_args& a = p.f;
  if (a.seq_offsets || a.q_limit > 0) {
    set_error("attention: S=%d exceeds the 272-token limit of the single-pass kernels and the key-chunked path takes neither "
              "ragged sequences nor q_limit (text / image sequences are always shorter)", a.S);
    return MDT_ERR_UNSUPPORTED;
  }
  if (a.hd != 64) {
    set_error("attention (long sequences): head_dim %d unsupported (64 only)", a.hd);
    return MDT_ERR_UNSUPPORTED;
  }
  if (a.dtype == MDT_BF16) return long_launch<bf16_t, 64>(st, p, bwd);
  return long_launch<float, 64>(st, p, bwd);
}

}  // namespace mdt
