// Shared pieces of the two attention kernel families (attention.hip, attention_v2.hip).
#pragma once
#include "common.hpp"

namespace mdt {

struct AttnParams {
  mdt_attn_fwd_args f;
  DropCfg drop;
  const void* dout; int64_t ld_dout;
  void* dqkv; int64_t ld_dqkv;
  float* d_dense_bias; float* d_sp_table; float* d_virt;
};

struct BiasCtx {
  int seq, h, S, H;
  const uint8_t* key_mask;
  const uint8_t* key_pad;
  const float* dense;
  const float* attn_bias;
  const int32_t* sp;
  const void* table;
  const void* virt;
};

template <typename T>
__device__ __forceinline__ float key_only_bias(const BiasCtx& b, int key) {
  if (key >= b.S) return -INFINITY;
  if (b.key_mask && !b.key_mask[(int64_t)b.seq * b.S + key]) return -INFINITY;
  if (b.key_pad && b.key_pad[(int64_t)b.seq * b.S + key]) return -INFINITY;
  return 0.f;
}

// additive bias of score (q, key), both < S
template <typename T, bool STRUCT>
__device__ __forceinline__ float pair_bias(const BiasCtx& b, int q, int key) {
  float v = 0.f;
  if (b.dense) v += b.dense[(((int64_t)b.seq * b.H + b.h) * b.S + q) * b.S + key];
  if constexpr (STRUCT) {
    v += 2.0f * b.attn_bias[((int64_t)b.seq * b.S + q) * b.S + key];  // graphormer_layers.py:93 and :108
    if (q >= 1 && key >= 1) {
      const int idx = b.sp[((int64_t)b.seq * (b.S - 1) + (q - 1)) * (b.S - 1) + (key - 1)];
      v += to_f32(((const T*)b.table)[(int64_t)idx * b.H + b.h]);
    } else {
      v += to_f32(((const T*)b.virt)[b.h]);  // row 0 (graph token as query) or column 0 (as key)
    }
  }
  return v;
}

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// Attention-probability dropout counters: element (bh, q, key) -> (bh*S + q)*S2 + key with S2 = S rounded
// up to even, so keys 2i and 2i+1 of one query share a 32-bit word of the mixer (common.hpp "dropout RNG").
__device__ __forceinline__ uint32_t attn_row_pairs(int bh, int S, int q) {
  return (uint32_t)(bh * S + q) * (uint32_t)((S + 1) >> 1);
}
__device__ __forceinline__ uint32_t attn_drop_word(const DropCfg& d, uint32_t row_pairs, int key) {
  return drop_mix((row_pairs + (uint32_t)(key >> 1)) ^ d.key);
}
__device__ __forceinline__ bool drop_keep_lo(const DropCfg& d, uint32_t w) { return (w & 0xFFFFu) >= d.thresh; }
__device__ __forceinline__ bool drop_keep_hi(const DropCfg& d, uint32_t w) { return (w >> 16) >= d.thresh; }
// generic (one element): the same decision as the word helpers above
__device__ __forceinline__ float attn_drop_scale(const DropCfg& d, int bh, int S, int q, int key) {
  const uint32_t w = attn_drop_word(d, attn_row_pairs(bh, S, q), key);
  return ((key & 1) ? drop_keep_hi(d, w) : drop_keep_lo(d, w)) ? d.inv_keep : 0.f;
}

int attention_v2_dispatch(hipStream_t st, const AttnParams& p, bool bwd);
int attention_v3_bwd_dispatch(hipStream_t st, const AttnParams& p);
int attention_long_dispatch(hipStream_t st, const AttnParams& p, bool bwd);     // S > 272: attention_long.hip

}  // namespace mdt
