// Fused self-attention for the three token spaces of mDT (comment text S = nb+L, image
// patches S = nb+P, discussion-tree graph S = N+1): scores, structural / padding bias,
// fp32 softmax and P@V in one kernel; backward recomputes P from the saved log-sum-exp.
//
// One workgroup (4 waves) per (sequence, head); sequences are short (S <= 256), so the
// whole key range of a 16-query tile lives in MFMA accumulators (NT tiles of 16 keys)
// and softmax is a single exact pass — no online rescaling.
//   bf16: v_mfma_f32_16x16x32_bf16; K / V (fwd), then K,V / Q,dO (bwd) staged in LDS
//         (144-B padded rows); k-major operands (V in P@V, K in dS@K, dO / Q in the
//         key-tile pass) are fetched with ds_read_b64_tr_b16.
//   fp32: v_mfma_f32_16x16x4_f32 straight from global memory (parity path, exact fp32).
// The Graphormer structural bias (modules/graphormer_layers.py:86-110) is evaluated on
// the fly from int32 spatial_pos + the [num_spatial, H] table and never materialised;
// its gradient is reduced in an LDS histogram and leaves with one atomic per bin.
//
// Replaces modules/multihead_attention.py:139-202 and the eager attention inside HF
// BertLayer / ViTLayer (call sites modules/multi_graphormer_fusion_layer.py:94-96,138-146).
#include <type_traits>

#include <stdlib.h>

#include "attention_common.hpp"

namespace mdt {

template <typename T> struct MM;
template <> struct MM<float> {
  static constexpr int KS = 4;
  typedef float frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return mfma_f32(a, b, c); }
};
template <> struct MM<bf16_t> {
  static constexpr int KS = 32;
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return mfma_bf16(a, b, c); }
};

// Row-major [rows][ld] operand.  fp32: global memory, row index clamped to rows-1
// (padded positions re-read the last real row: finite, and always masked).  bf16: an LDS
// image (zero-filled padding) or global memory for the wave-private 16-row tiles.
template <typename T> struct Src {
  const T* p;
  int64_t ld;
  int rows;
};

// fragment of a 16 x KS block whose rows are the MFMA row (A) or column (B) index and
// whose k runs along the contiguous axis:  element(rc, k) = p[rc*ld + k]
template <typename T, bool LDS>
__device__ __forceinline__ typename MM<T>::frag frag_kc(const Src<T>& s, int rc0, int k0, int lane) {
  int rc = rc0 + (lane & 15);
  if (rc > s.rows - 1) rc = s.rows - 1;
  if constexpr (std::is_same<T, float>::value) {
    return s.p[rc * s.ld + k0 + (lane >> 4)];
  } else {
    const bf16_t* a = s.p + rc * s.ld + k0 + 8 * (lane >> 4);
    if constexpr (LDS) return *(const __attribute__((address_space(3))) bf16x8*)LDS_PTR(a);
    else return *(const bf16x8*)a;
  }
}

// fragment of a KS x 16 block stored k-major: element(k, c) = p[k*ld + c]  (B operand)
template <typename T>
__device__ __forceinline__ typename MM<T>::frag frag_km(const Src<T>& s, int c0, int k0, int lane) {
  if constexpr (std::is_same<T, float>::value) {
    int k = k0 + (lane >> 4);
    if (k > s.rows - 1) k = s.rows - 1;
    return s.p[k * s.ld + c0 + (lane & 15)];
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const bf16_t* a = s.p + (k0 + g * 8 + q) * s.ld + c0 + pp * 4;
    const bf16x4 lo = lds_read_tr16(a);
    const bf16x4 hi = lds_read_tr16(a + 4 * s.ld);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

constexpr int IMG_LD = 72;  // bf16 LDS image row stride (64 + 8 elements = 144 B)

// Stage a [S][HD] bf16 operand (rows = sequence positions) into an LDS image, zero padded
// to rows_pad rows.
template <int HD>
__device__ __forceinline__ void stage_image(bf16_t* img, const bf16_t* g, int64_t g_ld, int S, int rows_pad, int tid) {
  constexpr int CH = HD / 8;  // 16-B chunks per row
  for (int e = tid; e < rows_pad * CH; e += 256) {
    const int r = e / CH, c = e - r * CH;
    bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (r < S) v = *(const bf16x8*)(g + r * g_ld + c * 8);
    *(bf16x8*)(img + r * IMG_LD + c * 8) = v;
  }
}

template <typename T>
__device__ __forceinline__ int scratch_ld(int s_pad32) { return s_pad32 + (std::is_same<T, float>::value ? 1 : 8); }

// ---------------------------------------------------------------------------- forward
template <typename T, int HD, int NT, bool STRUCT, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams P) {
  constexpr bool BF = !std::is_same<T, float>::value;
  constexpr int KS = MM<T>::KS;
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, seq = blockIdx.y;
  const int SL = a.S, D = a.H * HD;                      // SL: lse / dropout-counter geometry
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;   // this sequence's length
  constexpr int s_pad32 = (NT * 16 + 31) & ~31;  // key range every tile loop covers
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  const T* qkv = (const T*)a.qkv + row0 * a.ld_qkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv;  // row stride between consecutive positions

  Src<T> srcK, srcV;
  T* scratch;
  if constexpr (BF) {
    bf16_t* imgK = (bf16_t*)smem;
    bf16_t* imgV = imgK + s_pad32 * IMG_LD;
    stage_image<HD>(imgK, qkv + D, tld, S, s_pad32, tid);
    stage_image<HD>(imgV, qkv + 2 * D, tld, S, s_pad32, tid);
    srcK = Src<T>{imgK, IMG_LD, s_pad32};
    srcV = Src<T>{imgV, IMG_LD, s_pad32};
    scratch = (T*)(imgV + s_pad32 * IMG_LD) + wave * 16 * scratch_ld<T>(s_pad32);
    __syncthreads();
  } else {
    srcK = Src<T>{qkv + D, tld, S};
    srcV = Src<T>{qkv + 2 * D, tld, S};
    scratch = (T*)smem + wave * 16 * scratch_ld<T>(s_pad32);
  }
  const int sld = scratch_ld<T>(s_pad32);
  const Src<T> srcQ{qkv, tld, S};
  const Src<T> srcP{scratch, sld, 16};
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};

  float colb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) colb[t] = key_only_bias<T>(bc, t * 16 + (lane & 15));
  // zero the k-padding columns of the scratch rows once (columns >= NT*16 up to s_pad32)
  if constexpr (s_pad32 > NT * 16) {
    for (int e = lane; e < 16 * (s_pad32 - NT * 16); e += 64) {
      const int r = e / (s_pad32 - NT * 16), c = NT * 16 + e % (s_pad32 - NT * 16);
      scratch[r * sld + c] = from_f32<T>(0.f);
    }
  }

  const int n_qt = (S + 15) >> 4;
  for (int qt = wave; qt < n_qt; qt += 4) {
    const int q0 = qt * 16;
    f32x4 sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < HD; k0 += KS) {
      const typename MM<T>::frag fa = frag_kc<T, false>(srcQ, q0, k0, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) sc[t] = MM<T>::mma(fa, frag_kc<T, BF>(srcK, t * 16, k0, lane), sc[t]);
    }
    // bias + softmax (rows = (lane>>4)*4 + r, cols = t*16 + (lane&15))
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key = t * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sc[t][r] * a.scale + colb[t];
        if (a.dense_bias || STRUCT) {
          int q = q0 + (lane >> 4) * 4 + r;
          if (q > S - 1) q = S - 1;
          if (key < S) v += pair_bias<T, STRUCT>(bc, q, key);
        }
        sc[t][r] = v;
        mx[r] = fmaxf(mx[r], v);
      }
    }
    float sum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mx[r] = row16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (sc[t][r] == -INFINITY) ? 0.f : __expf(sc[t][r] - mx[r]);
        sc[t][r] = e;
        sum[r] += e;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sum[r] = row16_sum(sum[r]);
      const int q = q0 + (lane >> 4) * 4 + r;
      if ((lane & 15) == 0 && q < S && a.lse)
        a.lse[((int64_t)seq * a.H + h) * SL + q] = (sum[r] > 0.f) ? mx[r] + __logf(sum[r]) : -INFINITY;
      sum[r] = (sum[r] > 0.f) ? 1.0f / sum[r] : 0.f;
    }
    const int drop_bh = seq * a.H + h;   // counters: attention_common.hpp
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = sc[t][r] * sum[r];
        if constexpr (DROP) {
          const int q = q0 + (lane >> 4) * 4 + r, key = t * 16 + (lane & 15);
          pv *= attn_drop_scale(P.drop, drop_bh, SL, q, key);
        }
        scratch[((lane >> 4) * 4 + r) * sld + t * 16 + (lane & 15)] = from_f32<T>(pv);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // O = P @ V
    f32x4 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kmax = BF ? s_pad32 : NT * 16;
    for (int k0 = 0; k0 < kmax; k0 += KS) {
      const typename MM<T>::frag fp = frag_kc<T, true>(srcP, 0, k0, lane);
#pragma unroll
      for (int d = 0; d < ND; ++d) o[d] = MM<T>::mma(fp, frag_km<T>(srcV, d * 16, k0, lane), o[d]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // scratch is rewritten by the next q tile
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + (lane >> 4) * 4 + r;
        if (q < S)
          ((T*)a.out)[(row0 + (int64_t)q * a.pos_stride) * a.ld_out + h * HD + d * 16 + (lane & 15)] = from_f32<T>(o[d][r]);
      }
  }
}

// ---------------------------------------------------------------------------- backward
// Pass A (per 16-query tile): S, P, dP = dO V^T, delta = rowsum(P*dP), dS = P*(dP-delta),
//                             dQ = scale * dS K, bias gradients.
// Pass B (per 16-key tile):   S^T, P^T, dP^T = V dO^T, dS^T, dV = P^T dO, dK = scale * dS^T Q.
template <typename T, int HD, int NT, bool STRUCT, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams P) {
  constexpr bool BF = !std::is_same<T, float>::value;
  constexpr int KS = MM<T>::KS;
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, seq = blockIdx.y;
  const int SL = a.S, D = a.H * HD;                      // SL: lse / dropout-counter geometry
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;   // this sequence's length
  constexpr int s_pad32 = (NT * 16 + 31) & ~31;  // key range every tile loop covers
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  const T* qkv = (const T*)a.qkv + row0 * a.ld_qkv + h * HD;
  const T* dout = (const T*)P.dout + row0 * P.ld_dout + h * HD;
  T* dqkv = (T*)P.dqkv + row0 * P.ld_dqkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv, dld = a.pos_stride * P.ld_dout, gld = a.pos_stride * P.ld_dqkv;
  const int sld = scratch_ld<T>(s_pad32);

  // LDS carve: [delta S_pad32 f32][lse S_pad32 f32][hist (num_spatial+1) f32]
  //            [image0][image1] (bf16 only) [scratch 4 waves]
  float* s_delta = (float*)smem;
  float* s_lse = s_delta + s_pad32;
  float* s_hist = s_lse + s_pad32;
  const int nhist = STRUCT ? ((a.num_spatial + 1 + 3) & ~3) : 0;
  char* after = (char*)(s_hist + nhist);
  bf16_t* img0 = (bf16_t*)after;
  bf16_t* img1 = img0 + (BF ? s_pad32 * IMG_LD : 0);
  T* scratch = (T*)(img1 + (BF ? s_pad32 * IMG_LD : 0)) + wave * 16 * sld;
  const Src<T> srcX{scratch, sld, 16};

  for (int i = tid; i < s_pad32; i += 256) {
    s_lse[i] = (i < S) ? a.lse[((int64_t)seq * a.H + h) * SL + i] : 0.f;
    s_delta[i] = 0.f;
  }
  for (int i = tid; i < nhist; i += 256) s_hist[i] = 0.f;

  Src<T> srcK, srcV;
  if constexpr (BF) {
    stage_image<HD>(img0, qkv + D, tld, S, s_pad32, tid);
    stage_image<HD>(img1, qkv + 2 * D, tld, S, s_pad32, tid);
    srcK = Src<T>{img0, IMG_LD, s_pad32};
    srcV = Src<T>{img1, IMG_LD, s_pad32};
  } else {
    srcK = Src<T>{qkv + D, tld, S};
    srcV = Src<T>{qkv + 2 * D, tld, S};
  }
  if constexpr (s_pad32 > NT * 16) {
    for (int e = lane; e < 16 * (s_pad32 - NT * 16); e += 64) {
      const int r = e / (s_pad32 - NT * 16), c = NT * 16 + e % (s_pad32 - NT * 16);
      scratch[r * sld + c] = from_f32<T>(0.f);
    }
  }
  __syncthreads();

  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  const int drop_bh = seq * a.H + h;   // counters: attention_common.hpp
  const Src<T> gQ{qkv, tld, S};
  const Src<T> gDO{dout, dld, S};
  const int n_t = (S + 15) >> 4;
  const int kmax = BF ? s_pad32 : NT * 16;

  // ------------------------------------------------------------------ pass A
  {
    float colb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) colb[t] = key_only_bias<T>(bc, t * 16 + (lane & 15));
    for (int qt = wave; qt < n_t; qt += 4) {
      const int q0 = qt * 16;
      f32x4 sc[NT], dp[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int k0 = 0; k0 < HD; k0 += KS) {
        const typename MM<T>::frag fq = frag_kc<T, false>(gQ, q0, k0, lane);
        const typename MM<T>::frag fo = frag_kc<T, false>(gDO, q0, k0, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          sc[t] = MM<T>::mma(fq, frag_kc<T, BF>(srcK, t * 16, k0, lane), sc[t]);
          dp[t] = MM<T>::mma(fo, frag_kc<T, BF>(srcV, t * 16, k0, lane), dp[t]);
          if (t & 1) __builtin_amdgcn_sched_barrier(0);   // bound operand prefetch depth (registers -> occupancy)
        }
      }
      float del[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int key = t * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int q = q0 + (lane >> 4) * 4 + r;
          const bool qok = q < S;
          if (q > S - 1) q = S - 1;
          float v = sc[t][r] * a.scale + colb[t];
          if ((a.dense_bias || STRUCT) && key < S) v += pair_bias<T, STRUCT>(bc, q, key);
          const float l = s_lse[q];
          const float p = (v == -INFINITY || l == -INFINITY || !qok) ? 0.f : __expf(v - l);
          sc[t][r] = p;
          if constexpr (DROP) dp[t][r] *= attn_drop_scale(P.drop, drop_bh, SL, q, key);   // dP = dD * M / (1-p)
          del[r] += p * dp[t][r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        del[r] = row16_sum(del[r]);
        const int q = q0 + (lane >> 4) * 4 + r;
        if ((lane & 15) == 0 && q < S) s_delta[q] = del[r];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int key = t * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = sc[t][r] * (dp[t][r] - del[r]);
          const int q = q0 + (lane >> 4) * 4 + r;
          scratch[((lane >> 4) * 4 + r) * sld + key] = from_f32<T>(ds);
          if (q < S && key < S) {
            if (P.d_dense_bias) P.d_dense_bias[(((int64_t)seq * a.H + h) * S + q) * S + key] = ds;
            if constexpr (STRUCT) {
              if (P.d_sp_table && ds != 0.f) {
                if (q >= 1 && key >= 1) {
                  const int idx = a.spatial_pos[((int64_t)seq * (S - 1) + (q - 1)) * (S - 1) + (key - 1)];
                  if (idx != 0) atomicAdd(s_hist + idx, ds);  // nn.Embedding(padding_idx=0): row 0 gets no gradient
                } else {
                  atomicAdd(s_hist + a.num_spatial, ds);
                }
              }
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 dq[ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < kmax; k0 += KS) {
        const typename MM<T>::frag fs = frag_kc<T, true>(srcX, 0, k0, lane);
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = MM<T>::mma(fs, frag_km<T>(srcK, d * 16, k0, lane), dq[d]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q0 + (lane >> 4) * 4 + r;
          if (q < S) dqkv[(int64_t)q * gld + d * 16 + (lane & 15)] = from_f32<T>(dq[d][r] * a.scale);
        }
    }
  }
  __syncthreads();  // delta complete; K / V images no longer needed
  Src<T> srcQ, srcDO;
  if constexpr (BF) {
    stage_image<HD>(img0, qkv, tld, S, s_pad32, tid);
    stage_image<HD>(img1, dout, dld, S, s_pad32, tid);
    srcQ = Src<T>{img0, IMG_LD, s_pad32};
    srcDO = Src<T>{img1, IMG_LD, s_pad32};
    __syncthreads();
  } else {
    srcQ = gQ;
    srcDO = gDO;
  }
  if constexpr (STRUCT) {
    if (P.d_sp_table) {
      for (int i = tid; i <= a.num_spatial; i += 256) {
        const float v = s_hist[i];
        if (v != 0.f) {
          if (i < a.num_spatial) atomicAdd(P.d_sp_table + (int64_t)i * a.H + h, v);
          else if (P.d_virt) atomicAdd(P.d_virt + h, v);
        }
      }
    }
  }
  // ------------------------------------------------------------------ pass B
  {
    const Src<T> gK{qkv + D, tld, S};
    const Src<T> gV{qkv + 2 * D, tld, S};
    for (int kt = wave; kt < n_t; kt += 4) {
      const int key0 = kt * 16;
      f32x4 sc[NT], dp[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int k0 = 0; k0 < HD; k0 += KS) {
        const typename MM<T>::frag fk = frag_kc<T, false>(gK, key0, k0, lane);
        const typename MM<T>::frag fv = frag_kc<T, false>(gV, key0, k0, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          sc[t] = MM<T>::mma(fk, frag_kc<T, BF>(srcQ, t * 16, k0, lane), sc[t]);
          dp[t] = MM<T>::mma(fv, frag_kc<T, BF>(srcDO, t * 16, k0, lane), dp[t]);
          if (t & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      // rows = keys key0 + (lane>>4)*4 + r, cols = queries t*16 + (lane&15)
      float kb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) kb[r] = key_only_bias<T>(bc, key0 + (lane >> 4) * 4 + r);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int q = t * 16 + (lane & 15);
        const bool qok = q < S;
        const int qc = qok ? q : S - 1;
        const float l = s_lse[qc], de = s_delta[qc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + (lane >> 4) * 4 + r;
          float v = sc[t][r] * a.scale + kb[r];
          if ((a.dense_bias || STRUCT) && key < S) v += pair_bias<T, STRUCT>(bc, qc, key);
          const float p = (v == -INFINITY || l == -INFINITY || !qok) ? 0.f : __expf(v - l);
          float ds = dp[t][r];
          float pd = p;
          if constexpr (DROP) {
            const float m = attn_drop_scale(P.drop, drop_bh, SL, qc, key);
            ds *= m;                          // dP^T = dD^T * M / (1-p)
            pd *= m;                          // D^T  = P^T * M / (1-p)
          }
          sc[t][r] = pd;                      // (dropped) P^T, the operand of dV
          dp[t][r] = p * (ds - de);           // dS^T
        }
      }
      // dV = P^T dO
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[((lane >> 4) * 4 + r) * sld + t * 16 + (lane & 15)] = from_f32<T>(sc[t][r]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 acc[ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < kmax; k0 += KS) {
        const typename MM<T>::frag fs = frag_kc<T, true>(srcX, 0, k0, lane);
#pragma unroll
        for (int d = 0; d < ND; ++d) acc[d] = MM<T>::mma(fs, frag_km<T>(srcDO, d * 16, k0, lane), acc[d]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + (lane >> 4) * 4 + r;
          if (key < S) dqkv[(int64_t)key * gld + 2 * D + d * 16 + (lane & 15)] = from_f32<T>(acc[d][r]);
        }
      // dK = scale * dS^T Q
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[((lane >> 4) * 4 + r) * sld + t * 16 + (lane & 15)] = from_f32<T>(dp[t][r]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < ND; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < kmax; k0 += KS) {
        const typename MM<T>::frag fs = frag_kc<T, true>(srcX, 0, k0, lane);
#pragma unroll
        for (int d = 0; d < ND; ++d) acc[d] = MM<T>::mma(fs, frag_km<T>(srcQ, d * 16, k0, lane), acc[d]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + (lane >> 4) * 4 + r;
          if (key < S) dqkv[(int64_t)key * gld + D + d * 16 + (lane & 15)] = from_f32<T>(acc[d][r] * a.scale);
        }
    }
  }
}

// materialised structural bias (API parity with GraphAttnBias.forward)
template <typename T>
__global__ void graph_attn_bias_kernel(int nseq, int S, int H, const float* attn_bias, const int32_t* sp, const T* table,
                                       const T* virt, float* out) {
  const int64_t n = (int64_t)nseq * H * S * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int key = (int)(i % S);
    const int q = (int)((i / S) % S);
    const int h = (int)((i / ((int64_t)S * S)) % H);
    const int seq = (int)(i / ((int64_t)S * S * H));
    BiasCtx bc{seq, h, S, H, nullptr, nullptr, nullptr, attn_bias, sp, table, virt};
    out[i] = pair_bias<T, true>(bc, q, key);
  }
}

// Head-averaged attention probabilities (modules/multihead_attention.py:205-214, need_weights=True): recomputed from the
// saved q, k and the forward's log-sum-exp — one thread per (sequence, query, key), looping over the heads.  Module-level
// API only (the encoder layers pass need_weights=False); fp32 output [nseq, S, S], masked keys give 0.
template <typename T>
__global__ __launch_bounds__(256) void attn_mean_probs_kernel(mdt_attn_fwd_args a, float* out) {
  const int S = a.S, H = a.H, hd = a.hd;
  const int64_t n = (int64_t)a.nseq * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int key = (int)(i % S), q = (int)((i / S) % S), seq = (int)(i / ((int64_t)S * S));
    const T* qrow = (const T*)a.qkv + ((int64_t)seq * a.seq_stride + (int64_t)q * a.pos_stride) * a.ld_qkv;
    const T* krow = (const T*)a.qkv + ((int64_t)seq * a.seq_stride + (int64_t)key * a.pos_stride) * a.ld_qkv + (int64_t)H * hd;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
      BiasCtx bc{seq, h, S, H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
      float v = key_only_bias<T>(bc, key);
      if (v == 0.f) {
        float dot = 0.f;
        for (int d = 0; d < hd; ++d) dot += to_f32(qrow[h * hd + d]) * to_f32(krow[h * hd + d]);
        v = dot * a.scale + (a.attn_bias ? pair_bias<T, true>(bc, q, key) : pair_bias<T, false>(bc, q, key));
        const float l = a.lse[((int64_t)seq * H + h) * S + q];
        acc += (v == -INFINITY || l == -INFINITY) ? 0.f : __expf(v - l);
      }
    }
    out[i] = acc / (float)H;
  }
}

template <typename T, int HD, int NT, bool STRUCT, bool DROP>
static int launch_fwd(hipStream_t st, const AttnParams& p) {
  constexpr int s_pad32 = (NT * 16 + 31) & ~31;
  constexpr bool BF = !std::is_same<T, float>::value;
  const int sld = s_pad32 + (BF ? 8 : 1);
  size_t lds = (size_t)4 * 16 * sld * sizeof(T) + (BF ? (size_t)2 * s_pad32 * IMG_LD * 2 : 0);
  auto kern = attn_fwd_kernel<T, HD, NT, STRUCT, DROP>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("attention_fwd: cannot reserve %zu bytes of LDS", lds);
      return MDT_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(p.f.H, p.f.nseq), 256, lds, st, p);
  return check_launch("attention_fwd");
}

template <typename T, int HD, int NT, bool STRUCT, bool DROP>
static int launch_bwd(hipStream_t st, const AttnParams& p) {
  constexpr int s_pad32 = (NT * 16 + 31) & ~31;
  constexpr bool BF = !std::is_same<T, float>::value;
  const int sld = s_pad32 + (BF ? 8 : 1);
  const int nhist = STRUCT ? ((p.f.num_spatial + 1 + 3) & ~3) : 0;
  size_t lds = (size_t)(2 * s_pad32 + nhist) * 4 + (size_t)4 * 16 * sld * sizeof(T) +
               (BF ? (size_t)2 * s_pad32 * IMG_LD * 2 : 0);
  auto kern = attn_bwd_kernel<T, HD, NT, STRUCT, DROP>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("attention_bwd: cannot reserve %zu bytes of LDS", lds);
      return MDT_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(p.f.H, p.f.nseq), 256, lds, st, p);
  return check_launch("attention_bwd");
}

template <typename T, int HD, bool STRUCT, bool BWD>
static int dispatch_nt(hipStream_t st, const AttnParams& p) {
  const int nt = (p.f.S + 15) / 16;
#define ATT_CASE(N_)                                                                              \
  if (nt <= N_) {                                                                                 \
    if (p.f.drop_p > 0.f) return BWD ? launch_bwd<T, HD, N_, STRUCT, true>(st, p) : launch_fwd<T, HD, N_, STRUCT, true>(st, p); \
    return BWD ? launch_bwd<T, HD, N_, STRUCT, false>(st, p) : launch_fwd<T, HD, N_, STRUCT, false>(st, p);                    \
  }
  ATT_CASE(2) ATT_CASE(5) ATT_CASE(7) ATT_CASE(9) ATT_CASE(13) ATT_CASE(17)      // 17 tiles: ViT-L/14, 4 + 257 tokens
#undef ATT_CASE
  set_error("attention: S=%d exceeds the 272-token limit of the single-pass kernel", p.f.S);
  return MDT_ERR_UNSUPPORTED;
}

template <bool BWD>
static int dispatch(hipStream_t st, const AttnParams& p) {
  const mdt_attn_fwd_args& a = p.f;
  const bool st_bias = a.attn_bias != nullptr;
  const bool binned = a.seq_ids != nullptr || a.s_cap > 0;        // length bins of a ragged set: the v2 forward / v3-v4 backward only
  if (binned) {
    if (a.S > 272 || a.dtype != MDT_BF16 || a.hd != 64 || !a.seq_offsets || switches().attn_v1) {
      set_error("attention: seq_ids / s_cap are for ragged bf16 launches with head_dim 64 and S <= 272");
      return MDT_ERR_UNSUPPORTED;
    }
    return BWD ? attention_v3_bwd_dispatch(st, p) : attention_v2_dispatch(st, p, false);
  }
  if (a.S > 272) return attention_long_dispatch(st, p, BWD);      // discussion trees with more than 271 comments
  if (a.dtype == MDT_BF16) {
    // a plain dense bias (no structural terms) is only handled by the kernels in this file
    const bool dense_only = (a.dense_bias != nullptr || p.d_dense_bias != nullptr) && !st_bias;
    if (a.hd == 64 && !dense_only && !switches().attn_v1) {
      if (!BWD) return attention_v2_dispatch(st, p, false);              // forward: register-resident P
      // backward, measured at C2 shapes (profiles/round1_attention_v2.txt): S <= 112 -> whole-row v2,
      // longer sequences -> chunked v3 / one-pass v4; tiny graphs and short rows (S <= 80) stay on the LDS-scratch kernel below
      const char* force = switches().attn_bwd[0] ? switches().attn_bwd : nullptr;      // MDT_ATTN_BWD = "v1" | "v2" | "v3" for A/B runs
      const bool drop = a.drop_p > 0.f;                // with dropout the whole-row v2 falls to 1 wave / SIMD: chunked v3 wins
      if (BWD && a.S > 256 && !force) return attention_v3_bwd_dispatch(st, p);   // ViT-L/14: 4 + 257 tokens
      // (plain rows of up to 80 tokens take the one-pass kernel since round 4 — 3 x faster, 749 -> 225 us on 2048 ragged sequences
      // of 10-64 tokens — now that its short-row form sums delta = sum P o dP itself, in fp32, like the scratch kernel does
      // (attn_bwd_v4x); graphs with a structural bias stay here)
      // neither the scratch kernel nor the whole-row v2 backward knows of q_limit (they would read the out / lse rows the v2 FORWARD
      // skipped — unwritten memory): such launches take v3 / v4 whatever MDT_ATTN_BWD says (tests: ..._never_reads_what_forward_did_not_write)
      const bool v1 = force ? (!strcmp(force, "v1") && a.q_limit == 0) : (a.S <= 80 && a.q_limit == 0 && (st_bias || !switches().attn_exact_delta));
      const bool v2 = (force ? !strcmp(force, "v2") : !drop) && a.q_limit == 0 && a.S <= 112;
      if (!v1) return v2 ? attention_v2_dispatch(st, p, true) : attention_v3_bwd_dispatch(st, p);
    }
    if (a.hd == 64) return st_bias ? dispatch_nt<bf16_t, 64, true, BWD>(st, p) : dispatch_nt<bf16_t, 64, false, BWD>(st, p);
    set_error("attention(bf16): head_dim %d unsupported (64 only)", a.hd);
    return MDT_ERR_UNSUPPORTED;
  }
  if (a.hd == 64) return st_bias ? dispatch_nt<float, 64, true, BWD>(st, p) : dispatch_nt<float, 64, false, BWD>(st, p);
  if (a.hd == 16) return st_bias ? dispatch_nt<float, 16, true, BWD>(st, p) : dispatch_nt<float, 16, false, BWD>(st, p);
  set_error("attention(fp32): head_dim %d unsupported (16 or 64)", a.hd);
  return MDT_ERR_UNSUPPORTED;
}

static int check_args(const mdt_attn_fwd_args& a) {
  MDT_CHECK_ARG(a.dtype == MDT_F32 || a.dtype == MDT_BF16, "attention: bad dtype %d", a.dtype);
  MDT_CHECK_ARG(a.nseq >= 0 && a.S > 0 && a.H > 0 && a.hd > 0, "attention: bad shape nseq=%d S=%d H=%d hd=%d", a.nseq,
                a.S, a.H, a.hd);
  MDT_CHECK_ARG(a.qkv && a.out && a.lse, "attention: null qkv / out / lse");
  MDT_CHECK_ARG(a.ld_qkv >= 3 * a.H * a.hd && a.ld_out >= a.H * a.hd, "attention: row strides too small");
  if (a.dtype == MDT_BF16)
    MDT_CHECK_ARG(a.ld_qkv % 8 == 0 && ((uintptr_t)a.qkv & 15) == 0, "attention(bf16): qkv must be 16-byte aligned rows");
  if (a.attn_bias) MDT_CHECK_ARG(a.spatial_pos && a.sp_table && a.virt && a.num_spatial > 0, "attention: incomplete structural bias");
  MDT_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f, "attention: dropout p=%f out of [0,1)", a.drop_p);
  if (a.seq_offsets)
    MDT_CHECK_ARG(a.pos_stride == 1 && !a.key_mask && !a.key_pad && !a.dense_bias && !a.attn_bias,
                  "attention: ragged sequences (seq_offsets) take no masks / biases and need pos_stride == 1");
  MDT_CHECK_ARG(a.s_cap >= 0 && a.s_cap <= a.S, "attention: s_cap=%d outside [0, S=%d]", a.s_cap, a.S);
  MDT_CHECK_ARG(!a.seq_ids || (a.seq_offsets && (a.nseq_total == 0 || a.nseq_total >= a.nseq)),
                "attention: seq_ids need seq_offsets (and nseq_total >= nseq)");
  MDT_CHECK_ARG(a.drop_p == 0.f || (uint64_t)(a.nseq_total > a.nseq ? a.nseq_total : a.nseq) * a.H * a.S * (a.S + 1) < (1ull << 32),
                "attention: dropout counters are 32-bit (nseq*H*S*(S+1) must stay below 2^32)");
  return MDT_OK;
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_attention_fwd(void* stream, const mdt_attn_fwd_args* a) {
  MDT_CHECK_ARG(a, "attention_fwd: null args");
  if (a->nseq == 0) return MDT_OK;
  if (int e = check_args(*a)) return e;
  MDT_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, "attention_fwd: dropout p=%f out of [0,1)", a->drop_p);
  AttnParams p;
  memset(&p, 0, sizeof(p));
  p.f = *a;
  p.drop = make_drop(a->drop_p, a->drop_seed);
  return dispatch<false>((hipStream_t)stream, p);
}

extern "C" int mdt_attention_bwd(void* stream, const mdt_attn_bwd_args* a) {
  MDT_CHECK_ARG(a, "attention_bwd: null args");
  if (a->f.nseq == 0) return MDT_OK;
  if (int e = check_args(a->f)) return e;
  MDT_CHECK_ARG(a->dout && a->dqkv, "attention_bwd: null dout / dqkv");
  MDT_CHECK_ARG(a->ld_dqkv >= 3 * a->f.H * a->f.hd, "attention_bwd: ld_dqkv too small");
  if (a->f.dtype == MDT_BF16)
    MDT_CHECK_ARG(a->ld_dout % 8 == 0 && ((uintptr_t)a->dout & 15) == 0, "attention_bwd(bf16): dout must be 16-byte aligned rows");
  AttnParams p;
  p.f = a->f;
  p.drop = make_drop(a->f.drop_p, a->f.drop_seed);
  p.dout = a->dout; p.ld_dout = a->ld_dout; p.dqkv = a->dqkv; p.ld_dqkv = a->ld_dqkv;
  p.d_dense_bias = a->d_dense_bias; p.d_sp_table = a->d_sp_table; p.d_virt = a->d_virt;
  return dispatch<true>((hipStream_t)stream, p);
}

extern "C" int mdt_attention_mean_probs(void* stream, const mdt_attn_fwd_args* a, float* out) {
  MDT_CHECK_ARG(a && out, "attention_mean_probs: null args");
  if (a->nseq == 0) return MDT_OK;
  if (int e = check_args(*a)) return e;
  MDT_CHECK_ARG(!a->seq_offsets, "attention_mean_probs: ragged sequences are not supported (module-level API only)");
  const int64_t n = (int64_t)a->nseq * a->S * a->S;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (a->dtype == MDT_F32) hipLaunchKernelGGL((attn_mean_probs_kernel<float>), grid, 256, 0, (hipStream_t)stream, *a, out);
  else hipLaunchKernelGGL((attn_mean_probs_kernel<bf16_t>), grid, 256, 0, (hipStream_t)stream, *a, out);
  return check_launch("attention_mean_probs");
}

// Per-head attention weights (modules/multihead_attention.py:91-102,186-214: ``need_head_weights`` — the softmax probabilities
// BEFORE dropout of every head — and ``before_softmax`` — the raw scores q k^T * scale + bias with masked keys at -inf),
// recomputed like the head average above: one thread per (sequence, head, query, key); fp32 output [nseq, H, S, S].
// Module-level API only (nothing in mDT asks for them).
template <typename T, bool RAW>
__global__ __launch_bounds__(256) void attn_head_weights_kernel(mdt_attn_fwd_args a, float* out) {
  const int S = a.S, H = a.H, hd = a.hd;
  const int64_t n = (int64_t)a.nseq * H * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int key = (int)(i % S), q = (int)((i / S) % S), h = (int)((i / ((int64_t)S * S)) % H), seq = (int)(i / ((int64_t)S * S * H));
    const T* qrow = (const T*)a.qkv + ((int64_t)seq * a.seq_stride + (int64_t)q * a.pos_stride) * a.ld_qkv + h * hd;
    const T* krow = (const T*)a.qkv + ((int64_t)seq * a.seq_stride + (int64_t)key * a.pos_stride) * a.ld_qkv + (int64_t)H * hd + h * hd;
    BiasCtx bc{seq, h, S, H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
    float v = key_only_bias<T>(bc, key);
    if (v == 0.f) {
      float dot = 0.f;
      for (int d = 0; d < hd; ++d) dot += to_f32(qrow[d]) * to_f32(krow[d]);
      v = dot * a.scale + (a.attn_bias ? pair_bias<T, true>(bc, q, key) : pair_bias<T, false>(bc, q, key));
    }
    if constexpr (RAW) {
      out[i] = v;
    } else {
      const float l = a.lse[((int64_t)seq * H + h) * S + q];
      out[i] = (v == -INFINITY || l == -INFINITY) ? 0.f : __expf(v - l);
    }
  }
}

extern "C" int mdt_attention_head_weights(void* stream, const mdt_attn_fwd_args* a, int raw_scores, float* out) {
  MDT_CHECK_ARG(a && out, "attention_head_weights: null args");
  if (a->nseq == 0) return MDT_OK;
  if (int e = check_args(*a)) return e;
  MDT_CHECK_ARG(!a->seq_offsets, "attention_head_weights: ragged sequences are not supported (module-level API only)");
  MDT_CHECK_ARG(raw_scores || a->lse, "attention_head_weights: probabilities need the forward's log-sum-exp");
  const int64_t n = (int64_t)a->nseq * a->H * a->S * a->S;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == MDT_F32) {
    if (raw_scores) hipLaunchKernelGGL((attn_head_weights_kernel<float, true>), grid, 256, 0, st, *a, out);
    else hipLaunchKernelGGL((attn_head_weights_kernel<float, false>), grid, 256, 0, st, *a, out);
  } else {
    if (raw_scores) hipLaunchKernelGGL((attn_head_weights_kernel<bf16_t, true>), grid, 256, 0, st, *a, out);
    else hipLaunchKernelGGL((attn_head_weights_kernel<bf16_t, false>), grid, 256, 0, st, *a, out);
  }
  return check_launch("attention_head_weights");
}

extern "C" int mdt_graph_attn_bias(void* stream, int dtype, int nseq, int S, int H, const float* attn_bias,
                                   const int32_t* spatial_pos, const void* sp_table, const void* virt, float* out) {
  MDT_CHECK_ARG(attn_bias && spatial_pos && sp_table && virt && out, "graph_attn_bias: null pointer");
  if (nseq == 0) return MDT_OK;
  const int64_t n = (int64_t)nseq * H * S * S;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDT_F32) hipLaunchKernelGGL((graph_attn_bias_kernel<float>), grid, 256, 0, st, nseq, S, H, attn_bias, spatial_pos, (const float*)sp_table, (const float*)virt, out);
  else if (dtype == MDT_BF16) hipLaunchKernelGGL((graph_attn_bias_kernel<bf16_t>), grid, 256, 0, st, nseq, S, H, attn_bias, spatial_pos, (const bf16_t*)sp_table, (const bf16_t*)virt, out);
  else MDT_UNSUPPORTED("graph_attn_bias: dtype %d", dtype);
  return check_launch("graph_attn_bias");
}
