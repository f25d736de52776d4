// LayerNorm forward / backward: one wavefront per row, the row held in registers
// (two-pass mean / variance, no re-read), 16-byte vector loads, fp32 statistics.
// HBM-bound: algorithmic bytes = rows * D * sizeof(T) * 2 (fwd), * 3 (+ residual 4) (bwd).
// Replaces fairseq LayerNorm / HF nn.LayerNorm (modules/graphormer_graph_encoder_layer.py:
// 127-130,138-141; modules/multigraphormer_graph_encoder.py:400-403; HF BertLayer/ViTLayer).
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace mdt {

template <typename T> struct Vec;  // 16-byte vector of T
template <> struct Vec<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

// 8-byte vectors: D = 768 is 64 lanes x 12 elements = three 8-byte vectors per lane with EVERY lane busy; as 16-byte vectors it is one
// and a half (the second one on lanes 0-31 only: a quarter of the row's arithmetic slots idle).  Wave instructions still cover
// 512 contiguous bytes.
template <typename T> struct VecH;
template <> struct VecH<bf16_t> {
  static constexpr int N = 4;
  bf16x4 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

// NV = number of 16-B vectors per lane: D = NV * 64 * Vec<T>::N exactly, or masked tail.
// A wave walks rows wid, wid + nwaves, ...; the next row is requested before the current one is reduced, so a wave keeps
// two rows (and an 8-workgroup CU 64 of them, ~100 KiB) in flight — with one row per wave and no look-ahead the kernel
// moved 4.1 TB/s, bounded by bytes in flight over the ~2 us of an HBM round trip.
// Q8 (bf16 only): y ALSO leaves as fp8 (1: e4m3, 2: e5m2) for the 8-bit GEMM that consumes it — q8 = saturate(bf16(y) * *q8_scale),
// *q8_amax = max(*q8_amax, max |bf16(y)|), the bytes and the maximum mdt_fp8_quantize would make of y (fp8.hip), without its pass
template <typename T, int NV, int Q8 = 0, typename V = Vec<T>>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int64_t rows, int D, const T* __restrict__ x, int64_t ldx,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta,
                                                            float eps, T* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            uint8_t* __restrict__ q8 = nullptr, int64_t ldq = 0,
                                                            const float* __restrict__ q8_scale = nullptr, float* __restrict__ q8_amax = nullptr) {
  constexpr int VN = V::N;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float qmax = 0.f;
  const float qs = Q8 != 0 ? *q8_scale : 1.0f;
  __shared__ float s_qmax[4];
  if constexpr (Q8 != 0) {
    if (threadIdx.x < 4) s_qmax[threadIdx.x] = 0.f;
    __syncthreads();
  }
  if (row < rows) {
  V gv[NV], bv[NV], xv[NV], xn[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * VN;
    if (c < D) {
      xv[i] = *(const V*)(x + row * ldx + c);
      gv[i] = *(const V*)(gamma + c);
      bv[i] = *(const V*)(beta + c);
    }
  }
  for (;;) {
    const int64_t next = row + nwaves;
    if (next < rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * VN;
        if (c < D) xn[i] = *(const V*)(x + next * ldx + c);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < VN; ++j) s += xv[i].get(j);
      }
    }
    const float mu = wave_sum_dpp(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < VN; ++j) { const float d = xv[i].get(j) - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum_dpp(q) / (float)D + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    T* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) o.set(j, (xv[i].get(j) - mu) * rs * gv[i].get(j) + bv[i].get(j));
        *(V*)(yr + c) = o;
        if constexpr (Q8 != 0 && (VN == 8 || VN == 4)) {
          constexpr float FMAX = Q8 == 1 ? 448.0f : 57344.0f;
          // mirrors fp8.hip: a NaN / infinity passes through as NaN (the MFMA propagates it to the loss) and marks the running
          // maximum +inf, which fp8_scale_update leaves the scale alone for
          float qv[VN];
#pragma unroll
          for (int j = 0; j < VN; ++j) {
            const float r = o.get(j);                 // the ROUNDED output
            const bool bad = !(fabsf(r) <= 3.0e38f);
            qmax = bad ? __builtin_inff() : fmaxf(qmax, fabsf(r));
            const float xq = r * qs;
            qv[j] = (xq != xq) ? xq : fminf(fmaxf(xq, -FMAX), FMAX);
          }
          int w[VN / 4];
#pragma unroll
          for (int k = 0; k < VN / 4; ++k) {
            w[k] = 0;
            if constexpr (Q8 == 1) {
              w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4 * k], qv[4 * k + 1], w[k], false); w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4 * k + 2], qv[4 * k + 3], w[k], true);
            } else {
              w[k] = __builtin_amdgcn_cvt_pk_bf8_f32(qv[4 * k], qv[4 * k + 1], w[k], false); w[k] = __builtin_amdgcn_cvt_pk_bf8_f32(qv[4 * k + 2], qv[4 * k + 3], w[k], true);
            }
          }
          if constexpr (VN == 8) *(int2*)(q8 + row * ldq + c) = make_int2(w[0], w[1]);
          else *(int*)(q8 + row * ldq + c) = w[0];
        }
      }
    }
    if (next >= rows) break;
    row = next;
#pragma unroll
    for (int i = 0; i < NV; ++i) xv[i] = xn[i];
  }
  }
  if constexpr (Q8 != 0) {          // one atomic per workgroup (non-negative floats order like their bit patterns)
    qmax = wave_max(qmax);
    if (lane == 0) s_qmax[threadIdx.x >> 6] = qmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(s_qmax[0], s_qmax[1]), fmaxf(s_qmax[2], s_qmax[3]));
      if (m > 0.f) atomicMax((int*)q8_amax, __float_as_int(m));
    }
  }
}

// BERT embeddings and their LayerNorm in one pass (HF BertEmbeddings as run by the truncated BertModel,
// modules/multigraphormer_graph_encoder.py:325-329): x = T(word[ids] + type[types] + pos[pos_ids]) — rounded to the storage type
// exactly as mdt_bert_embed_rows leaves it — then y = LayerNorm(x).  The summed row never makes the round trip through HBM between
// the two; it is written out (xs) only when a backward pass will read it.  One wave per row, the next row's three gathers in flight.
template <typename T, int NV, typename V = Vec<T>>
__global__ __launch_bounds__(256) void bert_embed_ln_rows_kernel(int64_t rows, int D, const int32_t* __restrict__ ids,
                                                                 const int32_t* __restrict__ types, const int32_t* __restrict__ pos_ids,
                                                                 const T* __restrict__ word, const T* __restrict__ pos,
                                                                 const T* __restrict__ typ, const T* __restrict__ gamma,
                                                                 const T* __restrict__ beta, float eps, T* __restrict__ xs, int64_t ldxs,
                                                                 T* __restrict__ y, int64_t ldy, float* __restrict__ mean,
                                                                 float* __restrict__ rstd) {
  constexpr int VN = V::N;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  V gv[NV], bv[NV], wv[NV], pv[NV], tv[NV], wn[NV], pn[NV], tn[NV];
  auto request = [&](int64_t r, V* w_, V* p_, V* t_) {
    const int64_t wi = ids[r], pi = pos_ids[r], ti = types[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
        w_[i] = *(const V*)(word + wi * D + c);
        p_[i] = *(const V*)(pos + pi * D + c);
        t_[i] = *(const V*)(typ + ti * D + c);
      }
    }
  };
  request(row, wv, pv, tv);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * VN;
    if (c < D) { gv[i] = *(const V*)(gamma + c); bv[i] = *(const V*)(beta + c); }
  }
  for (;;) {
    const int64_t next = row + nwaves;
    if (next < rows) request(next, wn, pn, tn);
    V xv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          xv[i].set(j, wv[i].get(j) + tv[i].get(j) + pv[i].get(j));      // the order and the rounding of bert_embed_rows_kernel
          s += xv[i].get(j);
        }
        if (xs) *(V*)(xs + row * ldxs + c) = xv[i];
      }
    }
    const float mu = wave_sum_dpp(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < VN; ++j) { const float d = xv[i].get(j) - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum_dpp(q) / (float)D + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) o.set(j, (xv[i].get(j) - mu) * rs * gv[i].get(j) + bv[i].get(j));
        *(V*)(y + row * ldy + c) = o;
      }
    }
    if (next >= rows) break;
    row = next;
#pragma unroll
    for (int i = 0; i < NV; ++i) { wv[i] = wn[i]; pv[i] = pn[i]; tv[i] = tn[i]; }
  }
}

// Backward: each wave walks rows_per_wave consecutive rows (the next row's x / dy / residual-gradient vectors are
// requested before the current row is reduced), keeps per-lane partial dgamma / dbeta / column sums in registers,
// the block combines them through ONE [4 waves][row] LDS buffer used three times (16 KiB at D = 768: the 48 KiB of
// three buffers held a CU to three workgroups) and issues one float atomic per column.
template <typename T, int NV, bool TAIL, bool CS = TAIL, typename V = Vec<T>>   // TAIL: second (dropped) output and / or column sums; CS: the column sums
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int64_t rows, int D, const T* __restrict__ dy, int64_t lddy,
                                                            const T* __restrict__ x, int64_t ldx,
                                                            const T* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ add,
                                                            int64_t ldadd, T* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows_per_wave, T* __restrict__ dxd, int64_t lddxd,
                                                            DropCfg drop, float* __restrict__ colsum) {
  constexpr int VN = V::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [4 waves][NV*64*VN]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  const int64_t r1 = (r0 + rows_per_wave < rows) ? r0 + rows_per_wave : rows;
  float pg[NV][VN], pb[NV][VN], pc[CS ? NV : 1][CS ? VN : 1];
  V gv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * VN;
    if (c < D) gv[i] = *(const V*)(gamma + c);
#pragma unroll
    for (int j = 0; j < VN; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; if constexpr (CS) pc[i][j] = 0.f; }
  }
  V xv[NV], gy[NV], av[NV], xn[NV], gn[NV], an[NV];
  float mu = 0.f, rs = 0.f, mu_n = 0.f, rs_n = 0.f;
  if (r0 < r1) {
    mu = mean[r0]; rs = rstd[r0];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
        xv[i] = *(const V*)(x + r0 * ldx + c);
        gy[i] = *(const V*)(dy + r0 * lddy + c);
        if (add) av[i] = *(const V*)(add + r0 * ldadd + c);
      }
    }
  }
  for (int64_t row = r0; row < r1; ++row) {
    if (row + 1 < r1) {
      mu_n = mean[row + 1]; rs_n = rstd[row + 1];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * VN;
        if (c < D) {
          xn[i] = *(const V*)(x + (row + 1) * ldx + c);
          gn[i] = *(const V*)(dy + (row + 1) * lddy + c);
          if (add) an[i] = *(const V*)(add + (row + 1) * ldadd + c);
        }
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          const float xh = (xv[i].get(j) - mu) * rs;
          const float g = gy[i].get(j);
          const float gg = g * gv[i].get(j);
          s1 += gg;
          s2 += gg * xh;
          pg[i][j] += g * xh;
          pb[i][j] += g;
        }
      }
    }
    const float m1 = wave_sum_dpp(s1) / (float)D, m2 = wave_sum_dpp(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      if (c < D) {
        V o, od;
        float ds0 = 0.f, ds1 = 0.f;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          const float xh = (xv[i].get(j) - mu) * rs;
          float v = rs * (gy[i].get(j) * gv[i].get(j) - m1 - xh * m2);
          if (add) v += av[i].get(j);
          o.set(j, v);
          if constexpr (TAIL) {
            if (dxd) {   // second output: the gradient of the dense layer behind a hidden dropout
              if ((j & 1) == 0) drop_scale2(drop, (uint64_t)row * D + c + j, ds0, ds1);   // D and c are even
              v = o.get(j) * ((j & 1) ? ds1 : ds0);
              od.set(j, v);
              v = od.get(j);
            } else {
              v = o.get(j);
            }
            if constexpr (CS) pc[i][j] += v;
          }
        }
        *(V*)(dx + row * lddx + c) = o;
        if constexpr (TAIL) {
          if (dxd) *(V*)(dxd + row * lddxd + c) = od;
        }
      }
    }
    mu = mu_n; rs = rs_n;
#pragma unroll
    for (int i = 0; i < NV; ++i) { xv[i] = xn[i]; gy[i] = gn[i]; av[i] = an[i]; }
  }
  if (!dgamma && !dbeta && !colsum) return;
  const int W = NV * 64 * VN;
  // three rounds through the one buffer: dgamma, dbeta, column sums
#pragma unroll
  for (int round = 0; round < (CS ? 3 : 2); ++round) {
    float* dst = round == 0 ? dgamma : round == 1 ? dbeta : colsum;
    if (round) __syncthreads();           // the previous round's readers are done
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        const int c = (i * 64 + lane) * VN + j;
        red[wave * W + c] = round == 0 ? pg[i][j] : round == 1 ? pb[i][j] : pc[CS ? i : 0][CS ? j : 0];
      }
    __syncthreads();
    if (dst)
      for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dst + c, red[c] + red[W + c] + red[2 * W + c] + red[3 * W + c]);
  }
}

// The backward kernel for bf16 rows of 256 * NV elements (768: every LayerNorm of mDT-base; 1024: mDT-large) with its addresses in SCALAR registers: a wave owns whole
// rows, so a row's byte offset is wave-uniform — each tensor is a buffer descriptor based at the wave's first row, the row offset
// goes into the instruction's SGPR operand, the vector's place in the row into its 12-bit immediate (0 / 512 / 1024), and ONE VGPR
// (lane * 8) serves every load and store of the kernel.  Same arithmetic in the same order as layernorm_bwd_kernel<bf16_t, 3, ...,
// VecH>: what changes is the register budget (no 64-bit address per tensor and row, no vectors for an absent residual gradient).
template <int NV, bool DROPPED, bool CS, bool ADD>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(int64_t rows, const bf16_t* __restrict__ dy, int64_t lddy,
                                                                const bf16_t* __restrict__ x, int64_t ldx,
                                                                const bf16_t* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const bf16_t* __restrict__ add,
                                                                int64_t ldadd, bf16_t* __restrict__ dx, int64_t lddx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int rows_per_wave, bf16_t* __restrict__ dxd, int64_t lddxd,
                                                                DropCfg drop, float* __restrict__ colsum) {
  constexpr int D = 256 * NV, VN = 4;
  typedef VecH<bf16_t> V;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;  // [4 waves][D]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  const int64_t r1 = (r0 + rows_per_wave < rows) ? r0 + rows_per_wave : rows;
  const int nrow = r0 < r1 ? (int)(r1 - r0) : 0;
  auto desc = [&](const void* p, int64_t ld) {
    const int64_t bytes = (int64_t)nrow * ld * 2;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p + (nrow ? r0 * ld * 2 : 0)), 0,
                                             (unsigned)(p == nullptr ? 0 : bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_dy = desc(dy, lddy), rs_x = desc(x, ldx), rs_add = desc(ADD ? add : nullptr, ldadd),
                               rs_dx = desc(dx, lddx), rs_dxd = desc(DROPPED ? dxd : nullptr, lddxd);
  const int s_dy = (int)(lddy * 2), s_x = (int)(ldx * 2), s_add = (int)(ldadd * 2), s_dx = (int)(lddx * 2), s_dxd = (int)(lddxd * 2);
  const int voff = lane * 8;
  auto ld8 = [&](__amdgpu_buffer_rsrc_t rs, int i, int soff) {
    V v;
    v.v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + i * 512, soff, 0));
    return v;
  };
  auto st8 = [&](__amdgpu_buffer_rsrc_t rs, int i, int soff, const V& v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v.v), rs, voff + i * 512, soff, 0);
  };
  float pg[NV][VN], pb[NV][VN], pc[CS ? NV : 1][CS ? VN : 1];
  V gv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gv[i] = *(const V*)(gamma + (i * 64 + lane) * VN);
#pragma unroll
    for (int j = 0; j < VN; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; if constexpr (CS) pc[i][j] = 0.f; }
  }
  V xv[NV], gy[NV], av[ADD ? NV : 1], xn[NV], gn[NV], an[ADD ? NV : 1];
  float mu = 0.f, rs = 0.f, mu_n = 0.f, rs_n = 0.f;
  if (nrow) {
    mu = mean[r0]; rs = rstd[r0];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] = ld8(rs_x, i, 0);
      gy[i] = ld8(rs_dy, i, 0);
      if constexpr (ADD) av[i] = ld8(rs_add, i, 0);
    }
  }
  for (int k = 0; k < nrow; ++k) {
    const int64_t row = r0 + k;
    if (k + 1 < nrow) {
      mu_n = mean[row + 1]; rs_n = rstd[row + 1];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        xn[i] = ld8(rs_x, i, (k + 1) * s_x);
        gn[i] = ld8(rs_dy, i, (k + 1) * s_dy);
        if constexpr (ADD) an[i] = ld8(rs_add, i, (k + 1) * s_add);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        const float xh = (xv[i].get(j) - mu) * rs;
        const float g = gy[i].get(j);
        const float gg = g * gv[i].get(j);
        s1 += gg;
        s2 += gg * xh;
        pg[i][j] += g * xh;
        pb[i][j] += g;
      }
    }
    const float m1 = wave_sum_dpp(s1) / (float)D, m2 = wave_sum_dpp(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * VN;
      V o, od;
      float ds0 = 0.f, ds1 = 0.f;
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        const float xh = (xv[i].get(j) - mu) * rs;
        float v = rs * (gy[i].get(j) * gv[i].get(j) - m1 - xh * m2);
        if constexpr (ADD) v += av[i].get(j);
        o.set(j, v);
        if constexpr (DROPPED) {
          if ((j & 1) == 0) drop_scale2(drop, (uint64_t)row * D + c + j, ds0, ds1);
          v = o.get(j) * ((j & 1) ? ds1 : ds0);
          od.set(j, v);
          v = od.get(j);
        } else {
          v = o.get(j);
        }
        if constexpr (CS) pc[i][j] += v;
      }
      st8(rs_dx, i, k * s_dx, o);
      if constexpr (DROPPED) st8(rs_dxd, i, k * s_dxd, od);
    }
    mu = mu_n; rs = rs_n;
#pragma unroll
    for (int i = 0; i < NV; ++i) { xv[i] = xn[i]; gy[i] = gn[i]; if constexpr (ADD) av[i] = an[i]; }
  }
  if (!dgamma && !dbeta && !colsum) return;
  constexpr int W = D;
#pragma unroll
  for (int round = 0; round < (CS ? 3 : 2); ++round) {
    float* dst = round == 0 ? dgamma : round == 1 ? dbeta : colsum;
    if (round) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < VN; ++j)
        red[wave * W + (i * 64 + lane) * VN + j] = round == 0 ? pg[i][j] : round == 1 ? pb[i][j] : pc[CS ? i : 0][CS ? j : 0];
    __syncthreads();
    if (dst)
      for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dst + c, red[c] + red[W + c] + red[2 * W + c] + red[3 * W + c]);
  }
}

template <typename T>
static int ln_fwd_dispatch(hipStream_t st, int64_t rows, int D, const void* x, int64_t ldx, const void* gamma,
                           const void* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd, void* q8 = nullptr,
                           int64_t ldq = 0, int q8_fmt = 0, const float* q8_scale = nullptr, float* q8_amax = nullptr) {
  constexpr int VN = Vec<T>::N;
  const int nv = (D + 64 * VN - 1) / (64 * VN);
  // 8 resident workgroups per CU (32 waves, two rows in flight each); short inputs get one row per wave
  const int64_t wgs = (rows + 3) / 4;
  const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (D == 768) {          // three 8-byte vectors per lane, every lane busy (VecH)
#define LN_FWD_H(Q_) hipLaunchKernelGGL((layernorm_fwd_kernel<T, 3, Q_, VecH<T>>), grid, 256, 0, st, rows, D, (const T*)x, ldx, (const T*)gamma, (const T*)beta, \
                                        eps, (T*)y, ldy, mean, rstd, (uint8_t*)q8, ldq, q8_scale, q8_amax)
      if (q8 && q8_fmt == 0) LN_FWD_H(1);
      else if (q8) LN_FWD_H(2);
      else LN_FWD_H(0);
#undef LN_FWD_H
      return check_launch("layernorm_fwd");
    }
  }
#define LN_FWD(NV_)                                                                                          \
  if (q8 && q8_fmt == 0)                                                                                     \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV_, 1>), grid, 256, 0, st, rows, D, (const T*)x, ldx, (const T*)gamma, (const T*)beta, \
                       eps, (T*)y, ldy, mean, rstd, (uint8_t*)q8, ldq, q8_scale, q8_amax);                   \
  else if (q8)                                                                                               \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV_, 2>), grid, 256, 0, st, rows, D, (const T*)x, ldx, (const T*)gamma, (const T*)beta, \
                       eps, (T*)y, ldy, mean, rstd, (uint8_t*)q8, ldq, q8_scale, q8_amax);                   \
  else                                                                                                       \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV_>), grid, 256, 0, st, rows, D, (const T*)x, ldx,        \
                       (const T*)gamma, (const T*)beta, eps, (T*)y, ldy, mean, rstd, (uint8_t*)nullptr, (int64_t)0, (const float*)nullptr, (float*)nullptr)
  switch (nv) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    case 6: LN_FWD(6); break;
    case 8: LN_FWD(8); break;
    default: MDT_UNSUPPORTED("layernorm: D=%d not supported (vectors per lane %d)", D, nv);
  }
#undef LN_FWD
  return check_launch("layernorm_fwd");
}

template <typename T>
static int ln_bwd_dispatch(hipStream_t st, int64_t rows, int D, const void* dy, int64_t lddy, const void* x, int64_t ldx,
                           const void* gamma, const float* mean, const float* rstd, const void* add, int64_t ldadd,
                           void* dx, int64_t lddx, float* dgamma, float* dbeta, void* dxd, int64_t lddxd, DropCfg drop,
                           float* colsum) {
  constexpr int VN = Vec<T>::N;
  const int nv = (D + 64 * VN - 1) / (64 * VN);
  // ~2048 workgroups x 4 waves, at least 4 rows per wave so the column atomics stay few
  const int64_t want = switches().ln_bwd_wgs > 0 ? switches().ln_bwd_wgs : 2048;
  int rpw = (int)((rows + want * 4 - 1) / (want * 4));
  if (rpw < 4) rpw = 4;
  const unsigned grid = (unsigned)((rows + 4 * rpw - 1) / (4 * rpw));
  const bool tail = dxd != nullptr || colsum != nullptr;
  if constexpr (std::is_same<T, bf16_t>::value) {
    // row offsets inside a wave's block of rows are 32-bit scalars: rows_per_wave * row stride stays far below 2^31
    const int64_t ldmax = std::max(std::max(lddy, ldx), std::max(std::max(ldadd, lddx), lddxd));
    if ((D == 768 || D == 1024 || D == 512 || D == 256) && !switches().ln_generic && (int64_t)rpw * ldmax * 2 < (1ll << 31)) {
#define LN_BWD_S(NV_, DR_, CS_, ADD_) hipLaunchKernelGGL((layernorm_bwd_rows_kernel<NV_, DR_, CS_, ADD_>), grid, 256, (size_t)(4 * 256 * NV_ * 4), st, rows, \
                       (const T*)dy, lddy, (const T*)x, ldx, (const T*)gamma, mean, rstd, (const T*)add, ldadd,  \
                       (T*)dx, lddx, dgamma, dbeta, rpw, (T*)dxd, lddxd, drop, colsum)
#define LN_BWD_F(NV_)                                          \
      switch (form) {                                          \
        case 0: LN_BWD_S(NV_, false, false, false); break;     \
        case 1: LN_BWD_S(NV_, false, false, true); break;      \
        case 2: LN_BWD_S(NV_, false, true, false); break;      \
        case 3: LN_BWD_S(NV_, false, true, true); break;       \
        case 4: LN_BWD_S(NV_, true, false, false); break;      \
        case 5: LN_BWD_S(NV_, true, false, true); break;       \
        case 6: LN_BWD_S(NV_, true, true, false); break;       \
        default: LN_BWD_S(NV_, true, true, true); break;       \
      }
      const int form = (dxd ? 4 : 0) | (colsum ? 2 : 0) | (add ? 1 : 0);
      if (D == 768) { LN_BWD_F(3) } else if (D == 1024) { LN_BWD_F(4) } else if (D == 512) { LN_BWD_F(2) } else { LN_BWD_F(1) }
#undef LN_BWD_F
#undef LN_BWD_S
      return check_launch("layernorm_bwd");
    }
    if (D == 768) {
#define LN_BWD_H(TAIL_, CS_) hipLaunchKernelGGL((layernorm_bwd_kernel<T, 3, TAIL_, CS_, VecH<T>>), grid, 256, (size_t)(4 * 768 * 4), st, rows, D, \
                       (const T*)dy, lddy, (const T*)x, ldx, (const T*)gamma, mean, rstd, (const T*)add, ldadd,  \
                       (T*)dx, lddx, dgamma, dbeta, rpw, (T*)dxd, lddxd, drop, colsum)
      if (tail && !colsum) LN_BWD_H(true, false);
      else if (tail) LN_BWD_H(true, true);
      else LN_BWD_H(false, false);
#undef LN_BWD_H
      return check_launch("layernorm_bwd");
    }
  }
#define LN_BWD(NV_)                                                                                              \
  if (tail && !colsum)                                                                                           \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV_, true, false>), grid, 256, (size_t)(4 * NV_ * 64 * VN * 4), st, rows, D, \
                       (const T*)dy, lddy, (const T*)x, ldx, (const T*)gamma, mean, rstd, (const T*)add, ldadd,  \
                       (T*)dx, lddx, dgamma, dbeta, rpw, (T*)dxd, lddxd, drop, colsum);                          \
  else if (tail)                                                                                                 \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV_, true>), grid, 256, (size_t)(4 * NV_ * 64 * VN * 4), st, rows, D, \
                       (const T*)dy, lddy, (const T*)x, ldx, (const T*)gamma, mean, rstd, (const T*)add, ldadd,  \
                       (T*)dx, lddx, dgamma, dbeta, rpw, (T*)dxd, lddxd, drop, colsum);                          \
  else                                                                                                           \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV_, false>), grid, 256, (size_t)(4 * NV_ * 64 * VN * 4), st, rows, D, \
                       (const T*)dy, lddy, (const T*)x, ldx, (const T*)gamma, mean, rstd, (const T*)add, ldadd,  \
                       (T*)dx, lddx, dgamma, dbeta, rpw, (T*)dxd, lddxd, drop, colsum)
  switch (nv) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    case 6: LN_BWD(6); break;
    case 8: LN_BWD(8); break;
    default: MDT_UNSUPPORTED("layernorm: D=%d not supported (vectors per lane %d)", D, nv);
  }
#undef LN_BWD
  return check_launch("layernorm_bwd");
}

}  // namespace mdt

using namespace mdt;

static int ln_check(int dtype, int D, int64_t ld0, int64_t ld1, const void* p0, const void* p1) {
  const int vn = dtype == MDT_BF16 ? 8 : 4;
  MDT_CHECK_ARG(dtype == MDT_F32 || dtype == MDT_BF16, "layernorm: bad dtype %d", dtype);
  MDT_CHECK_ARG(D > 0 && D % vn == 0, "layernorm: D=%d must be a multiple of %d", D, vn);
  MDT_CHECK_ARG(ld0 % vn == 0 && ld1 % vn == 0, "layernorm: row strides must be multiples of %d elements", vn);
  MDT_CHECK_ARG((((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0, "layernorm: pointers must be 16-byte aligned");
  return MDT_OK;
}

extern "C" int mdt_layernorm_fwd(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx,
                                 const void* gamma, const void* beta, float eps, void* y, int64_t ldy, float* mean,
                                 float* rstd) {
  if (rows == 0) return MDT_OK;
  MDT_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd: null pointer");
  if (int e = ln_check(dtype, D, ldx, ldy, x, y)) return e;
  MDT_CHECK_ARG((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_fwd: gamma/beta must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return dtype == MDT_F32 ? ln_fwd_dispatch<float>(st, rows, D, x, ldx, gamma, beta, eps, y, ldy, mean, rstd)
                          : ln_fwd_dispatch<bf16_t>(st, rows, D, x, ldx, gamma, beta, eps, y, ldy, mean, rstd);
}

extern "C" int mdt_layernorm_fwd_q8(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx,
                                    const void* gamma, const void* beta, float eps, void* y, int64_t ldy, float* mean,
                                    float* rstd, void* q8_out, int64_t ld_q8, int q8_format, const float* q8_scale, float* q8_amax) {
  if (!q8_out) return mdt_layernorm_fwd(stream, dtype, rows, D, x, ldx, gamma, beta, eps, y, ldy, mean, rstd);
  if (rows == 0) return MDT_OK;
  MDT_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd_q8: null pointer");
  MDT_CHECK_ARG(dtype == MDT_BF16, "layernorm_fwd_q8: the fp8 copy exists for bf16 rows only (dtype %d)", dtype);
  MDT_CHECK_ARG(q8_format == 0 || q8_format == 1, "layernorm_fwd_q8: q8_format %d (0 = e4m3, 1 = e5m2)", q8_format);
  MDT_CHECK_ARG(q8_scale && q8_amax && ld_q8 >= D && ld_q8 % 8 == 0 && ((uintptr_t)q8_out & 7) == 0,
                "layernorm_fwd_q8: the fp8 copy needs its scale, its maximum slot and 8-byte aligned rows of at least D bytes");
  if (int e = ln_check(dtype, D, ldx, ldy, x, y)) return e;
  MDT_CHECK_ARG((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_fwd_q8: gamma/beta must be 16-byte aligned");
  return ln_fwd_dispatch<bf16_t>((hipStream_t)stream, rows, D, x, ldx, gamma, beta, eps, y, ldy, mean, rstd, q8_out, ld_q8, q8_format, q8_scale, q8_amax);
}

template <typename T>
static int embed_ln_dispatch(hipStream_t st, int64_t rows, int D, const int32_t* ids, const int32_t* types, const int32_t* pos_ids,
                             const void* word, const void* pos, const void* typ, const void* gamma, const void* beta, float eps,
                             void* xs, int64_t ldxs, void* y, int64_t ldy, float* mean, float* rstd) {
  constexpr int VN = Vec<T>::N;
  const int nv = (D + 64 * VN - 1) / (64 * VN);
  const int64_t wgs = (rows + 3) / 4;
  const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);
#define EL_(NV_, V_) hipLaunchKernelGGL((bert_embed_ln_rows_kernel<T, NV_, V_>), grid, 256, 0, st, rows, D, ids, types, pos_ids, (const T*)word, \
                                        (const T*)pos, (const T*)typ, (const T*)gamma, (const T*)beta, eps, (T*)xs, ldxs, (T*)y, ldy, mean, rstd)
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (D == 768) { EL_(3, VecH<T>); return check_launch("bert_embed_ln_rows"); }
  }
  switch (nv) {
    case 1: EL_(1, Vec<T>); break;
    case 2: EL_(2, Vec<T>); break;
    case 3: EL_(3, Vec<T>); break;
    case 4: EL_(4, Vec<T>); break;
    case 6: EL_(6, Vec<T>); break;
    case 8: EL_(8, Vec<T>); break;
    default: MDT_UNSUPPORTED("bert_embed_ln_rows: D=%d not supported (vectors per lane %d)", D, nv);
  }
#undef EL_
  return check_launch("bert_embed_ln_rows");
}

extern "C" int mdt_bert_embed_ln_rows(void* stream, int dtype, int64_t rows, const int32_t* ids, const int32_t* types,
                                      const int32_t* pos_ids, const void* word, const void* pos, const void* type, int D,
                                      const void* gamma, const void* beta, float eps, void* xs, int64_t ldxs, void* y, int64_t ldy,
                                      float* mean, float* rstd) {
  if (rows == 0) return MDT_OK;
  MDT_CHECK_ARG(ids && types && pos_ids && word && pos && type && gamma && beta && y, "bert_embed_ln_rows: null pointer");
  if (int e = ln_check(dtype, D, xs ? ldxs : ldy, ldy, xs ? xs : y, y)) return e;
  MDT_CHECK_ARG((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)word | (uintptr_t)pos | (uintptr_t)type) & 15) == 0,
                "bert_embed_ln_rows: tables, gamma and beta must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return dtype == MDT_F32 ? embed_ln_dispatch<float>(st, rows, D, ids, types, pos_ids, word, pos, type, gamma, beta, eps, xs, ldxs, y, ldy, mean, rstd)
                          : embed_ln_dispatch<bf16_t>(st, rows, D, ids, types, pos_ids, word, pos, type, gamma, beta, eps, xs, ldxs, y, ldy, mean, rstd);
}

extern "C" int mdt_layernorm_bwd(void* stream, int dtype, int64_t rows, int D, const void* dy, int64_t lddy,
                                 const void* x, int64_t ldx, const void* gamma, const float* mean, const float* rstd,
                                 const void* add, int64_t ldadd, void* dx, int64_t lddx, float* dgamma, float* dbeta,
                                 void* dxd, int64_t lddxd, float drop_p, uint64_t drop_seed, float* colsum) {
  if (rows == 0) return MDT_OK;
  MDT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd: dropout p=%f out of [0,1)", drop_p);
  MDT_CHECK_ARG(!dxd || rows * D < DROP_MAX_ELEMS, "layernorm_bwd: dropout site of %lld elements (limit 2^33)", (long long)(rows * D));
  const DropCfg drop = make_drop(dxd ? drop_p : 0.f, drop_seed);
  MDT_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  if (int e = ln_check(dtype, D, lddy, ldx, dy, x)) return e;
  if (int e = ln_check(dtype, D, lddx, add ? ldadd : lddx, dx, add ? add : dx)) return e;
  hipStream_t st = (hipStream_t)stream;
  return dtype == MDT_F32
             ? ln_bwd_dispatch<float>(st, rows, D, dy, lddy, x, ldx, gamma, mean, rstd, add, ldadd, dx, lddx, dgamma, dbeta, dxd, lddxd, drop, colsum)
             : ln_bwd_dispatch<bf16_t>(st, rows, D, dy, lddy, x, ldx, gamma, mean, rstd, add, ldadd, dx, lddx, dgamma, dbeta, dxd, lddxd, drop, colsum);
}
