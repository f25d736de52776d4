// bf16 attention, second generation: probabilities never leave the register file.
//
// Scores are computed TRANSPOSED — keys on the MFMA rows (registers), queries on the lanes —
// so the 16x16 fp32 accumulator tiles of S^T are already laid out as the B operand of the next
// product (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand"):
//   forward   S^T = K Q^T          ->  P^T (softmax down the registers + 2 shuffles)
//             O^T = V^T P^T            (V^T fetched from the row-major LDS image with ds_read_b64_tr_b16)
//   backward  pass A (queries on lanes):  dP^T = V dO^T, dS^T = P^T (dP^T - delta),  dQ^T = K^T dS^T
//             pass B (keys on lanes):     S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS
// A pair of 16-key (or 16-query) accumulator tiles forms one K = 32 operand: element j of lane
// (g, c) is row 4g + j of tile t0 (j < 4) or of tile t0+1 (j >= 4); the transposed-read operand
// is addressed with the same row permutation, so no data is ever shuffled between lanes.
// Compared with attention.hip this removes the P / dS round trip through LDS (28-52 two-byte
// ds_writes + wave fences per tile), halves the LDS footprint (4 workgroups per CU at S = 104)
// and reduces the softmax reductions from four shuffles to two.
// Same C ABI, same masks / structural bias / dropout semantics; attention.hip keeps the fp32
// parity path and remains selectable for bf16 with MDT_ATTN_V1=1.
#include "attention_common.hpp"

namespace mdt {

constexpr int V2_LD = 72;  // LDS image row stride in elements (144 B)

template <int HD>
__device__ __forceinline__ void v2_stage(bf16_t* img, const bf16_t* g, int64_t g_ld, int S, int rows_pad, int tid, int nthr = 256) {
  constexpr int CH = HD / 8;
  for (int e = tid; e < rows_pad * CH; e += nthr) {
    const int r = e / CH, c = e - r * CH;
    bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (r < S) v = *(const bf16x8*)(g + r * g_ld + c * 8);
    *(bf16x8*)(img + r * V2_LD + c * 8) = v;
  }
}

// The same staging in two halves: every 16-byte chunk a thread owns is REQUESTED before any is waited for.  The
// loop above waits for each of its loads before the LDS write of the same iteration (the compiler keeps a runtime
// trip count rolled), so a workgroup that stages two images in 2-4 iterations each pays 4-8 global latencies one
// after the other before its first MFMA; with the chunks held in registers (NCH x 4 per image) it pays one.
// The requests are unconditional — rows past S read row S - 1 again (S >= 1: the callers return on empty sequences)
// and become zeros when they are put: a load under a lane mask leaves the compiler a merge with the zero value
// that it resolves with a register copy behind s_waitcnt vmcnt(0), which is the serialisation this is here to remove.
template <int HD, int NCH>
__device__ __forceinline__ void v2_stage_req(bf16x8 (&v)[NCH], const bf16_t* g, int64_t g_ld, int S, int tid, int nthr) {
  constexpr int CH = HD / 8;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int e = tid + j * nthr, r = e / CH, c = e - r * CH;
    v[j] = *(const bf16x8*)(g + (r < S ? r : S - 1) * g_ld + c * 8);
  }
}
// `spare`: 16 bytes of LDS that chunks past the image land in — with the write under a lane mask instead, the compiler
// sinks the chunk's load into the masked block and waits for it there (one more latency)
template <int HD, int NCH>
__device__ __forceinline__ void v2_stage_put(bf16_t* img, const bf16x8 (&v)[NCH], int S, int rows_pad, int tid, int nthr, bf16_t* spare) {
  constexpr int CH = HD / 8;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int e = tid + j * nthr, r = e / CH, c = e - r * CH;
    bf16_t* dst = e < rows_pad * CH ? img + r * V2_LD + c * 8 : spare;
    *(bf16x8*)dst = r < S ? v[j] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}
// key-only bias, its mask bytes requested without a lane mask (same reason); key_only_bias_of turns them into 0 / -inf
struct KeyBytes { uint8_t km, kp; };
__device__ __forceinline__ KeyBytes key_only_bias_req(const BiasCtx& b, int key, const void* readable) {
  KeyBytes k{1, 0};
  if (b.key_mask || b.key_pad) {      // uniform; without masks (ragged rows, images) nothing is read
    const int64_t at = (int64_t)b.seq * b.S + (key < b.S ? key : b.S - 1);
    // one of the two absent: it reads a byte of `readable` (any mapped address) instead of branching around its load
    const uint8_t* pm = b.key_mask ? b.key_mask + at : (const uint8_t*)readable;
    const uint8_t* pp = b.key_pad ? b.key_pad + at : (const uint8_t*)readable;
    const uint8_t m = *pm, q = *pp;
    k = KeyBytes{(uint8_t)(b.key_mask ? m : 1), (uint8_t)(b.key_pad ? q : 0)};
  }
  return k;
}
__device__ __forceinline__ float key_only_bias_of(const BiasCtx& b, int key, KeyBytes k) {
  return (key < b.S && k.km && !k.kp) ? 0.f : -INFINITY;
}

// A / B fragment with k along the contiguous axis: element(rc, k) = p[rc*ld + k]
__device__ __forceinline__ bf16x8 v2_frag_lds(const bf16_t* img, int rc0, int k0, int lane) {
  const bf16_t* a = img + (rc0 + (lane & 15)) * V2_LD + k0 + 8 * (lane >> 4);
  return *(const __attribute__((address_space(3))) bf16x8*)LDS_PTR(a);
}
__device__ __forceinline__ bf16x8 v2_frag_glb(const bf16_t* p, int64_t ld, int rows, int rc0, int k0, int lane) {
  int rc = rc0 + (lane & 15);
  if (rc > rows - 1) rc = rows - 1;
  return *(const bf16x8*)(p + rc * ld + k0 + 8 * (lane >> 4));
}
// A operand X^T[d][k] for a K = 32 step whose k index runs over rows {t0*16 + 4g + j} (j < 4)
// and {(t0+1)*16 + 4g + j - 4} of the row-major image X[row][d]
__device__ __forceinline__ bf16x8 v2_frag_tr(const bf16_t* img, int t0, int d0, int lane) {
  const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const bf16_t* a = img + (t0 * 16 + 4 * g + q4) * V2_LD + d0 + pp * 4;
  const bf16x4 lo = lds_read_tr16(a);
  const bf16x4 hi = lds_read_tr16(a + 16 * V2_LD);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// B operand from two accumulator tiles (rows = k index)
__device__ __forceinline__ bf16x8 v2_pack(const f32x4& lo, const f32x4& hi) {
  return bf16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
}
__device__ __forceinline__ float col_max(float v) {  // over the 4 lanes that share a column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Output rows leave as 16-byte vectors, 64 contiguous bytes of a row per wave instruction.  An accumulator set x[d'][r] of a
// 16-row x 64-column result holds, in lane (g, c), columns 16 d' + 4 g + r of row c: four 8-byte pieces 32 bytes apart, and a
// wave store of one piece writes 32-byte fragments of 16 different rows (sixteen quarter-filled 128-byte lines per
// instruction; the stores of the one-pass backward ran at 2.8-3.5 TB/s for it).  One v_permlane16_swap per register between
// the column tiles 2 f and 2 f + 1 (odd 16-lane rows of the first trade places with even rows of the second — the exchange
// of gemm_epilogue.hpp) leaves lane (g, c) with columns 32 f + 16 (g & 1) + 8 (g >> 1) ... + 7 of row c: vector a (f = 0) and
// vector b (f = 1), at element offset rows4_off(g) and 32 further on — the four lanes of a row write 64 contiguous bytes in
// either instruction.  All 64 lanes must be active.  Inline asm with s_nop 1 for the reason given in gemm_epilogue.hpp
// (the builtin is folded by hipcc; VALU write -> swap hazard).
struct Row32 { bf16x8 a, b; };
__device__ __forceinline__ int rows4_off(int g) { return 16 * (g & 1) + 8 * (g >> 1); }
__device__ __forceinline__ Row32 rows4_exchange(const f32x4 (&x)[4], float scale) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const bf16x4 v = bf16x4{(bf16_t)(x[d][0] * scale), (bf16_t)(x[d][1] * scale), (bf16_t)(x[d][2] * scale), (bf16_t)(x[d][3] * scale)};
    const uint2 u = __builtin_bit_cast(uint2, v);
    lo[d] = u.x;
    hi[d] = u.y;
  }
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo[0]), "+v"(lo[1]));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(hi[0]), "+v"(hi[1]));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo[2]), "+v"(lo[3]));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(hi[2]), "+v"(hi[3]));
  Row32 r;
  r.a = __builtin_bit_cast(bf16x8, uint4{lo[0], hi[0], lo[1], hi[1]});
  r.b = __builtin_bit_cast(bf16x8, uint4{lo[2], hi[2], lo[3], hi[3]});
  return r;
}

// ---------------------------------------------------------------------------- forward
// Softmax in the exp2 domain (scores and key bias pre-multiplied by log2 e: v_exp_f32 is exp2), masked keys
// carry -inf so no element needs a compare; the probabilities stay UNNORMALISED (e <= 1) through P.V and the
// 16 output values of a lane are scaled by 1 / sum (and 1 / (1 - p)) at the end; dropout decides two
// neighbouring keys per mixer word.  Kernels without structural bias take no per-pair bias at all — a plain
// dense bias is served by attention.hip (see the dispatch there).
template <int HD, int NT, bool STRUCT, bool DROP, int NW>     // NW waves: 4, or 8 for long sequences (two workgroups per CU by LDS)
__global__ __launch_bounds__(NW * 64) void attn_fwd_v2_kernel(AttnParams P) {
  constexpr int ND = HD / 16, NP = (NT + 1) / 2, S_PAD = NP * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, seq = a.seq_ids ? a.seq_ids[blockIdx.y] : (int)blockIdx.y;
  const int SL = a.S, D = a.H * HD;                      // SL: lse / dropout-counter geometry
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;   // this sequence's length
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  if (S > NT * 16 || S <= 0) return;      // longer than this launch's bound (s_cap): never index past the images; empty: nothing to write
  const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv;
  bf16_t* imgK = (bf16_t*)smem;
  bf16_t* imgV = imgK + S_PAD * V2_LD;
  float* s_kb = (float*)(imgV + S_PAD * V2_LD);   // key-only bias (0 / -inf), [S_PAD]
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  // Ragged sequences: key tiles past this sequence's length are staged as zeros, their scores come out of the
  // (unguarded: a guard there costs the compiler 80 registers) MFMA loop as exact zeros and every later
  // per-element stage skips them, so they cost LDS reads and idle MFMA slots but no VALU work.
  const int ntk = (S + 15) >> 4;
  // The Q fragments of every query tile this wave will visit are requested BEFORE K / V are staged, so their global
  // latency overlaps the staging (measured: the waves of this kernel sat in s_waitcnt / s_barrier 68 % of their
  // cycles, mostly on the per-tile Q loads).  A wave owns tiles wave, wave + NW, ... : at most MAXT of them.
  constexpr int MAXT = (NT + NW - 1) / NW;
  int n_qt = (S + 15) >> 4;
  if (a.q_limit > 0 && ((a.q_limit + 15) >> 4) < n_qt) n_qt = (a.q_limit + 15) >> 4;   // only these query tiles are needed
  bf16x8 fq_all[MAXT][HD / 32];
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int qt_k = wave + NW * k;
    if (qt_k < n_qt) {
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) fq_all[k][ks] = v2_frag_glb(qkv, tld, S, qt_k * 16, ks * 32, lane);
    }
  }
  // ... and so are all K / V chunks and mask bytes of this thread (v2_stage_req): one global latency before the barrier
  {
    constexpr int NCH = (S_PAD * (HD / 8) + NW * 64 - 1) / (NW * 64), NKB = (S_PAD + NW * 64 - 1) / (NW * 64);
    bf16x8 ck[NCH], cv[NCH];
    KeyBytes kbv[NKB];
    v2_stage_req<HD, NCH>(ck, qkv + D, tld, S, tid, NW * 64);
    v2_stage_req<HD, NCH>(cv, qkv + 2 * D, tld, S, tid, NW * 64);
#pragma unroll
    for (int j = 0; j < NKB; ++j) kbv[j] = key_only_bias_req(bc, tid + j * NW * 64, qkv);
    v2_stage_put<HD, NCH>(imgK, ck, S, S_PAD, tid, NW * 64, (bf16_t*)(s_kb + S_PAD));
    v2_stage_put<HD, NCH>(imgV, cv, S, S_PAD, tid, NW * 64, (bf16_t*)(s_kb + S_PAD));
#pragma unroll
    for (int j = 0; j < NKB; ++j)
      if (tid + j * NW * 64 < S_PAD) s_kb[tid + j * NW * 64] = key_only_bias_of(bc, tid + j * NW * 64, kbv[j]);
  }
  __syncthreads();
  const int drop_bh = seq * a.H + h;   // counters: attention_common.hpp
  const float scale2 = a.scale * LOG2E;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int qt = wave + NW * k;
    if (qt >= n_qt) break;
    const int q0 = qt * 16;
    const int q = q0 + c;
    const int qc = q < S ? q : S - 1;
    bf16x8 fq[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) fq[ks] = fq_all[k][ks];
    f32x4 sc[2 * NP];
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t) sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        sc[t] = mfma_bf16(v2_frag_lds(imgK, t * 16, ks * 32, lane), fq[ks], sc[t]);
        if (ks == HD / 32 - 1 && (t & 1)) __builtin_amdgcn_sched_barrier(0);   // bound operand prefetch depth (registers)
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= ntk) continue;
      const f32x4 kb = *(const f32x4*)(s_kb + t * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = __builtin_fmaf(sc[t][r], scale2, kb[r]);     // kb is 0 or -inf: no scaling needed
        if constexpr (STRUCT) {
          const int key = t * 16 + 4 * g + r;
          if (key < S) v = __builtin_fmaf(pair_bias<bf16_t, true>(bc, qc, key), LOG2E, v);
        }
        sc[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = col_max(mx);
    const float mxc = (mx == -INFINITY) ? 0.f : mx;   // fully masked row: every exp2 below is exp2(-inf) = 0
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= ntk) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[t][r] - mxc);
        sc[t][r] = e;
        sum += e;
      }
    }
    sum = col_sum(sum);
    if (g == 0 && q < S) a.lse[((int64_t)seq * a.H + h) * SL + q] = (sum > 0.f) ? (mxc + __builtin_amdgcn_logf(sum)) * LN2 : -INFINITY;
    float inv = (sum > 0.f) ? __builtin_amdgcn_rcpf(sum) : 0.f;
    if constexpr (DROP) {
      inv *= P.drop.inv_keep;
      const uint32_t rp = attn_row_pairs(drop_bh, SL, qc) + 2 * g;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t >= ntk) continue;
        const uint32_t w0 = drop_mix((rp + 8 * t) ^ P.drop.key), w1 = drop_mix((rp + 8 * t + 1) ^ P.drop.key);
        sc[t][0] = drop_keep_lo(P.drop, w0) ? sc[t][0] : 0.f;
        sc[t][1] = drop_keep_hi(P.drop, w0) ? sc[t][1] : 0.f;
        sc[t][2] = drop_keep_lo(P.drop, w1) ? sc[t][2] : 0.f;
        sc[t][3] = drop_keep_hi(P.drop, w1) ? sc[t][3] : 0.f;
      }
    }
    f32x4 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      if (2 * pi >= ntk) continue;
      const bf16x8 fp = v2_pack(sc[2 * pi], sc[2 * pi + 1]);
#pragma unroll
      for (int d = 0; d < ND; ++d) o[d] = mfma_bf16(v2_frag_tr(imgV, 2 * pi, d * 16, lane), fp, o[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(ND == 4, "rows4_exchange: 64-column rows");
    const Row32 ov = rows4_exchange(o, inv);      // inv is per query = per lane column c: the same in the four lanes that trade
    if (q < S) {
      bf16_t* orow = (bf16_t*)a.out + (row0 + (int64_t)q * a.pos_stride) * a.ld_out + h * HD + rows4_off(g);
      *(bf16x8*)orow = ov.a;
      *(bf16x8*)(orow + 32) = ov.b;
    }
  }
}

// ---------------------------------------------------------------------------- backward
// Whole-row backward: two live accumulator sets of NT tiles + packed operands, so it only pays up to
// NT = 7 (S <= 112; beyond that it drops to 1 wave per SIMD and spills) — longer rows take the chunked
// v3 kernel below (profiles/round1_attention_v2.txt).
template <int HD, int NT, bool STRUCT, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_v2_kernel(AttnParams P) {
  constexpr int ND = HD / 16, NP = (NT + 1) / 2, S_PAD = NP * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, seq = a.seq_ids ? a.seq_ids[blockIdx.y] : (int)blockIdx.y;
  const int SL = a.S, D = a.H * HD;                      // SL: lse / dropout-counter geometry
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;   // this sequence's length
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;
  const bf16_t* dout = (const bf16_t*)P.dout + row0 * P.ld_dout + h * HD;
  bf16_t* dqkv = (bf16_t*)P.dqkv + row0 * P.ld_dqkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv, dld = a.pos_stride * P.ld_dout, gld = a.pos_stride * P.ld_dqkv;
  bf16_t* img0 = (bf16_t*)smem;
  bf16_t* img1 = img0 + S_PAD * V2_LD;
  float* s_kb = (float*)(img1 + S_PAD * V2_LD);
  float* s_lse = s_kb + S_PAD;
  float* s_delta = s_lse + S_PAD;
  float* s_hist = s_delta + S_PAD;
  const int nhist = STRUCT ? ((a.num_spatial + 1 + 3) & ~3) : 0;
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  v2_stage<HD>(img0, qkv + D, tld, S, S_PAD, tid);      // K
  v2_stage<HD>(img1, qkv + 2 * D, tld, S, S_PAD, tid);  // V
  for (int i = tid; i < S_PAD; i += 256) {
    s_kb[i] = key_only_bias<bf16_t>(bc, i);
    s_lse[i] = (i < S) ? a.lse[((int64_t)seq * a.H + h) * SL + i] : -INFINITY;
    s_delta[i] = 0.f;
  }
  for (int i = tid; i < nhist; i += 256) s_hist[i] = 0.f;
  __syncthreads();
  const int drop_bh = seq * a.H + h;   // counters: attention_common.hpp
  const int g = lane >> 4, c = lane & 15;
  const int n_t = (S + 15) >> 4;

  // ------------------------------------------------------------------ pass A: queries on lanes
  for (int qt = wave; qt < n_t; qt += 4) {
    const int q0 = qt * 16;
    const int q = q0 + c;
    const bool qok = q < S;
    const int qc = qok ? q : S - 1;
    bf16x8 fq[HD / 32], fo[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      fq[ks] = v2_frag_glb(qkv, tld, S, q0, ks * 32, lane);
      fo[ks] = v2_frag_glb(dout, dld, S, q0, ks * 32, lane);
    }
    f32x4 sc[2 * NP], dp[2 * NP];
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        sc[t] = mfma_bf16(v2_frag_lds(img0, t * 16, ks * 32, lane), fq[ks], sc[t]);
        dp[t] = mfma_bf16(v2_frag_lds(img1, t * 16, ks * 32, lane), fo[ks], dp[t]);
        if (ks == HD / 32 - 1) __builtin_amdgcn_sched_barrier(0);
      }
    const float l = s_lse[qc];
    float del = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        float v = sc[t][r] * a.scale + s_kb[key];
        if (STRUCT && key < S) v += pair_bias<bf16_t, STRUCT>(bc, qc, key);
        const float p = (v == -INFINITY || l == -INFINITY || !qok) ? 0.f : __expf(v - l);
        sc[t][r] = p;
        if constexpr (DROP) dp[t][r] *= attn_drop_scale(P.drop, drop_bh, SL, q, key);
        del += p * dp[t][r];
      }
    del = col_sum(del);
    if (g == 0 && qok) s_delta[q] = del;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = sc[t][r] * (dp[t][r] - del);
        sc[t][r] = ds;
        const int key = t * 16 + 4 * g + r;
        if (qok && key < S) {
          if (P.d_dense_bias) P.d_dense_bias[(((int64_t)seq * a.H + h) * S + q) * S + key] = ds;
          if constexpr (STRUCT) {
            if (P.d_sp_table && ds != 0.f) {
              if (q >= 1 && key >= 1) {
                const int idx = a.spatial_pos[((int64_t)seq * (S - 1) + (q - 1)) * (S - 1) + (key - 1)];
                if (idx != 0) atomicAdd(s_hist + idx, ds);
              } else {
                atomicAdd(s_hist + a.num_spatial, ds);
              }
            }
          }
        }
      }
    f32x4 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      const bf16x8 fs = v2_pack(sc[2 * pi], sc[2 * pi + 1]);
#pragma unroll
      for (int d = 0; d < ND; ++d) dq[d] = mfma_bf16(v2_frag_tr(img0, 2 * pi, d * 16, lane), fs, dq[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qok) {
      bf16_t* orow = dqkv + (int64_t)q * gld + 4 * g;
#pragma unroll
      for (int d = 0; d < ND; ++d)
        *(bf16x4*)(orow + d * 16) = bf16x4{(bf16_t)(dq[d][0] * a.scale), (bf16_t)(dq[d][1] * a.scale),
                                           (bf16_t)(dq[d][2] * a.scale), (bf16_t)(dq[d][3] * a.scale)};
    }
  }
  __syncthreads();   // delta complete; K / V images are free
  v2_stage<HD>(img0, qkv, tld, S, S_PAD, tid);    // Q
  v2_stage<HD>(img1, dout, dld, S, S_PAD, tid);   // dO
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
  __syncthreads();
  // ------------------------------------------------------------------ pass B: keys on lanes
  for (int kt = wave; kt < n_t; kt += 4) {
    const int key0 = kt * 16;
    const int key = key0 + c;
    const bool kok = key < S;
    const float kb = s_kb[key];
    bf16x8 fk[HD / 32], fv[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      fk[ks] = v2_frag_glb(qkv + D, tld, S, key0, ks * 32, lane);
      fv[ks] = v2_frag_glb(qkv + 2 * D, tld, S, key0, ks * 32, lane);
    }
    f32x4 sc[2 * NP], dp[2 * NP];
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        sc[t] = mfma_bf16(v2_frag_lds(img0, t * 16, ks * 32, lane), fk[ks], sc[t]);   // S[q][key]
        dp[t] = mfma_bf16(v2_frag_lds(img1, t * 16, ks * 32, lane), fv[ks], dp[t]);   // dP[q][key] = dO V^T
        if (ks == HD / 32 - 1) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = t * 16 + 4 * g + r;
        const bool qok = q < S;
        const int qc = qok ? q : S - 1;
        float v = sc[t][r] * a.scale + kb;
        if (STRUCT && kok) v += pair_bias<bf16_t, STRUCT>(bc, qc, key);
        const float l = s_lse[q];
        const float p = (v == -INFINITY || l == -INFINITY || !qok || !kok) ? 0.f : __expf(v - l);
        float ds = dp[t][r];
        float pd = p;
        if constexpr (DROP) {
          const float m = attn_drop_scale(P.drop, drop_bh, SL, qc, key);
          ds *= m;
          pd *= m;
        }
        sc[t][r] = pd;                          // (dropped) P, operand of dV
        dp[t][r] = p * (ds - s_delta[q]);       // dS
      }
    f32x4 dv[ND], dk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      const bf16x8 fp = v2_pack(sc[2 * pi], sc[2 * pi + 1]);
      const bf16x8 fs = v2_pack(dp[2 * pi], dp[2 * pi + 1]);
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        dv[d] = mfma_bf16(v2_frag_tr(img1, 2 * pi, d * 16, lane), fp, dv[d]);   // dO^T P
        dk[d] = mfma_bf16(v2_frag_tr(img0, 2 * pi, d * 16, lane), fs, dk[d]);   // Q^T dS
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kok) {
      bf16_t* krow = dqkv + (int64_t)key * gld + D + 4 * g;
      bf16_t* vrow = dqkv + (int64_t)key * gld + 2 * D + 4 * g;
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        *(bf16x4*)(krow + d * 16) = bf16x4{(bf16_t)(dk[d][0] * a.scale), (bf16_t)(dk[d][1] * a.scale),
                                           (bf16_t)(dk[d][2] * a.scale), (bf16_t)(dk[d][3] * a.scale)};
        *(bf16x4*)(vrow + d * 16) = bf16x4{(bf16_t)dv[d][0], (bf16_t)dv[d][1], (bf16_t)dv[d][2], (bf16_t)dv[d][3]};
      }
    }
  }
}


// ---------------------------------------------------------------------------- backward, chunked (v3)
// With the forward log-sum-exp saved and delta = rowsum(dO * O) (valid with dropout too: O = D V),
// P and dS are local to a key / query chunk, so the backward never needs a whole row of scores at
// once.  Both passes walk 64-wide chunks (two K = 32 operand pairs) with the v2 register-resident
// operands: 8 score / gradient accumulator tiles live instead of 2 x NT, which is what brings the
// kernel from 1 to 3-4 waves per SIMD — these sequences are short and the kernel is latency-bound.
//   pass A (queries on lanes): for each key chunk: S^T, dP^T -> dS^T -> dQ^T += K^T dS^T
//   pass B (keys on lanes):    for each query chunk: S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS
template <int HD, bool STRUCT, bool DROP, bool PF>   // PF: fragments of the next tile requested one tile ahead (16 more registers)
__device__ __forceinline__ void attn_bwd_v3_body(const AttnParams& P, int s_pad) {
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nw = nthr >> 6;      // 4 waves, or 8 for long sequences in the 128-register build
  const int h = blockIdx.x, seq = a.seq_ids ? a.seq_ids[blockIdx.y] : (int)blockIdx.y;
  const int SL = a.S, D = a.H * HD;                      // SL: lse / dropout-counter geometry
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;   // this sequence's length
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;
  const bf16_t* dout = (const bf16_t*)P.dout + row0 * P.ld_dout + h * HD;
  const bf16_t* outp = (const bf16_t*)a.out + row0 * a.ld_out + h * HD;
  bf16_t* dqkv = (bf16_t*)P.dqkv + row0 * P.ld_dqkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv, dld = a.pos_stride * P.ld_dout, old_ = a.pos_stride * a.ld_out,
                gld = a.pos_stride * P.ld_dqkv;
  bf16_t* img0 = (bf16_t*)smem;
  bf16_t* img1 = img0 + s_pad * V2_LD;
  float* s_kb = (float*)(img1 + s_pad * V2_LD);
  float* s_lse = s_kb + s_pad;
  float* s_delta = s_lse + s_pad;
  float* s_hist = s_delta + s_pad;
  const int nhist = STRUCT ? ((a.num_spatial + 1 + 3) & ~3) : 0;
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  const int s_live = (S + 63) & ~63;                    // rows this (possibly ragged) sequence really uses, in 64-key chunks
  if (s_live > s_pad) return;                           // longer than this launch's bound (s_cap)
  // Latency hiding (the waves of this kernel sat in s_waitcnt / s_barrier for half to two thirds of their cycles):
  // the Q / dO fragments of a wave's first query tile are requested before K / V are staged, and every later
  // tile's fragments while the previous tile is being computed; pass B treats its K / V fragments the same way.
  bf16x8 fq_n[HD / 32], fo_n[HD / 32];
  if constexpr (PF) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      fq_n[ks] = v2_frag_glb(qkv, tld, S, wave * 16, ks * 32, lane);
      fo_n[ks] = v2_frag_glb(dout, dld, S, wave * 16, ks * 32, lane);
    }
  }
  v2_stage<HD>(img0, qkv + D, tld, S, s_live, tid, nthr);      // K
  v2_stage<HD>(img1, qkv + 2 * D, tld, S, s_live, tid, nthr);  // V
  // s_lse holds lse * log2(e) (+inf for rows without a finite lse, so every p of such a row is exp2(-inf) = 0);
  // s_delta holds delta * (1 - p_drop): the 1 / (1 - p_drop) factor of the dropout mask is folded out of
  // dS and dV and applied once to the outputs.
  const float ik = DROP ? P.drop.inv_keep : 1.0f, rik = 1.0f / ik;
  for (int i = tid; i < s_live; i += nthr) {
    s_kb[i] = key_only_bias<bf16_t>(bc, i);
    float l = -INFINITY, de = 0.f;
    // rows beyond q_limit were not computed by the forward pass (out / lse unspecified) and carry no gradient:
    // treat them like rows past the end (lse -> +inf below, so every p of theirs is 0)
    if (i < S && (a.q_limit <= 0 || i < ((a.q_limit + 15) & ~15))) {
      l = a.lse[((int64_t)seq * a.H + h) * SL + i];
#pragma unroll
      for (int c8 = 0; c8 < HD / 8; ++c8) {
        const bf16x8 o = *(const bf16x8*)(outp + i * old_ + c8 * 8);
        const bf16x8 g_ = *(const bf16x8*)(dout + i * dld + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) de += (float)o[e] * (float)g_[e];
      }
    }
    s_lse[i] = (l == -INFINITY) ? INFINITY : l * LOG2E;
    s_delta[i] = de * rik;
  }
  for (int i = tid; i < nhist; i += nthr) s_hist[i] = 0.f;
  __syncthreads();
  const int drop_bh = seq * a.H + h;   // counters: attention_common.hpp
  const uint32_t s2h = (uint32_t)((SL + 1) >> 1);
  const float scale2 = a.scale * LOG2E;
  const int g = lane >> 4, c = lane & 15;
  const int n_t = (S + 15) >> 4;
  const int n_chunk = s_live >> 6;          // 64-wide chunks
  // q_limit: queries beyond it carry no gradient (dout = 0, dQ = 0): pass A visits only the needed query tiles, pass B
  // sums over the query chunks that contain them
  const int n_tq = (a.q_limit > 0 && ((a.q_limit + 15) >> 4) < n_t) ? (a.q_limit + 15) >> 4 : n_t;
  const int n_chunk_q = (a.q_limit > 0 && ((a.q_limit + 63) >> 6) < n_chunk) ? (a.q_limit + 63) >> 6 : n_chunk;

  // ------------------------------------------------------------------ pass A (queries on lanes)
  for (int qt = wave; qt < n_tq; qt += nw) {
    const int q0 = qt * 16;
    const int q = q0 + c;
    const bool qok = q < S;
    const int qc = qok ? q : S - 1;
    bf16x8 fq[HD / 32], fo[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      if constexpr (PF) { fq[ks] = fq_n[ks]; fo[ks] = fo_n[ks]; }
      else { fq[ks] = v2_frag_glb(qkv, tld, S, q0, ks * 32, lane); fo[ks] = v2_frag_glb(dout, dld, S, q0, ks * 32, lane); }
    }
    if (PF && qt + nw < n_tq) {
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        fq_n[ks] = v2_frag_glb(qkv, tld, S, q0 + 16 * nw, ks * 32, lane);
        fo_n[ks] = v2_frag_glb(dout, dld, S, q0 + 16 * nw, ks * 32, lane);
      }
    }
    const float l2 = s_lse[qc], del = s_delta[qc];
    const uint32_t rp = attn_row_pairs(drop_bh, SL, qc) + 2 * g;
    f32x4 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < n_chunk; ++ch) {
      const int t0 = ch * 4;
      f32x4 sc[4], dp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          sc[t] = mfma_bf16(v2_frag_lds(img0, (t0 + t) * 16, ks * 32, lane), fq[ks], sc[t]);
          dp[t] = mfma_bf16(v2_frag_lds(img1, (t0 + t) * 16, ks * 32, lane), fo[ks], dp[t]);
          if (ks == HD / 32 - 1 && (t & 1)) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 kb = *(const f32x4*)(s_kb + (t0 + t) * 16 + 4 * g);
        bool keep[4] = {true, true, true, true};
        if constexpr (DROP) {
          const uint32_t w0 = drop_mix((rp + 8 * (t0 + t)) ^ P.drop.key), w1 = drop_mix((rp + 8 * (t0 + t) + 1) ^ P.drop.key);
          keep[0] = drop_keep_lo(P.drop, w0); keep[1] = drop_keep_hi(P.drop, w0);
          keep[2] = drop_keep_lo(P.drop, w1); keep[3] = drop_keep_hi(P.drop, w1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(sc[t][r], scale2, kb[r]);
          const int key = (t0 + t) * 16 + 4 * g + r;
          if constexpr (STRUCT) {
            if (key < S) v = __builtin_fmaf(pair_bias<bf16_t, true>(bc, qc, key), LOG2E, v);
          }
          const float p = __builtin_amdgcn_exp2f(v - l2);
          const float dpe = keep[r] ? dp[t][r] : 0.f;
          const float ds = p * (dpe - del);           // true dS = ds / (1 - p_drop)
          sc[t][r] = ds;
          if constexpr (STRUCT) {
            if (qok && key < S) {
              const float dst = ds * ik;
              if (P.d_dense_bias) P.d_dense_bias[(((int64_t)seq * a.H + h) * S + q) * S + key] = dst;
              if (P.d_sp_table && dst != 0.f) {
                if (q >= 1 && key >= 1) {
                  const int idx = a.spatial_pos[((int64_t)seq * (S - 1) + (q - 1)) * (S - 1) + (key - 1)];
                  if (idx != 0) atomicAdd(s_hist + idx, dst);
                } else {
                  atomicAdd(s_hist + a.num_spatial, dst);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const bf16x8 fs = v2_pack(sc[2 * pi], sc[2 * pi + 1]);
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = mfma_bf16(v2_frag_tr(img0, t0 + 2 * pi, d * 16, lane), fs, dq[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (qok) {
      const float os = a.scale * ik;
      bf16_t* orow = dqkv + (int64_t)q * gld + 4 * g;
#pragma unroll
      for (int d = 0; d < ND; ++d)
        *(bf16x4*)(orow + d * 16) = bf16x4{(bf16_t)(dq[d][0] * os), (bf16_t)(dq[d][1] * os), (bf16_t)(dq[d][2] * os), (bf16_t)(dq[d][3] * os)};
    }
  }
  bf16x8 fk_n[HD / 32], fv_n[HD / 32];
  if constexpr (PF) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      fk_n[ks] = v2_frag_glb(qkv + D, tld, S, wave * 16, ks * 32, lane);
      fv_n[ks] = v2_frag_glb(qkv + 2 * D, tld, S, wave * 16, ks * 32, lane);
    }
  }
  __syncthreads();   // K / V images are free
  v2_stage<HD>(img0, qkv, tld, S, s_live, tid, nthr);    // Q
  v2_stage<HD>(img1, dout, dld, S, s_live, tid, nthr);   // dO
  if constexpr (STRUCT) {
    if (P.d_sp_table) {
      for (int i = tid; i <= a.num_spatial; i += nthr) {
        const float v = s_hist[i];
        if (v != 0.f) {
          if (i < a.num_spatial) atomicAdd(P.d_sp_table + (int64_t)i * a.H + h, v);
          else if (P.d_virt) atomicAdd(P.d_virt + h, v);
        }
      }
    }
  }
  __syncthreads();
  // ------------------------------------------------------------------ pass B (keys on lanes)
  // A lane holds four consecutive QUERIES of one key, so its four dropout decisions sit in four different mixer
  // words; the neighbouring lane (key ^ 1) needs the same four words (other half), so each lane of the pair
  // computes two of them and they trade through one DPP quad swap each.
  const int odd = c & 1;
  const uint32_t base_rp = (uint32_t)(drop_bh * SL) * s2h;
  for (int kt = wave; kt < n_t; kt += nw) {
    const int key0 = kt * 16;
    const int key = key0 + c;
    const bool kok = key < S;
    const float kb = s_kb[key];
    bf16x8 fk[HD / 32], fv[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      if constexpr (PF) { fk[ks] = fk_n[ks]; fv[ks] = fv_n[ks]; }
      else { fk[ks] = v2_frag_glb(qkv + D, tld, S, key0, ks * 32, lane); fv[ks] = v2_frag_glb(qkv + 2 * D, tld, S, key0, ks * 32, lane); }
    }
    if (PF && kt + nw < n_t) {
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        fk_n[ks] = v2_frag_glb(qkv + D, tld, S, key0 + 16 * nw, ks * 32, lane);
        fv_n[ks] = v2_frag_glb(qkv + 2 * D, tld, S, key0 + 16 * nw, ks * 32, lane);
      }
    }
    const uint32_t kh = base_rp + (uint32_t)(key >> 1);
    f32x4 dv[ND], dk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int ch = 0; ch < n_chunk_q; ++ch) {
      const int t0 = ch * 4;
      f32x4 sc[4], dp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          sc[t] = mfma_bf16(v2_frag_lds(img0, (t0 + t) * 16, ks * 32, lane), fk[ks], sc[t]);   // S[q][key]
          dp[t] = mfma_bf16(v2_frag_lds(img1, (t0 + t) * 16, ks * 32, lane), fv[ks], dp[t]);   // dP[q][key]
          if (ks == HD / 32 - 1 && (t & 1)) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int qb = (t0 + t) * 16 + 4 * g;
        const f32x4 l2v = *(const f32x4*)(s_lse + qb);
        const f32x4 dlv = *(const f32x4*)(s_delta + qb);
        bool keep[4] = {true, true, true, true};
        if constexpr (DROP) {
          const uint32_t ra = kh + (uint32_t)(qb + 2 * odd) * s2h;
          const uint32_t wa = drop_mix(ra ^ P.drop.key), wb = drop_mix((ra + s2h) ^ P.drop.key);
          const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
          const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0xB1, 0xF, 0xF, false);
          const uint32_t w[4] = {odd ? pa : wa, odd ? pb : wb, odd ? wa : pa, odd ? wb : pb};
#pragma unroll
          for (int r = 0; r < 4; ++r) keep[r] = ((w[r] >> (16 * odd)) & 0xFFFFu) >= P.drop.thresh;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(sc[t][r], scale2, kb);
          if constexpr (STRUCT) {
            const int q = qb + r;
            if (kok && q < S) v = __builtin_fmaf(pair_bias<bf16_t, true>(bc, q, key), LOG2E, v);
          }
          const float p = __builtin_amdgcn_exp2f(v - l2v[r]);
          const float dpv = keep[r] ? dp[t][r] : 0.f;
          sc[t][r] = keep[r] ? p : 0.f;
          dp[t][r] = p * (dpv - dlv[r]);
        }
      }
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const bf16x8 fp = v2_pack(sc[2 * pi], sc[2 * pi + 1]);
        const bf16x8 fs = v2_pack(dp[2 * pi], dp[2 * pi + 1]);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          dv[d] = mfma_bf16(v2_frag_tr(img1, t0 + 2 * pi, d * 16, lane), fp, dv[d]);
          dk[d] = mfma_bf16(v2_frag_tr(img0, t0 + 2 * pi, d * 16, lane), fs, dk[d]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kok) {
      const float os = a.scale * ik;
      bf16_t* krow = dqkv + (int64_t)key * gld + D + 4 * g;
      bf16_t* vrow = dqkv + (int64_t)key * gld + 2 * D + 4 * g;
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        *(bf16x4*)(krow + d * 16) = bf16x4{(bf16_t)(dk[d][0] * os), (bf16_t)(dk[d][1] * os), (bf16_t)(dk[d][2] * os), (bf16_t)(dk[d][3] * os)};
        *(bf16x4*)(vrow + d * 16) = bf16x4{(bf16_t)(dv[d][0] * ik), (bf16_t)(dv[d][1] * ik), (bf16_t)(dv[d][2] * ik), (bf16_t)(dv[d][3] * ik)};
      }
    }
  }
}

template <int HD, bool STRUCT, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_v3_kernel(AttnParams P, int s_pad) {
  attn_bwd_v3_body<HD, STRUCT, DROP, true>(P, s_pad);
}
// Short sequences (S <= 128: 38 KB of LDS, four workgroups fit a CU): the same body held to 128 registers (a handful
// spill) so that four waves per SIMD are resident — the ragged BERT sequences are latency-bound, not register-bound.
template <int HD, bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_v3_occ4_kernel(AttnParams P, int s_pad) {
  attn_bwd_v3_body<HD, false, DROP, false>(P, s_pad);   // the 128-register build has no room for the look-ahead
}

// ---------------------------------------------------------------------------- backward in ONE pass (v4)
// The two-pass kernels above compute S, dP, the softmax and the dropout decisions twice — once with queries on the
// lanes (dQ) and once with keys on the lanes (dK, dV) — because either gradient wants the 16 x 16 tiles of dS in the
// orientation the other one does not have.  Here the keys-on-lanes pass is the ONLY pass: besides dV^T += dO^T P and
// dK^T += Q^T dS it leaves every dS tile in LDS as bf16, TRANSPOSED ([key][query]: a lane holds four consecutive
// queries of one key = one ds_write_b64 per tile), and a light second phase reads those tiles back with
// ds_read_b64_tr_b16 as the B operand of dQ^T = K^T dS^T — the same row permutation as every other operand pair of
// this file, so still nothing is shuffled between lanes.  Against v3: 40 instead of 56 MFMAs per 16 x 64 block, the
// exp2 / mixer / select VALU stream once instead of twice, three operand stagings instead of four (V never enters
// LDS: its fragments come straight from global memory).  The price is the dS image: S x S x 2 bytes of LDS on top
// of two operand images (154 KiB at S = 201: one 16-wave workgroup per CU, the same 4 waves per SIMD as the 8-wave
// pairs of v3), so the host takes this kernel where that fits and pays (launch_v3 below).
//   phase 1 (keys on lanes, a wave owns key tiles, walks the queries in pairs of tiles):
//            S = Q K^T, dP = dO V^T -> P, dS -> dV^T, dK^T (registers) and dS^T -> LDS
//   phase 2 (queries on lanes, a wave owns query tiles): dQ^T = K^T dS^T with K staged where Q was
__device__ __forceinline__ bf16x8 v2_frag_tr_ld(const bf16_t* img, int ld, int t0, int c0, int lane, bool second) {
  const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const bf16_t* a = img + (t0 * 16 + 4 * g + q4) * ld + c0 + pp * 4;
  const bf16x4 lo = lds_read_tr16(a);
  bf16x4 hi = bf16x4{0, 0, 0, 0};
  if (second) hi = lds_read_tr16(a + 16 * ld);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// EXACT (rows of at most 6 tiles = 96 tokens, launch_v3): delta_i = sum_j P_ij dP_ij is formed HERE, in fp32, from the very P and dP
// that make dS — not as rowsum(dO o O) from the forward's bf16-rounded output.  The two are equal in exact arithmetic; with O
// rounded to 2^-9 the second one is off by dO . (O_bf16 - O), the same for every key of a row, which is harmless at the block's
// gradient scale but is what is LEFT of the query / key gradients where the softmax is nearly uniform (C4F fixture: |g_query| at
// 1e-3 ... 1e-5 of |g_value|, tests/test_real_shapes_gpu.py).  A wave owns one key tile and at most three query-tile pairs, so
// phase 1 runs as two sweeps with P and dP of all its pairs held in registers (48): sweep A forms them, sums p * dP over the
// wave's 16 keys (four DPP row rotations) and stores the row sums in ITS slab of the delta image in LDS; behind a workgroup
// barrier sweep B adds the slabs of all key tiles in tile order (bitwise reproducible) and forms dS.  The output of the forward is
// not read at all.
constexpr int V4_EXACT_PAIRS = 3;
template <int HD, bool DROP, bool EXACT>
__device__ __forceinline__ void attn_bwd_v4_body(const AttnParams& P, int rows_img, int ldq) {
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nw = nthr >> 6;
  const int h = blockIdx.x, seq = a.seq_ids ? a.seq_ids[blockIdx.y] : (int)blockIdx.y;
  const int SL = a.S, D = a.H * HD;
  const int S = a.seq_offsets ? a.seq_offsets[seq + 1] - a.seq_offsets[seq] : a.S;
  const int64_t row0 = a.seq_offsets ? (int64_t)a.seq_offsets[seq] : (int64_t)seq * a.seq_stride;
  const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;
  const bf16_t* dout = (const bf16_t*)P.dout + row0 * P.ld_dout + h * HD;
  const bf16_t* outp = (const bf16_t*)a.out + row0 * a.ld_out + h * HD;
  bf16_t* dqkv = (bf16_t*)P.dqkv + row0 * P.ld_dqkv + h * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv, dld = a.pos_stride * P.ld_dout, old_ = a.pos_stride * a.ld_out,
                gld = a.pos_stride * P.ld_dqkv;
  bf16_t* img0 = (bf16_t*)smem;                       // Q, then K
  bf16_t* img1 = img0 + rows_img * V2_LD;              // dO
  float* s_kb = (float*)(img1 + rows_img * V2_LD);
  float* s_lse = s_kb + rows_img;
  float* s_delta = s_lse + rows_img;                   // EXACT: [8 key tiles][rows_img] partial row sums, one slab per wave
  bf16_t* dsT = (bf16_t*)(s_delta + rows_img * (EXACT ? 8 : 1));   // [16 * tiles][ldq]: dS^T, key-major
  BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
  const int n_t = (S + 15) >> 4;
  const int rows_live = ((n_t + 1) >> 1) * 32;         // this sequence's rows, in pairs of tiles (zero rows past S)
  if (rows_live > rows_img || S <= 0) return;          // longer than this launch's bound (s_cap); empty: nothing to write
  const float ik = DROP ? P.drop.inv_keep : 1.0f, rik = 1.0f / ik;
  const int g = lane >> 4, c = lane & 15;
  // Everything this item reads from global memory is REQUESTED here, before anything is waited for: the wave's K / V
  // fragments (a wave owns at most one key tile: the host launches >= n_t waves), then per thread at most two 16-byte
  // chunks each of Q, dO and O (the host checks rows_img * 8 <= 2 * nthr), its lse value and mask bytes.  The rolled
  // staging loops this replaces waited for every load inside its own iteration — nine global latencies one after the
  // other per (sequence, head), which was most of the 22 us such an item lived (5-6 us of it arithmetic).
  const int kt = wave, key0 = kt * 16;
  bf16x8 fk[HD / 32], fv[HD / 32];
  if (kt < n_t) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      fk[ks] = v2_frag_glb(qkv + D, tld, S, key0, ks * 32, lane);
      fv[ks] = v2_frag_glb(qkv + 2 * D, tld, S, key0, ks * 32, lane);
    }
  }
  {
    static_assert(HD == 64, "delta reduction and staging assume 8 chunks per row");
    const int q_rows = (a.q_limit > 0 && ((a.q_limit + 15) & ~15) < S) ? ((a.q_limit + 15) & ~15) : S;   // rows the forward computed
    bf16x8 cq[2], cg[2], co[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {      // no lane masks around the requests: see v2_stage_req
      const int e = tid + j * nthr, r = e >> 3, c8 = e & 7;
      const int rc = r < S ? r : S - 1, ro = r < q_rows ? r : q_rows - 1;
      cq[j] = *(const bf16x8*)(qkv + rc * tld + c8 * 8);
      cg[j] = *(const bf16x8*)(dout + rc * dld + c8 * 8);
      if constexpr (!EXACT) co[j] = *(const bf16x8*)(outp + ro * old_ + c8 * 8);
      else co[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};       // delta starts at zero: phase 1 sums it
    }
    const int ti = tid < rows_live ? tid : 0;
    const float lv = a.lse[((int64_t)seq * a.H + h) * SL + (ti < q_rows ? ti : q_rows - 1)];
    const KeyBytes kbv = key_only_bias_req(bc, ti, qkv);
    // Q -> img0; dO -> img1 by the same (row, chunk) walk that forms delta = rowsum(dO * O): the 8 lanes of a row hold
    // its 8 chunks, three shuffles finish the row (rows_live * 8 is a multiple of 64: whole waves, no divergence)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * nthr, r = e >> 3, c8 = e & 7;
      if (e < rows_live * 8) {
        const bf16x8 z = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        const bf16x8 gv = r < S ? cg[j] : z;
        *(bf16x8*)(img0 + r * V2_LD + c8 * 8) = r < S ? cq[j] : z;
        *(bf16x8*)(img1 + r * V2_LD + c8 * 8) = gv;
        float de = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) de += (float)co[j][k] * (float)gv[k];
        de += __shfl_xor(de, 1, 64);
        de += __shfl_xor(de, 2, 64);
        de += __shfl_xor(de, 4, 64);
        if (c8 == 0) s_delta[r] = r < q_rows ? de * rik : 0.f;
      }
    }
    if (tid < rows_live) {
      s_kb[tid] = key_only_bias_of(bc, tid, kbv);
      s_lse[tid] = (tid >= q_rows || lv == -INFINITY) ? INFINITY : lv * LOG2E;
    }
  }
  __syncthreads();
  const int drop_bh = seq * a.H + h;
  const uint32_t s2h = (uint32_t)((SL + 1) >> 1);
  const float scale2 = a.scale * LOG2E;
  const int n_tq = (a.q_limit > 0 && ((a.q_limit + 15) >> 4) < n_t) ? (a.q_limit + 15) >> 4 : n_t;
  const int n_pair_q = (n_tq + 1) >> 1, n_pair_k = (n_t + 1) >> 1;

  // ------------------------------------------------------------------ phase 1 (keys on lanes)
  const int odd = c & 1;
  const uint32_t base_rp = (uint32_t)(drop_bh * SL) * s2h;
  if (kt < n_t) {
    const int key = key0 + c;
    const bool kok = key < S;
    const float kb = s_kb[key];
    const uint32_t kh = base_rp + (uint32_t)(key >> 1);
    bf16_t* ds_row = dsT + (key0 + c) * ldq + 4 * g;
    f32x4 dv[ND], dk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if constexpr (EXACT) {
      // ---- sweep A: P and dP of every pair (registers), delta += row sums of p * dP over this wave's keys
      f32x4 pA[V4_EXACT_PAIRS][2], dA[V4_EXACT_PAIRS][2];
      unsigned keepA[V4_EXACT_PAIRS];
#pragma unroll
      for (int pr = 0; pr < V4_EXACT_PAIRS; ++pr) {
        keepA[pr] = 0xFFu;
#pragma unroll
        for (int t = 0; t < 2; ++t) { pA[pr][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dA[pr][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (pr < n_pair_q) {
          const int t0 = 2 * pr;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
              pA[pr][t] = mfma_bf16(v2_frag_lds(img0, (t0 + t) * 16, ks * 32, lane), fk[ks], pA[pr][t]);   // S[q][key]
              dA[pr][t] = mfma_bf16(v2_frag_lds(img1, (t0 + t) * 16, ks * 32, lane), fv[ks], dA[pr][t]);   // dP[q][key]
            }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int qb = (t0 + t) * 16 + 4 * g;
            const f32x4 l2v = *(const f32x4*)(s_lse + qb);
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) {
              const uint32_t ra = kh + (uint32_t)(qb + 2 * odd) * s2h;
              const uint32_t wa = drop_mix(ra ^ P.drop.key), wb = drop_mix((ra + s2h) ^ P.drop.key);
              const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
              const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0xB1, 0xF, 0xF, false);
              const uint32_t w[4] = {odd ? pa : wa, odd ? pb : wb, odd ? wa : pa, odd ? wb : pb};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                keep[r] = ((w[r] >> (16 * odd)) & 0xFFFFu) >= P.drop.thresh;
                if (!keep[r]) keepA[pr] &= ~(1u << (4 * t + r));
              }
            }
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = __builtin_fmaf(pA[pr][t][r], scale2, kb);
              const float p = __builtin_amdgcn_exp2f(v - l2v[r]);
              const float dpv = keep[r] ? dA[pr][t][r] : 0.f;
              pA[pr][t][r] = p;
              dA[pr][t][r] = dpv;
              x[r] = row16_sum_dpp(p * dpv);          // over the 16 keys of this tile: every lane of the row group holds it
            }
            // this key tile's slab of partial sums (plain stores, summed in a FIXED order by the readers: LDS atomics would make
            // delta — and through one-ulp flips of dS the whole step — depend on the order the waves arrive in)
            if (c == 0) *(f32x4*)(s_delta + kt * rows_img + qb) = f32x4{x[0], x[1], x[2], x[3]};
          }
        }
      }
      __syncthreads();                             // every wave's row sums are in the delta image (the waves without a key tile meet this barrier below)
      // ---- sweep B: dS = p (dP - delta) with the finished delta; products as in the one-sweep form
#pragma unroll
      for (int pr = 0; pr < V4_EXACT_PAIRS; ++pr) {
        if (pr < n_pair_q) {
          const int t0 = 2 * pr;
          f32x4 sc[2], dp[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x4 dlv = *(const f32x4*)(s_delta + (t0 + t) * 16 + 4 * g);
            for (int w = 1; w < n_t; ++w) dlv += *(const f32x4*)(s_delta + w * rows_img + (t0 + t) * 16 + 4 * g);     // key tiles in order
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = pA[pr][t][r];
              dp[t][r] = p * (dA[pr][t][r] - dlv[r]);
              sc[t][r] = (keepA[pr] >> (4 * t + r)) & 1u ? p : 0.f;
            }
          }
          const bf16x8 fp = v2_pack(sc[0], sc[1]);
          const bf16x8 fs = v2_pack(dp[0], dp[1]);
          *(bf16x4*)(ds_row + t0 * 16) = bf16x4{fs[0], fs[1], fs[2], fs[3]};
          if (t0 + 1 < n_tq) *(bf16x4*)(ds_row + t0 * 16 + 16) = bf16x4{fs[4], fs[5], fs[6], fs[7]};
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            dv[d] = mfma_bf16(v2_frag_tr(img1, t0, d * 16, lane), fp, dv[d]);
            dk[d] = mfma_bf16(v2_frag_tr(img0, t0, d * 16, lane), fs, dk[d]);
          }
        }
      }
    }
    for (int pr = 0; pr < (EXACT ? 0 : n_pair_q); ++pr) {
      const int t0 = 2 * pr;
      f32x4 sc[2], dp[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) { sc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          sc[t] = mfma_bf16(v2_frag_lds(img0, (t0 + t) * 16, ks * 32, lane), fk[ks], sc[t]);   // S[q][key]
          dp[t] = mfma_bf16(v2_frag_lds(img1, (t0 + t) * 16, ks * 32, lane), fv[ks], dp[t]);   // dP[q][key]
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qb = (t0 + t) * 16 + 4 * g;
        const f32x4 l2v = *(const f32x4*)(s_lse + qb);
        const f32x4 dlv = *(const f32x4*)(s_delta + qb);
        bool keep[4] = {true, true, true, true};
        if constexpr (DROP) {
          const uint32_t ra = kh + (uint32_t)(qb + 2 * odd) * s2h;
          const uint32_t wa = drop_mix(ra ^ P.drop.key), wb = drop_mix((ra + s2h) ^ P.drop.key);
          const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
          const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0xB1, 0xF, 0xF, false);
          const uint32_t w[4] = {odd ? pa : wa, odd ? pb : wb, odd ? wa : pa, odd ? wb : pb};
#pragma unroll
          for (int r = 0; r < 4; ++r) keep[r] = ((w[r] >> (16 * odd)) & 0xFFFFu) >= P.drop.thresh;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_fmaf(sc[t][r], scale2, kb);
          const float p = __builtin_amdgcn_exp2f(v - l2v[r]);
          const float dpv = keep[r] ? dp[t][r] : 0.f;
          sc[t][r] = keep[r] ? p : 0.f;
          dp[t][r] = p * (dpv - dlv[r]);
        }
      }
      const bf16x8 fp = v2_pack(sc[0], sc[1]);
      const bf16x8 fs = v2_pack(dp[0], dp[1]);
      // dS^T[key][q]: the lane's four queries of either tile are contiguous — one 8-byte write per tile
      *(bf16x4*)(ds_row + t0 * 16) = bf16x4{fs[0], fs[1], fs[2], fs[3]};
      if (t0 + 1 < n_tq) *(bf16x4*)(ds_row + t0 * 16 + 16) = bf16x4{fs[4], fs[5], fs[6], fs[7]};
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        dv[d] = mfma_bf16(v2_frag_tr(img1, t0, d * 16, lane), fp, dv[d]);
        dk[d] = mfma_bf16(v2_frag_tr(img0, t0, d * 16, lane), fs, dk[d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(ND == 4, "rows4_exchange: 64-column rows");
    const Row32 kv = rows4_exchange(dk, a.scale * ik), vv = rows4_exchange(dv, ik);
    if (kok) {
      bf16_t* krow = dqkv + (int64_t)key * gld + D + rows4_off(g);
      bf16_t* vrow = dqkv + (int64_t)key * gld + 2 * D + rows4_off(g);
      *(bf16x8*)krow = kv.a;
      *(bf16x8*)(krow + 32) = kv.b;
      *(bf16x8*)vrow = vv.a;
      *(bf16x8*)(vrow + 32) = vv.b;
    }
  } else if constexpr (EXACT) {
    __syncthreads();                               // the delta barrier between the two sweeps of the waves that own a key tile
  }
  // K goes where Q was — from the registers that already hold it: the K fragments of the waves ARE the rows of K (lane
  // (g, c) of the owner of key tile kt holds chunks g and 4 + g of row 16 kt + c), so K is read from memory once per
  // item, not twice, and no load stands between the two phases.  Rows past S inside a tile hold a copy of the last row
  // (v2_frag_glb clamps) and meet dS^T rows that are exact zeros; the odd pair partner past n_t keeps Q's zero rows.
  __syncthreads();   // every dS^T tile is in LDS; the Q / dO images are free
  if (kt < n_t) {
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) *(bf16x8*)(img0 + (key0 + c) * V2_LD + ks * 32 + 8 * g) = fk[ks];
  }
  __syncthreads();
  // ------------------------------------------------------------------ phase 2 (queries on lanes): dQ^T = K^T dS^T
  for (int qt = wave; qt < n_tq; qt += nw) {
    const int q = qt * 16 + c;
    f32x4 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pk = 0; pk < n_pair_k; ++pk) {
      const bf16x8 fs = v2_frag_tr_ld(dsT, ldq, 2 * pk, qt * 16, lane, 2 * pk + 1 < n_t);
#pragma unroll
      for (int d = 0; d < ND; ++d) dq[d] = mfma_bf16(v2_frag_tr(img0, 2 * pk, d * 16, lane), fs, dq[d]);
    }
    const Row32 qv = rows4_exchange(dq, a.scale * ik);
    if (q < S) {
      bf16_t* orow = dqkv + (int64_t)q * gld + rows4_off(g);
      *(bf16x8*)orow = qv.a;
      *(bf16x8*)(orow + 32) = qv.b;
    }
  }
}

template <int HD, bool DROP>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_v4_kernel(AttnParams P, int rows_img, int ldq) {
  attn_bwd_v4_body<HD, DROP, false>(P, rows_img, ldq);
}
// rows of at most 96 tokens: 4 or 8 waves per workgroup, P / dP of three query-tile pairs in registers (no 128-register cap)
template <int HD, bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_v4x_kernel(AttnParams P, int rows_img, int ldq) {
  attn_bwd_v4_body<HD, DROP, true>(P, rows_img, ldq);
}

// ---------------------------------------------------------------------------- one-pass backward, persistent (v5)
// The one-pass kernel for LONG rows (ViT: 197 tokens, 13 tiles).  There its images fill the LDS of a CU — one workgroup per CU
// — and a (sequence, head) item runs request -> stage -> phase 1 -> K -> phase 2 -> stores with nothing beside it: measured
// per launch (profiles/round3_experiments/attn_prologue_and_store_exchange.log) 140 us of reads at 5.5 TB/s + 290 us of
// arithmetic + 100 us of stores = the 530 us it took; the two halves never overlap.  This form keeps the LDS layout and the
// arithmetic of v4 and changes who waits for whom:
//   * 8 waves of up to 256 registers instead of 16 of 128; a workgroup walks items blockIdx.x, + gridDim.x, ... and requests
//     the NEXT item's Q / dO / O chunks, lse and mask bytes (50 VGPRs) right after the barrier that ends the current item's
//     staging — they arrive under phase 1 — and its K / V fragments into the fragment registers as soon as phase 1 has
//     written K to LDS (under phase 2); no workgroup launch, kernel argument fetch or store drain stands between two items;
//   * a wave owns key tiles w and w + 8 and runs both through ONE query-pair loop: the Q / dO fragments (row-major for
//     S, dP; transposed for dV, dK), lse and delta vectors are read from LDS once for the two tiles — half the LDS read
//     traffic per MFMA of v4, which together with the VALU stream is what bounds phase 1.
// Operand values, summation orders and dropout decisions are those of v4: the outputs are bit-identical to it.
// Measured (profiles/round3_experiments/attn_v5_persistent.log, 512 x 197 x 12 heads, one call): 547 -> 487 us with dropout
// 0.3, 469 -> 375 us without; never slower than v4 on launches of one to three rounds.  The first version requested the next
// item only after phase 1 (it spilled otherwise) and gained 3 %: what spilled were loop-invariant offsets hoisted out of the
// item loop, not the chunks (v5_opaque), and a scratch reload drags s_waitcnt vmcnt(0) behind it.  Where the time of that
// first version went (phases switched off one by one): phase 1 310 us (195 without dropout), phase 2 31, staging / barriers
// / K 67, requests exposed 105.
template <int HD, bool DROP, int NU>
__device__ __forceinline__ void v5_phase1(const AttnParams& P, const bf16_t* img0, const bf16_t* img1, const float* s_kb, const float* s_lse,
                                          const float* s_delta, bf16_t* dsT, int ldq, const bf16x8 (&wk)[2][HD / 32], const bf16x8 (&wv)[2][HD / 32],
                                          int wave, int lane, int S, int SL, int drop_bh, int n_tq, bf16_t* dqkv, int64_t gld, int D, float ik) {
  constexpr int ND = HD / 16;
  const mdt_attn_fwd_args& a = P.f;
  const int g = lane >> 4, c = lane & 15, odd = c & 1;
  const uint32_t s2h = (uint32_t)((SL + 1) >> 1);
  const float scale2 = a.scale * LOG2E;
  const uint32_t base_rp = (uint32_t)(drop_bh * SL) * s2h;
  const int n_pair_q = (n_tq + 1) >> 1;
  int key[NU];
  float kb[NU];
  uint32_t kh[NU];
  bf16_t* ds_row[NU];
  f32x4 dv[NU][ND], dk[NU][ND];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    key[u] = (wave + 8 * u) * 16 + c;
    kb[u] = s_kb[key[u]];
    kh[u] = base_rp + (uint32_t)(key[u] >> 1);
    ds_row[u] = dsT + key[u] * ldq + 4 * g;
#pragma unroll
    for (int d = 0; d < ND; ++d) { dv[u][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[u][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  for (int pr = 0; pr < n_pair_q; ++pr) {
    const int t0 = 2 * pr;
    f32x4 sc[NU][2], dp[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) { sc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8 aq = v2_frag_lds(img0, (t0 + t) * 16, ks * 32, lane), ao = v2_frag_lds(img1, (t0 + t) * 16, ks * 32, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          sc[u][t] = mfma_bf16(aq, wk[u][ks], sc[u][t]);   // S[q][key]
          dp[u][t] = mfma_bf16(ao, wv[u][ks], dp[u][t]);   // dP[q][key]
        }
      }
    bf16x8 fp[NU], fs[NU];
    {
      f32x4 l2v[2], dlv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qb = (t0 + t) * 16 + 4 * g;
        l2v[t] = *(const f32x4*)(s_lse + qb);
        dlv[t] = *(const f32x4*)(s_delta + qb);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int qb = (t0 + t) * 16 + 4 * g;
          bool keep[4] = {true, true, true, true};
          if constexpr (DROP) {
            const uint32_t ra = kh[u] + (uint32_t)(qb + 2 * odd) * s2h;
            const uint32_t wa = drop_mix(ra ^ P.drop.key), wb = drop_mix((ra + s2h) ^ P.drop.key);
            const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
            const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0xB1, 0xF, 0xF, false);
            const uint32_t w[4] = {odd ? pa : wa, odd ? pb : wb, odd ? wa : pa, odd ? wb : pb};
#pragma unroll
            for (int r = 0; r < 4; ++r) keep[r] = ((w[r] >> (16 * odd)) & 0xFFFFu) >= P.drop.thresh;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = __builtin_fmaf(sc[u][t][r], scale2, kb[u]);
            const float p = __builtin_amdgcn_exp2f(v - l2v[t][r]);
            const float dpv = keep[r] ? dp[u][t][r] : 0.f;
            sc[u][t][r] = keep[r] ? p : 0.f;
            dp[u][t][r] = p * (dpv - dlv[t][r]);
          }
        }
        fp[u] = v2_pack(sc[u][0], sc[u][1]);
        fs[u] = v2_pack(dp[u][0], dp[u][1]);
        // dS^T[key][q]: the lane's four queries of either tile are contiguous — one 8-byte write per tile
        *(bf16x4*)(ds_row[u] + t0 * 16) = bf16x4{fs[u][0], fs[u][1], fs[u][2], fs[u][3]};
        if (t0 + 1 < n_tq) *(bf16x4*)(ds_row[u] + t0 * 16 + 16) = bf16x4{fs[u][4], fs[u][5], fs[u][6], fs[u][7]};
      }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const bf16x8 to = v2_frag_tr(img1, t0, d * 16, lane), tq = v2_frag_tr(img0, t0, d * 16, lane);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        dv[u][d] = mfma_bf16(to, fp[u], dv[u][d]);
        dk[u][d] = mfma_bf16(tq, fs[u], dk[u][d]);
      }
    }
  }
  static_assert(ND == 4, "rows4_exchange: 64-column rows");
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const Row32 kv = rows4_exchange(dk[u], a.scale * ik), vv = rows4_exchange(dv[u], ik);
    if (key[u] < S) {
      bf16_t* krow = dqkv + (int64_t)key[u] * gld + D + rows4_off(g);
      bf16_t* vrow = dqkv + (int64_t)key[u] * gld + 2 * D + rows4_off(g);
      *(bf16x8*)krow = kv.a;
      *(bf16x8*)(krow + 32) = kv.b;
      *(bf16x8*)vrow = vv.a;
      *(bf16x8*)(vrow + 32) = vv.b;
    }
  }
}

// The thread index through an empty asm: what is derived from the result is recomputed where it is used instead of being
// hoisted out of the item loop and kept — the hoisted LDS / global offsets of all four sections of v5's loop were what
// pushed it past 256 registers, and a scratch reload (a vector-memory instruction) drags s_waitcnt vmcnt(0) behind it.
__device__ __forceinline__ int v5_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// One int through the scalar cache (uniform address, memory no kernel of this library writes while it runs).  Behind the first
// global store of a kernel the compiler reads such tables with vector loads (the scalar cache is not coherent with them) and
// s_waitcnt vmcnt(0) — which in v5's loop would also wait for every store and request in flight, three times per item.
__device__ __forceinline__ int v5_sload(const int* p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}

template <int HD, bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_v5_kernel(AttnParams P, int rows_img, int ldq, int n_items) {
  constexpr int ND = HD / 16, NCH = 4;         // 16-byte chunks per thread and tensor: rows_img * 8 <= 4 * 512 (host check)
  static_assert(HD == 64, "staging assumes 8 chunks per row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mdt_attn_fwd_args& a = P.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SL = a.S, D = a.H * HD;
  const int64_t tld = a.pos_stride * a.ld_qkv, dld = a.pos_stride * P.ld_dout, old_ = a.pos_stride * a.ld_out, gld = a.pos_stride * P.ld_dqkv;
  bf16_t* img0 = (bf16_t*)smem;                       // Q, then K
  bf16_t* img1 = img0 + rows_img * V2_LD;              // dO
  float* s_kb = (float*)(img1 + rows_img * V2_LD);
  float* s_lse = s_kb + rows_img;
  float* s_delta = s_lse + rows_img;
  bf16_t* dsT = (bf16_t*)(s_delta + rows_img);         // [16 * tiles][ldq]: dS^T, key-major
  const float ik = DROP ? P.drop.inv_keep : 1.0f, rik = 1.0f / ik;

  // an item = (sequence, head); everything about it is uniform over the workgroup
  int seq, h, S, n_t, rows_live, q_rows;
  int64_t row0;
  auto item_of = [&](int it_) {
    const int it = __builtin_amdgcn_readfirstlane(it_);      // uniform: keeps the index loads below on the scalar unit (no vmcnt)
    const int si = it / a.H;
    h = it - si * a.H;
    seq = a.seq_ids ? v5_sload(a.seq_ids + si) : si;
    if (a.seq_offsets) {
      const int o0 = v5_sload(a.seq_offsets + seq), o1 = v5_sload(a.seq_offsets + seq + 1);
      S = o1 - o0;
      row0 = o0;
    } else {
      S = a.S;
      row0 = (int64_t)seq * a.seq_stride;
    }
    n_t = (S + 15) >> 4;
    rows_live = ((n_t + 1) >> 1) * 32;
    if (rows_live > rows_img || S <= 0) { S = 0; n_t = 0; rows_live = 0; }      // longer than this launch's bound (s_cap) / empty: nothing to do
    q_rows = (a.q_limit > 0 && ((a.q_limit + 15) & ~15) < S) ? ((a.q_limit + 15) & ~15) : S;   // rows the forward computed
  };
  // the registers an item arrives in
  bf16x8 cq[NCH], cg[NCH], co[NCH];
  float lv = 0.f;
  KeyBytes kbv{1, 0};
  bf16x8 wk[2][HD / 32], wv[2][HD / 32];
  auto request_rows = [&](int tid) {           // unconditional, clamped: see v2_stage_req
    if (S > 0) {
      const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;
      const bf16_t* dout = (const bf16_t*)P.dout + row0 * P.ld_dout + h * HD;
      const bf16_t* outp = (const bf16_t*)a.out + row0 * a.ld_out + h * HD;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int e = tid + j * 512, r = e >> 3, c8 = e & 7;
        const int rc = r < S ? r : S - 1;
        cq[j] = *(const bf16x8*)(qkv + rc * tld + c8 * 8);
        cg[j] = *(const bf16x8*)(dout + rc * dld + c8 * 8);
        co[j] = *(const bf16x8*)(outp + (r < q_rows ? r : q_rows - 1) * old_ + c8 * 8);
      }
      const int ti = tid < rows_live ? tid : 0;
      lv = a.lse[((int64_t)seq * a.H + h) * SL + (ti < q_rows ? ti : q_rows - 1)];
      BiasCtx bc{seq, h, S, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
      kbv = key_only_bias_req(bc, ti, qkv);
    }
  };
  auto request_frags = [&](int tid) {
    if (S > 0) {
      const int lane = tid & 63, wave = tid >> 6;
      const bf16_t* qkv = (const bf16_t*)a.qkv + row0 * a.ld_qkv + h * HD;

#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          wk[u][ks] = v2_frag_glb(qkv + D, tld, S, (wave + 8 * u) * 16, ks * 32, lane);      // rows clamp to S - 1: a tile past n_t reads valid memory
          wv[u][ks] = v2_frag_glb(qkv + 2 * D, tld, S, (wave + 8 * u) * 16, ks * 32, lane);
        }
    }
  };

  int it = blockIdx.x;
  if (it >= n_items) return;
  item_of(it);
  request_rows(tid);
  request_frags(tid);
  for (;;) {
    // ---- the item whose registers have arrived: stage it
    const int cS = S, c_nt = n_t, c_rows = rows_live, c_qrows = q_rows, c_seq = seq, c_h = h;
    const int64_t c_row0 = row0;
    const int tid1 = v5_opaque(tid);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int e = tid1 + j * 512, r = e >> 3, c8 = e & 7;
      if (e < c_rows * 8) {        // whole waves (c_rows * 8 is a multiple of 256)
        const bf16x8 z = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        const bf16x8 gv = r < cS ? cg[j] : z;
        *(bf16x8*)(img0 + r * V2_LD + c8 * 8) = r < cS ? cq[j] : z;
        *(bf16x8*)(img1 + r * V2_LD + c8 * 8) = gv;
        float de = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) de += (float)co[j][k] * (float)gv[k];
        de += __shfl_xor(de, 1, 64);
        de += __shfl_xor(de, 2, 64);
        de += __shfl_xor(de, 4, 64);
        if (c8 == 0) s_delta[r] = r < c_qrows ? de * rik : 0.f;
      }
    }
    if (tid1 < c_rows) {
      BiasCtx bc{c_seq, c_h, cS, a.H, a.key_mask, a.key_pad, a.dense_bias, a.attn_bias, a.spatial_pos, a.sp_table, a.virt};
      s_kb[tid1] = key_only_bias_of(bc, tid1, kbv);
      s_lse[tid1] = (tid1 >= c_qrows || lv == -INFINITY) ? INFINITY : lv * LOG2E;
    }
    // the K / V fragments count as arrived HERE on every path: phase 1 runs under a wave-uniform branch, and where the paths
    // meet again the compiler's wait bookkeeping keeps the worst case — it then waited for this item's dK / dV stores before
    // writing K to LDS
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) asm volatile("" ::"v"(wk[u][ks]), "v"(wv[u][ks]));
    __syncthreads();
    const int nxt = it + (int)gridDim.x;
    const bool more = nxt < n_items;
    if (more) { item_of(nxt); request_rows(v5_opaque(tid)); }      // Q / dO chunks, lse, mask bytes: arrive under phase 1
    bf16_t* dqkv = (bf16_t*)P.dqkv + c_row0 * P.ld_dqkv + c_h * HD;
    const int n_tq = (a.q_limit > 0 && ((a.q_limit + 15) >> 4) < c_nt) ? (a.q_limit + 15) >> 4 : c_nt;
    // ---- phase 1 (keys on lanes): this wave's key tiles wave and wave + 8
    const int lane1 = v5_opaque(lane);
    if (wave + 8 < c_nt) v5_phase1<HD, DROP, 2>(P, img0, img1, s_kb, s_lse, s_delta, dsT, ldq, wk, wv, wave, lane1, cS, SL, c_seq * a.H + c_h, n_tq, dqkv, gld, D, ik);
    else if (wave < c_nt) v5_phase1<HD, DROP, 1>(P, img0, img1, s_kb, s_lse, s_delta, dsT, ldq, wk, wv, wave, lane1, cS, SL, c_seq * a.H + c_h, n_tq, dqkv, gld, D, ik);
    __syncthreads();   // every dS^T tile is in LDS; the Q / dO images are free
    // K goes where Q was, from the fragment registers (see v4)
    const int tid2 = v5_opaque(tid), lane2 = tid2 & 63;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (wave + 8 * u < c_nt) {
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) *(bf16x8*)(img0 + ((wave + 8 * u) * 16 + (lane2 & 15)) * V2_LD + ks * 32 + 8 * (lane2 >> 4)) = wk[u][ks];
      }
    // ---- the next item's K / V fragments, into the registers just written out: they arrive under phase 2
    if (more) request_frags(tid2);
    __syncthreads();
    // ---- phase 2 (queries on lanes): dQ^T = K^T dS^T
    const int n_pair_k = (c_nt + 1) >> 1;
    const int lane3 = v5_opaque(lane);
    for (int qt = wave; qt < n_tq; qt += 8) {
      const int q = qt * 16 + (lane3 & 15);
      f32x4 dq[ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int pk = 0; pk < n_pair_k; ++pk) {
        const bf16x8 fs = v2_frag_tr_ld(dsT, ldq, 2 * pk, qt * 16, lane3, 2 * pk + 1 < c_nt);
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = mfma_bf16(v2_frag_tr(img0, 2 * pk, d * 16, lane3), fs, dq[d]);
      }
      const Row32 qv = rows4_exchange(dq, a.scale * ik);
      if (q < cS) {
        bf16_t* orow = dqkv + (int64_t)q * gld + rows4_off(lane3 >> 4);
        *(bf16x8*)orow = qv.a;
        *(bf16x8*)(orow + 32) = qv.b;
      }
    }
    if (!more) break;
    it = nxt;
    __syncthreads();   // the images and dS^T are free for the next item
  }
}

// LDS of the one-pass kernel for rows of up to S keys; 0 = does not fit
static size_t v4_lds_bytes(int S, int* rows_img, int* ldq) {
  const int n_t = (S + 15) / 16;
  *rows_img = ((n_t + 1) / 2) * 32;
  *ldq = (n_t & 1) ? 16 * n_t : 16 * n_t + 16;        // ldq / 2 = 8 (mod 16) banks: rows 0-7 of a transposed read fall on distinct 8-bank groups
  const size_t slabs = n_t <= 2 * V4_EXACT_PAIRS ? 7 : 0;      // attn_bwd_v4x: seven more slabs of per-key-tile row sums
  const size_t b = (size_t)2 * *rows_img * V2_LD * 2 + (size_t)(3 + slabs) * *rows_img * 4 + (size_t)16 * n_t * *ldq * 2;
  return b <= 160 * 1024 ? b : 0;
}

template <bool STRUCT, bool DROP>
static int launch_v3(hipStream_t st, const AttnParams& p) {
  const int cap = p.f.s_cap > 0 ? p.f.s_cap : p.f.S;          // longest sequence of this launch
  const int s_pad = (cap + 63) & ~63;
  const int nhist = STRUCT ? ((p.f.num_spatial + 1 + 3) & ~3) : 0;
  const size_t lds = (size_t)2 * s_pad * V2_LD * 2 + (size_t)3 * s_pad * 4 + (size_t)nhist * 4;
  if (lds > 160 * 1024) { set_error("attention_bwd_v3: S=%d needs %zu bytes of LDS", cap, lds); return MDT_ERR_UNSUPPORTED; }
  auto kern = attn_bwd_v3_kernel<64, STRUCT, DROP>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("attention_bwd_v3: cannot reserve %zu bytes of LDS", lds);
      return MDT_ERR_LAUNCH;
    }
  }
  if constexpr (!STRUCT) {
    // one-pass kernel wherever its dS image fits LDS (S <= 224); MDT_ATTN_ONEPASS=0 keeps the two-pass kernels.  In-call
    // A/B with dropout 0.1 (tools/attn_onepass_ab.py): ViT rows (512 x 201) 755 -> 525 us, padded BERT rows (2048 x 104)
    // 993 -> 792 us, ragged BERT rows (8-100 tokens) 592 -> 493 us; gradients equal to bf16 rounding of delta.
    const int op = switches().attn_onepass;
    int rows_img = 0, ldq = 0;
    const size_t lds4 = v4_lds_bytes(cap, &rows_img, &ldq);
    if (lds4 && op != 0) {
      auto k4 = attn_bwd_v4_kernel<64, DROP>;
      static bool attr_set = false;
      if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
          (void)hipGetLastError();
          set_error("attention_bwd_v4: cannot reserve 160 KiB of LDS");
          return MDT_ERR_LAUNCH;
        }
        attr_set = true;
      }
      const int n_t = (cap + 15) / 16;
      // long rows (one workgroup per CU by LDS): the persistent form; MDT_ATTN_ONEPASS=4 keeps v4 there
      if (n_t > 8 && op != 4 && rows_img * 8 <= 4 * 512 && n_t <= 16) {
        auto k5 = attn_bwd_v5_kernel<64, DROP>;
        static bool attr5 = false;
        if (!attr5) {
          if (hipFuncSetAttribute((const void*)k5, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("attention_bwd_v5: cannot reserve 160 KiB of LDS");
            return MDT_ERR_LAUNCH;
          }
          attr5 = true;
        }
        static int cus = 0;
        if (!cus) {
          int dev = 0;
          hipDeviceProp_t prop;
          if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
          if (cus <= 0) cus = 256;
        }
        const int64_t n_items = (int64_t)p.f.H * p.f.nseq;
        const int grid = (int)(n_items < cus ? n_items : cus);
        hipLaunchKernelGGL(k5, dim3(grid), 512, lds4, st, p, rows_img, ldq, (int)n_items);
        return check_launch("attention_bwd_v5");
      }
      const int waves = n_t <= 4 ? 4 : n_t <= 8 ? 8 : 16;
      if (rows_img * 8 > 2 * waves * 64 || n_t > waves) {      // two 16-byte chunks per thread and image, one key tile per wave
        set_error("attention_bwd_v4: %d image rows for %d waves", rows_img, waves);
        return MDT_ERR_UNSUPPORTED;
      }
      // rows of at most 96 tokens (three query-tile pairs): delta summed in the kernel from P and dP (MDT_ATTN_EXACT_DELTA=0: from the bf16 output)
      if (n_t <= 2 * V4_EXACT_PAIRS && switches().attn_exact_delta) {
        auto k4x = attn_bwd_v4x_kernel<64, DROP>;
        static bool attr_x = false;
        if (!attr_x) {
          if (hipFuncSetAttribute((const void*)k4x, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("attention_bwd_v4x: cannot reserve 160 KiB of LDS");
            return MDT_ERR_LAUNCH;
          }
          attr_x = true;
        }
        hipLaunchKernelGGL(k4x, dim3(p.f.H, p.f.nseq), waves * 64, lds4, st, p, rows_img, ldq);
        return check_launch("attention_bwd_v4x");
      }
      hipLaunchKernelGGL(k4, dim3(p.f.H, p.f.nseq), waves * 64, lds4, st, p, rows_img, ldq);
      return check_launch("attention_bwd_v4");
    }
    if (s_pad <= 128 && !switches().attn_no_occ4) {
      hipLaunchKernelGGL((attn_bwd_v3_occ4_kernel<64, DROP>), dim3(p.f.H, p.f.nseq), 256, lds, st, p, s_pad);
      return check_launch("attention_bwd_v3_occ4");
    }
    // long sequences (ViT: 13 tiles): LDS allows two workgroups per CU; 8 waves each in the 128-register build
    // = 4 waves per SIMD instead of 2
    if (s_pad > 128 && !switches().attn_no_w8) {      // in-call A/B at the ViT shape: +0.7 % on the step
      hipLaunchKernelGGL((attn_bwd_v3_occ4_kernel<64, DROP>), dim3(p.f.H, p.f.nseq), 512, lds, st, p, s_pad);
      return check_launch("attention_bwd_v3_w8");
    }
  }
  hipLaunchKernelGGL(kern, dim3(p.f.H, p.f.nseq), 256, lds, st, p, s_pad);
  return check_launch("attention_bwd_v3");
}

int attention_v3_bwd_dispatch(hipStream_t st, const AttnParams& p) {
  const bool s = p.f.attn_bias != nullptr, d = p.f.drop_p > 0.f;
  if (s && d) return launch_v3<true, true>(st, p);
  if (s) return launch_v3<true, false>(st, p);
  if (d) return launch_v3<false, true>(st, p);
  return launch_v3<false, false>(st, p);
}

template <int NT, bool STRUCT, bool DROP, bool BWD>
static int launch_v2(hipStream_t st, const AttnParams& p) {
  constexpr int S_PAD = ((NT + 1) / 2) * 32;
  const int nhist = (STRUCT && BWD) ? ((p.f.num_spatial + 1 + 3) & ~3) : 0;
  const size_t lds = (size_t)2 * S_PAD * V2_LD * 2 + (size_t)(BWD ? 3 : 1) * S_PAD * 4 + (size_t)nhist * 4 + (BWD ? 0 : 16);   // forward: + the staging's spare chunk
  if constexpr (BWD && NT > 7) {
    // the whole-row backward runs out of registers past 112 keys; attention.hip routes those to the chunked v3
    set_error("attention_v2: backward supports S <= 112 (got %d)", p.f.S);
    return MDT_ERR_UNSUPPORTED;
  } else {
    constexpr int NWF = (NT >= 13 && !STRUCT) ? 8 : 4;     // forward: 8 waves for the long (ViT) rows, 4 waves per SIMD
    const void* kern;
    if constexpr (BWD) kern = (const void*)attn_bwd_v2_kernel<64, NT, STRUCT, DROP>;
    else kern = (const void*)attn_fwd_v2_kernel<64, NT, STRUCT, DROP, NWF>;
    if (lds > 64 * 1024) {
      if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("attention_v2: cannot reserve %zu bytes of LDS", lds);
        return MDT_ERR_LAUNCH;
      }
    }
    if constexpr (BWD) hipLaunchKernelGGL((attn_bwd_v2_kernel<64, NT, STRUCT, DROP>), dim3(p.f.H, p.f.nseq), 256, lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_v2_kernel<64, NT, STRUCT, DROP, NWF>), dim3(p.f.H, p.f.nseq), NWF * 64, lds, st, p);
    return check_launch(BWD ? "attention_bwd_v2" : "attention_fwd_v2");
  }
}

template <bool STRUCT, bool DROP, bool BWD>
static int dispatch_v2_nt(hipStream_t st, const AttnParams& p) {
  const int nt = ((p.f.s_cap > 0 ? p.f.s_cap : p.f.S) + 15) / 16;
#define V2_CASE(N_) if (nt <= N_) return launch_v2<N_, STRUCT, DROP, BWD>(st, p);
  V2_CASE(2) V2_CASE(4) V2_CASE(5) V2_CASE(7) V2_CASE(9) V2_CASE(13) V2_CASE(17)
#undef V2_CASE
  set_error("attention_v2: S=%d exceeds 272", p.f.S);
  return MDT_ERR_UNSUPPORTED;
}

int attention_v2_dispatch(hipStream_t st, const AttnParams& p, bool bwd) {
  const bool s = p.f.attn_bias != nullptr, d = p.f.drop_p > 0.f;
#define V2_GO(S_, D_) return bwd ? dispatch_v2_nt<S_, D_, true>(st, p) : dispatch_v2_nt<S_, D_, false>(st, p);
  if (s && d) V2_GO(true, true)
  if (s) V2_GO(true, false)
  if (d) V2_GO(false, true)
  V2_GO(false, false)
#undef V2_GO
}

}  // namespace mdt
