/*
 * mdt_hip.h — C ABI of the MI355X-native (gfx950) mDT hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / a HIP stream
 * (as `void*` = hipStream_t) and NO torch types.  The caller owns every buffer,
 * workspace included; the library allocates nothing and never synchronises the host.
 * Its only state: a thread-local error string, the MDT_* environment switches read
 * once at the first launch (mdt_reload_env re-reads them) and the pointer to the
 * caller's tile-queue buffer (mdt_gemm_set_tile_queue, optional).  Entry points are
 * re-entrant, so forward and backward may run on different host threads and any call
 * sequence can be captured in a hipGraph.  (The profiling switch MDT_GEMM_STAMP=1 is
 * the one exception: it allocates, synchronises and prints — never set in production.)
 *
 * Return value: 0 on success, a negative mdt_status otherwise; the message is
 * available through mdt_last_error_string() on the calling thread.
 *
 * The reference has no native boundary for this path (it is eager PyTorch); each
 * function cites the reference operator sequence it replaces (paths relative to
 * /root/reference/mDT/src).  INTEGRATION.md shows the reference-side binding.
 */
#ifndef MDT_HIP_H
#define MDT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDT_ABI_VERSION 1

typedef enum { MDT_F32 = 0, MDT_BF16 = 1 } mdt_dtype;

typedef enum {
  MDT_OK = 0,
  MDT_ERR_ARG = -1,          /* shape / alignment / null-pointer contract violated */
  MDT_ERR_UNSUPPORTED = -2,  /* valid request this build has no kernel for */
  MDT_ERR_LAUNCH = -3        /* hipLaunch / hipMemsetAsync reported an error */
} mdt_status;

/* GEMM epilogue selector (bit flags) */
enum {
  MDT_EPI_BIAS = 1,      /* + bias[n] */
  MDT_EPI_GELU = 2,      /* erf-GELU; if aux != NULL the pre-activation is stored there */
  MDT_EPI_RESIDUAL = 4,  /* + residual[m, n] */
  MDT_EPI_DGELU = 8,     /* * gelu'(aux[m, n]) (backward of MDT_EPI_GELU) */
  MDT_EPI_ACCUM = 16,    /* C += result (plain read-modify-write, fp32 or T) */
  MDT_EPI_ATOMIC = 32,   /* C (fp32) += result with float atomics (split-K weight gradients) */
  MDT_EPI_COLSUM = 128,  /* colsum[n] += sum_m C[m, n] (fp32 atomics) — the bias gradient of the layer whose
                            output gradient this GEMM produces, fused instead of a separate pass */
  MDT_EPI_AUX_GRAD = 256, /* with MDT_EPI_GELU and aux != NULL: aux receives d out / d pre-activation
                            (GELU'(u), times the dropout scale of the element when MDT_EPI_DROPOUT is set)
                            instead of u — the backward pass is then a plain MDT_EPI_MULAUX */
  MDT_EPI_MULAUX = 512,  /* * aux[m, n] (backward of an epilogue that saved its derivative) */
  MDT_EPI_ASUM = 1024,   /* colsum[m] += sum_k op(A)[m, k] (fp32 atomics): with trans_a = 1 the column sums of the STORED
                            A — the bias gradient db = colsum(dY) riding on the weight-gradient GEMM dW = dY^T X that
                            streams dY anyway (one extra MFMA against a vector of ones per A fragment, in the workgroups
                            of the first tile column only).  Needs trans_a = 1, MDT_EPI_ATOMIC and a colsum buffer of M
                            floats; excludes MDT_EPI_COLSUM (the two share the buffer argument) */
  MDT_EPI_DROPOUT = 64   /* inverted dropout on the value after bias / GELU and before the residual add
                            (with MDT_EPI_DGELU: on the incoming gradient, before the GELU' factor);
                            element (m, n) of site `drop_seed` uses counter m*N + n */
};

int mdt_abi_version(void);
const char* mdt_last_error_string(void);
/* sha256 (hex) over csrc/ and this header at the time the library was linked (build.py source_hash()): the Python side
 * refuses a libmdt_hip.so that was not built from the sources next to it. */
const char* mdt_source_hash(void);

/* ------------------------------------------------------------------ GEMM
 * C[M,N] = epilogue( alpha * op(A)[M,K] @ op(B)[K,N] )
 *   trans_a = 0: A stored [M,K] (lda = row stride);  1: stored [K,M]
 *   trans_b = 0: B stored [N,K] (nn.Linear weight, y = x W^T);  1: stored [K,N]
 * Replaces nn.Linear forward and its autograd (modules/multihead_attention.py:134-137,203;
 * modules/graphormer_graph_encoder_layer.py:135-137; HF BertLayer/ViTLayer dense layers
 * called from modules/multi_graphormer_fusion_layer.py:94-96,138-146).
 * dtype: element type of A, B, bias, residual, aux.  out_dtype: element type of C.
 * split_k > 1 requires MDT_EPI_ATOMIC and a pre-zeroed (or accumulating) fp32 C.
 */
int mdt_gemm(void* stream, int dtype, int out_dtype, int trans_a, int trans_b,
             int64_t M, int64_t N, int64_t K,
             const void* A, int64_t lda, const void* B, int64_t ldb,
             void* C, int64_t ldc, int epilogue, float alpha,
             const void* bias, const void* residual, int64_t ldr,
             void* aux, int64_t ldaux, int split_k, float drop_p, uint64_t drop_seed, float* colsum);
/* Dynamic tile queue of the persistent GEMM (MDT_GEMM_DYNAMIC=1, for nodes where RCCL kernels hold compute units during
 * backward): the queue heads live in a CALLER-OWNED device buffer of mdt_gemm_tile_queue_bytes() bytes, zero-initialised
 * by the caller and registered once (NULL unregisters); the library allocates nothing.  Without it the static tile walk
 * is used. */
size_t mdt_gemm_tile_queue_bytes(void);
int mdt_gemm_set_tile_queue(void* zeroed_device_buffer, size_t bytes);
/* Re-read the MDT_* environment switches (tuning / diagnostics).  They are read once, at the first launch; a host that
 * changes them afterwards (tests, A/B tools) calls this. */
void mdt_reload_env(void);

/* Inverted dropout as a stand-alone op: y[m, n] = x[m, n] * keep(seed, m*D + n) / (1 - p).  The mask is
 * a pure function of (seed, counter), so calling it on the gradient with the same seed is the backward
 * (FairseqDropout / nn.Dropout in modules/graphormer_graph_encoder_layer.py:127,136,138,
 * modules/multigraphormer_graph_encoder.py:403 and inside HF BertLayer / ViTLayer). */
int mdt_dropout(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx, void* y, int64_t ldy,
                float p, uint64_t seed);
/* Test hook: mask[i] = 1 if counter i of site `seed` is kept. */
int mdt_dropout_mask(void* stream, int64_t n, float p, uint64_t seed, uint8_t* mask);

/* Column sums: out[n] (+)= sum_m w[m] * X[m,n]  (bias gradients; w = NULL means 1, otherwise an
 * int32 row weight — token-type embedding gradient).  out is fp32, accumulated with atomics —
 * zero it first unless accumulating. */
int mdt_colsum(void* stream, int dtype, int64_t M, int64_t N, const void* X, int64_t ldx, float* out,
               const int32_t* row_weight);

/* ------------------------------------------------------------------ LayerNorm
 * y = (x - mean) * rstd * gamma + beta, rows of width D (fairseq LayerNorm eps 1e-5,
 * HF LayerNorm eps 1e-12: modules/graphormer_graph_encoder_layer.py:127-130,138-141;
 * modules/multigraphormer_graph_encoder.py:400-403).  mean / rstd are fp32 [rows].
 */
int mdt_layernorm_fwd(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx,
                      const void* gamma, const void* beta, float eps,
                      void* y, int64_t ldy, float* mean, float* rstd);
/* mdt_layernorm_fwd (bf16 rows) whose output ALSO leaves as fp8 for the 8-bit GEMM that consumes it (fp8 operands, below):
 * q8_out u8[rows, D] (rows of ld_q8 bytes) = saturate(q8_format, y * *q8_scale), *q8_amax = max(*q8_amax, max |y|) — bit for bit
 * what mdt_fp8_quantize makes of y, without its pass over y.  q8_out NULL: plain mdt_layernorm_fwd. */
int mdt_layernorm_fwd_q8(void* stream, int dtype, int64_t rows, int D, const void* x, int64_t ldx, const void* gamma,
                         const void* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd, void* q8_out,
                         int64_t ld_q8, int q8_format, const float* q8_scale, float* q8_amax);
/* dx = LN'(dy) (+ add[m,:] if add != NULL);  dgamma/dbeta fp32, atomically accumulated.
 * Optional fused tail for the layer that FEEDS this LayerNorm through a hidden dropout + residual:
 *   dxd (may be NULL) = dx * keep(drop_seed, m*D + n) / (1 - drop_p)   — gradient of the dense output,
 *   colsum (may be NULL, fp32[D]) += column sums of dxd (of dx when dxd is NULL) — that layer's bias gradient. */
int mdt_layernorm_bwd(void* stream, int dtype, int64_t rows, int D, const void* dy, int64_t lddy,
                      const void* x, int64_t ldx, const void* gamma, const float* mean, const float* rstd,
                      const void* add, int64_t ldadd, void* dx, int64_t lddx, float* dgamma, float* dbeta,
                      void* dxd, int64_t lddxd, float drop_p, uint64_t drop_seed, float* colsum);

/* ------------------------------------------------------------------ attention
 * Self-attention over `nseq` independent sequences of `S` tokens, `H` heads of `hd`,
 * q|k|v packed per token: qkv[row, 0:D | D:2D | 2D:3D], D = H*hd.  Row of token p of
 * sequence s = s*seq_stride + p*pos_stride (so batch-major and fairseq's time-major
 * [T,B,C] layouts need no copy).  scores = scale * q.k + bias, softmax in fp32,
 * out[row, h*hd:(h+1)*hd] = P @ v.  lse[s,h,p] (fp32) is saved for backward.
 *
 * Bias / mask sources, all optional (NULL):
 *   key_mask   u8[nseq,S]   1 = key may be attended (HF additive mask, quirk 9 of
 *                           −65504 / finfo.min ≡ excluded)
 *   dense_bias f32[nseq,H,S,S]   additive (modules/multihead_attention.py:173-174)
 *   structural (modules/graphormer_layers.py:86-110 fused, never materialised):
 *     attn_bias f32[nseq,S,S] (added TWICE, :93 and :108), spatial_pos i32[nseq,S-1,S-1],
 *     sp_table T[num_spatial,H], virt T[H] (graph-token virtual distance)
 *   key_pad    u8[nseq,S]   1 = padded key → −inf (multihead_attention.py:180-187)
 */
typedef struct {
  int dtype;
  int nseq, S, H, hd;
  int64_t seq_stride, pos_stride; /* in rows */
  float scale;
  const void* qkv; int64_t ld_qkv;
  void* out; int64_t ld_out;
  float* lse;
  const uint8_t* key_mask;
  const float* dense_bias;
  const float* attn_bias;
  const int32_t* spatial_pos;
  const void* sp_table;
  const void* virt;
  const uint8_t* key_pad;
  int num_spatial;
  float drop_p;              /* attention-probability dropout (0 = off); counter ((s*H + h)*S + q)*S2 + key, S2 = S rounded up to even */
  uint64_t drop_seed;
  const int32_t* seq_offsets; /* NULL, or i32[nseq + 1]: RAGGED sequences — sequence s is the S_s = off[s+1] - off[s] <= S
                                 consecutive rows starting at row off[s] (seq_stride unused, pos_stride must be 1, no masks /
                                 biases: every packed token is valid).  S stays the bound that sizes lse [nseq,H,S] and the
                                 dropout counters.  This is how padded token positions of a comment batch are never computed. */
  int q_limit;                /* 0 = every query row.  n > 0: only the first n rows of each sequence are needed as QUERIES
                                 (keys / values are always the whole sequence): forward leaves `out` / `lse` of the other rows
                                 unwritten (backward never reads them), backward assumes their `dout` is zero and does not
                                 write their dQ — the caller zeroes the dQ third of dqkv beforehand; dK / dV are written for
                                 every row.  Kernels may round n up (they work in 16-row tiles) or ignore it. */
  const int32_t* seq_ids;     /* NULL, or i32[nseq] (ragged bf16 launches only): this launch covers the sequences
                                 seq_ids[0 .. nseq) of a ragged set of `nseq_total` sequences — seq_offsets has nseq_total + 1
                                 entries, lse is [nseq_total, H, S] and the dropout counters use the sequence's own index, so a
                                 set may be processed as several launches (length bins) with bit-identical results. */
  int s_cap;                  /* 0, or: no sequence of THIS launch is longer than s_cap (<= S) rows.  Kernels size their LDS
                                 images and unrolled key loops by it instead of by S (a bin of short comments then runs the
                                 small kernels); S keeps defining the lse / dropout-counter geometry.  A longer sequence is
                                 skipped, not computed wrongly — the caller's host-side lengths are the contract. */
  int nseq_total;             /* with seq_ids: size of the whole ragged set (0 = nseq) */
} mdt_attn_fwd_args;
int mdt_attention_fwd(void* stream, const mdt_attn_fwd_args* a);

typedef struct {
  mdt_attn_fwd_args f;        /* same tensors as forward (out = forward output, lse filled) */
  const void* dout; int64_t ld_dout;
  void* dqkv; int64_t ld_dqkv;
  float* d_dense_bias;        /* f32[nseq,H,S,S] or NULL */
  float* d_sp_table;          /* f32[num_spatial,H], atomically accumulated, or NULL */
  float* d_virt;              /* f32[H], atomically accumulated, or NULL */
} mdt_attn_bwd_args;
int mdt_attention_bwd(void* stream, const mdt_attn_bwd_args* a);
/* Head-averaged attention probabilities, fp32 [nseq, S, S] (modules/multihead_attention.py:205-214, need_weights=True):
 * recomputed from the qkv buffer and the log-sum-exp mdt_attention_fwd wrote (same args struct; out / dropout fields are
 * not read).  Masked keys give 0. */
int mdt_attention_mean_probs(void* stream, const mdt_attn_fwd_args* a, float* out);
/* Per-head weights of the same attention (need_head_weights / before_softmax of modules/multihead_attention.py:91-102,186-214):
 * out fp32 [nseq, H, S, S] = softmax probabilities before dropout (raw_scores = 0; needs a->lse) or the scores
 * q k^T * scale + bias with masked keys at -inf (raw_scores = 1).  Dense sequences only. */
int mdt_attention_head_weights(void* stream, const mdt_attn_fwd_args* a, int raw_scores, float* out);

/* Materialise the [nseq,H,S,S] structural bias (API parity with GraphAttnBias.forward,
 * modules/graphormer_layers.py:86-110); the fused encoder path never calls this. */
int mdt_graph_attn_bias(void* stream, int dtype, int nseq, int S, int H, const float* attn_bias,
                        const int32_t* spatial_pos, const void* sp_table, const void* virt,
                        float* out);

/* ------------------------------------------------------------------ row movers
 * dst[di(r), :] = alpha * a[ai(r), :] + beta * b[bi(r), :] (+ dst if accumulate)
 * for r in [0, nrows); an index array may be NULL and then a two-level affine map is
 * applied: row = (r / inner) * stride + r % inner + offset (inner = 1: r*stride + offset).
 * A negative index skips the row (dst) or contributes zero (a, b).  This one kernel is the bottleneck-token
 * exchange between the text, image and graph token spaces
 * (modules/multigraphormer_graph_encoder.py:339,363-371,425,435;
 *  modules/multi_graphormer_fusion_layer.py:37-66) on precomputed CSR indices — the
 * reference's boolean-mask indexing (implicit nonzero + host sync) is gone.
 */
int mdt_row_axpby(void* stream, int dtype, int64_t nrows, int D,
                  void* dst, int64_t ldd, const int32_t* di, int64_t d_inner, int64_t d_stride, int64_t d_off,
                  const void* a, int64_t lda, const int32_t* ai, int64_t a_inner, int64_t a_stride, int64_t a_off, float alpha,
                  const void* b, int64_t ldb, const int32_t* bi, int64_t b_inner, int64_t b_stride, int64_t b_off, float beta,
                  int accumulate);
/* fp32 table[idx[r], :] += src[r, :]  (embedding backward; atomics).  idx < 0 skipped. */
int mdt_row_scatter_add_f32(void* stream, int dtype, int64_t nrows, int D, float* table, int64_t ldt,
                            const int32_t* idx, const void* src, int64_t lds, int64_t s_stride, int64_t s_off);

/* BERT embeddings: out[r,:] = word[ids[r]] + pos[r % L] + type[types[r]]  (pre-LayerNorm sum;
 * HF BertEmbeddings, call site modules/multigraphormer_graph_encoder.py:325-329).
 * Row r of the [M, L] id matrix goes to out row (r / L) * out_seq_stride + out_off + r % L. */
int mdt_bert_embed_sum(void* stream, int dtype, int64_t M, int L, const int32_t* ids, const int32_t* types,
                       const void* word, const void* pos, const void* type, int D,
                       void* out, int64_t ldo, int64_t out_seq_stride, int64_t out_off);

/* Ragged form of the same sum: one output row per VALID token (padded positions are never materialised);
 * out[r,:] = word[ids[r]] + pos[pos_ids[r]] + type[types[r]]. */
int mdt_bert_embed_rows(void* stream, int dtype, int64_t rows, const int32_t* ids, const int32_t* types,
                        const int32_t* pos_ids, const void* word, const void* pos, const void* type, int D,
                        void* out, int64_t ldo);
/* The same sum followed by its LayerNorm in ONE pass (HF BertEmbeddings incl. its LayerNorm, as run by the truncated
 * BertModel at modules/multigraphormer_graph_encoder.py:325-329): xs[r,:] = T(word + type + pos) exactly as mdt_bert_embed_rows
 * leaves it (xs may be NULL: nobody will read the sum), y[r,:] = LayerNorm(xs[r,:]) * gamma + beta, mean / rstd fp32[rows] (may be
 * NULL) — bit-identical to mdt_bert_embed_rows + mdt_layernorm_fwd. */
int mdt_bert_embed_ln_rows(void* stream, int dtype, int64_t rows, const int32_t* ids, const int32_t* types,
                           const int32_t* pos_ids, const void* word, const void* pos, const void* type, int D,
                           const void* gamma, const void* beta, float eps, void* xs, int64_t ldxs, void* y, int64_t ldy,
                           float* mean, float* rstd);

/* ViT patch gather (Conv2d k=s=p is a pure re-index, modules/multigraphormer_graph_encoder.py:333):
 * cols[(i*np + py*gw + px), c*p*p + dy*p + dx] = img[i, c, py*p+dy, px*p+dx]  (img fp32 → T). */
int mdt_vit_patchify(void* stream, int dtype, int I, int C, int HW, int p, const float* img, void* cols, int64_t ldc);
/* tokens[i, off + 0] = cls + pos[0]; tokens[i, off + 1 + j] = patches[i*np + j] + pos[1 + j]. */
int mdt_vit_assemble(void* stream, int dtype, int I, int np, int D, const void* patches, int64_t ldp,
                     const void* cls, const void* pos, void* tokens, int64_t ldt, int64_t seq_stride, int64_t off);
/* The three steps above as ONE launch for bf16 weights and 16 x 16 patches (a GEMM whose A loader gathers from
 * pixel_values; replaces HF ViTEmbeddings as invoked at modules/multigraphormer_graph_encoder.py:332-335):
 * tokens[i, off + 0] = cls + pos[0]; tokens[i, off + 1 + j] = Conv2d(img[i])[:, py, px] + bias + pos[1 + j], j = py*gw + px,
 * w bf16 [D, C*p*p] with row stride ldw (the Conv2d weight viewed flat), one rounding to bf16 at the end.
 * MDT_ERR_UNSUPPORTED for other patch sizes or D not a multiple of 128: the caller takes the three-launch route. */
int mdt_vit_patch_embed(void* stream, int I, int C, int HW, int p, const float* img, const void* w, int64_t ldw,
                        const void* bias, const void* cls, const void* pos, int D, void* tokens, int64_t ldt,
                        int64_t seq_stride, int64_t off);

/* Graph node features (modules/graphormer_layers.py:39-50) on the padded [B, T] grid:
 * x[b,0] = graph_token; x[b,1+n] = (node_row[b,n] >= 0 ? src[node_row[b,n]] : 0)
 *                                  + in_emb[in_degree[b,n]] + out_emb[out_degree[b,n]]. */
int mdt_graph_node_feature(void* stream, int dtype, int B, int T, int D, const void* src, int64_t lds,
                           const int32_t* node_row, const int32_t* in_degree, const int32_t* out_degree,
                           const void* in_emb, const void* out_emb, const void* graph_token, void* x, int64_t ldx);

/* Head tail (models/multi_modal_discussion_transformer.py:265-274): logits[m, c] =
 * 0.5 * (cls(pooled_text[m]) + cls(pooled_bn[m])), pooled = tanh(pre-activation) given. */
int mdt_tanh_fwd(void* stream, int dtype, int64_t n, const void* x, void* y);
int mdt_tanh_bwd(void* stream, int dtype, int64_t n, const void* y, const void* dy, void* dx);

/* Weighted 2-class cross entropy in fp16 arithmetic + counters
 * (criterions/hatespeech_loss.py:95-118, quirk 14): logits T[M,2] gathered by rows[r];
 * out_loss f32[1] (sum), counters i32[4] = ncorrect, num_positive_correct, total_positive,
 * num_pred_positive; dlogits T[M,2] (zero for unlabelled rows) scaled by grad_scale. */
int mdt_node_ce(void* stream, int dtype, int64_t M, int nlab, const void* logits, const int32_t* rows,
                const int32_t* targets, float w_neg, float w_pos, int fp16_loss, float grad_scale,
                float* out_loss, int32_t* counters, void* dlogits);

/* ------------------------------------------------------------------ fp8 operands (BASELINE.json configs[4])
 * Per-tensor-scaled OCP fp8 for the big k-contiguous GEMMs of the encoder blocks (every nn.Linear forward and, against a
 * transposed weight copy, every input gradient): C[M,N] (bf16) = epilogue((1/scale_a)(1/scale_b) * A[M,K] B[N,K]^T), A in
 * e4m3 (a_format 0: activations) or e5m2 (1: gradients), B in e4m3, fp32 accumulation on v_mfma_f32_16x16x128_f8f6f4 (4-wave
 * kernel, K % 128 == 0, K >= 640) or v_mfma_f32_16x16x32_{fp8,bf8}_fp8 (8-wave kernel, the rest),
 * same epilogues as mdt_gemm (bf16 store forms).  inv_scale_a / inv_scale_b are DEVICE floats (delayed scaling keeps
 * every scale on the device).  Returns MDT_ERR_UNSUPPORTED for shapes the 8-bit kernel is not built for (N % 256, K % 64,
 * K < 256, unaligned rows): the caller then stays in bf16. */
int mdt_gemm_fp8(void* stream, int a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                 int64_t ldb, void* C, int64_t ldc, int epilogue, const float* inv_scale_a, const float* inv_scale_b,
                 const void* bias, const void* residual, int64_t ldr, void* aux, int64_t ldaux, float drop_p,
                 uint64_t drop_seed, float* colsum);
/* mdt_gemm_fp8 whose output ALSO leaves as fp8, for the next 8-bit GEMM to consume without a quantisation pass of its own
 * ("quantise inside the producer": the GELU forward hands fc2 its operand, fc2's input gradient hands fc1's): q8_out u8[M, N]
 * (rows of ld_q8 bytes) = saturate(q8_format, bf16(C) * *q8_scale), *q8_amax = max(*q8_amax, max |bf16(C)|) — bit for bit what
 * mdt_fp8_quantize makes of C.  q8_out NULL: plain mdt_gemm_fp8.  MDT_ERR_UNSUPPORTED when no kernel writes the copy for this
 * epilogue / shape (only the block-MFMA kernel does, for bias + GELU + saved derivative → e4m3 and saved-derivative multiply +
 * column sums → e5m2): the caller then quantises C itself. */
int mdt_gemm_fp8_q8(void* stream, int a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                    int64_t ldb, void* C, int64_t ldc, int epilogue, const float* inv_scale_a, const float* inv_scale_b,
                    const void* bias, const void* residual, int64_t ldr, void* aux, int64_t ldaux, float drop_p,
                    uint64_t drop_seed, float* colsum, void* q8_out, int64_t ld_q8, int q8_format, const float* q8_scale,
                    float* q8_amax);
/* dst u8[rows, cols] = saturate_fp8(src * *scale_dev) (fmt 0: e4m3, |x| <= 448; 1: e5m2, |x| <= 57344; scale_dev NULL: 1);
 * *amax_dev = max(*amax_dev, max |src|) (NULL: not tracked) — the input of the next step's scale. */
int mdt_fp8_quantize(void* stream, int src_dtype, int fmt, int64_t rows, int64_t cols, const void* src, int64_t ld_src,
                     void* dst, int64_t ld_dst, const float* scale_dev, float* amax_dev);
/* Delayed scaling, all sites at once: scale[i] = fmt_max[i] / (amax[i] * margin), inv_scale[i] = 1 / scale[i] for every
 * site whose amax is positive and finite (others keep their scale); amax[i] = 0. */
int mdt_fp8_scale_update(void* stream, int n, float* amax, float* scale, float* inv_scale, const float* fmt_max, float margin);

/* Community-contrastive loss on the global discussion embeddings (criterions/contrastive_loss.py:76-180):
 * emb T[B, D] (row stride ld), y / hard_y f32[B] (community label of every tree and of its polar-opposite community);
 * sim = scale * normalize(emb) normalize(emb)^T, weighted BCE against [y_i == y_j] with the reference's soft-negative
 * weights (adaptive != 0: 2 * #hard / #soft per row, applied along the LAST axis as the reference's broadcast does;
 * otherwise soft_negative_weight), diagonal excluded.  out_loss f32[1] (sum over the B x B pairs), counters i32[4] =
 * ncorrect, positive_correct, total_positive, pred_positive as the reference defines them (:153-161);
 * d_emb T[B, D] = grad_scale * dL/d emb (NULL: forward only).  workspace: mdt_contrastive_loss_workspace_bytes(B, D)
 * bytes owned by the caller. */
size_t mdt_contrastive_loss_workspace_bytes(int B, int D);
int mdt_contrastive_loss(void* stream, int dtype, int B, int D, const void* emb, int64_t ld, const float* y,
                         const float* hard_y, float scale, float soft_negative_weight, int adaptive, void* workspace,
                         float grad_scale, float* out_loss, int32_t* counters, void* d_emb, int64_t ldd);

/* Elementwise cast / transpose helpers for the bf16 weight shadow copies. */
int mdt_cast(void* stream, int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst);
int mdt_transpose2d(void* stream, int src_dtype, int dst_dtype, int64_t rows, int64_t cols,
                    const void* src, int64_t lds, void* dst, int64_t ldd);

/* ------------------------------------------------------------------ optimiser (the step after the path)
 * Fused Adam with FairSeq semantics over one tensor: g = grad * (*grad_scale if given);
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= wd*lr*p; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v)+eps).
 * `master` (fp32, optional) is the high-precision copy of a bf16 `param`.  grad_scale is a DEVICE scalar
 * (e.g. 1 / global sample size) so no host synchronisation is needed. */
int mdt_adam_step(void* stream, int dtype, int64_t n, void* param, float* master, const float* grad, float* m,
                  float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  const float* grad_scale);

/* Multi-tensor form: ONE launch over a device-resident table of same-dtype parameter tensors.  chunk_first[t] = first
 * 4096-element chunk of tensor t, chunk_first[n_tensors] = total_chunks (int64, device). */
typedef struct {
  void* param;          /* T[numel] */
  float* master;        /* fp32 copy or NULL */
  const float* grad;    /* fp32 */
  float* m;
  float* v;
  int64_t numel;
} mdt_adam_tensor;
int mdt_adam_step_multi(void* stream, int dtype, int n_tensors, const mdt_adam_tensor* table_dev,
                        const int64_t* chunk_first_dev, int64_t total_chunks, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, const float* grad_scale);

/* ------------------------------------------------------------------ packer (host, C++)
 * Native replacement of preprocess_item + collator (data/pyg_datasets/pre_processing.py:18-69,
 * data/collator.py:69-179): integer tensors are bit-exact with the reference.
 * All outputs are caller-allocated (pinned) host buffers sized for B x nmax.
 */
int mdt_pack_structure(int B, const int64_t* n_nodes, const int64_t* const* parents, int nmax,
                       int spatial_pos_max, float* attn_bias /*[B,nmax+1,nmax+1]*/,
                       int32_t* spatial_pos /*[B,nmax,nmax]*/, int64_t* in_degree /*[B,nmax]*/);
/* Same, with the (hops up, hops down) pairs of a tree given explicitly: updown[b] = i64[n_b, n_b, 2] or NULL (derive
 * them from parents[b]).  The reference stores this matrix per graph (``distance_matrix``,
 * experiments/hateful_discussions/datasets/hateful_discussions.py:148-165) and feeds it to preprocess_item; for a
 * well-formed tree it is a function of the parent array, for discussions with repeated comment ids it is not. */
int mdt_pack_structure_ud(int B, const int64_t* n_nodes, const int64_t* const* parents, const int64_t* const* updown,
                          int nmax, int spatial_pos_max, float* attn_bias, int32_t* spatial_pos, int64_t* in_degree);

/* ------------------------------------------------------------------ image front end
 * Replaces the ViTImageProcessor call of the reference's dataset builder
 * (mDT/experiments/hateful_discussions/datasets/hateful_discussions.py:47-49,168-184): PIL bilinear (antialiased) resize to
 * out_size x out_size, x * rescale (double, rounded once to float), (x - mean) / std (float) — for a batch of decoded RGB images
 * of different sizes, on the device, byte-exact against PIL's resampler.
 *   mdt_resize_plan_ksize / mdt_resize_plan (HOST): the taps of one axis — bounds[2 * out_size] = (first source sample,
 *     tap count) per output sample, coeffs[out_size * ksize] 22-bit fixed-point weights (Pillow Resample.c precompute_coeffs +
 *     normalize_coeffs_8bpc restated).
 *   mdt_image_norm_lut (HOST): lut[3 * 256] = ((float)(u * rescale) - mean[c]) / std[c].
 *   mdt_image_preprocess (DEVICE pointers): pixels = the images' HWC uint8 bytes back to back; desc[n][8] (int64) =
 *     {pixel byte offset, tmp byte offset, H, W, horizontal plan offset, horizontal ksize, vertical plan offset, vertical ksize};
 *     plan (int32) = at each plan offset the bounds followed by the coeffs of that axis; tmp = scratch of sum_i H_i * out_size * 3
 *     bytes; out = [n, 3, out_size, out_size] in out_dtype (may be NULL), out_u8 = [n, out_size, out_size, 3] resized bytes (may
 *     be NULL); max_h = the largest H. */
int mdt_resize_plan_ksize(int in_size, int out_size);
int mdt_resize_plan(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int ksize);
int mdt_image_norm_lut(double rescale, const float* mean3, const float* std3, float* lut);
int mdt_image_preprocess(void* stream, int n_images, int max_h, const uint8_t* pixels, const int64_t* desc, const int32_t* plan,
                         uint8_t* tmp, const float* lut, int out_dtype, void* out, uint8_t* out_u8, int out_size);

#ifdef __cplusplus
}
#endif
#endif /* MDT_HIP_H */
