// ViT patch embedding as ONE launch (the reference runs HF ViTEmbeddings: Conv2d(3, D, k = s = 16) + [CLS] +
// position add, modules/multigraphormer_graph_encoder.py:332-335).
//
// A stride-equals-kernel convolution is a GEMM whose A operand is a pure re-index of the image: row (image, py, px), column
// (c, dy, dx).  The two-launch route (mdt_vit_patchify + mdt_gemm + mdt_vit_assemble) writes that re-index out as a bf16 matrix
// (154 MB at 512 images), reads it back, writes the patch rows (154 MB) and reads them back to add the position table.  Here the
// GEMM's A loader gathers from pixel_values itself: a 64-k step of a row is FOUR runs of 16 contiguous fp32 (dy .. dy+3 at one
// channel), adjacent rows (px, px+1) continue the same image line, so a wave's 64 lanes read 2 KiB contiguous per dy.  The pixels
// pass through registers (fp32 -> bf16, round to nearest even — what mdt_vit_patchify stores) into the same XOR-swizzled LDS image
// the LDS-DMA route fills (gemm_tiles.hpp load_frag<false, 128>), one k-step ahead of the MFMAs; the weight tile comes by LDS-DMA.
// The epilogue adds bias and position row and stores straight into the token matrix ([CLS] rows written by the tile that owns an
// image's first patch): one rounding instead of two.  HBM-side: 308 MB of pixels read once, 155 MB of tokens written once.
#include "common.hpp"
#include "gemm_tiles.hpp"

namespace mdt {

struct PatchParams {
  const float* img; const bf16_t* w; const bf16_t* bias; const bf16_t* cls; const bf16_t* pos; bf16_t* tokens;
  int64_t ldw, ldt, seq_stride, off, M;
  int C, HW, gw, np, D, K, tiles_m, tiles_n;
};

template <int P>
__global__ __launch_bounds__(256) void vit_patch_embed_kernel(PatchParams p) {
  static_assert(P == 16, "a 64-k step = four image lines of one channel");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (A 16K + B 16K)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * T_BM, n0 = (int64_t)tn * T_BN;
  const int nk = p.K / T_BK;

  // A: thread t owns row t >> 1 of the tile, 8-k half t & 1 of each of the step's four image lines
  const int arow = tid >> 1, half = tid & 1;
  int64_t grow = m0 + arow;
  if (grow >= p.M) grow = p.M - 1;                 // rows past M: a duplicate of the last row, never stored
  const int64_t ai = grow / p.np;
  const int aj = (int)(grow - ai * p.np);
  const int apy = aj / p.gw, apx = aj - apy * p.gw;
  const float* abase = p.img + ((ai * p.C * p.HW + apy * P) * (int64_t)p.HW + apx * P + half * 8);
  char* const a_dst = smem + arow * 128;
  const int a_swz = swz_kc(arow);

  const int64_t ldw_b = p.ldw * 2;
  const char* b_base = (const char*)p.w + n0 * ldw_b;
  const int64_t b_bytes = ((int64_t)p.D - n0) * ldw_b;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (unsigned)(b_bytes > 0xFFFFFFF0ll ? 0xFFFFFFF0ll : b_bytes), 0x00020000);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 px[4][2];
  auto a_request = [&](int kt) {
    const int k0 = kt * T_BK;
    const int c = k0 / (P * P), dy0 = (k0 % (P * P)) / P;
    const float* s = abase + ((int64_t)c * p.HW + dy0) * p.HW;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      px[dy][0] = *(const f32x4*)(s + (int64_t)dy * p.HW);
      px[dy][1] = *(const f32x4*)(s + (int64_t)dy * p.HW + 4);
    }
  };
  auto a_park = [&](char* stage) {
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = (bf16_t)px[dy][0][e]; v[4 + e] = (bf16_t)px[dy][1][e]; }
      *(bf16x8*)(stage + (a_dst - smem) + (((dy * 2 + half) ^ a_swz) * 16)) = v;
    }
  };

  a_request(0);
  stage_tile<false, 128, 4>(rsB, ldw_b, 0, 0, smem + T_TILE_BYTES, wave, lane);
  a_park(smem);
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * T_TILE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * T_TILE_BYTES;
    // own LDS-DMA pieces of step kt landed and own parked pixels are written (vmcnt(0) / lgkmcnt(0) in front of the barrier);
    // behind it: everybody's are, and nobody still reads the stage step kt + 1 goes to
    __syncthreads();
    if (kt + 1 < nk) {
      a_request(kt + 1);
      stage_tile<false, 128, 4>(rsB, ldw_b, (int64_t)(kt + 1) * T_BK, 0, nxt + T_TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = load_frag<false, 128>(cur, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag<false, 128>(cur + T_TILE_BYTES, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
    if (kt + 1 < nk) a_park(nxt);
  }

  // epilogue: the wave's 64 x 64 block through LDS into row segments of 8 columns per lane
  __syncthreads();
  float* ws = (float*)(smem + wave * 16384);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ws[(i * 16 + (lane >> 4) * 4 + r) * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int c8 = (lane & 7) * 8;
  const int64_t gc = n0 + wc * 64 + c8;
  float bias[8];
  {
    const bf16x8 b = *(const bf16x8*)(p.bias + gc);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (float)b[e];
  }
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 8 + (lane >> 3);
    const int64_t gr = m0 + wr * 64 + row;
    if (gr >= p.M) continue;
    const int64_t i = gr / p.np;
    const int j = (int)(gr - i * p.np);
    const f32x4 lo = *(const f32x4*)(ws + row * 64 + c8);
    const f32x4 hi = *(const f32x4*)(ws + row * 64 + c8 + 4);
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const bf16x8 pe = *(const bf16x8*)(p.pos + (int64_t)(j + 1) * p.D + gc);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + bias[e] + (float)pe[e]);
    bf16_t* t = p.tokens + (i * p.seq_stride + p.off) * p.ldt + gc;
    *(bf16x8*)(t + (int64_t)(j + 1) * p.ldt) = o;
    if (j == 0) {
      const bf16x8 c0 = *(const bf16x8*)(p.cls + gc), p0 = *(const bf16x8*)(p.pos + gc);
      bf16x8 oc;
#pragma unroll
      for (int e = 0; e < 8; ++e) oc[e] = (bf16_t)((float)c0[e] + (float)p0[e]);
      *(bf16x8*)t = oc;
    }
  }
}

}  // namespace mdt

using namespace mdt;

extern "C" int mdt_vit_patch_embed(void* stream, int I, int C, int HW, int p, const float* img, const void* w, int64_t ldw,
                                   const void* bias, const void* cls, const void* pos, int D, void* tokens, int64_t ldt,
                                   int64_t seq_stride, int64_t off) {
  if (I == 0) return MDT_OK;
  MDT_CHECK_ARG(img && w && bias && cls && pos && tokens && I > 0 && C > 0 && p > 0 && HW % p == 0, "vit_patch_embed: bad arguments");
  const int gw = HW / p, np = gw * gw, K = C * p * p;
  if (p != 16 || D % 128 != 0 || HW % 4 != 0) MDT_UNSUPPORTED("vit_patch_embed: patch %d, D %d (16 x 16 patches, D a multiple of 128)", p, D);
  MDT_CHECK_ARG(ldw >= K && ldw % 8 == 0 && ldt >= D && ldt % 8 == 0 && seq_stride >= off + np + 1 && off >= 0,
                "vit_patch_embed: ldw %lld, ldt %lld, seq_stride %lld, off %lld", (long long)ldw, (long long)ldt, (long long)seq_stride, (long long)off);
  MDT_CHECK_ARG((((uintptr_t)img | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)cls | (uintptr_t)pos | (uintptr_t)tokens) & 15) == 0,
                "vit_patch_embed: pointers must be 16-byte aligned");
  PatchParams q;
  q.img = img; q.w = (const bf16_t*)w; q.bias = (const bf16_t*)bias; q.cls = (const bf16_t*)cls; q.pos = (const bf16_t*)pos;
  q.tokens = (bf16_t*)tokens; q.ldw = ldw; q.ldt = ldt; q.seq_stride = seq_stride; q.off = off; q.M = (int64_t)I * np;
  q.C = C; q.HW = HW; q.gw = gw; q.np = np; q.D = D; q.K = K;
  q.tiles_m = (int)((q.M + T_BM - 1) / T_BM); q.tiles_n = D / T_BN;
  MDT_CHECK_ARG((int64_t)q.tiles_m * q.tiles_n < (1ll << 31), "vit_patch_embed: too many tiles");
  hipLaunchKernelGGL((vit_patch_embed_kernel<16>), dim3((unsigned)(q.tiles_m * q.tiles_n)), 256, (size_t)(4 * T_TILE_BYTES), (hipStream_t)stream, q);
  return check_launch("vit_patch_embed");
}
