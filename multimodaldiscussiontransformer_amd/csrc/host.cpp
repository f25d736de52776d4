// Host-side pieces of the C ABI: error plumbing and the native packer (tree structure
// tensors), compiled with hipcc as plain C++.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "common.hpp"

namespace mdt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MDT_ERR_LAUNCH;
  }
  return MDT_OK;
}

static Switches g_sw;
static bool g_sw_valid = false;

static void read_switches() {
  auto flag = [](const char* n) { return getenv(n) != nullptr; };
  auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
  auto str = [](const char* n, char* dst, size_t cap) {
    const char* v = getenv(n);
    dst[0] = 0;
    if (v) { strncpy(dst, v, cap - 1); dst[cap - 1] = 0; }
  };
  g_sw.gemm_pp_dist = num("MDT_GEMM_PP_DIST", 4);
  g_sw.gemm_persist = num("MDT_GEMM_PERSIST", 1);
  g_sw.gemm_dynamic = num("MDT_GEMM_DYNAMIC", 0) != 0;
  g_sw.gemm_group = num("MDT_GEMM_GROUP", -1);
  g_sw.gemm_stamp = flag("MDT_GEMM_STAMP");
  g_sw.gemm_no_spec = flag("MDT_GEMM_NO_SPEC");
  g_sw.attn_exact_delta = num("MDT_ATTN_EXACT_DELTA", 1) != 0;
  g_sw.gemm_diag = num("MDT_GEMM_DIAG", 0);
  str("MDT_GEMM_TILE", g_sw.gemm_tile, sizeof(g_sw.gemm_tile));
  g_sw.gemm_no_pp = flag("MDT_GEMM_NO_PP");
  g_sw.gemm_w4 = num("MDT_GEMM_W4", 2);
  g_sw.gemm_f8w = num("MDT_GEMM_F8W", 1);
  g_sw.attn_v1 = flag("MDT_ATTN_V1");
  str("MDT_ATTN_BWD", g_sw.attn_bwd, sizeof(g_sw.attn_bwd));
  g_sw.attn_no_occ4 = flag("MDT_ATTN_NO_OCC4");
  g_sw.attn_no_w8 = flag("MDT_ATTN_NO_W8");
  g_sw.attn_onepass = num("MDT_ATTN_ONEPASS", -1);
  g_sw.ln_generic = flag("MDT_LN_GENERIC");
  g_sw.ln_bwd_wgs = num("MDT_LN_BWD_WGS", 2048);
  g_sw_valid = true;
}

const Switches& switches() {
  if (!g_sw_valid) read_switches();
  return g_sw;
}

}  // namespace mdt

extern "C" void mdt_reload_env(void) { mdt::read_switches(); }
extern "C" int mdt_abi_version(void) { return MDT_ABI_VERSION; }
extern "C" const char* mdt_last_error_string(void) { return mdt::g_err; }

// --------------------------------------------------------------------------- packer
// (up, down) hops through the lowest common ancestor → 21-bucket spatial index
// (ascending Cantor value of the sorted pair, anything with a component > 5 shares the
// (5,5) bucket), tree distance, undirected degree — data/pyg_datasets/pre_processing.py:
// 18-69 and experiments/hateful_discussions/datasets/hateful_discussions.py:242-264 —
// then the collator's shifts, padding and distance clipping (data/collator.py:38-66,
// 122-126, 156-164), written straight into the padded batch tensors.
namespace {

struct SpatialTable {
  int t[6][6];
  SpatialTable() {
    std::vector<int> vals;
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) vals.push_back((a + b) * (a + b + 1) / 2 + a);
    std::sort(vals.begin(), vals.end());
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        const int lo = std::min(a, b), hi = std::max(a, b);
        const int c = (lo + hi) * (lo + hi + 1) / 2 + lo;
        t[a][b] = (int)(std::lower_bound(vals.begin(), vals.end(), c) - vals.begin());
      }
  }
};
const SpatialTable kTable;

}  // namespace

extern "C" int mdt_pack_structure_ud(int B, const int64_t* n_nodes, const int64_t* const* parents,
                                     const int64_t* const* updown, int nmax, int spatial_pos_max, float* attn_bias,
                                     int32_t* spatial_pos, int64_t* in_degree) {
  MDT_CHECK_ARG(B >= 0 && nmax >= 1 && n_nodes && parents && attn_bias && spatial_pos && in_degree,
                "pack_structure: bad arguments");
  const int T = nmax + 1;
  const float ninf = -std::numeric_limits<float>::infinity();
  std::vector<int> depth, anc_pos;
  for (int b = 0; b < B; ++b) {
    const int n = (int)n_nodes[b];
    const int64_t* par = parents[b];
    MDT_CHECK_ARG(n >= 1 && n <= nmax, "pack_structure: tree %d has %d nodes (nmax %d)", b, n, nmax);
    float* ab = attn_bias + (int64_t)b * T * T;
    int32_t* sp = spatial_pos + (int64_t)b * nmax * nmax;
    int64_t* deg = in_degree + (int64_t)b * nmax;
    depth.assign(n, 0);
    for (int i = 0; i < n; ++i) {
      MDT_CHECK_ARG(par[i] < i && par[i] >= -1 && (i == 0) == (par[i] < 0),
                    "pack_structure: tree %d node %d: parents must precede children (root first)", b, i);
      depth[i] = par[i] < 0 ? 0 : depth[par[i]] + 1;
    }
    // padding: columns beyond the tree are -inf on every row, rows beyond it are 0 over real columns
    for (int i = 0; i < T; ++i)
      for (int j = 0; j < T; ++j) ab[(int64_t)i * T + j] = (j <= n) ? 0.f : ninf;
    std::fill(sp, sp + (int64_t)nmax * nmax, 0);
    std::fill(deg, deg + nmax, (int64_t)0);
    for (int i = 0; i < n; ++i) {
      deg[i] += 1;  // the collator's +1 shift
      if (par[i] >= 0) { deg[i] += 1; deg[par[i]] += 1; }
    }
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        int up = 0, down = 0;
        if (updown && updown[b]) {   // the dataset's own distance_matrix (hateful_discussions.py:148-165), taken as given
          const int64_t* e = updown[b] + ((int64_t)i * n + j) * 2;
          MDT_CHECK_ARG(e[0] >= 0 && e[1] >= 0 && e[0] < (1 << 20) && e[1] < (1 << 20), "pack_structure: tree %d: negative hop count", b);
          up = (int)e[0];
          down = (int)e[1];
        } else {
          int a = i, c = j;  // walk both to their lowest common ancestor
          while (a != c) {
            if (depth[a] >= depth[c]) { a = (int)par[a]; ++up; }
            else { c = (int)par[c]; ++down; }
          }
        }
        const int bucket = (up <= 5 && down <= 5) ? kTable.t[up][down] : kTable.t[5][5];
        sp[(int64_t)i * nmax + j] = bucket + 1;
        if (up + down >= spatial_pos_max) ab[(int64_t)(i + 1) * T + (j + 1)] = ninf;
      }
    }
  }
  return MDT_OK;
}

extern "C" int mdt_pack_structure(int B, const int64_t* n_nodes, const int64_t* const* parents, int nmax,
                                  int spatial_pos_max, float* attn_bias, int32_t* spatial_pos, int64_t* in_degree) {
  return mdt_pack_structure_ud(B, n_nodes, parents, nullptr, nmax, spatial_pos_max, attn_bias, spatial_pos, in_degree);
}

// --------------------------------------------------------------------------- image front end (host part)
// The reference hands every image to HuggingFace's ViT image processor (experiments/hateful_discussions/datasets/
// hateful_discussions.py:47-49, 168-184): PIL bilinear resize to 224 x 224 — for a reduction that is a triangle filter whose
// support grows with the scale (antialiasing), run as a horizontal then a vertical pass in 22-bit fixed point with the
// intermediate image rounded to 8 bits — then x * (1 / 255) in double rounded to float, (x - mean) / std in float.
// mdt_resize_plan restates PIL's coefficient computation for one axis (Pillow src/libImaging/Resample.c: precompute_coeffs
// + normalize_coeffs_8bpc, bilinear filter) so that the device passes of csrc/image.hip reproduce the resized bytes exactly;
// mdt_image_norm_lut the 3 x 256 possible results of the rescale / normalise arithmetic.
extern "C" int mdt_resize_plan_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;        // bilinear: filter support 1
  return (int)ceil(support) * 2 + 1;
}

extern "C" int mdt_resize_plan(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int ksize) {
  MDT_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && coeffs, "mdt_resize_plan: bad arguments");
  MDT_CHECK_ARG(ksize == mdt_resize_plan_ksize(in_size, out_size), "mdt_resize_plan: ksize %d, expected %d", ksize,
                mdt_resize_plan_ksize(in_size, out_size));
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  std::vector<double> k((size_t)ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double w = t < 1.0 ? 1.0 - t : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    int32_t* kk = coeffs + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[x] : 0.0;
      kk[x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << 22)) : (int32_t)(0.5 + v * (double)(1 << 22));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return MDT_OK;
}

extern "C" int mdt_image_norm_lut(double rescale, const float* mean3, const float* std3, float* lut) {
  MDT_CHECK_ARG(mean3 && std3 && lut, "mdt_image_norm_lut: null pointer");
  for (int c = 0; c < 3; ++c)
    for (int u = 0; u < 256; ++u) {
      const float x = (float)((double)u * rescale);        // rescale in double, ONE rounding to float
      lut[c * 256 + u] = (x - mean3[c]) / std3[c];         // normalise in float
    }
  return MDT_OK;
}
