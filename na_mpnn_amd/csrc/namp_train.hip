// Training (backward) entry points of libnamp_hip.so — see include/namp.h "training" section and namp_train.h.
#include "../../include/namp.h"
#include "namp_train.h"
#include "namp_train_dw.h"

#include <cstdarg>
#include <cstdio>
#include <mutex>

int namp_internal_fail(int code, const char* msg);      // namp.hip

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return namp_internal_fail(code, buf);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define REQUIRE_PTR(p)                                                                         \
  do {                                                                                         \
    if ((p) == nullptr) return fail(NAMP_EINVAL, "%s: null pointer argument '%s'", __func__, #p); \
    if (!aligned16(p)) return fail(NAMP_EINVAL, "%s: '%s' is not 16-byte aligned", __func__, #p); \
  } while (0)
#define REQUIRE(cond, ...)                                   \
  do {                                                       \
    if (!(cond)) return fail(NAMP_EINVAL, __VA_ARGS__);      \
  } while (0)
#define CHECK_LAUNCH()                                                                          \
  do {                                                                                          \
    hipError_t e_ = hipGetLastError();                                                          \
    if (e_ != hipSuccess) return fail(NAMP_ELAUNCH, "%s: %s", __func__, hipGetErrorString(e_)); \
  } while (0)

std::once_flag g_once;
hipError_t g_attr_err = hipSuccess;

int ensure_attributes() {
  std::call_once(g_once, [] {
    auto set = [](const void* f) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NAMP_IMG_BYTES);
      if (e != hipSuccess) g_attr_err = e;
    };
#define NAMP_SET3(M, ...) set((const void*)(edge_chain_bwd_kernel<M, 0 __VA_ARGS__>)); set((const void*)(edge_chain_bwd_kernel<M, 1 __VA_ARGS__>)); \
                          set((const void*)(edge_chain_bwd_kernel<M, 2 __VA_ARGS__>))
    NAMP_SET3(BWD_ENC_MSG); NAMP_SET3(BWD_DEC_MSG); NAMP_SET3(BWD_ROWS); NAMP_SET3(BWD_EDGE_LN);
    set((const void*)(embed_ln_bwd_kernel<1>)); set((const void*)(embed_ln_bwd_kernel<2>));
    auto set_dw = [](const void* f) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
      if (e != hipSuccess) g_attr_err = e;
    };
#define NAMP_SET_DWR(M, P) set_dw((const void*)(edge_bwd_dw_kernel<M, P, false, 1>)); set_dw((const void*)(edge_bwd_dw_kernel<M, P, true, 1>)); \
                           set_dw((const void*)(edge_bwd_dw_kernel<M, P, false, 2>)); set_dw((const void*)(edge_bwd_dw_kernel<M, P, true, 2>))
    NAMP_SET_DWR(BWD_ENC_MSG, 1); NAMP_SET_DWR(BWD_DEC_MSG, 1);
    auto set_dw16 = [](const void* f) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DW16_LDS);
      if (e != hipSuccess) g_attr_err = e;
    };
#define NAMP_SET_DW16(M) set_dw16((const void*)(edge_bwd_dw16_kernel<M, false, 1>)); set_dw16((const void*)(edge_bwd_dw16_kernel<M, true, 1>)); \
                         set_dw16((const void*)(edge_bwd_dw16_kernel<M, false, 2>)); set_dw16((const void*)(edge_bwd_dw16_kernel<M, true, 2>))
    NAMP_SET_DW16(BWD_ENC_MSG); NAMP_SET_DW16(BWD_DEC_MSG);
  });
  if (g_attr_err != hipSuccess)
    return fail(NAMP_ELAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(g_attr_err));
  return NAMP_OK;
}

}  // namespace

extern "C" {

int namp_train_edge_bwd(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                        const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                        const float* W2_img, const float* W3t_img, const float* W2t_img, const float* W1t_img,
                        const float* b2, const float* g_out, float* A1, float* A2, float* G1, float* G2, float* G3,
                        float* g_hE, const float* g_hE_in, float* g_Pa, float* g_Pj0, float* g_Pj1, float* S3, float* w3, int x3, int B,
                        int N, int K, void* stream) {
  REQUIRE(mode >= 0 && mode <= 2, "namp_train_edge_bwd: mode=%d must be 0 (enc message), 1 (dec message) or 2 (enc edge)", mode);
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pj0); REQUIRE_PTR(W1_img); REQUIRE_PTR(W2_img);
  REQUIRE_PTR(W2t_img); REQUIRE_PTR(W1t_img); REQUIRE_PTR(b2); REQUIRE_PTR(g_out);
  REQUIRE_PTR(A1); REQUIRE_PTR(G1); REQUIRE_PTR(G2); REQUIRE_PTR(g_hE);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_edge_bwd: null E_idx");
  (void)S3; (void)w3;                                        // message modes: layer 3 is a residue-level product (see namp.h)
  if (mode == 2) { REQUIRE_PTR(W3t_img); REQUIRE_PTR(A2); }
  if (mode == 1) { REQUIRE_PTR(Pj1); REQUIRE(rank != nullptr, "namp_train_edge_bwd: decoder message needs rank"); }
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= NAMP_MAX_K, "namp_train_edge_bwd: bad dims B=%d N=%d K=%d", B, N, K);
  int rc = ensure_attributes();
  if (rc) return rc;
  EdgeBwdArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.rank = rank; a.Pa = Pa; a.Pj0 = Pj0; a.Pj1 = Pj1;
  a.W1_img = W1_img; a.W2_img = W2_img; a.W3t_img = W3t_img; a.W2t_img = W2t_img; a.W1t_img = W1t_img; a.b2 = b2;
  if (mode == 2) a.g_rows = g_out; else a.g_node = g_out;
  REQUIRE(mode != 1 || (g_Pj0 == nullptr) == (g_Pj1 == nullptr), "namp_train_edge_bwd: decoder message needs g_Pj1 with g_Pj0");
  a.g_Pa = g_Pa; a.g_Pj0 = g_Pj0; a.g_Pj1 = g_Pj1;
  a.A1 = A1; a.A2 = A2; a.G1 = G1; a.G2 = G2; a.G3 = G3; a.g_hE = g_hE; a.S3 = S3; a.w3 = w3;
  a.G = B * N; a.N = N; a.K = K; a.E = (long)a.G * K;
  a.acc_hE = (x3 & 4) ? 1 : 0;                               // bit 2 of the precision argument: g_hE = g_hE_in + this launch's dL/dh_E
  a.g_hE_in = g_hE_in;
  REQUIRE(!a.acc_hE || g_hE_in != nullptr, "namp_train_edge_bwd: the accumulate flag needs g_hE_in");
  a.gpa_tiles = (x3 & 8) ? 1 : 0;                            // bit 3: g_Pa holds per-tile sums [E/16][128] (needs K % 16 == 0)
  REQUIRE(!a.gpa_tiles || (K % 16) == 0, "namp_train_edge_bwd: per-tile g_Pa sums need K %% 16 == 0 (K=%d)", K);
  x3 &= 3;
  const int grid = (int)((a.E + 127) / 128);
  hipStream_t s = (hipStream_t)stream;
  // x3: precision code — 0 exact fp32 MFMA, 1 split-bf16 products, 2 plain bf16 products (mixed precision)
#define NAMP_LAUNCH_BWD(M)                                                                                              \
  do {                                                                                                                    \
    if (x3 == 2) hipLaunchKernelGGL((edge_chain_bwd_kernel<M, 2>), dim3(grid), dim3(512), 2 * NAMP_IMG_BYTES, s, a);      \
    else if (x3) hipLaunchKernelGGL((edge_chain_bwd_kernel<M, 1>), dim3(grid), dim3(512), 2 * NAMP_IMG_BYTES, s, a);      \
    else hipLaunchKernelGGL((edge_chain_bwd_kernel<M, 0>), dim3(grid), dim3(512), 2 * NAMP_IMG_BYTES, s, a);              \
  } while (0)
  if (mode == 0) NAMP_LAUNCH_BWD(BWD_ENC_MSG);
  else if (mode == 1) NAMP_LAUNCH_BWD(BWD_DEC_MSG);
  else NAMP_LAUNCH_BWD(BWD_ROWS);
  CHECK_LAUNCH();
  return NAMP_OK;
}

// Workgroups of the persistent message backward (namp_train_edge_bwd_dw): one per CU, never more than there are 128-row rounds.
static int dw_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    return v;
  }();
  return n;
}

long namp_train_edge_bwd_dw_rows(int B, int N, int K) {
  if (B < 1 || N < 1 || K < 1) return 0;
  return (((long)B * N * K + DW_ROWS - 1) / DW_ROWS) * DW_ROWS;
}

int namp_train_edge_bwd_dw_groups(int B, int N, int K) {
  if (B < 1 || N < 1 || K < 1) return 0;
  const long rounds = ((long)B * N * K + DW_ROWS - 1) / DW_ROWS;
  return (int)(rounds < dw_cus() ? rounds : dw_cus());
}

int namp_train_edge_bwd_dw(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                           const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                           const float* W2_img, const float* W2t_img, const float* W1t_img, const float* b2, const float* g_out,
                           float* G1, float* g_hE, const float* g_hE_in, float* g_Pa, float* dW_part, float* db_part, int x3, int B,
                           int N, int K, void* stream) {
  REQUIRE(mode == 0 || mode == 1, "namp_train_edge_bwd_dw: mode=%d must be 0 (enc message) or 1 (dec message)", mode);
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pj0); REQUIRE_PTR(W1_img); REQUIRE_PTR(W2_img);
  REQUIRE_PTR(W2t_img); REQUIRE_PTR(W1t_img); REQUIRE_PTR(b2); REQUIRE_PTR(g_out);
  REQUIRE_PTR(G1); REQUIRE_PTR(g_hE); REQUIRE_PTR(dW_part); REQUIRE_PTR(db_part);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_edge_bwd_dw: null E_idx");
  if (mode == 1) { REQUIRE_PTR(Pj1); REQUIRE(rank != nullptr, "namp_train_edge_bwd_dw: decoder message needs rank"); }
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= NAMP_MAX_K, "namp_train_edge_bwd_dw: bad dims B=%d N=%d K=%d", B, N, K);
  int rc = ensure_attributes();
  if (rc) return rc;
  EdgeBwdDwArgs aa = {};
  EdgeBwdArgs& a = aa.b;
  a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.rank = rank; a.Pa = Pa; a.Pj0 = Pj0; a.Pj1 = Pj1;
  a.W1_img = W1_img; a.W2_img = W2_img; a.W2t_img = W2t_img; a.W1t_img = W1t_img; a.b2 = b2; a.g_node = g_out;
  a.G1 = G1; a.g_hE = g_hE; a.g_Pa = g_Pa;
  a.G = B * N; a.N = N; a.K = K; a.E = (long)a.G * K;
  a.acc_hE = (x3 & 4) ? 1 : 0;
  a.g_hE_in = g_hE_in;
#ifdef DW_EXP_STAMPS
  { const char* e = getenv("NAMP_DW_STAMPS"); a.S3 = e ? (float*)strtoull(e, nullptr, 0) : nullptr; }   // device address of the stamp buffer (tools/dw_time.py)
#endif
  REQUIRE(!a.acc_hE || g_hE_in != nullptr, "namp_train_edge_bwd_dw: the accumulate flag needs g_hE_in");
  a.gpa_tiles = (x3 & 8) ? 1 : 0;
  REQUIRE(!a.gpa_tiles || (K % 16) == 0, "namp_train_edge_bwd_dw: per-tile g_Pa sums need K %% 16 == 0 (K=%d)", K);
  x3 &= 3;
  REQUIRE(x3 == 1 || x3 == 2, "namp_train_edge_bwd_dw: precision code %d (1 = split-bf16, 2 = bf16 products)", x3);
  aa.dW_part = dW_part; aa.db_part = db_part;
  aa.nrounds = (a.E + DW_ROWS - 1) / DW_ROWS;
  const int grid = namp_train_edge_bwd_dw_groups(B, N, K);
  hipStream_t s = (hipStream_t)stream;
  // bf16 products: the weight-stationary kernel (all four images resident); split-bf16: the LDS-DMA ring of half images
  const bool acc_ = a.acc_hE != 0;
  const int gpa_ = a.gpa_tiles ? 1 : 2;
  REQUIRE(g_Pa != nullptr, "namp_train_edge_bwd_dw: null g_Pa");
  REQUIRE(x3 != 2 || mode != 0 || mask != nullptr || mask_attend != nullptr, "namp_train_edge_bwd_dw: the bf16 encoder launch needs mask or mask_attend");
#define NAMP_LAUNCH_DWR(M, P)                                                                                                       \
  do {                                                                                                                                \
    if (acc_ && gpa_ == 1) hipLaunchKernelGGL((edge_bwd_dw_kernel<M, P, true, 1>), dim3(grid), dim3(64 * DW_WAVES), DW_LDS, s, aa);    \
    else if (acc_) hipLaunchKernelGGL((edge_bwd_dw_kernel<M, P, true, 2>), dim3(grid), dim3(64 * DW_WAVES), DW_LDS, s, aa);            \
    else if (gpa_ == 1) hipLaunchKernelGGL((edge_bwd_dw_kernel<M, P, false, 1>), dim3(grid), dim3(64 * DW_WAVES), DW_LDS, s, aa);      \
    else hipLaunchKernelGGL((edge_bwd_dw_kernel<M, P, false, 2>), dim3(grid), dim3(64 * DW_WAVES), DW_LDS, s, aa);                     \
  } while (0)
#define NAMP_LAUNCH_DW16(M)                                                                                                          \
  do {                                                                                                                                \
    if (acc_ && gpa_ == 1) hipLaunchKernelGGL((edge_bwd_dw16_kernel<M, true, 1>), dim3(grid), dim3(64 * DW_WAVES), DW16_LDS, s, aa);   \
    else if (acc_) hipLaunchKernelGGL((edge_bwd_dw16_kernel<M, true, 2>), dim3(grid), dim3(64 * DW_WAVES), DW16_LDS, s, aa);           \
    else if (gpa_ == 1) hipLaunchKernelGGL((edge_bwd_dw16_kernel<M, false, 1>), dim3(grid), dim3(64 * DW_WAVES), DW16_LDS, s, aa);     \
    else hipLaunchKernelGGL((edge_bwd_dw16_kernel<M, false, 2>), dim3(grid), dim3(64 * DW_WAVES), DW16_LDS, s, aa);                    \
  } while (0)
  if (mode == 0) {
    if (x3 == 2) NAMP_LAUNCH_DW16(BWD_ENC_MSG);
    else NAMP_LAUNCH_DWR(BWD_ENC_MSG, 1);
  } else {
    if (x3 == 2) NAMP_LAUNCH_DW16(BWD_DEC_MSG);
    else NAMP_LAUNCH_DWR(BWD_DEC_MSG, 1);
  }
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_edge_update_bwd_groups(int B, int N, int K) {
  if (B < 1 || N < 1 || K < 1) return 0;
  return (int)(((long)B * N * K + 127) / 128);
}

int namp_train_edge_update_bwd(const float* h_E, const int32_t* E_idx, const float* Pa, const float* Pc, const float* W1_img,
                               const float* W2_img, const float* W3_img, const float* W3t_img, const float* W2t_img,
                               const float* W1t_img, const float* b2, const float* b3, const float* ln_g, float drop_p,
                               uint32_t drop_seed, long drop_row0, const float* g_out, float* A1, float* A2, float* G1, float* G2, float* G3,
                               float* g_hE, float* g_Pa, float* g_Pc, float* dgb_part, int x3, int B, int N, int K, void* stream) {
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(W1_img); REQUIRE_PTR(W2_img); REQUIRE_PTR(W3_img);
  REQUIRE_PTR(W3t_img); REQUIRE_PTR(W2t_img); REQUIRE_PTR(W1t_img); REQUIRE_PTR(b2); REQUIRE_PTR(b3); REQUIRE_PTR(ln_g);
  REQUIRE_PTR(g_out); REQUIRE_PTR(A1); REQUIRE_PTR(A2); REQUIRE_PTR(G1); REQUIRE_PTR(G2); REQUIRE_PTR(G3); REQUIRE_PTR(g_hE);
  REQUIRE_PTR(dgb_part);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_edge_update_bwd: null E_idx");
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "namp_train_edge_update_bwd: drop_p=%g must be in [0,1)", (double)drop_p);
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= NAMP_MAX_K, "namp_train_edge_update_bwd: bad dims B=%d N=%d K=%d", B, N, K);
  REQUIRE(drop_row0 >= 0, "namp_train_edge_update_bwd: drop_row0=%ld", drop_row0);
  int rc = ensure_attributes();
  if (rc) return rc;
  EdgeBwdArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.Pa = Pa; a.Pj0 = Pc;
  a.W1_img = W1_img; a.W2_img = W2_img; a.W3_img = W3_img; a.W3t_img = W3t_img; a.W2t_img = W2t_img; a.W1t_img = W1t_img;
  a.b2 = b2; a.b3 = b3; a.ln_g = ln_g; a.g_rows = g_out;
  if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_seed = drop_seed; a.drop_scale = 1.0f / (1.0f - drop_p); }
  a.drop_row0 = drop_row0;
  a.A1 = A1; a.A2 = A2; a.G1 = G1; a.G2 = G2; a.G3 = G3; a.g_hE = g_hE; a.g_Pa = g_Pa; a.g_Pj0 = g_Pc; a.dgb_part = dgb_part;
  a.G = B * N; a.N = N; a.K = K; a.E = (long)a.G * K;
  a.gpa_tiles = (x3 & 8) ? 1 : 0;
  REQUIRE(!a.gpa_tiles || (K % 16) == 0, "namp_train_edge_update_bwd: per-tile g_Pa sums need K %% 16 == 0 (K=%d)", K);
  x3 &= 3;
  {
    const int grid = namp_train_edge_update_bwd_groups(B, N, K);
    hipStream_t s = (hipStream_t)stream;
    NAMP_LAUNCH_BWD(BWD_EDGE_LN);
  }
  CHECK_LAUNCH();
  return NAMP_OK;
}

static int scatter_rows_impl(const void* G1, bool bf16_rows, const int32_t* rev_edge, const int32_t* rev_off, const uint8_t* sel,
                             float* out0, float* out1, int G, void* stream) {
  REQUIRE_PTR(G1); REQUIRE_PTR(out0);
  if (!rev_edge || !rev_off) return fail(NAMP_EINVAL, "namp_train_scatter_rows: null reverse adjacency");
  REQUIRE((sel == nullptr) == (out1 == nullptr), "namp_train_scatter_rows: sel and out1 go together");
  REQUIRE(G >= 1, "namp_train_scatter_rows: G=%d", G);
  if (bf16_rows) hipLaunchKernelGGL(scatter_rows_kernel<true>, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)G1,
                                    rev_edge, rev_off, sel, out0, out1, G);
  else hipLaunchKernelGGL(scatter_rows_kernel<false>, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)G1, rev_edge,
                          rev_off, sel, out0, out1, G);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_scatter_rows(const float* G1, const int32_t* rev_edge, const int32_t* rev_off, const uint8_t* sel,
                            float* out0, float* out1, int G, void* stream) {
  return scatter_rows_impl(G1, false, rev_edge, rev_off, sel, out0, out1, G, stream);
}

int namp_train_scatter_rows_bf16(const void* G1, const int32_t* rev_edge, const int32_t* rev_off, const uint8_t* sel,
                                 float* out0, float* out1, int G, void* stream) {
  return scatter_rows_impl(G1, true, rev_edge, rev_off, sel, out0, out1, G, stream);
}

int namp_train_tail_groups(int G) { return G < 1 ? 0 : (G + 16 * TAIL_T - 1) / (16 * TAIL_T); }

static int tail_attr() {
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] {
    hipError_t e = hipFuncSetAttribute((const void*)tail_train_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_train_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS);
    err = e;
  });
  if (err != hipSuccess) return fail(NAMP_ELAUNCH, "hipFuncSetAttribute(tail_train): %s", hipGetErrorString(err));
  return NAMP_OK;
}

int namp_train_tail_fwd(const float* h_V, const float* dh, const int32_t* mask, const float* ln1_g, const float* ln1_b,
                        const float* Win_ximg, const float* b_in, const float* Wout_ximg, const float* b_out, const float* ln2_g,
                        const float* ln2_b, float drop_p, uint32_t seed1, uint32_t seed2, float* out, float* x1, float* z, float* y,
                        int G, void* stream) {
  REQUIRE_PTR(h_V); REQUIRE_PTR(dh); REQUIRE_PTR(ln1_g); REQUIRE_PTR(ln1_b); REQUIRE_PTR(Win_ximg); REQUIRE_PTR(b_in);
  REQUIRE_PTR(Wout_ximg); REQUIRE_PTR(b_out); REQUIRE_PTR(ln2_g); REQUIRE_PTR(ln2_b); REQUIRE_PTR(out); REQUIRE_PTR(x1);
  REQUIRE_PTR(z); REQUIRE_PTR(y);
  REQUIRE(G >= 1, "namp_train_tail_fwd: G=%d", G);
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "namp_train_tail_fwd: drop_p=%g must be in [0,1)", (double)drop_p);
  int rc = tail_attr();
  if (rc) return rc;
  TailTrainArgs a = {};
  a.hV = h_V; a.dh = dh; a.mask = mask; a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.b_in = b_in; a.b_out = b_out; a.ln2_g = ln2_g; a.ln2_b = ln2_b;
  a.WA_ximg = Win_ximg; a.WB_ximg = Wout_ximg; a.out = out; a.x1 = x1; a.z = z; a.y = y; a.G = G;
  if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.seed1 = seed1; a.seed2 = seed2; a.drop_scale = 1.0f / (1.0f - drop_p); }
  hipLaunchKernelGGL(tail_train_fwd_kernel, dim3(namp_train_tail_groups(G)), dim3(512), TAIL_LDS, (hipStream_t)stream, a);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_tail_bwd(const float* h_V, const float* dh, const int32_t* mask, const float* ln1_g, const float* ln2_g,
                        const float* WoutT_ximg, const float* WinT_ximg, float drop_p, uint32_t seed1, uint32_t seed2,
                        const float* x1, const float* z, const float* y, const float* g_out, float* g_hV, float* g_dh, float* g_f,
                        float* g_z, float* h, float* part, int G, void* stream) {
  REQUIRE_PTR(h_V); REQUIRE_PTR(dh); REQUIRE_PTR(ln1_g); REQUIRE_PTR(ln2_g); REQUIRE_PTR(WoutT_ximg); REQUIRE_PTR(WinT_ximg);
  REQUIRE_PTR(x1); REQUIRE_PTR(z); REQUIRE_PTR(y); REQUIRE_PTR(g_out); REQUIRE_PTR(g_hV); REQUIRE_PTR(g_dh); REQUIRE_PTR(g_f);
  REQUIRE_PTR(g_z); REQUIRE_PTR(h); REQUIRE_PTR(part);
  REQUIRE(G >= 1, "namp_train_tail_bwd: G=%d", G);
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "namp_train_tail_bwd: drop_p=%g must be in [0,1)", (double)drop_p);
  int rc = tail_attr();
  if (rc) return rc;
  TailTrainArgs a = {};
  a.hV = h_V; a.dh = dh; a.mask = mask; a.ln1_g = ln1_g; a.ln2_g = ln2_g; a.WA_ximg = WoutT_ximg; a.WB_ximg = WinT_ximg;
  a.x1 = (float*)x1; a.z = (float*)z; a.y = (float*)y; a.g_out = g_out; a.g_hV = g_hV; a.g_dh = g_dh; a.g_f = g_f; a.g_z = g_z; a.h = h;
  a.part = part; a.G = G;
  if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.seed1 = seed1; a.seed2 = seed2; a.drop_scale = 1.0f / (1.0f - drop_p); }
  hipLaunchKernelGGL(tail_train_bwd_kernel, dim3(namp_train_tail_groups(G)), dim3(512), TAIL_LDS, (hipStream_t)stream, a);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_ln_rows_groups(long rows) {
  if (rows <= 0) return 0;
  long n = (rows + 63) / 64;               // >= 8 rows per sub-group pass, <= 8 workgroups per CU
  if (n > 2048) n = 2048;
  return (int)n;
}

int namp_train_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* out, long rows, void* stream) {
  REQUIRE_PTR(x); REQUIRE_PTR(gamma); REQUIRE_PTR(beta); REQUIRE_PTR(out);
  REQUIRE(rows >= 1, "namp_train_ln_rows_fwd: rows=%ld", rows);
  hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3(namp_train_ln_rows_groups(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out,
                     rows);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_ln_rows_bwd(const float* x, const float* g, const float* gamma, float* gx, float* dgb_part, long rows, void* stream) {
  REQUIRE_PTR(x); REQUIRE_PTR(g); REQUIRE_PTR(gamma); REQUIRE_PTR(gx); REQUIRE_PTR(dgb_part);
  REQUIRE(rows >= 1, "namp_train_ln_rows_bwd: rows=%ld", rows);
  hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3(namp_train_ln_rows_groups(rows)), dim3(256), 0, (hipStream_t)stream, x, g, gamma, gx,
                     dgb_part, rows);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_reduce_sum(const NampReduce* seg, int nseg, void* stream) {
  REQUIRE(seg != nullptr, "namp_reduce_sum: null segment table");
  REQUIRE(nseg >= 1 && nseg <= NAMP_REDUCE_MAX, "namp_reduce_sum: nseg=%d must be in [1,%d]", nseg, NAMP_REDUCE_MAX);
  ReduceArgs ra = {};
  ra.nseg = nseg;
  long blocks = 0;
  for (int s = 0; s < nseg; ++s) {
    const NampReduce& d = seg[s];
    REQUIRE(d.src != nullptr && d.dst != nullptr, "namp_reduce_sum: segment %d: null src / dst", s);
    REQUIRE(d.A >= 1 && d.Mb >= 1 && d.n >= 1 && d.sa >= 0 && d.sn >= 1, "namp_reduce_sum: segment %d: A=%ld Mb=%ld n=%d sa=%ld sn=%ld", s,
            (long)d.A, (long)d.Mb, d.n, (long)d.sa, (long)d.sn);
    ReduceSeg& g = ra.seg[s];
    g.src = d.src; g.dst = d.dst; g.A = d.A; g.Mb = d.Mb; g.sa = d.sa; g.sn = d.sn; g.n = d.n;
    const bool vec = d.Mb % 4 == 0 && d.sa % 4 == 0 && d.sn % 4 == 0 && aligned16(d.src) && aligned16(d.dst);
    g.vec = vec ? (d.n >= 64 ? 16 : 4) : 0;
    ra.first_block[s] = (int)blocks;
    const long per_block = vec ? 4 * (256 / g.vec) : 256;      // outputs per workgroup
    blocks += (d.A * d.Mb + per_block - 1) / per_block;
    REQUIRE(blocks < (1L << 30), "namp_reduce_sum: too many outputs");
  }
  ra.first_block[nseg] = (int)blocks;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ra);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_pos_features(const int32_t* R_idx, const int32_t* chain, const int32_t* E_idx, const float* pos_w, const float* pos_b,
                            int32_t* d_out, float* E_pos, int B, int L, int K, void* stream) {
  if (!R_idx || !chain || !E_idx || !d_out) return fail(NAMP_EINVAL, "namp_train_pos_features: null pointer argument");
  REQUIRE_PTR(pos_w); REQUIRE_PTR(pos_b); REQUIRE_PTR(E_pos);
  REQUIRE(B >= 1 && L >= 1 && K >= 1 && K <= L, "namp_train_pos_features: bad dims B=%d L=%d K=%d", B, L, K);
  const long E = (long)B * L * K;
  long blocks = (E + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pos_features_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, R_idx, chain, E_idx, pos_w, pos_b, d_out, E_pos,
                     E, L, K);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_pos_grad_groups(long edges) {
  if (edges <= 0) return 0;
  long n = (edges + 8 * POS_GRAD_WAVES - 1) / (8 * POS_GRAD_WAVES);      // >= 8 rows per wave
  if (n > 1024) n = 1024;
  return (int)n;
}

int namp_train_pos_grad(const float* g, const float* Wedge, int ld, const int32_t* d, float* part, long edges, void* stream) {
  REQUIRE_PTR(g); REQUIRE_PTR(Wedge); REQUIRE_PTR(part);
  if (!d) return fail(NAMP_EINVAL, "namp_train_pos_grad: null class index");
  REQUIRE(edges >= 1 && ld >= POS_DIM, "namp_train_pos_grad: edges=%ld ld=%d", edges, ld);
  hipLaunchKernelGGL(pos_grad_kernel, dim3(namp_train_pos_grad_groups(edges)), dim3(64 * POS_GRAD_WAVES), 0, (hipStream_t)stream, g, Wedge, ld, d,
                     part, edges);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_reverse_adjacency(const int32_t* E_idx, int32_t* offsets, int32_t* edges, int32_t* ws, int B, int N, int K, void* stream) {
  if (!E_idx || !offsets || !edges || !ws) return fail(NAMP_EINVAL, "namp_train_reverse_adjacency: null pointer argument");
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= N, "namp_train_reverse_adjacency: bad dims B=%d N=%d K=%d", B, N, K);
  const long G = (long)B * N, E = G * K;
  REQUIRE(E < (1L << 31), "namp_train_reverse_adjacency: %ld edges do not fit int32", E);
  hipStream_t s = (hipStream_t)stream;
  int32_t* counts = ws;                    // [G]
  int32_t* cursor = ws + G;                // [G]
  int32_t* tmp = ws + 2 * G;               // [E]
  hipError_t e_ = hipMemsetAsync(counts, 0, (size_t)G * sizeof(int32_t), s);
  if (e_ != hipSuccess) return fail(NAMP_ELAUNCH, "namp_train_reverse_adjacency: hipMemsetAsync: %s", hipGetErrorString(e_));
  long blocks = (E + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(radj_count_kernel, dim3((unsigned)blocks), dim3(256), 0, s, E_idx, counts, E, N, K);
  hipLaunchKernelGGL(radj_scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, cursor, (int)G);
  hipLaunchKernelGGL(radj_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, E_idx, cursor, tmp, E, N, K);
  hipLaunchKernelGGL(radj_sort_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, s, offsets, tmp, edges, (int)G);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_rows_groups(long rows) {
  if (rows <= 0) return 0;
  long n = (rows + 63) / 64;               // >= 16 rows per wave
  if (n > 256) n = 256;
  return (int)n;
}

int namp_train_class_sums(const float* g, const int32_t* idx, int nclass, long rows, float* part, void* stream) {
  REQUIRE_PTR(g); REQUIRE_PTR(part);
  if (!idx) return fail(NAMP_EINVAL, "namp_train_class_sums: null class index");
  REQUIRE(nclass >= 1 && nclass <= CLASS_SUMS_MAX && rows >= 1, "namp_train_class_sums: nclass=%d (1..%d) rows=%ld", nclass, CLASS_SUMS_MAX, rows);
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] { err = hipFuncSetAttribute((const void*)class_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * CLASS_SUMS_MAX * NAMP_H * 4); });
  if (err != hipSuccess) return fail(NAMP_ELAUNCH, "hipFuncSetAttribute(class_sums): %s", hipGetErrorString(err));
  hipLaunchKernelGGL(class_sums_kernel, dim3(namp_train_rows_groups(rows)), dim3(256), (size_t)4 * nclass * NAMP_H * 4, (hipStream_t)stream, g, idx, nclass,
                     rows, part);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_wcolsum(const float* g, const float* w, long rows, float* part, void* stream) {
  REQUIRE_PTR(g); REQUIRE_PTR(part);
  if (!w) return fail(NAMP_EINVAL, "namp_train_wcolsum: null row weights");
  REQUIRE(rows >= 1, "namp_train_wcolsum: rows=%ld", rows);
  hipLaunchKernelGGL(wcolsum_kernel, dim3(namp_train_rows_groups(rows)), dim3(256), 0, (hipStream_t)stream, g, w, rows, part);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_wgrad_chunks(long rows) {
  if (rows <= 0) return 0;
  long n = (rows + 511) / 512;             // >= 512 rows (32 MFMA steps) per workgroup, <= 2 workgroups per CU (the split-bf16
  if (n > 512) n = 512;                    // kernel's occupancy): every chunk costs 64 KB of partials the caller has to add up
  return (int)n;
}

int namp_train_wgrad(const float* G, const float* A, int gelu_A, int x3, long rows, float* dW_part, float* db_part, void* stream) {
  REQUIRE(!(gelu_A && (x3 & 3)), "namp_train_wgrad: the split-bf16 form takes activations (gelu_A = 0)");
  REQUIRE_PTR(G); REQUIRE_PTR(A); REQUIRE_PTR(dW_part);
  REQUIRE(rows >= 1, "namp_train_wgrad: rows=%ld", rows);
  const int nchunk = namp_train_wgrad_chunks(rows);
  long per = (rows + nchunk - 1) / nchunk;
  per = (per + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  // bits 4 / 5 of the precision argument: G / A are bf16 row tensors (the mixed-precision backward's outputs)
  const bool g16 = (x3 & 16) != 0, a16 = (x3 & 32) != 0;
  x3 &= 3;
  REQUIRE(!(g16 || a16) || (x3 == 2 && g16), "namp_train_wgrad: bf16 row tensors belong to precision code 2, and G must be one of them");
  if (g16 && a16) hipLaunchKernelGGL(wgrad_bf16_kernel<true>, dim3(nchunk), dim3(256), 0, s, (const __bf16*)G, (const void*)A, rows, per, dW_part, db_part);
  else if (g16) hipLaunchKernelGGL(wgrad_bf16_kernel<false>, dim3(nchunk), dim3(256), 0, s, (const __bf16*)G, (const void*)A, rows, per, dW_part, db_part);
  else if (x3 == 2) hipLaunchKernelGGL(wgrad_x3_kernel<false>, dim3(nchunk), dim3(256), 0, s, G, A, rows, per, dW_part, db_part);
  else if (x3) hipLaunchKernelGGL(wgrad_x3_kernel<true>, dim3(nchunk), dim3(256), 0, s, G, A, rows, per, dW_part, db_part);
  else if (gelu_A) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(nchunk), dim3(256), 0, s, G, A, rows, per, dW_part, db_part);
  else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(nchunk), dim3(256), 0, s, G, A, rows, per, dW_part, db_part);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_wgrad_ln(const float* G, const float* Y, const float* ln_stats, const float* ln_g, const float* ln_b, int x3, long rows,
                        float* dW_part, float* db_part, void* stream) {
  REQUIRE_PTR(G); REQUIRE_PTR(Y); REQUIRE_PTR(ln_g); REQUIRE_PTR(ln_b); REQUIRE_PTR(dW_part);
  if (!ln_stats || ((uintptr_t)ln_stats & 7)) return fail(NAMP_EINVAL, "namp_train_wgrad_ln: ln_stats is null or not 8-byte aligned");
  REQUIRE(x3 == 1 || x3 == 2, "namp_train_wgrad_ln: precision code %d (1 = split-bf16, 2 = bf16 products)", x3);
  REQUIRE(rows >= 1, "namp_train_wgrad_ln: rows=%ld", rows);
  const int nchunk = namp_train_wgrad_chunks(rows);
  long per = (rows + nchunk - 1) / nchunk;
  per = (per + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  if (x3 == 2) hipLaunchKernelGGL(wgrad_x3_ln_kernel<false>, dim3(nchunk), dim3(256), 0, s, G, Y, ln_stats, ln_g, ln_b, rows, per, dW_part, db_part);
  else hipLaunchKernelGGL(wgrad_x3_ln_kernel<true>, dim3(nchunk), dim3(256), 0, s, G, Y, ln_stats, ln_g, ln_b, rows, per, dW_part, db_part);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_embed_ln_bwd_groups(long rows) {
  if (rows <= 0) return 0;
  const long tiles = (rows + 15) / 16, wgs = (tiles + 7) / 8;
  const int cus = dw_cus();
  return (int)(wgs < cus ? wgs : cus);
}

long namp_train_g16_elems(long rows) { return rows <= 0 ? 0 : (rows + FEATW_TILE - 1) / FEATW_TILE * (long)(NAMP_H * FEATW_TILE); }

int namp_train_embed_ln_bwd(const float* g, const float* Y, const float* Wt_img, const float* ln_g, float* g_pre, float* ln_stats,
                            float* dgb_part, void* g16, int x3, long rows, void* stream) {
  REQUIRE_PTR(g); REQUIRE_PTR(Y); REQUIRE_PTR(Wt_img); REQUIRE_PTR(ln_g); REQUIRE_PTR(g_pre); REQUIRE_PTR(dgb_part);
  if (!ln_stats || ((uintptr_t)ln_stats & 7)) return fail(NAMP_EINVAL, "namp_train_embed_ln_bwd: ln_stats is null or not 8-byte aligned");
  REQUIRE(x3 == 1 || x3 == 2, "namp_train_embed_ln_bwd: precision code %d (1 = split-bf16, 2 = bf16 products)", x3);
  REQUIRE(rows >= 1, "namp_train_embed_ln_bwd: rows=%ld", rows);
  int rc = ensure_attributes();
  if (rc) return rc;
  EmbedLnBwdArgs a = {};
  a.g = g; a.y = Y; a.Wt_img = Wt_img; a.ln_g = ln_g; a.g_pre = g_pre; a.stats = ln_stats; a.dgb_part = dgb_part; a.E = rows;
  if (g16 && ((uintptr_t)g16 & 15)) return fail(NAMP_EINVAL, "namp_train_embed_ln_bwd: g16 is not 16-byte aligned");
  a.g16 = (__bf16*)g16; a.g16_plane = namp_train_g16_elems(rows);
  const int grid = namp_train_embed_ln_bwd_groups(rows);
  hipStream_t s = (hipStream_t)stream;
  if (x3 == 2) hipLaunchKernelGGL(embed_ln_bwd_kernel<2>, dim3(grid), dim3(512), NAMP_BIMG_BYTES, s, a);
  else hipLaunchKernelGGL(embed_ln_bwd_kernel<1>, dim3(grid), dim3(512), NAMP_IMG_BYTES, s, a);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_wgrad_multi(const float* const* G, const float* const* A, int n, int x3, long rows, int chunks, float* const* dW_part,
                           float* const* db_part, void* stream) {
  REQUIRE(G && A && dW_part && db_part, "namp_train_wgrad_multi: null pointer table");
  REQUIRE(n >= 1 && n <= 8, "namp_train_wgrad_multi: n=%d must be in [1,8]", n);
  const int accumulate = (x3 & 64) ? 1 : 0;          // bit 6: add to the partials of an earlier launch over other rows (same chunk count or more)
  x3 &= ~64;
  REQUIRE(x3 == 1 || x3 == 2, "namp_train_wgrad_multi: precision code %d (1 = split-bf16, 2 = bf16 products of fp32 rows)", x3);
  REQUIRE(rows >= 1, "namp_train_wgrad_multi: rows=%ld", rows);
  WgradMulti m = {};
  for (int q = 0; q < 8; ++q) {
    const int s_ = q < n ? q : 0;
    REQUIRE_PTR(G[s_]); REQUIRE_PTR(A[s_]); REQUIRE_PTR(dW_part[s_]);
    m.G[q] = G[s_]; m.A[q] = A[s_]; m.dW[q] = dW_part[s_]; m.db[q] = db_part[s_];
  }
  REQUIRE(chunks >= 0 && chunks <= 512, "namp_train_wgrad_multi: chunks=%d must be in [0,512] (0 = namp_train_wgrad_chunks(rows))", chunks);
  const int nchunk = chunks ? chunks : namp_train_wgrad_chunks(rows);
  long per = (rows + nchunk - 1) / nchunk;
  per = (per + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  // few workgroups (residue-level contractions: 47 chunks x <= 8): the launch lasts as long as one workgroup's chain of dependent steps — two
  // groups of four waves split the chain (KS = 2).  Chip-filling launches keep four waves, two workgroups per CU.
  const bool ks2 = (long)nchunk * n <= 1024;            // cfg5: 984 -> 650 us per step over the 28 residue-level launches (36 -> 23 us each)
  per = (per + 63) / 64 * 64;
  if (ks2) {
    if (x3 == 2) hipLaunchKernelGGL((wgrad_x3_multi_kernel<false, 2>), dim3(nchunk, n), dim3(512), 0, s, m, rows, per, accumulate);
    else hipLaunchKernelGGL((wgrad_x3_multi_kernel<true, 2>), dim3(nchunk, n), dim3(512), 0, s, m, rows, per, accumulate);
  } else if (x3 == 2) hipLaunchKernelGGL((wgrad_x3_multi_kernel<false, 1>), dim3(nchunk, n), dim3(256), 0, s, m, rows, per, accumulate);
  else hipLaunchKernelGGL((wgrad_x3_multi_kernel<true, 1>), dim3(nchunk, n), dim3(256), 0, s, m, rows, per, accumulate);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_feat_wgrad_chunks(long edges) {
  if (edges <= 0) return 0;
  long n = (edges + 4095) / 4096;          // finer chunks than the dense cost needs: the block sparsity makes workgroups uneven
  if (n > 128) n = 128;
  return (int)n;
}

long namp_train_feat_wgrad_ws_ints(long edges) { return edges <= 0 ? 0 : 2 * ((edges + FEATW_TILE - 1) / FEATW_TILE); }

int namp_train_feat_wgrad(const float* X18, const float* M18, const int32_t* E_idx, const float* E_pos, const float* g_pre, const void* g16,
                          float* dW_part, int32_t* tile_ws, int x3, int B, int L, int K, void* stream) {
  if (!tile_ws) return fail(NAMP_EINVAL, "namp_train_feat_wgrad: null tile workspace (namp_train_feat_wgrad_ws_ints int32s)");
  REQUIRE_PTR(X18); REQUIRE_PTR(E_pos); REQUIRE_PTR(g_pre); REQUIRE_PTR(dW_part);
  const bool packed = (M18 == nullptr);            // X18 = [G][18][4] (x, y, z, mask): split-bf16 / bf16 launches only
  REQUIRE(!packed || x3 != 0, "namp_train_feat_wgrad: the exact-fp32 launch takes X18 [G][18][3] and M18 [G][18] (M18 = NULL means the packed form)");
  REQUIRE(packed || aligned16(M18), "namp_train_feat_wgrad: 'M18' is not 16-byte aligned");
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_feat_wgrad: null E_idx");
  REQUIRE(B >= 1 && L >= 1 && K >= 1 && K <= L, "namp_train_feat_wgrad: bad dims B=%d L=%d K=%d", B, L, K);
  REQUIRE(!g16 || (packed && x3 != 0 && aligned16(g16)), "namp_train_feat_wgrad: g16 (bf16 tiles of g_pre) goes with the packed atoms and precision code 1 / 2");
  const long E = (long)B * L * K;
  const int nchunk = namp_train_feat_wgrad_chunks(E);
  long per = (E + nchunk - 1) / nchunk;
  per = (per + FEATW_TILE - 1) / FEATW_TILE * FEATW_TILE;
  const long ntile = (E + FEATW_TILE - 1) / FEATW_TILE;
  hipLaunchKernelGGL(tile_presence_kernel, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, (hipStream_t)stream, packed ? X18 + 3 : M18,
                     packed ? 4 : 1, E_idx, E, L, K, tile_ws);
  const __bf16* g16p = (const __bf16*)g16;
  const long g16_plane = namp_train_g16_elems(E);
#define NAMP_FEATW(MID_, PK_, S16_) hipLaunchKernelGGL((feat_wgrad_x3_kernel<MID_, PK_, S16_>), dim3(FEATW_GRID_X, nchunk), dim3(256), 0, (hipStream_t)stream, \
                                                       X18, M18, E_idx, E_pos, g_pre, tile_ws, E, per, L, K, dW_part, g16p, g16_plane)
  if (x3 == 2) { if (g16p) hipLaunchKernelGGL(feat_wgrad_t16_kernel, dim3(FEATW_GRID_X, nchunk), dim3(256), 0, (hipStream_t)stream, X18, E_idx, E_pos, tile_ws,
                                               E, per, L, K, dW_part, g16p);
                 else if (packed) NAMP_FEATW(false, true, false); else NAMP_FEATW(false, false, false); }   // mixed precision
  else if (x3) { if (g16p) NAMP_FEATW(true, true, true); else if (packed) NAMP_FEATW(true, true, false); else NAMP_FEATW(true, false, false); }
#undef NAMP_FEATW
  else
    hipLaunchKernelGGL(feat_wgrad_kernel, dim3(41, nchunk), dim3(256), 0, (hipStream_t)stream, X18, M18, E_idx, E_pos, g_pre,
                       tile_ws, E, per, L, K, dW_part);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_loss_smoothed(int backward, const int32_t* S, const float* log_probs, const float* protein_mask, const float* dna_mask,
                             const float* rna_mask, const float* protein_restypes, const float* dna_restypes, const float* rna_restypes,
                             const float* eps_scale3, double weight, const int32_t* ppm_mask, const double* aligned_ppm,
                             double* loss, const double* g_loss, float* g_log_probs, long G, int V, void* stream) {
  if (!S) return fail(NAMP_EINVAL, "namp_train_loss_smoothed: null S");
  REQUIRE_PTR(protein_mask); REQUIRE_PTR(dna_mask); REQUIRE_PTR(rna_mask);
  if (!protein_restypes || !dna_restypes || !rna_restypes || !eps_scale3) return fail(NAMP_EINVAL, "namp_train_loss_smoothed: null restype tables");
  REQUIRE(G >= 1 && V >= 1 && V <= 64, "namp_train_loss_smoothed: bad dims G=%ld V=%d", G, V);
  REQUIRE((ppm_mask == nullptr) == (aligned_ppm == nullptr), "namp_train_loss_smoothed: ppm_mask and aligned_ppm go together");
  LossArgs a = {};
  a.S = S; a.log_probs = log_probs; a.pm[0] = protein_mask; a.pm[1] = dna_mask; a.pm[2] = rna_mask;
  a.rm[0] = protein_restypes; a.rm[1] = dna_restypes; a.rm[2] = rna_restypes;
  for (int k = 0; k < 3; ++k) a.eps_scale[k] = eps_scale3[k];
  a.one_minus_w = 1.0 - weight; a.ppm_mask = ppm_mask; a.aligned_ppm = aligned_ppm; a.G = G; a.V = V;
  const unsigned grid = (unsigned)((G + 255) / 256);
  if (backward) {
    if (!g_loss || !g_log_probs) return fail(NAMP_EINVAL, "namp_train_loss_smoothed: backward needs g_loss and g_log_probs");
    hipLaunchKernelGGL(loss_smoothed_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (double*)nullptr, g_loss, g_log_probs);
  } else {
    if (!log_probs || !loss) return fail(NAMP_EINVAL, "namp_train_loss_smoothed: forward needs log_probs and loss");
    hipLaunchKernelGGL(loss_smoothed_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, loss, (const double*)nullptr, (float*)nullptr);
  }
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_adam_chunk(void) { return NAMP_ADAM_CHUNK; }

int namp_train_adam_step(const int32_t* blk_tensor, const long long* blk_off, const long long* numel, const unsigned long long* ptrs,
                         int ntensors, int nblocks, float max_norm, double beta1, double beta2, float step_size, float bias_correction2_sqrt,
                         float eps, float* ws, void* stream) {
  if (!blk_tensor || !blk_off || !numel || !ptrs) return fail(NAMP_EINVAL, "namp_train_adam_step: null plan");
  REQUIRE(ntensors >= 1 && nblocks >= 1, "namp_train_adam_step: ntensors=%d nblocks=%d", ntensors, nblocks);
  REQUIRE(max_norm <= 0.f || ws != nullptr, "namp_train_adam_step: clipping needs a workspace of nblocks + 2 floats");
  AdamPlan p = {blk_tensor, blk_off, numel, ptrs, ntensors, nblocks};
  hipStream_t s = (hipStream_t)stream;
  const float* coef2 = nullptr;
  if (max_norm > 0.f) {
    hipLaunchKernelGGL(adam_sqnorm_kernel, dim3(nblocks), dim3(256), 0, s, p, ws + 2);
    hipLaunchKernelGGL(adam_coef_kernel, dim3(1), dim3(256), 0, s, (const float*)(ws + 2), nblocks, max_norm, ws);
    coef2 = ws;
  }
  hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks), dim3(256), 0, s, p, coef2, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                     bias_correction2_sqrt, eps);
  CHECK_LAUNCH();
  return NAMP_OK;
}

}  // extern "C"
