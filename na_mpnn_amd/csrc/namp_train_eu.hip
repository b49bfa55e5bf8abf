// Entry point of the EncLayer edge update's two-launch backward (namp_train_eu.h) — its own translation unit so that it compiles
// beside namp_train.hip.  See include/namp.h "training".
#include "../../include/namp.h"
#define NAMP_TRAIN_EDGE_ONLY
#include "namp_train_eu.h"

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
    auto set = [](const void* f, int bytes) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e != hipSuccess) g_attr_err = e;
    };
    set((const void*)(edge_update_bwd_a16_kernel<false>), EUA_LDS);
    set((const void*)(edge_update_bwd_a16_kernel<true>), EUA_LDS);
    set((const void*)(edge_update_bwd_b16_kernel<1>), EUB_LDS);
    set((const void*)(edge_update_bwd_b16_kernel<2>), EUB_LDS);
  });
  if (g_attr_err != hipSuccess)
    return fail(NAMP_ELAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(g_attr_err));
  return NAMP_OK;
}

}  // namespace

extern "C" {

// The edge update's backward as two persistent launches that own their weight gradients (mixed precision only: x3 & 3 == 2; bit 3 = per-tile
// g_Pa sums).  Row buffers G2, G1 (bf16) and g_hE (fp32) hold namp_train_edge_bwd_dw_rows() rows, g_Pa one row per 16 of those (bit 3) or
// [G][128] zeroed; dW_part = launch A's slab [groups][128][128] (dW3) followed by launch B's [groups][2][128][128] (0 = dW2, 1 = dW1b);
// db_part = [groups][128] (db3) followed by [groups][128] (db2); dgb_part [groups][2][128]; groups = namp_train_edge_bwd_dw_groups().
int namp_train_edge_update_bwd_dw(const float* h_E, const int32_t* E_idx, const float* Pa, const float* Pc, const float* W1_img,
                                  const float* W2_img, const float* W3_img, const float* W3t_img, const float* W2t_img,
                                  const float* W1t_img, const float* b2, const float* b3, const float* ln_g, float drop_p,
                                  uint32_t drop_seed, const float* g_out, float* G2, float* G1, float* g_hE, float* g_Pa,
                                  float* dW_part, float* db_part, float* dgb_part, int x3, int B, int N, int K, void* stream) {
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(W1_img); REQUIRE_PTR(W2_img); REQUIRE_PTR(W3_img);
  REQUIRE_PTR(W3t_img); REQUIRE_PTR(W2t_img); REQUIRE_PTR(W1t_img); REQUIRE_PTR(b2); REQUIRE_PTR(b3); REQUIRE_PTR(ln_g);
  REQUIRE_PTR(g_out); REQUIRE_PTR(G2); REQUIRE_PTR(G1); REQUIRE_PTR(g_hE); REQUIRE_PTR(g_Pa);
  REQUIRE_PTR(dW_part); REQUIRE_PTR(db_part); REQUIRE_PTR(dgb_part);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_edge_update_bwd_dw: null E_idx");
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "namp_train_edge_update_bwd_dw: drop_p=%g must be in [0,1)", (double)drop_p);
  REQUIRE(B >= 1 && N >= 1 && K >= 1 && K <= NAMP_MAX_K, "namp_train_edge_update_bwd_dw: bad dims B=%d N=%d K=%d", B, N, K);
  REQUIRE((x3 & 3) == 2, "namp_train_edge_update_bwd_dw: precision code %d (bf16 products = 2 only)", x3 & 3);
  REQUIRE(!(x3 & 8) || (K % 16) == 0, "namp_train_edge_update_bwd_dw: per-tile g_Pa sums need K %% 16 == 0 (K=%d)", K);
  int rc = ensure_attributes();
  if (rc) return rc;
  const int grid = namp_train_edge_bwd_dw_groups(B, N, K);
  hipStream_t s = (hipStream_t)stream;
  const long E = (long)B * N * K;
  const long nrounds = (E + DW_ROWS - 1) / DW_ROWS;
  // ---- launch A: LayerNorm / dropout backward, dW3 / db3, d ln -> dL/dx rows (in g_hE), G2 rows
  EdgeUpdAArgs ua = {};
  {
    EdgeBwdArgs& a = ua.b;
    a.hE = h_E; a.E_idx = E_idx; a.Pa = Pa; a.Pj0 = Pc;
    a.W1_img = W1_img; a.W2_img = W2_img; a.W3_img = W3_img; a.W3t_img = W3t_img;
    a.b2 = b2; a.b3 = b3; a.ln_g = ln_g; a.g_rows = g_out;
    if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_seed = drop_seed; a.drop_scale = 1.0f / (1.0f - drop_p); }
    a.G2 = G2; a.g_hE = g_hE;
    a.G = B * N; a.N = N; a.K = K; a.E = E;
    ua.dW_part = dW_part;
    ua.db_part = db_part; ua.dgb_part = dgb_part; ua.nrounds = nrounds;
  }
  if (drop_p > 0.f) hipLaunchKernelGGL((edge_update_bwd_a16_kernel<true>), dim3(grid), dim3(64 * DW_WAVES), EUA_LDS, s, ua);
  else hipLaunchKernelGGL((edge_update_bwd_a16_kernel<false>), dim3(grid), dim3(64 * DW_WAVES), EUA_LDS, s, ua);
  CHECK_LAUNCH();
  // ---- launch B: dW2 / db2, dW1b, dL/dh_E = dL/dx + W1b^T g1, G1 rows, g_Pa
  EdgeBwdDwArgs ub = {};
  {
    EdgeBwdArgs& a = ub.b;
    a.hE = h_E; a.E_idx = E_idx; a.Pa = Pa; a.Pj0 = Pc;
    a.W1_img = W1_img; a.W2t_img = W2t_img; a.W1t_img = W1t_img;
    a.G2 = G2; a.G1 = G1; a.g_hE = g_hE; a.g_hE_in = g_hE; a.g_Pa = g_Pa;
    a.G = B * N; a.N = N; a.K = K; a.E = E;
    a.gpa_tiles = (x3 & 8) ? 1 : 0;
    ub.dW_part = dW_part + (long)grid * NAMP_H * NAMP_H;
    ub.db_part = db_part + (long)grid * NAMP_H;
    ub.nrounds = nrounds;
  }
  if (x3 & 8) hipLaunchKernelGGL((edge_update_bwd_b16_kernel<1>), dim3(grid), dim3(64 * DW_WAVES), EUB_LDS, s, ub);
  else hipLaunchKernelGGL((edge_update_bwd_b16_kernel<2>), dim3(grid), dim3(64 * DW_WAVES), EUB_LDS, s, ub);
  CHECK_LAUNCH();
  return NAMP_OK;
}

}  // extern "C"
