// libnamp_hip.so — C ABI (include/namp.h) over the gfx950 kernels in namp_kernels.h.
// Host code here only validates arguments, carves the caller's workspace and enqueues
// launches on the caller's stream; it owns no device memory and never synchronises.
#include "../../include/namp.h"
#include "namp_kernels.h"
#include <atomic>
#include "namp_bf16s32.h"
#include "namp_bf16p.h"
#include "namp_node_w.h"
#include "namp_order.h"

#include <cstdarg>
#include <cstdlib>
#include <initializer_list>
#include <cstdio>
#include <mutex>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace

// namp_persist.hip (its own translation unit: the persistent kernel takes minutes to compile)
__attribute__((visibility("hidden"))) int namp_internal_launch_persistent(const struct PersistArgs* a, int x3, int grid, int block, hipStream_t s);

// error text setter for namp_train.hip (same thread-local buffer; not part of the ABI)
__attribute__((visibility("hidden"))) int namp_internal_fail(int code, const char* msg) { return fail(code, "%s", msg); }

namespace {

// which launches of the bf16-storage path run in their round-6 form: bit 0 messages, 1 edge update, 2 message + embedding (namp_bf16p.h),
// 3 residue update (namp_node_w.h), 4 the edge update's two-pass LayerNorm (equality test), 5 (any precision) the two-part edge-feature launch of small batches
static std::atomic<int> g_bf16p{[] { const char* e = getenv("NAMP_BF16P"); return e ? atoi(e) : 43; }()};

// ---- optional per-kernel timing (bench.py): thread-local, off by default --------------------
struct ProfRec { int kind; hipEvent_t a, b; };
thread_local bool g_prof_on = false;
thread_local std::vector<ProfRec> g_prof;

struct ProfScope {
  hipStream_t s; ProfRec r; bool on;
  ProfScope(int kind, hipStream_t st) : s(st), on(g_prof_on) {
    if (!on) return;
    r.kind = kind;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    g_prof.push_back(r);
  }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define REQUIRE_PTR(p)                                                                         \
  do {                                                                                         \
    if ((p) == nullptr) return fail(NAMP_EINVAL, "%s: null pointer argument '%s'", __func__, #p); \
    if (!aligned16(p)) return fail(NAMP_EINVAL, "%s: '%s' is not 16-byte aligned", __func__, #p); \
  } while (0)
#define OPTIONAL_PTR(p)                                                                        \
  do {                                                                                         \
    if ((p) != nullptr && !aligned16(p))                                                       \
      return fail(NAMP_EINVAL, "%s: '%s' is not 16-byte aligned", __func__, #p);               \
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

int check_dims(const char* fn, long B, long N, long K) {
  if (B < 1 || N < 1 || K < 1) return fail(NAMP_EINVAL, "%s: B, N, K must be >= 1 (got %ld, %ld, %ld)", fn, B, N, K);
  if (K > NAMP_MAX_K) return fail(NAMP_EINVAL, "%s: K=%ld exceeds NAMP_MAX_K=%d", fn, K, NAMP_MAX_K);
  if (B * N * K > (1L << 31) / NAMP_HIDDEN * 4) return fail(NAMP_EINVAL, "%s: B*N*K=%ld too large", fn, B * N * K);
  return NAMP_OK;
}

std::once_flag g_attr_once;
hipError_t g_attr_err = hipSuccess;

void set_lds_attributes() {
  auto set = [](const void* f, int bytes) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) g_attr_err = e;
  };
  set((const void*)edge_mlp_kernel<MODE_ENC_MSG, 0>, 2 * NAMP_IMG_BYTES);
  set((const void*)edge_mlp_kernel<MODE_DEC_MSG, 0>, 2 * NAMP_IMG_BYTES);
  set((const void*)edge_mlp_kernel<MODE_ENC_MSG, 4>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_DEC_MSG, 4>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_ENC_MSG, 8>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_DEC_MSG, 8>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_ENC_MSG, 16>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_DEC_MSG, 16>, EDGE_TAIL_LDS);
  set((const void*)edge_mlp_kernel<MODE_ENC_EDGE, 0>, 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 0, true>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 0, true>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 4, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 8, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 16, true>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_EDGE, 0, true>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 4, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 8, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 16, false, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, false, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, false, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, false, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 0, PREC_X3>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 0, PREC_X3>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_ENC_EDGE, 0, PREC_X3>), 2 * NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 4, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 8, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 16, PREC_X3>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 4, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 8, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_DEC_MSG, 16, PREC_X3, PRE_EDGE>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 4, PREC_X3, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 8, PREC_X3, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)(edge_mlp_kernel<MODE_ENC_MSG, 16, PREC_X3, PRE_EMBED>), EDGE_TAIL_LDS);
  set((const void*)edge_mlp_bf16_persistent_kernel<MODE_ENC_MSG>, 3 * NAMP_BIMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_mlp_bf16_persistent_kernel<MODE_DEC_MSG>, 3 * NAMP_BIMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_mlp_bf16_persistent_kernel<MODE_ENC_EDGE>, 3 * NAMP_BIMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_mlp_bf16s32_kernel<MODE_ENC_MSG>, BF16S32_LDS);
  set((const void*)edge_mlp_bf16s32_kernel<MODE_DEC_MSG>, BF16S32_LDS);
  set((const void*)edge_mlp_bf16s32_kernel<MODE_ENC_EDGE>, BF16S32_LDS);
  set((const void*)(edge_mlp_bf16s32_kernel<MODE_ENC_MSG, true>), BF16S32_LDS);
  set((const void*)node_update_w_kernel<false>, NODEW_LDS);
  set((const void*)node_update_w_kernel<true>, NODEW_LDS);
  set((const void*)node_linear_w_kernel<1>, NODEW_LDS);
  set((const void*)node_linear_w_kernel<2>, NODEW_LDS);
  set((const void*)decoding_order_kernel, 65536);
  set((const void*)work_lists_kernel, 65536);
  set((const void*)logits_mfma_kernel, LOGITS_MFMA_LDS(48));
  set((const void*)edge_mlp_bf16p_kernel<MODE_ENC_MSG>, BF16P_LDS);
  set((const void*)edge_mlp_bf16p_kernel<MODE_DEC_MSG>, BF16P_LDS);
  set((const void*)edge_mlp_bf16p_kernel<MODE_ENC_EDGE>, BF16P_LDS);
  set((const void*)(edge_mlp_bf16p_kernel<MODE_ENC_EDGE, false, true>), BF16P_LDS);
  set((const void*)(edge_mlp_bf16p_kernel<MODE_ENC_MSG, true>), BF16P_LDS);
  set((const void*)edge_mlp_kernel<MODE_EMBED, 0>, NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_EMBED, 0, PREC_BF16>), NAMP_IMG_BYTES);
  set((const void*)(edge_mlp_kernel<MODE_EMBED, 0, PREC_X3>), NAMP_IMG_BYTES);
  set((const void*)node_update_kernel, NODE_TAIL_LDS);
  set((const void*)(node_update_multi_kernel<2, 0>), NODE_MULTI_LDS(2));
  set((const void*)(node_update_multi_kernel<2, 1>), NODE_MULTI_LDS_X3(2));
  set((const void*)(node_update_multi_kernel<2, 2>), NODE_MULTI_LDS_X3(2));
  set((const void*)dec_sample_kernel<0, false, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<1, false, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<2, false, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<0, true, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<1, true, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<2, true, 8>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<0, false, 12>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<1, false, 12>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<0, true, 12>, SAMPLE_LDS);
  set((const void*)dec_sample_kernel<1, true, 12>, SAMPLE_LDS);
  set((const void*)edge_mlp_x3_persistent_kernel<MODE_ENC_MSG>, 2 * NAMP_IMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_mlp_x3_persistent_kernel<MODE_DEC_MSG>, 2 * NAMP_IMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_mlp_x3_persistent_kernel<MODE_ENC_EDGE>, 2 * NAMP_IMG_BYTES + 2048 + 12 * 512);
  set((const void*)edge_features_kernel<0>, FEAT_LDS);
  set((const void*)edge_features_kernel<1>, FEAT_LDS);
  set((const void*)edge_features_kernel<2>, FEAT_LDS);
  set((const void*)feat_finish_kernel<0>, NAMP_IMG_BYTES);
  set((const void*)feat_finish_kernel<1>, NAMP_IMG_BYTES);
  set((const void*)knn_kernel, 8192 * 8 + 64);
  set((const void*)knn_select_kernel, (8192 + 4096) * 8 + 1024 + 64);
}

int ensure_attributes() {
  std::call_once(g_attr_once, set_lds_attributes);
  if (g_attr_err != hipSuccess)
    return fail(NAMP_ELAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(g_attr_err));
  return NAMP_OK;
}

// The message kernels update their own residues in-launch (fused tail) while the batch is small enough
// that this beats a separate 16-residue-per-workgroup node_update launch: every workgroup of the fused
// form re-streams the 768 KiB of FFN + projection weights for <= 12 residues.
#ifndef NAMP_FUSED_TAIL_MAX_RESIDUES
#define NAMP_FUSED_TAIL_MAX_RESIDUES 2500   // tools/batch_sweep.py: 3000 / 4000 residues run 5 / 10 % faster unfused, 2000 run 5 % slower
#endif

struct EdgeGeom { int tpn, nwaves, npw, grid; };
EdgeGeom edge_geom(int G, int K) {
  EdgeGeom e;
  e.tpn = (K + 15) / 16;
  e.nwaves = e.tpn * (12 / e.tpn > 0 ? 12 / e.tpn : 1);
  e.npw = e.nwaves / e.tpn;
  e.grid = (G + e.npw - 1) / e.npw;
  return e;
}

template <int MODE, int TAIL = 0, int PREC = PREC_F32, int PRE = PRE_NONE>
int launch_edge(EdgeArgs a, hipStream_t s) {
  int rc = ensure_attributes();
  if (rc) return rc;
  const EdgeGeom e = edge_geom(a.G, a.K);
  a.TPN = e.tpn;
  const int lds = TAIL ? EDGE_TAIL_LDS : (MODE == MODE_EMBED) ? NAMP_IMG_BYTES : 2 * NAMP_IMG_BYTES;
  hipLaunchKernelGGL((edge_mlp_kernel<MODE, TAIL, PREC, PRE>), dim3(e.grid), dim3(e.nwaves * 64), lds, s, a);
  return NAMP_OK;
}

// precision dispatch: flags bit 0 of the layer struct selects the bf16 message GEMMs (throughput mode)
int device_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// bf16 throughput mode on a grid larger than two waves of workgroups: persistent workgroups (weights resident in LDS,
// no barriers in the tile loop) instead of one workgroup per <= 12 tiles
template <int MODE>
int launch_edge_bf16_persistent(EdgeArgs a, hipStream_t s) {
  int rc = ensure_attributes();
  if (rc) return rc;
  const EdgeGeom e = edge_geom(a.G, a.K);
  a.TPN = e.tpn;
  if ((long)a.G * e.tpn >= (1L << 31)) return fail(NAMP_EINVAL, "edge launch: %ld row tiles exceed 2^31", (long)a.G * e.tpn);
  hipLaunchKernelGGL((edge_mlp_bf16_persistent_kernel<MODE>), dim3(device_cus()), dim3(768), 3 * NAMP_BIMG_BYTES + 2048 + 12 * 512, s, a);
  return NAMP_OK;
}

// split-bf16 mode on a grid larger than two waves of workgroups: persistent workgroups, ring turning across rounds
template <int MODE>
int launch_edge_x3_persistent(EdgeArgs a, hipStream_t s) {
  int rc = ensure_attributes();
  if (rc) return rc;
  const EdgeGeom e = edge_geom(a.G, a.K);
  a.TPN = e.tpn;
  if ((long)a.G * e.tpn >= (1L << 31)) return fail(NAMP_EINVAL, "edge launch: %ld row tiles exceed 2^31", (long)a.G * e.tpn);
  hipLaunchKernelGGL((edge_mlp_x3_persistent_kernel<MODE>), dim3(device_cus()), dim3(768), 2 * NAMP_IMG_BYTES + 2048 + 12 * 512, s, a);
  return NAMP_OK;
}

// bf16 STORAGE variant (h_E and the gathered tables as bf16 rows in fragment order B): large batches of the bf16 throughput mode, on
// v_mfma_f32_32x32x16_bf16 (namp_bf16s32.h; images from namp_pack_image_bf16_32)
template <int MODE>
int launch_edge_bf16s(EdgeArgs a, hipStream_t s) {
  int rc = ensure_attributes();
  if (rc) return rc;
  const EdgeGeom e = edge_geom(a.G, a.K);
  a.TPN = e.tpn;
  if ((long)a.G * e.tpn >= (1L << 31)) return fail(NAMP_EINVAL, "edge launch: %ld row tiles exceed 2^31", (long)a.G * e.tpn);
  // 8 waves per CU (up to 256 VGPRs each), one 32-row pair of tiles per wave and step
  // NAMP_BF16P (A/B switch, bit mask over 1 = messages, 2 = edge update, 4 = message + embedding): the round-6 sequencing (namp_bf16p.h)
  const int bf16p = g_bf16p.load(std::memory_order_relaxed);
  if constexpr (MODE == MODE_ENC_MSG) {
    if (a.eW1_img && (bf16p & 4)) {
      hipLaunchKernelGGL((edge_mlp_bf16p_kernel<MODE_ENC_MSG, true>), dim3(device_cus()), dim3(512), BF16P_LDS, s, a);
      return NAMP_OK;
    }
  }
  if constexpr (MODE == MODE_ENC_EDGE) {
    if ((bf16p & 2) && (bf16p & 16)) {        // bit 4: LayerNorm 3 with the round-3 kernel's two-pass variance (the bit-equality test)
      hipLaunchKernelGGL((edge_mlp_bf16p_kernel<MODE_ENC_EDGE, false, true>), dim3(device_cus()), dim3(512), BF16P_LDS, s, a);
      return NAMP_OK;
    }
  }
  if (!(MODE == MODE_ENC_MSG && a.eW1_img) && (bf16p & (MODE == MODE_ENC_EDGE ? 2 : 1))) {
    hipLaunchKernelGGL((edge_mlp_bf16p_kernel<MODE>), dim3(device_cus()), dim3(512), BF16P_LDS, s, a);
    return NAMP_OK;
  }
  if constexpr (MODE == MODE_ENC_MSG) {
    if (a.eW1_img) {                     // fused edge embedding: a.hE = fp32 E, a.hE16_out = the bf16 rows
      hipLaunchKernelGGL((edge_mlp_bf16s32_kernel<MODE_ENC_MSG, true>), dim3(device_cus()), dim3(512), BF16S32_LDS, s, a);
      return NAMP_OK;
    }
  }
  hipLaunchKernelGGL((edge_mlp_bf16s32_kernel<MODE>), dim3(device_cus()), dim3(512), BF16S32_LDS, s, a);
  return NAMP_OK;
}

void launch_cvt_tables(const float* const* src, __bf16* const* dst, int n, long rows, hipStream_t s) {
  CvtTables c = {};
  c.n = n; c.rows = rows;
  for (int i = 0; i < n; ++i) { c.src[i] = src[i]; c.dst[i] = dst[i]; }
  const long total = rows * 16;
  hipLaunchKernelGGL(cvt_tables_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, c);
}

// precision of the per-edge GEMMs from a layer's flags: bf16 throughput mode, split-bf16 (fp32-equivalent) or exact fp32
#define NAMP_FLAG_BF16 1
#define NAMP_FLAG_X3 2
inline int prec_of(int64_t flags) { return (flags & NAMP_FLAG_BF16) ? PREC_BF16 : (flags & NAMP_FLAG_X3) ? PREC_X3 : PREC_F32; }
inline const float* pick_img(int prec, const float* img, const float* bimg, const float* ximg) {
  return prec == PREC_BF16 ? bimg : prec == PREC_X3 ? ximg : img;
}

template <int MODE, int TAIL>
int launch_edge_prec(const EdgeArgs& a, int prec, hipStream_t s) {
  if (prec == PREC_BF16 && TAIL == 0 && MODE != MODE_EMBED && edge_geom(a.G, a.K).grid > 2 * device_cus())
    return launch_edge_bf16_persistent<MODE == MODE_EMBED ? MODE_ENC_MSG : MODE>(a, s);
  if (prec == PREC_BF16) return launch_edge<MODE, TAIL, PREC_BF16>(a, s);
  if (prec == PREC_X3 && TAIL == 0 && MODE != MODE_EMBED && edge_geom(a.G, a.K).grid > 2 * device_cus())
    return launch_edge_x3_persistent<MODE == MODE_EMBED ? MODE_ENC_MSG : MODE>(a, s);
  if (prec == PREC_X3) return launch_edge<MODE, TAIL, PREC_X3>(a, s);
  return launch_edge<MODE, TAIL, PREC_F32>(a, s);
}

// fused-tail flavour by residues per workgroup: <= 4 and <= 6 take the VALU tails (the latter needs 128
// LayerNorm threads per residue: 2*npw <= waves), more take the 16-row MFMA tail
template <int MODE>
int launch_edge_tail(const EdgeArgs& a, int prec, hipStream_t s) {
  const EdgeGeom e = edge_geom(a.G, a.K);
  if (e.npw <= 4) return launch_edge_prec<MODE, 4>(a, prec, s);
  if (e.npw <= 8 && 2 * e.npw <= e.nwaves) return launch_edge_prec<MODE, 8>(a, prec, s);
  return launch_edge_prec<MODE, 16>(a, prec, s);
}

// a PRE stage (edge update of the previous layer / edge embedding) fused in front of the message phase (fp32-class only)
template <int MODE, int PRE = PRE_EDGE>
int launch_edge_tail_fused(const EdgeArgs& a, int prec, hipStream_t s) {
  const EdgeGeom e = edge_geom(a.G, a.K);
  if (prec == PREC_X3) {
    if (e.npw <= 4) return launch_edge<MODE, 4, PREC_X3, PRE>(a, s);
    if (e.npw <= 8 && 2 * e.npw <= e.nwaves) return launch_edge<MODE, 8, PREC_X3, PRE>(a, s);
    return launch_edge<MODE, 16, PREC_X3, PRE>(a, s);
  }
  if (e.npw <= 4) return launch_edge<MODE, 4, PREC_F32, PRE>(a, s);
  if (e.npw <= 8 && 2 * e.npw <= e.nwaves) return launch_edge<MODE, 8, PREC_F32, PRE>(a, s);
  return launch_edge<MODE, 16, PREC_F32, PRE>(a, s);
}

void fill_tail(NodeTail& t, const float* ln1_g, const float* ln1_b, const float* Win_img, const float* b_in,
               const float* Wout_img, const float* b_out, const float* ln2_g, const float* ln2_b, const float* hV,
               const int32_t* mask, float* hV_out, const NampProj* proj, int nproj, const int32_t* S) {
  t.hV = hV; t.mask = mask; t.ln1_g = ln1_g; t.ln1_b = ln1_b; t.Win_img = Win_img; t.b_in = b_in;
  t.Wout_img = Wout_img; t.b_out = b_out; t.ln2_g = ln2_g; t.ln2_b = ln2_b; t.hV_out = hV_out; t.S = S;
  t.head_w = nullptr; t.head_b = nullptr; t.log_probs = nullptr; t.logits = nullptr; t.vocab = 0;
  t.m3_img = nullptr; t.m3_b = nullptr;
  t.nproj = nproj;
  for (int i = 0; i < 8; ++i) {
    if (i < nproj) { t.p[i].img = proj[i].img; t.p[i].bias = proj[i].bias; t.p[i].tok = proj[i].tok; t.p[i].out = proj[i].out; }
    else { t.p[i].img = nullptr; t.p[i].bias = nullptr; t.p[i].tok = nullptr; t.p[i].out = nullptr; }
  }
}

int check_proj(const char* fn, const NampProj* proj, int nproj, const int32_t* S) {
  if (nproj < 0 || nproj > 8 || (nproj > 0 && !proj)) return fail(NAMP_EINVAL, "%s: nproj=%d must be in [0,8]", fn, nproj);
  for (int i = 0; i < nproj; ++i) {
    if (!proj[i].img || !proj[i].out) return fail(NAMP_EINVAL, "%s: proj[%d] has a null image / output", fn, i);
    if (!aligned16(proj[i].img) || !aligned16(proj[i].out) || !aligned16(proj[i].bias) || !aligned16(proj[i].tok))
      return fail(NAMP_EINVAL, "%s: proj[%d] pointers must be 16-byte aligned", fn, i);
    if (proj[i].tok && !S) return fail(NAMP_EINVAL, "%s: proj[%d].tok given but S is null", fn, i);
  }
  return NAMP_OK;
}

// bf16-storage path (encdec_bf16_storage): the residue kernels write the tables the edge launches gather directly as bf16 rows in
// fragment order (and skip the fp32 copy) instead of a conversion launch per stage.  Set around the producer call; `honoured`
// tells the caller whether the launch that ran supports it (the one-tile fp32 residue kernel does not).
struct Out16Req { __bf16* p[8]; int n; bool honoured; };
thread_local Out16Req* g_out16 = nullptr;
// Set by encdec_bf16_storage around its residue-level launches: the bf16 throughput mode runs them as plain bf16 products (hi . hi of the
// x3 images, fp32 accumulation) — the reference's AMP autocasts the whole model (na_run.py:216-218).  NAMP_BF16S_RESIDUE_X3=1 restores the
// split-bf16 (fp32-equivalent) residue GEMMs of rounds 2-3 for A/B runs (measured: profiles/r03e, r04*).
thread_local bool g_residue_x1 = false;

int launch_node_linear(const float* X, const int32_t* S, int G_out, int G_src, int N,
                       const NampProj* proj, int nproj, const NampProj* pre, hipStream_t s, bool x3 = false, unsigned* zero = nullptr) {
  NodeLinearArgs a;
  a.X = X; a.S = S; a.G_out = G_out; a.G_src = G_src; a.N = N; a.nproj = nproj; a.zero = zero;
  a.pre.img = pre ? pre->img : nullptr; a.pre.bias = pre ? pre->bias : nullptr;
  a.pre.tok = nullptr; a.pre.out = pre ? pre->out : nullptr;
  for (int i = 0; i < 8; ++i) {
    const NampProj& p = proj[i < nproj ? i : 0];
    a.p[i].img = p.img; a.p[i].bias = p.bias; a.p[i].tok = p.tok; a.p[i].out = p.out;
    a.out16[i] = (g_out16 && i < nproj && i < g_out16->n) ? g_out16->p[i] : nullptr;
    if (a.out16[i]) a.p[i].out = nullptr;
  }
  if (g_out16) g_out16->honoured = true;
  const int units = ((G_out + 15) / 16) * nproj;
  if (x3 && G_out >= 4096 && (g_bf16p.load(std::memory_order_relaxed) & 8) && ensure_attributes() == NAMP_OK) {
    // large batches: one tile per wave through all blocks, weights through an LDS ring (namp_node_w.h: node_linear_w_kernel)
    if (g_residue_x1) hipLaunchKernelGGL(node_linear_w_kernel<2>, dim3((G_out + NODEW_ROWS - 1) / NODEW_ROWS), dim3(NODEW_THREADS), NODEW_LDS, s, a);
    else hipLaunchKernelGGL(node_linear_w_kernel<1>, dim3((G_out + NODEW_ROWS - 1) / NODEW_ROWS), dim3(NODEW_THREADS), NODEW_LDS, s, a);
    return NAMP_OK;
  }
  if (x3 && g_residue_x1) hipLaunchKernelGGL(node_linear_kernel<2>, dim3((units + 3) / 4), dim3(256), 0, s, a);
  else if (x3) hipLaunchKernelGGL(node_linear_kernel<1>, dim3((units + 3) / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(node_linear_kernel<0>, dim3((units + 3) / 4), dim3(256), 0, s, a);
  return NAMP_OK;
}

int launch_node_update(const float* ln1_g, const float* ln1_b, const float* Win_img, const float* b_in,
                       const float* Wout_img, const float* b_out, const float* ln2_g, const float* ln2_b,
                       const float* hV, const float* partial, const float* m3_img, const float* m3_b, const int32_t* mask,
                       float* hV_out, const NampProj* proj, int nproj, const int32_t* S, int G, int TPN, hipStream_t s,
                       bool x3 = false) {
  int rc = ensure_attributes();
  if (rc) return rc;
  NodeUpdateArgs a;
  fill_tail(a.t, ln1_g, ln1_b, Win_img, b_in, Wout_img, b_out, ln2_g, ln2_b, hV, mask, hV_out, proj, nproj, S);
  a.t.m3_img = partial ? m3_img : nullptr; a.t.m3_b = m3_b;      // (x3: the x3 image of W3)
  a.partial = partial; a.G = G; a.TPN = TPN;
  const bool multi = x3 || G >= 32 * 2 * device_cus();
  for (int i = 0; i < 8; ++i) {
    a.out16[i] = (multi && g_out16 && i < nproj && i < g_out16->n) ? g_out16->p[i] : nullptr;
    if (a.out16[i]) a.t.p[i].out = nullptr;
  }
  if (multi && g_out16) g_out16->honoured = true;
  // large batches: 2 tiles per workgroup share every weight fragment (the one-tile form re-streams 768 KiB per 16 rows;
  // 4 tiles would halve the stream again but spill — measured in the split-bf16 form too: 87 spilled VGPRs, 224 vs 165 us)
  if (x3 && g_residue_x1 && a.t.m3_img && !a.t.head_w && nproj <= 4 && G >= 2048 && (g_bf16p.load(std::memory_order_relaxed) & 8))
    // bf16 throughput mode, large batch: one tile per wave end to end, weight blocks through an LDS ring (namp_node_w.h)
    hipLaunchKernelGGL(node_update_w_kernel<false>, dim3((G + NODEW_ROWS - 1) / NODEW_ROWS), dim3(NODEW_THREADS), NODEW_LDS, s, a);
  else if (x3 && !g_residue_x1 && a.t.m3_img && !a.t.head_w && nproj <= 4 && G >= 2048 && (g_bf16p.load(std::memory_order_relaxed) & 8))
    // split-bf16 (parity) mode, large batch: the same structure with hi / mid planes as separate ring entries
    hipLaunchKernelGGL(node_update_w_kernel<true>, dim3((G + NODEW_ROWS - 1) / NODEW_ROWS), dim3(NODEW_THREADS), NODEW_LDS, s, a);
  else if (x3 && g_residue_x1)      // bf16 throughput mode: hi . hi products out of the same x3 images
    hipLaunchKernelGGL((node_update_multi_kernel<2, 2>), dim3((G + 31) / 32), dim3(512), NODE_MULTI_LDS_X3(2), s, a);
  else if (x3)                 // every image is an x3 image (node_update_x3_ok below): the multi-tile kernel only
    hipLaunchKernelGGL((node_update_multi_kernel<2, 1>), dim3((G + 31) / 32), dim3(512), NODE_MULTI_LDS_X3(2), s, a);
  else if (G >= 32 * 2 * device_cus())
    hipLaunchKernelGGL((node_update_multi_kernel<2, 0>), dim3((G + 31) / 32), dim3(512), NODE_MULTI_LDS(2), s, a);
  else
    hipLaunchKernelGGL(node_update_kernel, dim3((G + 15) / 16), dim3(512), NODE_TAIL_LDS, s, a);
  return NAMP_OK;
}

// rows_per_table = output rows that share one lookup table of N rows (gather_nodes: N*K; gather_edges: K)
int launch_gather(const float* nodes, const float* nbrs, const int32_t* idx, float* out,
                  long B, int N, int K, int C1, int C2, hipStream_t s, int rows_per_table = 0) {
  const long rows = rows_per_table ? B * rows_per_table : B * N * K;      // rows_per_table: B tables of N rows, that many lookups each
  const int NK = rows_per_table ? rows_per_table : N * K;
  if (((C1 | C2) & 3) == 0) {
    const long total = rows * ((C1 + C2) >> 2);
#ifndef NAMP_GATHER_UNROLL
#define NAMP_GATHER_UNROLL 2      // rows in flight per thread: 1 / 2 / 4 / 8 measure 4.69 / 4.72 / 4.12 / 3.81 TB/s at the cfg3 shape
#endif
    long blocks = (total + 256L * NAMP_GATHER_UNROLL - 1) / (256L * NAMP_GATHER_UNROLL);
#ifndef NAMP_GATHER_MAXBLK
#define NAMP_GATHER_MAXBLK (256 * 16)
#endif
    if (blocks > NAMP_GATHER_MAXBLK) blocks = NAMP_GATHER_MAXBLK;
    blocks = (blocks + 7) & ~7L;                       // a multiple of 8: one contiguous row range per XCD (gather_cat_kernel)
    hipLaunchKernelGGL(gather_cat_kernel<NAMP_GATHER_UNROLL>, dim3((unsigned)blocks), dim3(256), 0, s, nodes, nbrs, idx, out, rows,
                       NK, N, C1, C2);
  } else {
    const long total = rows * (C1 + C2);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gather_cat_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, nodes, nbrs, idx, out, rows,
                       NK, N, C1, C2);
  }
  return NAMP_OK;
}

// bump allocator over the caller's workspace
struct Carver {
  char* base; size_t size, off;
  Carver(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  float* take(size_t floats) {
    size_t bytes = (floats * 4 + 255) & ~size_t(255);
    if (off + bytes > size) return nullptr;
    float* r = (float*)(base + off);
    off += bytes;
    return r;
  }
};

size_t tbl(size_t G) { return ((G * NAMP_HIDDEN * 4 + 255) & ~size_t(255)); }

// ---- persistent forward (encdec_persistent_kernel): switch + in-flight guard ---------------------------------------------
// Every workgroup of a persistent launch must be resident, and two such launches running at once on different streams could
// each hold part of the chip while waiting for the rest.  Launches on ONE stream are ordered; a launch on another stream is
// allowed only once the previous persistent launch has completed (hipEventQuery), otherwise the caller gets the launch chain.
std::mutex g_persist_mutex;
// Off by default: measured on MI355X (profiles/r02_persistent.md) the six in-kernel grid barriers cost more (9 us each) than
// the launch boundaries and h_E round trips they replace (cfg2, x3: 0.360 ms persistent vs 0.332 ms for the chain).
int g_persist_on = [] { const char* e = getenv("NAMP_PERSISTENT"); return (e && e[0] == '1') ? 1 : 0; }();
struct PersistInFlight { hipStream_t stream = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
PersistInFlight g_persist_dev[16];

bool persist_acquire(hipStream_t s, int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  *dev_out = dev;
  std::lock_guard<std::mutex> lk(g_persist_mutex);
  if (!g_persist_on) return false;
  PersistInFlight& f = g_persist_dev[dev];
  if (!f.ev && hipEventCreateWithFlags(&f.ev, hipEventDisableTiming) != hipSuccess) { f.ev = nullptr; return false; }
  if (f.pending && f.stream != s) {
    if (hipEventQuery(f.ev) != hipSuccess) { (void)hipGetLastError(); return false; }
    f.pending = false;
  }
  return true;
}

void persist_mark(hipStream_t s, int dev) {
  std::lock_guard<std::mutex> lk(g_persist_mutex);
  PersistInFlight& f = g_persist_dev[dev];
  if (f.ev && hipEventRecord(f.ev, s) == hipSuccess) { f.stream = s; f.pending = true; }
}

}  // namespace

extern "C" {

int namp_abi_version(void) { return NAMP_ABI_VERSION; }
const char* namp_last_error(void) { return g_err; }

int namp_pack_image(const float* W, int ld, int col0, int out_f, int in_f, float* img, void* stream) {
  if (!W || !img) return fail(NAMP_EINVAL, "namp_pack_image: null pointer");
  REQUIRE(out_f > 0 && in_f > 0 && (out_f % 16) == 0 && (in_f % 16) == 0 && col0 >= 0 && ld >= col0 + in_f,
          "namp_pack_image: out_f=%d in_f=%d must be multiples of 16 inside ld=%d (col0=%d)", out_f, in_f, ld, col0);
  const int total = out_f * in_f;
  hipLaunchKernelGGL(pack_image_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ld, col0,
                     out_f, in_f, img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_image_x3(const float* W, int ld, int col0, void* img, void* stream) {
  REQUIRE_PTR(W); REQUIRE_PTR(img);
  REQUIRE(ld >= 128 && col0 >= 0 && col0 + 128 <= ld, "namp_pack_image_x3: block [128 x 128] at column %d does not fit ld=%d", col0, ld);
  hipLaunchKernelGGL(pack_image_x3_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W, ld, col0, (__bf16*)img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_image_x3_general(const float* W, int ld, int col0, int out_f, int in_f, void* img, void* stream) {
  REQUIRE_PTR(W); REQUIRE_PTR(img);
  REQUIRE(out_f > 0 && in_f > 0 && (out_f % 16) == 0 && (in_f % 32) == 0 && col0 >= 0 && ld >= col0 + in_f,
          "namp_pack_image_x3_general: out_f=%d (x16) in_f=%d (x32) must fit ld=%d at column %d", out_f, in_f, ld, col0);
  const int total = out_f * in_f;
  hipLaunchKernelGGL(pack_image_x3_general_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ld, col0, out_f,
                     in_f, (__bf16*)img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_images(const NampPack* table_dev, int ndesc, int nblocks, void* stream) {
  REQUIRE_PTR(table_dev);
  REQUIRE(ndesc >= 1 && nblocks >= 1, "namp_pack_images: ndesc=%d nblocks=%d", ndesc, nblocks);
  static_assert(sizeof(NampPack) == sizeof(PackDesc), "NampPack (include/namp.h) and PackDesc (namp_kernels.h) must have one layout");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)table_dev, ndesc);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_feat_x3(const float* W, int ld, float* img, void* stream) {
  REQUIRE_PTR(W); REQUIRE_PTR(img);
  REQUIRE(ld >= 5200, "namp_pack_feat_x3: edge_embedding.weight rows hold 5200 columns, ld=%d", ld);
  hipLaunchKernelGGL(pack_feat_x3_kernel, dim3((54 * 12288 + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ld, img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_image_bf16(const float* W, int ld, int col0, void* img, void* stream) {
  if (!W || !img) return fail(NAMP_EINVAL, "namp_pack_image_bf16: null pointer");
  REQUIRE(col0 >= 0 && ld >= col0 + 128, "namp_pack_image_bf16: block [0:128, %d:%d) outside ld=%d", col0, col0 + 128, ld);
  hipLaunchKernelGGL(pack_image_bf16_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W, ld, col0, (__bf16*)img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_pack_image_bf16_32(const float* W, int ld, int col0, void* img, void* stream) {
  if (!W || !img) return fail(NAMP_EINVAL, "namp_pack_image_bf16_32: null pointer");
  REQUIRE(col0 >= 0 && ld >= col0 + 128, "namp_pack_image_bf16_32: block [0:128, %d:%d) outside ld=%d", col0, col0 + 128, ld);
  hipLaunchKernelGGL(pack_image_bf16_32_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W, ld, col0, (__bf16*)img);
  CHECK_LAUNCH();
  return NAMP_OK;
}

// The bf16-storage message launch on its own (per-kernel timing and parity): rows and tables in fragment order B, images from
// namp_pack_image_bf16_32.  mode 0 = encoder message, 1 = decoder message.  partial: [G][TPN][128] K-sums + [G][TPN] weight sums.
int namp_bf16s_message(int mode, const void* hE16, const int32_t* E_idx, const int32_t* mask, const int32_t* rank,
                       const void* Pa16, const void* Pj016, const void* Pj116, const void* W1_img, const void* W2_img, const float* b2,
                       float* partial, int B_dec, int B_enc, int N, int K, void* stream) {
  REQUIRE(mode == 0 || mode == 1, "namp_bf16s_message: mode=%d", mode);
  if (!hE16 || !E_idx || !Pa16 || !Pj016 || !W1_img || !W2_img || !b2 || !partial) return fail(NAMP_EINVAL, "namp_bf16s_message: null pointer");
  REQUIRE(mode == 0 || (rank && Pj116), "namp_bf16s_message: decoder message needs rank and Pj116");
  int rc = check_dims(__func__, B_dec, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE16 = (const __bf16*)hE16; a.E_idx = E_idx; a.mask = mask; a.rank = rank; a.Pa16 = (const __bf16*)Pa16; a.Pj016 = (const __bf16*)Pj016;
  a.Pj116 = (const __bf16*)Pj116; a.W1_img = (const float*)W1_img; a.W2_img = (const float*)W2_img; a.b2 = b2; a.partial = partial;
  a.G = B_dec * N; a.G_enc = B_enc * N; a.N = N; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof_(mode ? NAMP_KIND_DEC_MESSAGE : NAMP_KIND_ENC_MESSAGE, s);
  rc = mode ? launch_edge_bf16s<MODE_DEC_MSG>(a, s) : launch_edge_bf16s<MODE_ENC_MSG>(a, s);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_gather_nodes_f32(const float* nodes, const int32_t* idx, float* out, int B, int N, int K, int C,
                          void* stream) {
  if (!nodes || !idx || !out) return fail(NAMP_EINVAL, "namp_gather_nodes_f32: null pointer");
  REQUIRE(B >= 0 && N >= 0 && K >= 0 && C >= 1, "namp_gather_nodes_f32: bad dims B=%d N=%d K=%d C=%d", B, N, K, C);
  if ((long)B * N * K == 0) return NAMP_OK;               // empty input: nothing to do
  if ((C & 3) == 0) { REQUIRE_PTR(nodes); REQUIRE_PTR(out); }
  ProfScope prof_(NAMP_KIND_GATHER, (hipStream_t)stream);
  launch_gather(nodes, nullptr, idx, out, B, N, K, 0, C, (hipStream_t)stream);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_gather_rows_f32(const float* tables, const int32_t* idx, float* out, long T, int R, int M, int C, void* stream) {
  if (!tables || !idx || !out) return fail(NAMP_EINVAL, "namp_gather_rows_f32: null pointer");
  REQUIRE(T >= 0 && R >= 0 && M >= 0 && C >= 1, "namp_gather_rows_f32: bad dims T=%ld R=%d M=%d C=%d", T, R, M, C);
  if (T * M == 0) return NAMP_OK;
  if ((C & 3) == 0) { REQUIRE_PTR(tables); REQUIRE_PTR(out); }
  ProfScope prof_(NAMP_KIND_GATHER, (hipStream_t)stream);
  // launch_gather(B, N, K, ..., rows_per_table): rows = B*N*K output rows, tables of N rows, one table per rows_per_table rows
  launch_gather(tables, nullptr, idx, out, T, R, 1, 0, C, (hipStream_t)stream, M);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_gather_edges_f32(const float* edges, const int32_t* idx, float* out, int B, int N, int K, int C, void* stream) {
  REQUIRE(B >= 0 && N >= 0 && K >= 0, "namp_gather_edges_f32: bad dims B=%d N=%d K=%d", B, N, K);
  return namp_gather_rows_f32(edges, idx, out, (long)B * N, N, K, C, stream);      // every (b, i) owns the table edges[b, i]
}

int namp_cat_neighbors_nodes_f32(const float* h_nodes, const float* h_neighbors, const int32_t* idx, float* out,
                                 int B, int N, int K, int C1, int C2, void* stream) {
  if (!h_nodes || !h_neighbors || !idx || !out) return fail(NAMP_EINVAL, "namp_cat_neighbors_nodes_f32: null pointer");
  REQUIRE(B >= 0 && N >= 0 && K >= 0 && C1 >= 1 && C2 >= 1, "namp_cat_neighbors_nodes_f32: bad dims");
  if ((long)B * N * K == 0) return NAMP_OK;
  if (((C1 | C2) & 3) == 0) { REQUIRE_PTR(h_nodes); REQUIRE_PTR(h_neighbors); REQUIRE_PTR(out); }
  ProfScope prof_(NAMP_KIND_GATHER, (hipStream_t)stream);
  launch_gather(h_nodes, h_neighbors, idx, out, B, N, K, C1, C2, (hipStream_t)stream);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_node_linear(const float* X, const int32_t* S, int B_out, int B_src, int N, const NampProj* proj,
                     int nproj, const NampProj* pre, void* stream) {
  REQUIRE_PTR(X);
  if (pre) { REQUIRE_PTR(pre->img); OPTIONAL_PTR(pre->bias); OPTIONAL_PTR(pre->out);
             REQUIRE(B_out == B_src, "namp_node_linear: a pre-linear stage needs B_out == B_src"); }
  REQUIRE(proj != nullptr && nproj >= 1 && nproj <= 8, "namp_node_linear: nproj=%d must be in [1,8]", nproj);
  REQUIRE(B_out >= 1 && B_src >= 1 && N >= 1, "namp_node_linear: bad dims");
  for (int i = 0; i < nproj; ++i) {
    REQUIRE_PTR(proj[i].img); REQUIRE_PTR(proj[i].out); OPTIONAL_PTR(proj[i].bias); OPTIONAL_PTR(proj[i].tok);
    REQUIRE(!(proj[i].tok && !S), "namp_node_linear: proj[%d].tok given but S is null", i);
  }
  ProfScope prof_(NAMP_KIND_NODE_LINEAR, (hipStream_t)stream);
  launch_node_linear(X, S, B_out * N, B_src * N, N, proj, nproj, pre, (hipStream_t)stream);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_edge_embed(const float* We_img, const float* We_b, const float* E, float* h_E, int B, int N, int K,
                    void* stream) {
  REQUIRE_PTR(We_img); REQUIRE_PTR(We_b); REQUIRE_PTR(E); REQUIRE_PTR(h_E);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = E; a.hE_out = h_E; a.W1_img = We_img; a.b1 = We_b;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_EDGE_EMBED, (hipStream_t)stream);
  rc = launch_edge<MODE_EMBED>(a, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

// The same launch with the product evaluation chosen by the training precision code: 0 exact fp32 MFMA (fp32 image), 1 split-bf16
// products (x3 image), 2 plain bf16 products (bf16 image) — the edge embedding W_e and its data gradient in the training step ran on
// the fp32 matrix pipe (1/16 of the bf16 rate) whatever the step's precision: 0.39 ms per launch at cfg5.
int namp_edge_embed_prec(const float* We_img, const float* We_b, const float* E, float* h_E, int prec, int B, int N, int K,
                         void* stream) {
  REQUIRE_PTR(We_img); REQUIRE_PTR(We_b); REQUIRE_PTR(E); REQUIRE_PTR(h_E);
  REQUIRE(prec >= 0 && prec <= 2, "namp_edge_embed_prec: precision code %d (0 fp32, 1 split-bf16, 2 bf16)", prec);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = E; a.hE_out = h_E; a.W1_img = We_img; a.b1 = We_b;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_EDGE_EMBED, (hipStream_t)stream);
  if (prec == 1) rc = launch_edge<MODE_EMBED, 0, PREC_X3>(a, (hipStream_t)stream);
  else if (prec == 2) rc = launch_edge<MODE_EMBED, 0, PREC_BF16>(a, (hipStream_t)stream);
  else rc = launch_edge<MODE_EMBED>(a, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_edge_embed_ln(const float* We_img, const float* We_b, const float* ln_g, const float* ln_b, const float* Y, float* h_E, int prec,
                       int B, int N, int K, void* stream) {
  REQUIRE_PTR(We_img); REQUIRE_PTR(We_b); REQUIRE_PTR(ln_g); REQUIRE_PTR(ln_b); REQUIRE_PTR(Y); REQUIRE_PTR(h_E);
  REQUIRE(prec >= 0 && prec <= 2, "namp_edge_embed_ln: precision code %d (0 fp32, 1 split-bf16, 2 bf16)", prec);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = Y; a.hE_out = h_E; a.W1_img = We_img; a.b1 = We_b; a.ln_g = ln_g; a.ln_b = ln_b;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_EDGE_EMBED, (hipStream_t)stream);
  if (prec == 1) rc = launch_edge<MODE_EMBED, 0, PREC_X3>(a, (hipStream_t)stream);
  else if (prec == 2) rc = launch_edge<MODE_EMBED, 0, PREC_BF16>(a, (hipStream_t)stream);
  else rc = launch_edge<MODE_EMBED>(a, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

// namp_node_linear with split-bf16 (1) or plain bf16 (2) products: every image is then an x3 image (namp_pack_image_x3; the bf16
// evaluation reads its hi plane only).  No pre-stage, no token tables (the training path's hoisted first-layer tables and their data
// gradients: [B*N,128] x [128,128] products that ran as exact fp32 MFMA).
int namp_node_linear_prec(const float* X, int G, const NampProj* proj, int nproj, int prec, void* stream) {
  REQUIRE_PTR(X);
  REQUIRE(proj != nullptr && nproj >= 1 && nproj <= 8, "namp_node_linear_prec: nproj=%d must be in [1,8]", nproj);
  REQUIRE(G >= 1, "namp_node_linear_prec: G=%d", G);
  REQUIRE(prec >= 0 && prec <= 2, "namp_node_linear_prec: precision code %d (0 fp32, 1 split-bf16, 2 bf16)", prec);
  for (int i = 0; i < nproj; ++i) {
    REQUIRE_PTR(proj[i].img); REQUIRE_PTR(proj[i].out); OPTIONAL_PTR(proj[i].bias);
    REQUIRE(proj[i].tok == nullptr, "namp_node_linear_prec: token tables are not supported");
  }
  ProfScope prof_(NAMP_KIND_NODE_LINEAR, (hipStream_t)stream);
  const bool prev = g_residue_x1;
  g_residue_x1 = (prec == 2);
  launch_node_linear(X, nullptr, G, G, G, proj, nproj, nullptr, (hipStream_t)stream, prec != 0);
  g_residue_x1 = prev;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_node_linear_sum(const float* const* X, const float* const* img, int n, float* out, int G, int prec, void* stream) {
  REQUIRE(X && img, "namp_node_linear_sum: null pointer table");
  REQUIRE(n >= 1 && n <= 8, "namp_node_linear_sum: n=%d must be in [1,8]", n);
  REQUIRE(G >= 1, "namp_node_linear_sum: G=%d", G);
  REQUIRE(prec >= 0 && prec <= 2, "namp_node_linear_sum: precision code %d (0 fp32, 1 split-bf16, 2 bf16)", prec);
  REQUIRE_PTR(out);
  NodeLinearSumArgs a = {};
  for (int q = 0; q < 8; ++q) {
    const int s_ = q < n ? q : 0;
    REQUIRE_PTR(X[s_]); REQUIRE_PTR(img[s_]);
    a.X[q] = X[s_]; a.img[q] = img[s_];
  }
  a.out = out; a.G = G; a.n = n;
  ProfScope prof_(NAMP_KIND_NODE_LINEAR, (hipStream_t)stream);
  const dim3 grid(((G + 15) / 16 + 3) / 4);
  if (prec == 2) hipLaunchKernelGGL(node_linear_sum_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (prec == 1) hipLaunchKernelGGL(node_linear_sum_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(node_linear_sum_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_enc_message(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* mask,
                     const int32_t* mask_attend, const float* Pa, const float* Pc, float* partial, int B, int N,
                     int K, void* stream) {
  REQUIRE(w != nullptr, "namp_enc_message: null weights");
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(partial);
  REQUIRE_PTR(w->W1b_img); REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_enc_message: null E_idx");
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.Pa = Pa; a.Pj0 = Pc;
  const int prec = prec_of(w->flags);
  if (prec == PREC_BF16) { REQUIRE_PTR(w->W1b_bimg); REQUIRE_PTR(w->W2_bimg); REQUIRE_PTR(w->W3_bimg); }
  if (prec == PREC_X3) { REQUIRE_PTR(w->W1b_ximg); REQUIRE_PTR(w->W2_ximg); REQUIRE_PTR(w->W3_ximg); }
  a.W1_img = pick_img(prec, w->W1b_img, w->W1b_bimg, w->W1b_ximg); a.W2_img = pick_img(prec, w->W2_img, w->W2_bimg, w->W2_ximg);
  a.W3_img = pick_img(prec, w->W3_img, w->W3_bimg, w->W3_ximg);
  a.b2 = w->b2; a.b3 = w->b3;
  a.partial = partial; a.G = a.G_enc = B * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_ENC_MESSAGE, (hipStream_t)stream);
  rc = launch_edge_prec<MODE_ENC_MSG, 0>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_enc_edge_update(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx, const float* Pa,
                         const float* Pc, float* h_E_out, int B, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_enc_edge_update: null weights");
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(h_E_out);
  REQUIRE_PTR(w->W11b_img); REQUIRE_PTR(w->W12_img); REQUIRE_PTR(w->W13_img); REQUIRE_PTR(w->b12); REQUIRE_PTR(w->b13);
  REQUIRE_PTR(w->ln3_g); REQUIRE_PTR(w->ln3_b);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_enc_edge_update: null E_idx");
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = h_E; a.hE_out = h_E_out; a.E_idx = E_idx; a.Pa = Pa; a.Pj0 = Pc;
  const int prec = prec_of(w->flags);
  if (prec == PREC_BF16) { REQUIRE_PTR(w->W11b_bimg); REQUIRE_PTR(w->W12_bimg); REQUIRE_PTR(w->W13_bimg); }
  if (prec == PREC_X3) { REQUIRE_PTR(w->W11b_ximg); REQUIRE_PTR(w->W12_ximg); REQUIRE_PTR(w->W13_ximg); }
  a.W1_img = pick_img(prec, w->W11b_img, w->W11b_bimg, w->W11b_ximg); a.W2_img = pick_img(prec, w->W12_img, w->W12_bimg, w->W12_ximg);
  a.W3_img = pick_img(prec, w->W13_img, w->W13_bimg, w->W13_ximg);
  a.b2 = w->b12; a.b3 = w->b13;
  a.ln_g = w->ln3_g; a.ln_b = w->ln3_b; a.G = a.G_enc = B * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_ENC_EDGE, (hipStream_t)stream);
  rc = launch_edge_prec<MODE_ENC_EDGE, 0>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_train_edge_fwd(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                        const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                        const float* W2_img, const float* W3_img, const float* b2, const float* b3,
                        const float* ln_g, const float* ln_b, float drop_p, uint32_t drop_seed, float* out,
                        int x3, int B, int N, int K, void* stream) {
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "namp_train_edge_fwd: drop_p=%g must be in [0,1)", (double)drop_p);
  REQUIRE((ln_g == nullptr) == (ln_b == nullptr), "namp_train_edge_fwd: ln_g and ln_b go together");
  REQUIRE(mode == 2 || (ln_g == nullptr && drop_p == 0.f), "namp_train_edge_fwd: LayerNorm3 / dropout belong to mode 2");
  OPTIONAL_PTR(ln_g); OPTIONAL_PTR(ln_b);
  REQUIRE(mode >= 0 && mode <= 2, "namp_train_edge_fwd: mode=%d must be 0 (enc message), 1 (dec message) or 2 (enc edge)", mode);
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pj0); REQUIRE_PTR(W1_img); REQUIRE_PTR(W2_img); REQUIRE_PTR(W3_img);
  REQUIRE_PTR(b2); REQUIRE_PTR(b3); REQUIRE_PTR(out);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_train_edge_fwd: null E_idx");
  if (mode == MODE_DEC_MSG) { REQUIRE_PTR(Pj1); REQUIRE(rank != nullptr, "namp_train_edge_fwd: decoder message needs rank"); }
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  EdgeArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.rank = rank; a.Pa = Pa; a.Pj0 = Pj0; a.Pj1 = Pj1;
  a.W1_img = W1_img; a.W2_img = W2_img; a.W3_img = W3_img; a.b2 = b2; a.b3 = b3;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  const int prec = x3 == 2 ? PREC_BF16 : x3 ? PREC_X3 : PREC_F32;        // x3 = 2: plain bf16 products (mixed-precision training)
  if (mode == MODE_ENC_MSG) { a.partial = out; ProfScope p_(NAMP_KIND_ENC_MESSAGE, s); rc = launch_edge_prec<MODE_ENC_MSG, 0>(a, prec, s); }
  else if (mode == MODE_DEC_MSG) { a.partial = out; ProfScope p_(NAMP_KIND_DEC_MESSAGE, s); rc = launch_edge_prec<MODE_DEC_MSG, 0>(a, prec, s); }
  else {
    a.hE_out = out; a.ln_g = ln_g; a.ln_b = ln_b;                                 // ln_g null: bare message
    if (drop_p > 0.f) { a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); a.drop_seed = drop_seed; a.drop_scale = 1.0f / (1.0f - drop_p); }
    ProfScope p_(NAMP_KIND_ENC_EDGE, s);
    rc = launch_edge_prec<MODE_ENC_EDGE, 0>(a, prec, s);
  }
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_node_update(const float* ln1_g, const float* ln1_b, const float* Win_img, const float* b_in,
                     const float* Wout_img, const float* b_out, const float* ln2_g, const float* ln2_b,
                     const float* h_V, const float* partial, const float* m3_img, const float* m3_b, const int32_t* mask,
                     float* h_V_out, const NampProj* proj, int nproj, const int32_t* S, int G, int K, void* stream) {
  REQUIRE_PTR(ln1_g); REQUIRE_PTR(ln1_b); REQUIRE_PTR(Win_img); REQUIRE_PTR(b_in); REQUIRE_PTR(Wout_img);
  REQUIRE_PTR(b_out); REQUIRE_PTR(ln2_g); REQUIRE_PTR(ln2_b); REQUIRE_PTR(h_V); REQUIRE_PTR(h_V_out);
  OPTIONAL_PTR(partial); OPTIONAL_PTR(m3_img); OPTIONAL_PTR(m3_b);
  REQUIRE(!m3_img || (partial && m3_b), "namp_node_update: m3_img needs partial (K-sums + weight sums) and m3_b");
  REQUIRE(G >= 1 && K >= 1 && K <= NAMP_MAX_K, "namp_node_update: bad dims G=%d K=%d", G, K);
  int rc = check_proj(__func__, proj, nproj, S);
  if (rc) return rc;
  ProfScope prof_(NAMP_KIND_NODE_UPDATE, (hipStream_t)stream);
  rc = launch_node_update(ln1_g, ln1_b, Win_img, b_in, Wout_img, b_out, ln2_g, ln2_b, h_V, partial, m3_img, m3_b, mask, h_V_out,
                              proj, nproj, S, G, (K + 15) / 16, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

// Residue update of the unfused (large-batch) paths: as split-bf16 products (multi-tile kernel) when the precision is not exact
// fp32 and the layer carries x3 images of the FFN and of every projected block; exact fp32 otherwise.
static int node_update_auto(int64_t flags, const float* Win_ximg, const float* Wout_ximg, const float* const* proj_ximg,
                            const float* ln1_g, const float* ln1_b, const float* Win_img, const float* b_in,
                            const float* Wout_img, const float* b_out, const float* ln2_g, const float* ln2_b,
                            const float* h_V, const float* partial, const float* m3_img, const float* m3_ximg, const float* m3_b,
                            const int32_t* mask, float* h_V_out,
                            const NampProj* proj, int nproj, const int32_t* S, int G, int K, void* stream) {
#ifndef NAMP_NODE_X3_MIN_RESIDUES
#define NAMP_NODE_X3_MIN_RESIDUES 2500   // the whole unfused regime: 3-12 % per forward at 3,000-16,000 residues (tools/batch_sweep.py)
#endif
  bool x3 = prec_of(flags) != PREC_F32 && Win_ximg && Wout_ximg && G >= NAMP_NODE_X3_MIN_RESIDUES && nproj <= 8 && (!m3_img || m3_ximg);
  for (int i = 0; x3 && i < nproj; ++i) x3 = proj_ximg && proj_ximg[i] != nullptr;
  if (!x3)
    return namp_node_update(ln1_g, ln1_b, Win_img, b_in, Wout_img, b_out, ln2_g, ln2_b, h_V, partial, m3_img, m3_b, mask, h_V_out,
                            proj, nproj, S, G, K, stream);
  OPTIONAL_PTR(m3_ximg); OPTIONAL_PTR(m3_b);
  REQUIRE_PTR(ln1_g); REQUIRE_PTR(ln1_b); REQUIRE_PTR(Win_ximg); REQUIRE_PTR(b_in); REQUIRE_PTR(Wout_ximg); REQUIRE_PTR(b_out);
  REQUIRE_PTR(ln2_g); REQUIRE_PTR(ln2_b); REQUIRE_PTR(h_V); REQUIRE_PTR(h_V_out); OPTIONAL_PTR(partial);
  REQUIRE(G >= 1 && K >= 1 && K <= NAMP_MAX_K, "node_update: bad dims G=%d K=%d", G, K);
  NampProj px[8];
  for (int i = 0; i < nproj; ++i) {
    px[i] = proj[i]; px[i].img = proj_ximg[i];
    REQUIRE_PTR(px[i].img); REQUIRE_PTR(px[i].out); OPTIONAL_PTR(px[i].bias); OPTIONAL_PTR(px[i].tok);
    REQUIRE(!(px[i].tok && !S), "node_update: proj[%d].tok given but S is null", i);
  }
  ProfScope prof_(NAMP_KIND_NODE_UPDATE, (hipStream_t)stream);
  int rc = launch_node_update(ln1_g, ln1_b, Win_ximg, b_in, Wout_ximg, b_out, ln2_g, ln2_b, h_V, partial, m3_img ? m3_ximg : nullptr,
                              m3_b, mask, h_V_out, px, nproj, S, G, (K + 15) / 16, (hipStream_t)stream, true);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_dec_message(const NampDecLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* rank,
                     const float* Pa, const float* Pbw, const float* Pfw, float* partial, int B_dec, int B_enc,
                     int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_dec_message: null weights");
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pbw); REQUIRE_PTR(Pfw); REQUIRE_PTR(partial);
  REQUIRE_PTR(w->W1e_img); REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  if (!E_idx || !rank) return fail(NAMP_EINVAL, "namp_dec_message: null E_idx / rank");
  int rc = check_dims(__func__, B_dec, N, K);
  if (rc) return rc;
  REQUIRE(B_enc >= 1 && B_dec % B_enc == 0, "namp_dec_message: B_dec=%d must be a multiple of B_enc=%d", B_dec, B_enc);
  EdgeArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.rank = rank; a.Pa = Pa; a.Pj0 = Pbw; a.Pj1 = Pfw;
  const int prec = prec_of(w->flags);
  if (prec == PREC_BF16) { REQUIRE_PTR(w->W1e_bimg); REQUIRE_PTR(w->W2_bimg); REQUIRE_PTR(w->W3_bimg); }
  if (prec == PREC_X3) { REQUIRE_PTR(w->W1e_ximg); REQUIRE_PTR(w->W2_ximg); REQUIRE_PTR(w->W3_ximg); }
  a.W1_img = pick_img(prec, w->W1e_img, w->W1e_bimg, w->W1e_ximg); a.W2_img = pick_img(prec, w->W2_img, w->W2_bimg, w->W2_ximg);
  a.W3_img = pick_img(prec, w->W3_img, w->W3_bimg, w->W3_ximg);
  a.b2 = w->b2; a.b3 = w->b3;
  a.partial = partial; a.G = B_dec * N; a.G_enc = B_enc * N; a.N = N; a.K = K;
  ProfScope prof_(NAMP_KIND_DEC_MESSAGE, (hipStream_t)stream);
  rc = launch_edge_prec<MODE_DEC_MSG, 0>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_enc_message_update(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* mask,
                            const int32_t* mask_attend, const float* Pa, const float* Pc, const float* h_V,
                            float* h_V_out, const NampProj* proj, int nproj, int B, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_enc_message_update: null weights");
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(h_V); REQUIRE_PTR(h_V_out);
  REQUIRE_PTR(w->W1b_img); REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  REQUIRE_PTR(w->Win_img); REQUIRE_PTR(w->Wout_img); REQUIRE_PTR(w->b_in); REQUIRE_PTR(w->b_out);
  REQUIRE_PTR(w->ln1_g); REQUIRE_PTR(w->ln1_b); REQUIRE_PTR(w->ln2_g); REQUIRE_PTR(w->ln2_b);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_enc_message_update: null E_idx");
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  if ((rc = check_proj(__func__, proj, nproj, nullptr))) return rc;
  EdgeArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.Pa = Pa; a.Pj0 = Pc;
  const int prec = prec_of(w->flags);
  if (prec == PREC_BF16) { REQUIRE_PTR(w->W1b_bimg); REQUIRE_PTR(w->W2_bimg); REQUIRE_PTR(w->W3_bimg); }
  if (prec == PREC_X3) { REQUIRE_PTR(w->W1b_ximg); REQUIRE_PTR(w->W2_ximg); REQUIRE_PTR(w->W3_ximg); }
  a.W1_img = pick_img(prec, w->W1b_img, w->W1b_bimg, w->W1b_ximg); a.W2_img = pick_img(prec, w->W2_img, w->W2_bimg, w->W2_ximg);
  a.W3_img = pick_img(prec, w->W3_img, w->W3_bimg, w->W3_ximg);
  a.b2 = w->b2; a.b3 = w->b3;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  fill_tail(a.tail, w->ln1_g, w->ln1_b, w->Win_img, w->b_in, w->Wout_img, w->b_out, w->ln2_g, w->ln2_b, h_V, mask,
            h_V_out, proj, nproj, nullptr);
  a.tail.m3_img = w->W3_img; a.tail.m3_b = w->b3;                    // layer 3 of the message MLP runs per residue (NodeTail)
  ProfScope prof_(NAMP_KIND_ENC_MESSAGE, (hipStream_t)stream);
  rc = launch_edge_tail<MODE_ENC_MSG>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_enc_edge_message_update(const NampEncLayerW* w_prev, const float* ePa, const float* ePc, float* h_E,
                                 const NampEncLayerW* w, const int32_t* E_idx, const int32_t* mask,
                                 const int32_t* mask_attend, const float* Pa, const float* Pc, const float* h_V,
                                 float* h_V_out, const NampProj* proj, int nproj, int B, int N, int K, void* stream) {
  REQUIRE(w_prev != nullptr && w != nullptr, "namp_enc_edge_message_update: null weights");
  REQUIRE((w->flags & NAMP_FLAG_BF16) == 0 && (w_prev->flags & NAMP_FLAG_BF16) == 0,
          "namp_enc_edge_message_update: fp32-class precisions only (use the separate launches in bf16 mode)");
  const int prec = prec_of(w->flags);
  REQUIRE(prec == prec_of(w_prev->flags), "namp_enc_edge_message_update: both layers must use the same precision");
  if (prec == PREC_X3) {
    REQUIRE_PTR(w_prev->W11b_ximg); REQUIRE_PTR(w_prev->W12_ximg); REQUIRE_PTR(w_prev->W13_ximg);
    REQUIRE_PTR(w->W1b_ximg); REQUIRE_PTR(w->W2_ximg); REQUIRE_PTR(w->W3_ximg);
  }
  REQUIRE_PTR(h_E); REQUIRE_PTR(ePa); REQUIRE_PTR(ePc); REQUIRE_PTR(Pa); REQUIRE_PTR(Pc); REQUIRE_PTR(h_V); REQUIRE_PTR(h_V_out);
  REQUIRE_PTR(w_prev->W11b_img); REQUIRE_PTR(w_prev->W12_img); REQUIRE_PTR(w_prev->W13_img); REQUIRE_PTR(w_prev->b12);
  REQUIRE_PTR(w_prev->b13); REQUIRE_PTR(w_prev->ln3_g); REQUIRE_PTR(w_prev->ln3_b);
  REQUIRE_PTR(w->W1b_img); REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  REQUIRE_PTR(w->Win_img); REQUIRE_PTR(w->Wout_img); REQUIRE_PTR(w->b_in); REQUIRE_PTR(w->b_out);
  REQUIRE_PTR(w->ln1_g); REQUIRE_PTR(w->ln1_b); REQUIRE_PTR(w->ln2_g); REQUIRE_PTR(w->ln2_b);
  if (!E_idx) return fail(NAMP_EINVAL, "namp_enc_edge_message_update: null E_idx");
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  if ((rc = check_proj(__func__, proj, nproj, nullptr))) return rc;
  for (int i = 0; i < nproj; ++i)
    REQUIRE(proj[i].out != ePa && proj[i].out != ePc && proj[i].out != Pa && proj[i].out != Pc,
            "namp_enc_edge_message_update: projection %d writes a table this launch still gathers", i);
  EdgeArgs a = {};
  a.hE = h_E; a.hE_out = h_E; a.E_idx = E_idx; a.mask = mask; a.mask_attend = mask_attend; a.Pa = Pa; a.Pj0 = Pc;
  a.ePa = ePa; a.ePj = ePc;
  a.eW1_img = pick_img(prec, w_prev->W11b_img, nullptr, w_prev->W11b_ximg); a.eW2_img = pick_img(prec, w_prev->W12_img, nullptr, w_prev->W12_ximg);
  a.eW3_img = pick_img(prec, w_prev->W13_img, nullptr, w_prev->W13_ximg);
  a.eb2 = w_prev->b12; a.eb3 = w_prev->b13; a.ln_g = w_prev->ln3_g; a.ln_b = w_prev->ln3_b;
  a.W1_img = pick_img(prec, w->W1b_img, nullptr, w->W1b_ximg); a.W2_img = pick_img(prec, w->W2_img, nullptr, w->W2_ximg);
  a.W3_img = pick_img(prec, w->W3_img, nullptr, w->W3_ximg); a.b2 = w->b2; a.b3 = w->b3;
  a.G = a.G_enc = B * N; a.N = N; a.K = K;
  fill_tail(a.tail, w->ln1_g, w->ln1_b, w->Win_img, w->b_in, w->Wout_img, w->b_out, w->ln2_g, w->ln2_b, h_V, mask,
            h_V_out, proj, nproj, nullptr);
  a.tail.m3_img = w->W3_img; a.tail.m3_b = w->b3;
  ProfScope prof_(NAMP_KIND_ENC_EDGE_MESSAGE, (hipStream_t)stream);
  rc = launch_edge_tail_fused<MODE_ENC_MSG>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_dec_message_update(const NampDecLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* rank,
                            const float* Pa, const float* Pbw, const float* Pfw, const float* h_V, const int32_t* mask,
                            float* h_V_out, const NampProj* proj, int nproj, const int32_t* S,
                            const float* head_w, const float* head_b, float* log_probs, float* logits, int vocab,
                            int B_dec, int B_enc, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_dec_message_update: null weights");
  if (head_w) {
    REQUIRE_PTR(head_w);
    REQUIRE(head_b && log_probs && vocab >= 1 && vocab <= 64, "namp_dec_message_update: output head needs head_b, log_probs and 1 <= vocab <= 64");
  }
  REQUIRE_PTR(h_E); REQUIRE_PTR(Pa); REQUIRE_PTR(Pbw); REQUIRE_PTR(Pfw); REQUIRE_PTR(h_V); REQUIRE_PTR(h_V_out);
  REQUIRE_PTR(w->W1e_img); REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  REQUIRE_PTR(w->Win_img); REQUIRE_PTR(w->Wout_img); REQUIRE_PTR(w->b_in); REQUIRE_PTR(w->b_out);
  REQUIRE_PTR(w->ln1_g); REQUIRE_PTR(w->ln1_b); REQUIRE_PTR(w->ln2_g); REQUIRE_PTR(w->ln2_b);
  if (!E_idx || !rank) return fail(NAMP_EINVAL, "namp_dec_message_update: null E_idx / rank");
  int rc = check_dims(__func__, B_dec, N, K);
  if (rc) return rc;
  REQUIRE(B_enc >= 1 && B_dec % B_enc == 0, "namp_dec_message_update: B_dec=%d must be a multiple of B_enc=%d", B_dec, B_enc);
  if ((rc = check_proj(__func__, proj, nproj, S))) return rc;
  EdgeArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.rank = rank; a.Pa = Pa; a.Pj0 = Pbw; a.Pj1 = Pfw;
  const int prec = prec_of(w->flags);
  if (prec == PREC_BF16) { REQUIRE_PTR(w->W1e_bimg); REQUIRE_PTR(w->W2_bimg); REQUIRE_PTR(w->W3_bimg); }
  if (prec == PREC_X3) { REQUIRE_PTR(w->W1e_ximg); REQUIRE_PTR(w->W2_ximg); REQUIRE_PTR(w->W3_ximg); }
  a.W1_img = pick_img(prec, w->W1e_img, w->W1e_bimg, w->W1e_ximg); a.W2_img = pick_img(prec, w->W2_img, w->W2_bimg, w->W2_ximg);
  a.W3_img = pick_img(prec, w->W3_img, w->W3_bimg, w->W3_ximg);
  a.b2 = w->b2; a.b3 = w->b3;
  a.G = B_dec * N; a.G_enc = B_enc * N; a.N = N; a.K = K;
  fill_tail(a.tail, w->ln1_g, w->ln1_b, w->Win_img, w->b_in, w->Wout_img, w->b_out, w->ln2_g, w->ln2_b, h_V, mask,
            h_V_out, proj, nproj, S);
  a.tail.m3_img = w->W3_img; a.tail.m3_b = w->b3;
  a.tail.head_w = head_w; a.tail.head_b = head_b; a.tail.log_probs = log_probs; a.tail.logits = logits; a.tail.vocab = vocab;
  ProfScope prof_(NAMP_KIND_DEC_MESSAGE, (hipStream_t)stream);
  rc = launch_edge_tail<MODE_DEC_MSG>(a, prec, (hipStream_t)stream);
  if (rc) return rc;
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_logits_log_softmax(const float* Wout_w, const float* Wout_b, const float* h_V, float* log_probs,
                            float* logits, int G, int vocab, void* stream) {
  REQUIRE_PTR(Wout_w); REQUIRE_PTR(h_V);
  if (!Wout_b || !log_probs) return fail(NAMP_EINVAL, "namp_logits_log_softmax: null pointer");
  REQUIRE(G >= 1 && vocab >= 1 && vocab <= 64, "namp_logits_log_softmax: vocab=%d must be in [1,64]", vocab);
  ProfScope prof_(NAMP_KIND_LOGITS, (hipStream_t)stream);
  if (G >= 4096 && vocab <= 48 && aligned16(Wout_w) && aligned16(h_V) && aligned16(log_probs) && (!logits || aligned16(logits))) {
    // large batches: 16-row tiles on the exact-fp32 matrix pipe (namp_order.h: logits_mfma_kernel)
    int rc = ensure_attributes();
    if (rc) return rc;
    const int ntile = (G + 15) / 16;
    int grid = (ntile + 3) / 4;
    if (grid > 4 * device_cus()) grid = 4 * device_cus();
    hipLaunchKernelGGL(logits_mfma_kernel, dim3(grid), dim3(256), LOGITS_MFMA_LDS(vocab), (hipStream_t)stream, h_V, Wout_w, Wout_b, log_probs,
                       logits, G, vocab);
    CHECK_LAUNCH();
    return NAMP_OK;
  }
  int per_wg = (G + 2 * device_cus() - 1) / (2 * device_cus());            // ~2 workgroups per CU; W_out is staged once per workgroup
  per_wg = (per_wg + 15) / 16 * 16;                                         // 4 waves x 4 residues per pass
  hipLaunchKernelGGL(logits_kernel, dim3((G + per_wg - 1) / per_wg), dim3(256), (size_t)32 * vocab * 16, (hipStream_t)stream, h_V,
                     Wout_w, Wout_b, log_probs, logits, G, vocab, per_wg);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_profile_enable(int on) {
  g_prof_on = (on != 0);
  return NAMP_OK;
}

int namp_profile_collect(float* ms_per_kind, int32_t* launches_per_kind, int nkinds) {
  if (!ms_per_kind || !launches_per_kind || nkinds < NAMP_NUM_KINDS)
    return fail(NAMP_EINVAL, "namp_profile_collect: need arrays of at least %d entries", NAMP_NUM_KINDS);
  for (int i = 0; i < nkinds; ++i) { ms_per_kind[i] = 0.f; launches_per_kind[i] = 0; }
  int rc = NAMP_OK;
  for (ProfRec& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
      rc = fail(NAMP_ELAUNCH, "namp_profile_collect: event query failed");
    else { ms_per_kind[r.kind] += ms; launches_per_kind[r.kind] += 1; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  return rc;
}

size_t namp_featurize_workspace_bytes(int B, int L) {
  if (B < 1 || L < 1) return 0;
  const size_t G = (size_t)B * L;
  return ((G * 54 * 4 + 255) & ~size_t(255)) + ((G * 4 + 255) & ~size_t(255)) + ((G * 3 * 4 + 255) & ~size_t(255)) + 4096;
}

// extra workspace bytes with which namp_featurize can split its edge-feature launch in two parts when only one of E / h_E is requested
// (0 when the batch is too large for the split to apply)
size_t namp_featurize_split_bytes(int B, int L, int top_k) {
  if (B < 1 || L < 1 || top_k < 1) return 0;
  const int K = top_k < L ? top_k : L;
  const int G = B * L;
  EdgeGeom e = edge_geom(G, K);
  if (G <= device_cus()) e.grid = G;
  if (e.grid > device_cus()) return 0;
  return 3 * ((((size_t)G * K * NAMP_HIDDEN * 4 + 255) & ~size_t(255)) + 256) + 2048;
}

static int featurize_impl(const NampModelW* w, const float* X, const int32_t* X_m, const int32_t* mask, const int32_t* R_idx,
                          const int32_t* chain_labels, const int32_t* protein_mask, const int32_t* dna_mask,
                          const int32_t* rna_mask, int top_k, int ref_atom, int32_t* E_idx, float* E, float* h_E, void* ws,
                          size_t ws_bytes, int B, int L, const OrderJob* job, void* stream) {
  REQUIRE(w != nullptr, "namp_featurize: null weights");
  REQUIRE_PTR(X); REQUIRE_PTR(ws); OPTIONAL_PTR(E); OPTIONAL_PTR(h_E);
  OPTIONAL_PTR(w->feat.Wedge_ximg);
  if (!w->feat.Wedge_ximg) REQUIRE_PTR(w->feat.Wedge_img);
  REQUIRE_PTR(w->feat.pos_w); REQUIRE_PTR(w->feat.pos_b);
  OPTIONAL_PTR(w->feat.ln_g); OPTIONAL_PTR(w->feat.ln_b);
  REQUIRE((w->feat.ln_g == nullptr) == (w->feat.ln_b == nullptr), "namp_featurize: norm_edges weight and bias go together");
  REQUIRE(w->feat.ln_g || !h_E, "namp_featurize: h_E needs norm_edges (pre-LayerNorm output is E only)");
  if (!X_m || !mask || !R_idx || !chain_labels || !protein_mask || !dna_mask || !rna_mask || !E_idx)
    return fail(NAMP_EINVAL, "namp_featurize: null pointer argument");
  REQUIRE(E || h_E, "namp_featurize: at least one of E / h_E must be requested");
  REQUIRE(B >= 1 && L >= 1 && top_k >= 1, "namp_featurize: bad dims B=%d L=%d top_k=%d", B, L, top_k);
  REQUIRE(L <= 8192, "namp_featurize: L=%d exceeds the 8192 residues the in-LDS neighbour sort handles", L);
  REQUIRE(ref_atom >= 0 && ref_atom < 16, "namp_featurize: ref_atom=%d out of range", ref_atom);
  const int K = top_k < L ? top_k : L;
  int rc = check_dims(__func__, B, L, K);
  if (rc) return rc;
  if ((rc = ensure_attributes())) return rc;
  if (h_E) { if (w->feat.Wedge_ximg) REQUIRE_PTR(w->We_ximg); else REQUIRE_PTR(w->We_img); REQUIRE_PTR(w->We_b); }
  const int G = B * L;
  hipStream_t s = (hipStream_t)stream;
  Carver c(ws, ws_bytes);
  float* X18 = c.take((size_t)G * 54);
  uint32_t* M18 = (uint32_t*)c.take((size_t)G);
  float* P = c.take((size_t)G * 3);
  if (!P) return fail(NAMP_EWORKSPACE, "namp_featurize: workspace too small (%zu bytes)", ws_bytes);
  {
    ProfScope prof_(NAMP_KIND_FEATURES, s);
    hipLaunchKernelGGL(prep_atoms_kernel, dim3((G + 255) / 256), dim3(256), 0, s, X, X_m, protein_mask, dna_mask, rna_mask,
                       X18, M18, P, G, ref_atom);
    FeatArgs a = {};
    a.X18 = X18; a.M18 = M18; a.E_idx = E_idx; a.R_idx = R_idx; a.chain = chain_labels;
    const bool x3 = w->feat.Wedge_ximg != nullptr;
    a.Wedge_img = x3 ? w->feat.Wedge_ximg : w->feat.Wedge_img; a.pos_w = w->feat.pos_w; a.pos_b = w->feat.pos_b; a.ln_g = w->feat.ln_g; a.ln_b = w->feat.ln_b;
    a.We_img = h_E ? (x3 ? w->We_ximg : w->We_img) : nullptr; a.We_b = h_E ? w->We_b : nullptr; a.E_out = E; a.hE_out = h_E;
    a.G = G; a.L = L; a.K = K;
    EdgeGeom e = edge_geom(G, K);
    a.TPN = e.tpn;
    // A chain of up to ~250 residues (the design call of the README demos: cfg1, 97 residues): at 12 waves per workgroup the launch is a handful of
    // workgroups that each walk the UNION of their 4-6 residues' atom-pair chunks (a protein residue needs 10 of the 54, a nucleotide 39): 169 us at
    // 97 residues.  One residue per workgroup — its own chunks only, every workgroup on a CU of its own — 115 us.  Beyond one round of the chip it loses
    // (1,000 residues: 177 -> 281 us: a workgroup's 140 KB of LDS allow one per CU).
    if (G <= device_cus()) { e.npw = 1; e.nwaves = e.tpn; e.grid = G; }
    // (Round 6 tried two row tiles per wave beyond that — every weight fragment read from LDS feeding both, six waves per workgroup, bit-identical
    // rows: 159 -> 212 us at 1,000 residues, 2.42 -> 3.18 ms at a 31,100-token batch (profiles/r06b): the launch is not bound by its LDS reads
    // either; with half the waves it hides less of everything else.  Removed.)
    // At most one round of the chip (one complex of up to ~1,000 residues): the launch lasts as long as its longest chain of atom-pair chunks —
    // 39-54 for a workgroup that holds a nucleotide, ~10 for protein residues, ~2.5 us each — while the protein workgroups' CUs idle.  In parts:
    // the long blocks (feat_rank_blocks, an extra workgroup of the neighbour-search launch) become 2-4 items each, every n-th chunk, leaving
    // pre-LayerNorm partial rows that feat_finish_kernel adds, normalises and embeds; the short ones stay one item; longest first in dispatch
    // order.  Partial rows go to the output buffers themselves (E and h_E both asked for: two parts without workspace), the
    // others to the workspace's tail when the caller sized it with namp_featurize_split_bytes.  Bit 5 of the namp_set_bf16p mask (A/B switch).
    dim3 grid(e.grid);
    FeatRank fr = {};
    if ((g_bf16p.load(std::memory_order_relaxed) & 32) && e.grid <= device_cus() && e.grid <= 256 && w->feat.ln_g) {
      int np = 2 + ((g_bf16p.load(std::memory_order_relaxed) >> 6) & 3);          // bits 6-7: one or two parts more
      if (np > 4) np = 4;
      const size_t rows = (size_t)G * K * NAMP_HIDDEN;
      int32_t* order = (int32_t*)c.take(260);
      float* bufs[4] = {h_E ? h_E : E, (E && h_E) ? E : nullptr, nullptr, nullptr};
      int have = bufs[1] ? 2 : 1;
      while (order && have < np) { float* t = c.take(rows); if (!t) break; bufs[have++] = t; }
      if (order && have >= 2) {
        a.nparts = have; a.order = order;
        for (int q = 0; q < have; ++q) a.pbuf[q] = bufs[q];
        grid = dim3(e.grid * have);                                    // sized for every block being long; the surplus exits at once
        fr.M18 = M18; fr.order = order; fr.G = G; fr.npw = e.npw; fr.nblk = e.grid;
      }
    }
    // neighbour lists (one workgroup per residue; one more ranks the residue blocks for a launch in parts)
    int Lp2 = 1; while (Lp2 < L) Lp2 <<= 1;
    int Kp2 = 64; while (Kp2 < K) Kp2 <<= 1;
    const bool full_sort = getenv("NAMP_KNN_FULL_SORT") != nullptr;      // debugging / test switch: sort whole rows
    const dim3 knn_grid(G + (fr.order ? 1 : 0));
    if (2 * Kp2 <= Lp2 && !full_sort)      // selection pays when the final sort is at most half a row
      hipLaunchKernelGGL(knn_select_kernel, knn_grid, dim3(256), ((size_t)L + Kp2) * 8 + 1024 + 64, s, P, mask, E_idx, L, K, Kp2, fr);
    else
      hipLaunchKernelGGL(knn_kernel, knn_grid, dim3(256), std::max((size_t)Lp2 * 8 + 64, (size_t)1024), s, P, mask, E_idx, L, Lp2, K, fr);
    if (job) { a.ord = *job; grid.x += job->B; }                       // the decoding orders: the launch's first workgroups
    // NampModelW.reserved == 2 with an x3 image: plain bf16 products on its hi half (mixed-precision training)
    if (x3 && w->reserved == 2) hipLaunchKernelGGL(edge_features_kernel<2>, grid, dim3(e.nwaves * 64), FEAT_LDS, s, a);
    else if (x3) hipLaunchKernelGGL(edge_features_kernel<1>, grid, dim3(e.nwaves * 64), FEAT_LDS, s, a);
    else hipLaunchKernelGGL(edge_features_kernel<0>, grid, dim3(e.nwaves * 64), FEAT_LDS, s, a);
    if (a.nparts >= 2) {
      if (x3) hipLaunchKernelGGL(feat_finish_kernel<1>, dim3(e.grid), dim3(e.nwaves * 64), NAMP_IMG_BYTES, s, a);
      else hipLaunchKernelGGL(feat_finish_kernel<0>, dim3(e.grid), dim3(e.nwaves * 64), NAMP_IMG_BYTES, s, a);
    }
  }
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_featurize(const NampModelW* w, const float* X, const int32_t* X_m, const int32_t* mask, const int32_t* R_idx,
                   const int32_t* chain_labels, const int32_t* protein_mask, const int32_t* dna_mask,
                   const int32_t* rna_mask, int top_k, int ref_atom, int32_t* E_idx, float* E, float* h_E, void* ws,
                   size_t ws_bytes, int B, int L, void* stream) {
  return featurize_impl(w, X, X_m, mask, R_idx, chain_labels, protein_mask, dna_mask, rna_mask, top_k, ref_atom, E_idx, E, h_E, ws, ws_bytes, B, L,
                        nullptr, stream);
}

// namp_featurize + namp_decoding_order in the same launches: the sort depends on (mask, chain_mask, randn) only and its first consumer is the
// decoder, so its B_order workgroups ride in front of the edge-feature launch (one more launch, or a side stream with two cross-stream
// hand-overs, cost score() from coordinates ~12 us at 1,000 residues).
int namp_featurize_ordered(const NampModelW* w, const float* X, const int32_t* X_m, const int32_t* mask, const int32_t* R_idx,
                           const int32_t* chain_labels, const int32_t* protein_mask, const int32_t* dna_mask,
                           const int32_t* rna_mask, int top_k, int ref_atom, int32_t* E_idx, float* E, float* h_E, void* ws,
                           size_t ws_bytes, int B, int L, const float* order_mask, const float* order_chain_mask, const float* randn,
                           int64_t* order64, int32_t* order32, int32_t* rank32, int B_order, void* stream) {
  if (!order_mask || !randn || !rank32) return fail(NAMP_EINVAL, "namp_featurize_ordered: null mask / randn / rank");
  REQUIRE(B_order >= 1 && B >= 1 && B_order % B == 0 && L >= 1 && L <= 8192, "namp_featurize_ordered: bad dims B_order=%d B=%d L=%d (L <= 8192)", B_order, B, L);
  int P2 = 2;
  while (P2 < L) P2 <<= 1;
  const OrderJob o = {order_mask, order_chain_mask, randn, order64, order32, rank32, B_order, B, L, P2};
  return featurize_impl(w, X, X_m, mask, R_idx, chain_labels, protein_mask, dna_mask, rna_mask, top_k, ref_atom, E_idx, E, h_E, ws, ws_bytes, B, L,
                        &o, stream);
}

// Sampler workspace for a model of n_dec decoder layers: Pfw[L] + Pa0 on the encoder side; Pa[L-1] + Pv[L-1] + h[L] on the sample-stream
// side; the first-layer tables Z1_l = W1e_l . h_E of every edge (round 5) and one zero row; the 4 KiB tail holds the level walk's grid-barrier
// words.  (Round 5 sized every workspace for NAMP_MAX_LAYERS = 8 layers: 8 h_E-sized tables where the shipped model carves 3.)
static size_t sample_ws_bytes(int B_enc, int B_dec, int N, int K, int n_dec) {
  if (B_enc < 1 || B_dec < 1 || N < 1 || K < 1 || n_dec < 1 || n_dec > NAMP_MAX_LAYERS) return 0;
  const size_t Ge = (size_t)B_enc * N, Gd = (size_t)B_dec * N;
  const size_t L = (size_t)n_dec;
  return (L + 1) * tbl(Ge) + (3 * L - 2) * tbl(Gd) + L * tbl(Ge * (size_t)K) + 512 + NAMP_HIDDEN * 64 * 4 + 4096;
}
size_t namp_sample_workspace_bytes_n(int B_enc, int B_dec, int N, int K, int n_dec) { return sample_ws_bytes(B_enc, B_dec, N, K, n_dec); }
size_t namp_sample_workspace_bytes(int B_enc, int B_dec, int N, int K) { return sample_ws_bytes(B_enc, B_dec, N, K, NAMP_MAX_LAYERS); }

static int sample_prepare(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                          const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                          const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                          const int32_t* group_first, const int32_t* group_last, const float* sym_weights,
                          const float* pair_bias,
                          float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                          void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream,
                          SampleArgs* a_out, int* nwaves_out) {
  REQUIRE(w != nullptr, "namp_decoder_sample: null weights");
  REQUIRE((group_first == nullptr) == (group_last == nullptr), "namp_decoder_sample: group_first and group_last go together");
  REQUIRE(w->n_dec >= 1 && w->n_dec <= NAMP_MAX_LAYERS, "namp_decoder_sample: supports 1..%d decoder layers (got %d)", NAMP_MAX_LAYERS, w->n_dec);
  REQUIRE(w->vocab >= 1 && w->vocab <= 64, "namp_decoder_sample: vocab=%d must be in [1,64]", w->vocab);
  REQUIRE_PTR(h_V_enc); REQUIRE_PTR(h_E); REQUIRE_PTR(ws);
  if (!E_idx || !mask || !chain_mask || !S_true || !bias || !order || !rank || !uniform || !S_out || !probs_out || !logp_out)
    return fail(NAMP_EINVAL, "namp_decoder_sample: null pointer argument");
  REQUIRE(temperature > 0.f, "namp_decoder_sample: temperature must be > 0");
  int rc = check_dims(__func__, B_dec, N, K);
  if (rc) return rc;
  REQUIRE(B_enc >= 1 && B_dec % B_enc == 0, "namp_decoder_sample: B_dec=%d must be a multiple of B_enc=%d", B_dec, B_enc);
  if ((rc = ensure_attributes())) return rc;
  const int Gd = B_dec * N, Ge = B_enc * N;
  Carver c(ws, ws_bytes);
  const int nd = w->n_dec;
  float* Pfw[NAMP_MAX_LAYERS]; for (int l = 0; l < nd; ++l) Pfw[l] = c.take((size_t)Ge * NAMP_HIDDEN);
  float* Pa0 = c.take((size_t)Ge * NAMP_HIDDEN);
  float *Pa[NAMP_MAX_LAYERS] = {}, *Pv[NAMP_MAX_LAYERS] = {}, *hs[NAMP_MAX_LAYERS] = {};
  for (int l = 0; l + 1 < nd; ++l) { Pa[l] = c.take((size_t)Gd * NAMP_HIDDEN); Pv[l] = c.take((size_t)Gd * NAMP_HIDDEN); }
  for (int l = 0; l < nd; ++l) hs[l] = c.take((size_t)Gd * NAMP_HIDDEN);
  float* Z1[NAMP_MAX_LAYERS] = {};
  for (int l = 0; l < nd; ++l) Z1[l] = c.take((size_t)Ge * K * NAMP_HIDDEN);
  float* zero_row = c.take(NAMP_HIDDEN);
  float* head_wT = c.take((size_t)NAMP_HIDDEN * 64);
  if (!hs[nd - 1] || !zero_row || !head_wT) return fail(NAMP_EWORKSPACE, "namp_decoder_sample: workspace too small (%zu bytes)", ws_bytes);
  // static tables from the encoder output: Pfw_l = W1v_l . h_V^enc, Pa_0 = W1a_0 . h_V^enc + b1
  NampProj pf[NAMP_MAX_LAYERS + 1];
  int nf = 0;
  for (int l = 0; l < w->n_dec; ++l) pf[nf++] = {w->dec[l].W1v_img, nullptr, nullptr, Pfw[l]};
  pf[nf++] = {w->dec[0].W1a_img, w->dec[0].b1, nullptr, Pa0};
  for (int q0 = 0; q0 < nf; q0 += 8)                         // (a node_linear launch takes up to 8 blocks)
    if ((rc = namp_node_linear(h_V_enc, nullptr, B_enc, B_enc, N, pf + q0, nf - q0 < 8 ? nf - q0 : 8, nullptr, stream))) return rc;

  // the first-layer product of every decoder layer on every edge: Z1_l = W1e_l . h_E (h_E is the encoder's output: static during the walk)
  {
    hipError_t ez = hipMemsetAsync(zero_row, 0, NAMP_HIDDEN * sizeof(float), (hipStream_t)stream);
    if (ez != hipSuccess) return fail(NAMP_ELAUNCH, "namp_decoder_sample: hipMemsetAsync: %s", hipGetErrorString(ez));
    for (int l = 0; l < nd; ++l) {
      const NampDecLayerW* D = &w->dec[l];
      const bool x3l = prec_of(D->flags) == PREC_X3;
      const float* img = x3l ? D->W1e_ximg : D->W1e_img;
      REQUIRE_PTR(img);
      if ((rc = namp_edge_embed_prec(img, zero_row, h_E, Z1[l], x3l ? 1 : 0, B_enc, N, K, stream))) return rc;
    }
  }

  REQUIRE_PTR(w->Wout_w); REQUIRE_PTR(w->Wout_b);
  hipLaunchKernelGGL(head_transpose_kernel, dim3(NAMP_HIDDEN * 64 / 256), dim3(256), 0, (hipStream_t)stream, w->Wout_w, head_wT, w->vocab);

  SampleArgs a = {};
  a.hE = h_E; a.E_idx = E_idx; a.mask_true = mask; a.chain_mask = chain_mask; a.S_true = S_true; a.bias = bias; a.order = order; a.rank = rank;
  a.uniform = uniform; a.S_forced = S_forced; a.group_first = group_first; a.group_last = group_last;
  a.sym_w = sym_weights; a.pair_bias = pair_bias; a.head_wT = head_wT; a.head_b = w->Wout_b; a.S_out = S_out;
  a.probs_out = probs_out; a.logp_out = logp_out; a.special = special_tokens; a.inv_T = 1.0f / temperature;
  a.B_dec = B_dec; a.B_enc = B_enc; a.N = N; a.K = K; a.TPN = (K + 15) / 16; a.n_layers = w->n_dec; a.vocab = w->vocab;
  // 8 waves per workgroup (256 VGPRs per lane: no scratch) serve 8 / TPN streams; K > 128 falls back to the 12-wave form
  const int maxw = a.TPN <= 8 ? 8 : 12;
  int slots = maxw / a.TPN; if (slots > NAMP_SAMPLE_SLOTS) slots = NAMP_SAMPLE_SLOTS; if (slots < 1) slots = 1;
  a.slots = slots;
  int nwaves = slots * a.TPN; if (nwaves < 8) nwaves = 8;
  REQUIRE(nwaves <= 12, "namp_decoder_sample: K=%d needs %d waves per workgroup (max 12)", K, nwaves);
  for (int l = 0; l < w->n_dec; ++l) {
    const NampDecLayerW* D = &w->dec[l];
    SampleLayer& L = a.l[l];
    const int prec = prec_of(D->flags) == PREC_X3 ? PREC_X3 : PREC_F32;       // the sampler has no bf16 mode: fp32-class only
    L.Z1 = Z1[l]; L.W2_img = pick_img(prec, D->W2_img, nullptr, D->W2_ximg);
    L.W3_img = pick_img(prec, D->W3_img, nullptr, D->W3_ximg); L.b2 = D->b2; L.b3 = D->b3; L.tok = D->tok;
    REQUIRE_PTR(L.W2_img); REQUIRE_PTR(L.W3_img);
    L.Pfw = Pfw[l];
    L.Pa = (l == 0) ? Pa0 : Pa[l - 1];
    L.Pv = (l == 0) ? Pfw[0] : Pv[l - 1];
    // split-bf16 mode with <= 8 waves per workgroup: the residue tail runs as an MFMA tile on the x3 images (node_tail_x3_rows)
    const bool tail_x3 = prec == PREC_X3 && maxw == 8;
    if (tail_x3) {
      REQUIRE(D->Win_ximg && D->Wout_ximg && aligned16(D->Win_ximg) && aligned16(D->Wout_ximg),
              "namp_decoder_sample: split-bf16 mode needs dec[%d].Win_ximg / Wout_ximg", l);
      REQUIRE(l + 1 >= w->n_dec || (w->dec[l + 1].W1a_ximg && w->dec[l + 1].W1v_ximg),
              "namp_decoder_sample: split-bf16 mode needs dec[%d].W1a_ximg / W1v_ximg", l + 1);
    }
    NampProj pn[2] = {{}, {}};
    int np = 0;
    if (l + 1 < w->n_dec) {
      const NampDecLayerW* Dn = &w->dec[l + 1];
      pn[0] = {tail_x3 ? Dn->W1a_ximg : Dn->W1a_img, Dn->b1, nullptr, Pa[l]};
      pn[1] = {tail_x3 ? Dn->W1v_ximg : Dn->W1v_img, nullptr, nullptr, Pv[l]};
      np = 2;
    }
    fill_tail(L.tail, D->ln1_g, D->ln1_b, tail_x3 ? D->Win_ximg : D->Win_img, D->b_in, tail_x3 ? D->Wout_ximg : D->Wout_img, D->b_out,
              D->ln2_g, D->ln2_b, (l == 0) ? h_V_enc : hs[l - 1], mask_dec, hs[l], pn, np, nullptr);
  }
  *a_out = a;
  *nwaves_out = nwaves;
  return NAMP_OK;
}

// the sampler kernel by (mode: 0 sequential walk / 1 one level per launch / 2 persistent level walk, precision, waves per workgroup)
static void launch_sample(int mode, bool x3, int nwaves, int grid, hipStream_t s, const SampleArgs& a, const int32_t* work,
                          const int32_t* work_n, int nwork, const int32_t* level_off = nullptr, unsigned* sync = nullptr) {
  const dim3 g(grid), b(nwaves * 64);
#define NAMP_LS(MD, X3, W) hipLaunchKernelGGL((dec_sample_kernel<MD, X3, W>), g, b, SAMPLE_LDS, s, a, work, work_n, nwork, level_off, sync)
  if (nwaves <= 8) {
    if (mode == 2)      { if (x3) NAMP_LS(2, true, 8); else NAMP_LS(2, false, 8); }
    else if (mode == 1) { if (x3) NAMP_LS(1, true, 8); else NAMP_LS(1, false, 8); }
    else                { if (x3) NAMP_LS(0, true, 8); else NAMP_LS(0, false, 8); }
  } else {
    if (mode == 1) { if (x3) NAMP_LS(1, true, 12); else NAMP_LS(1, false, 12); }
    else           { if (x3) NAMP_LS(0, true, 12); else NAMP_LS(0, false, 12); }
  }
#undef NAMP_LS
}

int namp_decoder_sample(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                        const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                        const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                        const int32_t* group_first, const int32_t* group_last, const float* sym_weights,
                        const float* pair_bias,
                        float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                        void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream) {
  SampleArgs a; int nwaves = 0;
  int rc = sample_prepare(w, h_V_enc, h_E, E_idx, mask, mask_dec, chain_mask, S_true, bias, order, rank, uniform, S_forced, group_first,
                          group_last, sym_weights, pair_bias, temperature, special_tokens, S_out, probs_out, logp_out, ws,
                          ws_bytes, B_dec, B_enc, N, K, stream, &a, &nwaves);
  if (rc) return rc;
  ProfScope prof_(NAMP_KIND_DEC_MESSAGE, (hipStream_t)stream);
  launch_sample(0, prec_of(w->dec[0].flags) == PREC_X3, nwaves, (B_dec + a.slots - 1) / a.slots, (hipStream_t)stream, a, nullptr, nullptr, 0);
  CHECK_LAUNCH();
  return NAMP_OK;
}


int namp_decoding_order(const float* mask, const float* chain_mask, const float* randn, int64_t* order64, int32_t* order32, int32_t* rank32,
                        int B, int B_mask, int L, void* stream) {
  if (!mask || !randn || !rank32) return fail(NAMP_EINVAL, "namp_decoding_order: null mask / randn / rank");      // (4-byte accesses: any alignment)
  REQUIRE(B >= 1 && B_mask >= 1 && B % B_mask == 0 && L >= 1 && L <= 8192, "namp_decoding_order: bad dims B=%d B_mask=%d L=%d (L <= 8192)", B, B_mask, L);
  int P2 = 2;
  while (P2 < L) P2 <<= 1;
  int rc = ensure_attributes();
  if (rc) return rc;
  const OrderJob o = {mask, chain_mask, randn, order64, order32, rank32, B, B_mask, L, P2};
  hipLaunchKernelGGL(decoding_order_kernel, dim3(B), dim3(P2 >= 1024 ? 512 : 256), (size_t)P2 * 8, (hipStream_t)stream, o);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_sample_work_lists(const int32_t* level, int32_t* work, int32_t* level_off, int32_t* n_levels, int B_dec, int N, void* stream) {
  if (!level || !work || !level_off || !n_levels) return fail(NAMP_EINVAL, "namp_sample_work_lists: null pointer argument");
  REQUIRE(B_dec >= 1 && N >= 1 && N <= 16000 && (long)B_dec * N < (1L << 31), "namp_sample_work_lists: bad dims B_dec=%d N=%d", B_dec, N);
  int rc = ensure_attributes();
  if (rc) return rc;
  hipLaunchKernelGGL(work_lists_kernel, dim3(1), dim3(1024), (size_t)(N + 2) * 4, (hipStream_t)stream, level, B_dec * N, N, work, level_off, n_levels);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_sample_levels_dep(const int32_t* E_idx, const int32_t* order, const int32_t* rank, const int32_t* dep_idx, int D,
                           const int32_t* group_first, const int32_t* group_last, int32_t* level,
                           int B_dec, int B_enc, int N, int K, void* stream) {
  if (!E_idx || !order || !rank || !level) return fail(NAMP_EINVAL, "namp_sample_levels: null pointer argument");
  REQUIRE(B_dec >= 1 && B_enc >= 1 && B_dec % B_enc == 0 && N >= 1 && K >= 1 && N <= 16384,
          "namp_sample_levels: bad dims B_dec=%d B_enc=%d N=%d K=%d", B_dec, B_enc, N, K);
  REQUIRE(dep_idx == nullptr || D >= 1, "namp_sample_levels_dep: D=%d", D);
  REQUIRE((group_first == nullptr) == (group_last == nullptr), "namp_sample_levels_dep: group_first and group_last go together");
  hipLaunchKernelGGL(sample_levels_kernel, dim3(B_dec), dim3(64), (size_t)N * 4, (hipStream_t)stream, E_idx, order, rank, dep_idx, D,
                     group_first, group_last, level, B_enc, N, K);
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_sample_levels(const int32_t* E_idx, const int32_t* order, const int32_t* rank, int32_t* level, int B_dec, int B_enc,
                       int N, int K, void* stream) {
  return namp_sample_levels_dep(E_idx, order, rank, nullptr, 0, nullptr, nullptr, level, B_dec, B_enc, N, K, stream);
}

int namp_decoder_sample_levels(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                               const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                               const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                               const int32_t* group_first, const int32_t* group_last, const float* sym_weights, const float* pair_bias,
                               const int32_t* work, const int32_t* work_n, const int32_t* level_counts, int n_levels,
                               float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                               void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream) {
  REQUIRE(work != nullptr && level_counts != nullptr && n_levels >= 1, "namp_decoder_sample_levels: work list / level counts missing");
  REQUIRE((work_n != nullptr) == (group_first != nullptr), "namp_decoder_sample_levels: work_n goes with group_first / group_last");
  long total = 0;
  for (int l = 0; l < n_levels; ++l) { REQUIRE(level_counts[l] >= 0, "namp_decoder_sample_levels: negative level count"); total += level_counts[l]; }
  REQUIRE(work_n ? (total >= 1 && total <= (long)B_dec * N) : (total == (long)B_dec * N),
          "namp_decoder_sample_levels: level counts sum to %ld, expected %sB_dec*N = %ld", total, work_n ? "<= " : "", (long)B_dec * N);
  SampleArgs a; int nwaves = 0;
  int rc = sample_prepare(w, h_V_enc, h_E, E_idx, mask, mask_dec, chain_mask, S_true, bias, order, rank, uniform, S_forced, group_first,
                          group_last, sym_weights, pair_bias, temperature, special_tokens, S_out, probs_out, logp_out, ws, ws_bytes,
                          B_dec, B_enc, N, K, stream, &a, &nwaves);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(S_out, 0xFF, (size_t)B_dec * N * sizeof(int32_t), s);          // every token "not drawn" (-1)
  if (e != hipSuccess) return fail(NAMP_ELAUNCH, "namp_decoder_sample_levels: hipMemsetAsync: %s", hipGetErrorString(e));
  ProfScope prof_(NAMP_KIND_DEC_MESSAGE, s);
  const bool x3 = prec_of(w->dec[0].flags) == PREC_X3;
  long off = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int cnt = level_counts[l];
    if (cnt == 0) continue;
    launch_sample(1, x3, nwaves, (cnt + a.slots - 1) / a.slots, s, a, work + 2 * off, work_n ? work_n + off : nullptr, cnt);
    off += cnt;
  }
  CHECK_LAUNCH();
  return NAMP_OK;
}

int namp_decoder_sample_walk_grid(int B_dec, int N, int K) {
  if (B_dec < 1 || N < 1 || K < 1 || K > 128) return 0;                     // K > 128 (12-wave workgroups): no persistent form
  const int tpn = (K + 15) / 16;
  int slots = 8 / tpn; if (slots > NAMP_SAMPLE_SLOTS) slots = NAMP_SAMPLE_SLOTS; if (slots < 1) slots = 1;
  long g = ((long)B_dec * N + slots - 1) / slots;
  // every workgroup must be resident at once (155 KiB of LDS: one per CU) — half the device's CUs at most, so that a partitioned
  // device (e.g. 32 CUs per partition) still co-schedules them
  const int cap = device_cus() / 2 < NAMP_WALK_MAX_GRID ? device_cus() / 2 : NAMP_WALK_MAX_GRID;
  return (int)(g < cap ? g : (cap < 1 ? 1 : cap));
}

int namp_decoder_sample_walk(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                             const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                             const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                             const int32_t* group_first, const int32_t* group_last, const float* sym_weights, const float* pair_bias,
                             const int32_t* work, const int32_t* work_n, int nwork, const int32_t* level_off,
                             const int32_t* close, const int32_t* close_off, float* zbuf,
                             float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                             void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream) {
  REQUIRE(work != nullptr && level_off != nullptr, "namp_decoder_sample_walk: work list / level offsets missing");
  REQUIRE((close != nullptr) == (close_off != nullptr) && (close != nullptr) == (zbuf != nullptr),
          "namp_decoder_sample_walk: close, close_off and zbuf go together");
  REQUIRE(close == nullptr || group_first != nullptr, "namp_decoder_sample_walk: the deferred group draw needs group_first / group_last");
  REQUIRE((work_n != nullptr) == (group_first != nullptr), "namp_decoder_sample_walk: work_n goes with group_first / group_last");
  REQUIRE(nwork >= 1 && nwork <= (long)B_dec * N && (work_n || nwork == (long)B_dec * N),
          "namp_decoder_sample_walk: nwork=%d work items for B_dec*N = %ld visits", nwork, (long)B_dec * N);
  const int grid = namp_decoder_sample_walk_grid(B_dec, N, K);
  REQUIRE(grid >= 1, "namp_decoder_sample_walk: no persistent form for B_dec=%d N=%d K=%d (K <= 128)", B_dec, N, K);
  SampleArgs a; int nwaves = 0;
  int rc = sample_prepare(w, h_V_enc, h_E, E_idx, mask, mask_dec, chain_mask, S_true, bias, order, rank, uniform, S_forced, group_first,
                          group_last, sym_weights, pair_bias, temperature, special_tokens, S_out, probs_out, logp_out, ws, ws_bytes,
                          B_dec, B_enc, N, K, stream, &a, &nwaves);
  if (rc) return rc;
  REQUIRE(nwaves <= 8, "namp_decoder_sample_walk: needs the 8-wave workgroup form (K <= 128)");
  // grid-barrier words: the last 4 KiB of the workspace (namp_sample_workspace_bytes reserves them behind the tables)
  REQUIRE(ws_bytes >= sample_ws_bytes(B_enc, B_dec, N, K, w->n_dec), "namp_decoder_sample_walk: workspace too small");
  unsigned* sync = (unsigned*)((char*)ws + sample_ws_bytes(B_enc, B_dec, N, K, w->n_dec) - 4096);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(S_out, 0xFF, (size_t)B_dec * N * sizeof(int32_t), s);          // every token "not drawn" (-1)
  if (e == hipSuccess) e = hipMemsetAsync(sync, 0, NAMP_SYNC_WORDS * sizeof(unsigned), s);
  if (e == hipSuccess && close) {
    const unsigned long long defer[3] = {(unsigned long long)zbuf, (unsigned long long)close, (unsigned long long)close_off};
    e = hipMemcpyAsync(sync + NAMP_SYNC_DEFER, defer, sizeof(defer), hipMemcpyHostToDevice, s);     // (pageable source: staged before the call returns)
  }
  if (e != hipSuccess) return fail(NAMP_ELAUNCH, "namp_decoder_sample_walk: hipMemsetAsync / hipMemcpyAsync: %s", hipGetErrorString(e));
  ProfScope prof_(NAMP_KIND_DEC_MESSAGE, s);
  int g2 = (nwork + a.slots - 1) / a.slots;
  if (const char* e_ = getenv("NAMP_WALK_GRID")) { const int v = atoi(e_); if (v >= 1 && v < g2) g2 = v; }      // (measurement switch)
  launch_sample(2, prec_of(w->dec[0].flags) == PREC_X3, nwaves, g2 < grid ? g2 : grid, s, a, work, work_n, nwork, level_off, sync);
  CHECK_LAUNCH();
  return NAMP_OK;
}

#ifdef NAMP_ABL_STAMPS
extern "C" int namp_debug_stamps(long long* out16, int reset) {
  long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(namp_stamp_acc), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(namp_stamp_acc), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef NAMP_ABL_WSTAMPS
// copies the per-wave event log of workgroup 0 out and clears it: counts[8], log[8][NAMP_WS_EVENTS][2]
extern "C" int namp_debug_wstamps(int* counts8, long long* log, int reset) {
  if (hipMemcpyFromSymbol(counts8, HIP_SYMBOL(namp_wstamp_n), 8 * sizeof(int)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(log, HIP_SYMBOL(namp_wstamp_log), sizeof(long long) * 8 * NAMP_WS_EVENTS * 2) != hipSuccess) return -1;
  int z[8] = {0};
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(namp_wstamp_n), z, sizeof(z)) != hipSuccess) return -1;
  return NAMP_WS_EVENTS;
}
#endif

int namp_set_bf16p(int mask) { return g_bf16p.exchange(mask & 255); }

int namp_set_persistent(int on) {
  std::lock_guard<std::mutex> lk(g_persist_mutex);
  const int prev = g_persist_on;
  g_persist_on = on ? 1 : 0;
  return prev;
}

int namp_persistent_status(const void* ws, size_t ws_bytes, int B, int N, int K, int32_t* code) {
  if (!ws || !code) return fail(NAMP_EINVAL, "namp_persistent_status: null pointer");
  REQUIRE(ws_bytes >= NAMP_SYNC_WORDS * 4, "namp_persistent_status: workspace too small");
  (void)B; (void)N; (void)K;
  unsigned words[NAMP_SYNC_WORDS];
  hipError_t e = hipMemcpy(words, ws, sizeof(words), hipMemcpyDeviceToHost);      // synchronous by design (a test / debug hook)
  if (e != hipSuccess) return fail(NAMP_ELAUNCH, "namp_persistent_status: hipMemcpy: %s", hipGetErrorString(e));
  *code = (int32_t)words[NAMP_SYNC_TIMEOUT];
  return NAMP_OK;
}

size_t namp_workspace_bytes(int B_enc, int B_dec, int N, int K) {
  if (B_enc < 1 || B_dec < 1 || N < 1 || K < 1) return 0;
  const size_t Ge = (size_t)B_enc * N, Gd = (size_t)B_dec * N, tpn = (K + 15) / 16;
  const size_t enc = (2 + 8 + tpn + 1) * tbl(Ge);            // (+1: the K-sums' weight sums behind `partial`)
  const size_t dec = (2 + 4 + tpn + 1) * tbl(Gd) + NAMP_MAX_LAYERS * tbl(Ge);
  return (enc > dec ? enc : dec) + 4096;
}

int namp_fused_tail_max_residues(void) { return NAMP_FUSED_TAIL_MAX_RESIDUES; }

int namp_enc_layer_fwd(const NampEncLayerW* w, const float* h_V, const float* h_E, const int32_t* E_idx,
                       const int32_t* mask, const int32_t* mask_attend, float* h_V_out, float* h_E_out, void* ws,
                       size_t ws_bytes, int B, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_enc_layer_fwd: null weights");
  REQUIRE_PTR(h_V); REQUIRE_PTR(h_E); REQUIRE_PTR(h_V_out); REQUIRE_PTR(h_E_out); REQUIRE_PTR(ws);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  const int G = B * N, tpn = (K + 15) / 16;
  Carver c(ws, ws_bytes);
  float* Pa = c.take((size_t)G * NAMP_HIDDEN);
  float* Pc = c.take((size_t)G * NAMP_HIDDEN);
  float* partial = c.take((size_t)G * tpn * (NAMP_HIDDEN + 1) + 3);      // K-sums [G][tpn][128] + weight sums [G][tpn]
  if (!partial) return fail(NAMP_EWORKSPACE, "namp_enc_layer_fwd: workspace too small (%zu bytes)", ws_bytes);
  NampProj p1[2] = {{w->W1a_img, w->b1, nullptr, Pa}, {w->W1c_img, nullptr, nullptr, Pc}};
  if ((rc = namp_node_linear(h_V, nullptr, B, B, N, p1, 2, nullptr, stream))) return rc;
  float* Pa2 = c.take((size_t)G * NAMP_HIDDEN);
  float* Pc2 = c.take((size_t)G * NAMP_HIDDEN);
  if (!Pc2) return fail(NAMP_EWORKSPACE, "namp_enc_layer_fwd: workspace too small (%zu bytes)", ws_bytes);
  NampProj p2[2] = {{w->W11a_img, w->b11, nullptr, Pa2}, {w->W11c_img, nullptr, nullptr, Pc2}};
  if (G <= NAMP_FUSED_TAIL_MAX_RESIDUES) {
    if ((rc = namp_enc_message_update(w, h_E, E_idx, mask, mask_attend, Pa, Pc, h_V, h_V_out, p2, 2, B, N, K, stream)))
      return rc;
  } else {
    if ((rc = namp_enc_message(w, h_E, E_idx, mask, mask_attend, Pa, Pc, partial, B, N, K, stream))) return rc;
    if ((rc = namp_node_update(w->ln1_g, w->ln1_b, w->Win_img, w->b_in, w->Wout_img, w->b_out, w->ln2_g, w->ln2_b, h_V,
                               partial, w->W3_img, w->b3, mask, h_V_out, p2, 2, nullptr, G, K, stream)))
      return rc;
  }
  return namp_enc_edge_update(w, h_E, E_idx, Pa2, Pc2, h_E_out, B, N, K, stream);
}

int namp_dec_layer_fwd(const NampDecLayerW* w, const float* h_V, const float* h_ESV, const int32_t* mask_V,
                       const float* mask_attend, float* h_V_out, void* ws, size_t ws_bytes, int B, int N, int K,
                       void* stream) {
  REQUIRE(w != nullptr, "namp_dec_layer_fwd: null weights");
  REQUIRE_PTR(h_V); REQUIRE_PTR(h_ESV); REQUIRE_PTR(h_V_out); REQUIRE_PTR(ws); OPTIONAL_PTR(mask_attend);
  REQUIRE_PTR(w->W1a_img); REQUIRE_PTR(w->W1e_img); REQUIRE_PTR(w->W1s_img); REQUIRE_PTR(w->W1v_img); REQUIRE_PTR(w->b1);
  REQUIRE_PTR(w->W2_img); REQUIRE_PTR(w->W3_img); REQUIRE_PTR(w->b2); REQUIRE_PTR(w->b3);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  const int G = B * N, tpn = (K + 15) / 16;
  Carver c(ws, ws_bytes);
  float* Pa = c.take((size_t)G * NAMP_HIDDEN);
  float* partial = c.take((size_t)G * tpn * (NAMP_HIDDEN + 1) + 3);      // K-sums [G][tpn][128] + weight sums [G][tpn]
  if (!partial) return fail(NAMP_EWORKSPACE, "namp_dec_layer_fwd: workspace too small (%zu bytes)", ws_bytes);
  NampProj pa = {w->W1a_img, w->b1, nullptr, Pa};
  if ((rc = namp_node_linear(h_V, nullptr, B, B, N, &pa, 1, nullptr, stream))) return rc;
  DecCtxArgs a = {};
  a.ctx = h_ESV; a.mask_attend = mask_attend; a.Pa = Pa;
  a.W1e_img = w->W1e_img; a.W1s_img = w->W1s_img; a.W1v_img = w->W1v_img; a.W2_img = w->W2_img; a.W3_img = w->W3_img;
  a.b2 = w->b2; a.b3 = w->b3; a.partial = partial; a.G = G; a.K = K; a.TPN = tpn;
  {
    ProfScope prof_(NAMP_KIND_DEC_MESSAGE, (hipStream_t)stream);
    const long tiles = (long)G * tpn;
    hipLaunchKernelGGL(dec_ctx_message_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    CHECK_LAUNCH();
  }
  return namp_node_update(w->ln1_g, w->ln1_b, w->Win_img, w->b_in, w->Wout_img, w->b_out, w->ln2_g, w->ln2_b, h_V, partial,
                          nullptr, nullptr, mask_V, h_V_out, nullptr, 0, nullptr, G, K, stream);     // partial holds whole messages here
}

int namp_encoder_fwd(const NampModelW* w, const float* V, const float* E, const int32_t* E_idx,
                     const int32_t* mask, float* h_V, float* h_E, void* ws, size_t ws_bytes, int B, int N, int K,
                     void* stream) {
  REQUIRE(w != nullptr, "namp_encoder_fwd: null weights");
  REQUIRE(w->n_enc >= 1 && w->n_enc <= NAMP_MAX_LAYERS, "namp_encoder_fwd: n_enc=%d out of range", w->n_enc);
  REQUIRE_PTR(V); OPTIONAL_PTR(E); REQUIRE_PTR(h_V); REQUIRE_PTR(h_E); REQUIRE_PTR(ws);
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  const int G = B * N, tpn = (K + 15) / 16;
  Carver c(ws, ws_bytes);
  float* hv[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* P[8];
  for (int i = 0; i < 8; ++i) P[i] = c.take((size_t)G * NAMP_HIDDEN);
  float* partial = c.take((size_t)G * tpn * (NAMP_HIDDEN + 1) + 3);      // K-sums [G][tpn][128] + weight sums [G][tpn]
  if (!partial) return fail(NAMP_EWORKSPACE, "namp_encoder_fwd: workspace too small (%zu bytes)", ws_bytes);
  const bool fused = G <= NAMP_FUSED_TAIL_MAX_RESIDUES;
  const bool chain_edges = fused && (w->enc[0].flags & NAMP_FLAG_BF16) == 0;

  // h_V = W_v.V + b (model_utils.py:88) chained with enc[0]'s Pa / Pc tables in one launch
  const NampEncLayerW* L0 = &w->enc[0];
  NampProj pre = {w->Wv_img, w->Wv_b, nullptr, hv[0]};
  NampProj p0[2] = {{L0->W1a_img, L0->b1, nullptr, P[0]}, {L0->W1c_img, nullptr, nullptr, P[1]}};
  if (!fused && prec_of(L0->flags) == PREC_X3 && w->Wv_ximg && L0->W1a_ximg && L0->W1c_ximg && aligned16(w->Wv_ximg) && aligned16(L0->W1a_ximg) &&
      aligned16(L0->W1c_ximg) && w->Wv_b && L0->b1) {
    // large batch, split-bf16 mode: the two chained residue-level products as split-bf16 products too (exact fp32 MFMA runs at 1/16 of the
    // bf16 rate: 70 us per launch at 32,000 residues, profiles/r05h_bench_cfg4_kernel_stats.md)
    const NampProj prex = {w->Wv_ximg, w->Wv_b, nullptr, hv[0]};
    const NampProj px[2] = {{L0->W1a_ximg, L0->b1, nullptr, P[0]}, {L0->W1c_ximg, nullptr, nullptr, P[1]}};
    ProfScope prof_(NAMP_KIND_NODE_LINEAR, (hipStream_t)stream);
    if ((rc = launch_node_linear(V, nullptr, G, G, N, px, 2, &prex, (hipStream_t)stream, true))) return rc;
    CHECK_LAUNCH();
  } else if ((rc = namp_node_linear(V, nullptr, B, B, N, p0, 2, &pre, stream))) return rc;
  if (E && (rc = namp_edge_embed(w->We_img, w->We_b, E, h_E, B, N, K, stream))) return rc;   // E == NULL: h_E already = W_e.E + b
  int cur = 0;     // hv[cur] holds the layer input
  for (int l = 0; l < w->n_enc; ++l) {
    const NampEncLayerW* L = &w->enc[l];
    const bool last = (l + 1 == w->n_enc);
    float* out = last ? h_V : hv[cur ^ 1];
    // message tables of layer l live in P[tm], P[tm+1] (ping-pong {0,1} / {4,5}: with the fused tail the
    // next layer's tables are written while other workgroups still gather this layer's); the
    // edge-update tables in P[2], P[3].
    // With the fused tail and fp32 weights, layer l-1's edge update rides in front of layer l's message phase (one
    // launch, h_E read once): the edge tables then ping-pong too ({2,3} / {6,7}).
    const int tm = (l & 1) ? 4 : 0, tn = (l & 1) ? 0 : 4;
    const int te = chain_edges ? ((l & 1) ? 6 : 2) : 2, tp = (l & 1) ? 2 : 6;
    NampProj pe[4] = {{L->W11a_img, L->b11, nullptr, P[te]}, {L->W11c_img, nullptr, nullptr, P[te + 1]}, {}, {}};
    int np = 2;
    if (!last) {
      const NampEncLayerW* Ln = &w->enc[l + 1];
      pe[2] = {Ln->W1a_img, Ln->b1, nullptr, P[tn]};
      pe[3] = {Ln->W1c_img, nullptr, nullptr, P[tn + 1]};
      np = 4;
    }
    if (chain_edges && l > 0) {
      if ((rc = namp_enc_edge_message_update(&w->enc[l - 1], P[tp], P[tp + 1], h_E, L, E_idx, mask, nullptr, P[tm], P[tm + 1],
                                             hv[cur], out, pe, np, B, N, K, stream)))
        return rc;
    } else if (fused) {
      if ((rc = namp_enc_message_update(L, h_E, E_idx, mask, nullptr, P[tm], P[tm + 1], hv[cur], out, pe, np, B, N, K,
                                        stream)))
        return rc;
    } else {
      if ((rc = namp_enc_message(L, h_E, E_idx, mask, nullptr, P[tm], P[tm + 1], partial, B, N, K, stream))) return rc;
      const float* pex[4] = {L->W11a_ximg, L->W11c_ximg, last ? nullptr : w->enc[l + 1].W1a_ximg, last ? nullptr : w->enc[l + 1].W1c_ximg};
      if ((rc = node_update_auto(L->flags, L->Win_ximg, L->Wout_ximg, pex, L->ln1_g, L->ln1_b, L->Win_img, L->b_in, L->Wout_img,
                                 L->b_out, L->ln2_g, L->ln2_b, hv[cur], partial, L->W3_img, L->W3_ximg, L->b3, mask, out, pe, np, nullptr, G, K,
                                 stream)))
        return rc;
    }
    if (!chain_edges || last) {
      if ((rc = namp_enc_edge_update(L, h_E, E_idx, P[te], P[te + 1], h_E, B, N, K, stream))) return rc;
    }
    cur ^= 1;
  }
  return NAMP_OK;
}

// bf16 throughput mode on a large batch (unfused residue tail), whole path with bf16 STORAGE of h_E and of the gathered
// tables (fragment order B, see namp_bf16s32.h).  The caller's h_E buffer is used as the bf16 store (its first half);
// it does not hold fp32 h_E afterwards.
static int encdec_bf16_storage(const NampModelW* w, const float* V, const float* E, const int32_t* E_idx, const int32_t* mask,
                               const int32_t* S, const int32_t* rank, float* h_V, float* h_E, float* log_probs, float* logits,
                               void* ws, size_t ws_bytes, int B, int N, int K, void* stream) {
  const int G = B * N, tpn = (K + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_attributes();
  if (rc) return rc;
  static const bool residue_x3 = [] { const char* e = getenv("NAMP_BF16S_RESIDUE_X3"); return e && atoi(e) != 0; }();   // A/B switch
  struct X1Scope { bool prev; X1Scope(bool on) : prev(g_residue_x1) { g_residue_x1 = on; } ~X1Scope() { g_residue_x1 = prev; } } x1scope_(!residue_x3);
  Carver c(ws, ws_bytes);
  float* hv[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* P[6];
  for (int i = 0; i < 6; ++i) P[i] = c.take((size_t)G * NAMP_HIDDEN);
  float* Pfw[NAMP_MAX_LAYERS];
  for (int l = 0; l < w->n_dec; ++l) Pfw[l] = c.take((size_t)G * NAMP_HIDDEN);
  float* partial = c.take((size_t)G * tpn * (NAMP_HIDDEN + 1) + 3);      // K-sums [G][tpn][128] + weight sums [G][tpn]
  __bf16* T16[4 + NAMP_MAX_LAYERS];                       // bf16 tables: 0,1 message / Pa,Pbw; 2,3 edge update; 4.. Pfw_l
  for (int i = 0; i < 4 + w->n_dec; ++i) T16[i] = (__bf16*)c.take((size_t)G * NAMP_HIDDEN / 2);
  if (!T16[3 + w->n_dec]) return fail(NAMP_EWORKSPACE, "namp_encdec_fwd: workspace too small (%zu bytes)", ws_bytes);
  __bf16* h16 = (__bf16*)h_E;
  auto cvt = [&](std::initializer_list<const float*> src, std::initializer_list<__bf16*> dst) {
    const float* sp[4]; __bf16* dp[4]; int n = 0;
    for (auto p : src) sp[n++] = p;
    n = 0;
    for (auto p : dst) dp[n++] = p;
    launch_cvt_tables(sp, dp, n, G, s);
  };
  // encoder
  const NampEncLayerW* L0 = &w->enc[0];
  NampProj pre = {w->Wv_img, w->Wv_b, nullptr, hv[0]};
  NampProj p0[2] = {{L0->W1a_img, L0->b1, nullptr, P[0]}, {L0->W1c_img, nullptr, nullptr, P[1]}};
  {
    Out16Req rq = {{T16[0], T16[1]}, 2, false};
    g_out16 = &rq;
    if (w->Wv_ximg && L0->W1a_ximg && L0->W1c_ximg && aligned16(w->Wv_ximg) && aligned16(L0->W1a_ximg) && aligned16(L0->W1c_ximg)) {
      // split-bf16 products instead of two chained fp32 MFMA GEMMs per unit (the fp32 form is MFMA-bound at this size)
      const NampProj prex = {w->Wv_ximg, w->Wv_b, nullptr, hv[0]};
      const NampProj px[2] = {{L0->W1a_ximg, L0->b1, nullptr, P[0]}, {L0->W1c_ximg, nullptr, nullptr, P[1]}};
      ProfScope prof_(NAMP_KIND_NODE_LINEAR, s);
      rc = (w->Wv_b && L0->b1) ? launch_node_linear(V, nullptr, G, G, N, px, 2, &prex, s, true)
                               : fail(NAMP_EINVAL, "namp_encdec_fwd: null W_v / W1 bias");
    } else {
      rc = namp_node_linear(V, nullptr, B, B, N, p0, 2, &pre, stream);
    }
    g_out16 = nullptr;
    if (rc) return rc;
    CHECK_LAUNCH();
    if (!rq.honoured) cvt({P[0], P[1]}, {T16[0], T16[1]});
  }
  static const bool fuse_embed = [] { const char* e = getenv("NAMP_BF16S_SEPARATE_EMBED"); return !(e && atoi(e) != 0); }();   // A/B switch
  const bool emb_fused = fuse_embed && w->We_simg != nullptr;
  if (!emb_fused) {
    EdgeArgs a = {};
    a.hE = E; a.hE16_out = h16; a.W1_img = w->We_bimg ? w->We_bimg : w->We_img; a.b1 = w->We_b; a.G = a.G_enc = G; a.N = N; a.K = K;
    ProfScope prof_(NAMP_KIND_EDGE_EMBED, s);
    // bf16 MFMA for the embedding too (the fp32 MFMA form is MFMA-bound: 8,192 pipe cycles per tile against 512)
    if ((rc = (w->We_bimg ? launch_edge<MODE_EMBED, 0, PREC_BF16>(a, s) : launch_edge<MODE_EMBED>(a, s)))) return rc;
  }
  int cur = 0;
  for (int l = 0; l < w->n_enc; ++l) {
    const NampEncLayerW* L = &w->enc[l];
    const bool last = (l + 1 == w->n_enc);
    float* out = last ? h_V : hv[cur ^ 1];
    REQUIRE_PTR(L->W1b_simg); REQUIRE_PTR(L->W2_simg); REQUIRE_PTR(L->W11b_simg); REQUIRE_PTR(L->W12_simg); REQUIRE_PTR(L->W13_simg);
    {
      EdgeArgs a = {};
      a.hE16 = h16; a.E_idx = E_idx; a.mask = mask; a.Pa16 = T16[0]; a.Pj016 = T16[1];
      a.W1_img = L->W1b_simg; a.W2_img = L->W2_simg; a.b2 = L->b2; a.b3 = L->b3;
      a.partial = partial; a.G = a.G_enc = G; a.N = N; a.K = K;
      if (l == 0 && emb_fused) { a.hE = E; a.hE16_out = h16; a.eW1_img = w->We_simg; a.eb2 = w->We_b; }   // h_E = W_e.E + b_e in this launch
      ProfScope prof_(NAMP_KIND_ENC_MESSAGE, s);
      if ((rc = launch_edge_bf16s<MODE_ENC_MSG>(a, s))) return rc;
    }
    NampProj pe[4] = {{L->W11a_img, L->b11, nullptr, P[2]}, {L->W11c_img, nullptr, nullptr, P[3]}, {}, {}};
    int np = 2;
    if (!last) {
      const NampEncLayerW* Ln = &w->enc[l + 1];
      pe[2] = {Ln->W1a_img, Ln->b1, nullptr, P[0]};
      pe[3] = {Ln->W1c_img, nullptr, nullptr, P[1]};
      np = 4;
    }
    const float* pex[4] = {L->W11a_ximg, L->W11c_ximg, last ? nullptr : w->enc[l + 1].W1a_ximg, last ? nullptr : w->enc[l + 1].W1c_ximg};
    {
      Out16Req rq = {{T16[2], T16[3], T16[0], T16[1]}, np, false};
      g_out16 = &rq;
      rc = node_update_auto(L->flags, L->Win_ximg, L->Wout_ximg, pex, L->ln1_g, L->ln1_b, L->Win_img, L->b_in, L->Wout_img, L->b_out,
                            L->ln2_g, L->ln2_b, hv[cur], partial, L->W3_img, L->W3_ximg, L->b3, mask, out, pe, np, nullptr, G, K, stream);
      g_out16 = nullptr;
      if (rc) return rc;
      if (!rq.honoured) {
        if (last) cvt({P[2], P[3]}, {T16[2], T16[3]});
        else cvt({P[2], P[3], P[0], P[1]}, {T16[2], T16[3], T16[0], T16[1]});
      }
    }
    {
      EdgeArgs a = {};
      a.hE16 = h16; a.hE16_out = h16; a.E_idx = E_idx; a.Pa16 = T16[2]; a.Pj016 = T16[3];
      a.W1_img = L->W11b_simg; a.W2_img = L->W12_simg; a.W3_img = L->W13_simg; a.b2 = L->b12; a.b3 = L->b13;
      a.ln_g = L->ln3_g; a.ln_b = L->ln3_b; a.G = a.G_enc = G; a.N = N; a.K = K;
      ProfScope prof_(NAMP_KIND_ENC_EDGE, s);
      if ((rc = launch_edge_bf16s<MODE_ENC_EDGE>(a, s))) return rc;
    }
    cur ^= 1;
  }
  // decoder tables from h_V_enc: Pfw_l, layer 0's Pa / Pbw
  const NampDecLayerW* D0 = &w->dec[0];
  NampProj pf[NAMP_MAX_LAYERS + 2];
  int nf = 0;
  for (int l = 0; l < w->n_dec; ++l) pf[nf++] = {w->dec[l].W1v_img, nullptr, nullptr, Pfw[l]};
  pf[nf++] = {D0->W1a_img, D0->b1, nullptr, P[0]};
  pf[nf++] = {D0->W1v_img, nullptr, D0->tok, P[1]};
  {
    Out16Req rq = {{}, nf <= 8 ? nf : 0, false};
    for (int l = 0; l < w->n_dec && l < 8; ++l) rq.p[l] = T16[4 + l];
    if (nf <= 8) { rq.p[nf - 2] = T16[0]; rq.p[nf - 1] = T16[1]; }
    g_out16 = rq.n ? &rq : nullptr;
    bool xok = nf <= 8 && D0->W1a_ximg && D0->W1v_ximg && aligned16(D0->W1a_ximg) && aligned16(D0->W1v_ximg);
    for (int l = 0; l < w->n_dec && xok; ++l) xok = w->dec[l].W1v_ximg && aligned16(w->dec[l].W1v_ximg);
    if (xok) {
      NampProj pfx[8];
      for (int i = 0; i < nf; ++i) pfx[i] = pf[i];
      for (int l = 0; l < w->n_dec; ++l) pfx[l].img = w->dec[l].W1v_ximg;
      pfx[nf - 2].img = D0->W1a_ximg; pfx[nf - 1].img = D0->W1v_ximg;
      ProfScope prof_(NAMP_KIND_NODE_LINEAR, s);
      rc = check_proj(__func__, pf, nf, S);
      if (!rc) rc = launch_node_linear(h_V, S, G, G, N, pfx, nf, nullptr, s, true);
    } else {
      rc = namp_node_linear(h_V, S, B, B, N, pf, nf, nullptr, stream);
    }
    g_out16 = nullptr;
    if (rc) return rc;
    CHECK_LAUNCH();
    if (!rq.honoured) {
      cvt({P[0], P[1]}, {T16[0], T16[1]});
      for (int l = 0; l < w->n_dec; l += 4) {
        const float* sp[4]; __bf16* dp[4]; int n = 0;
        for (int q = l; q < w->n_dec && q < l + 4; ++q) { sp[n] = Pfw[q]; dp[n] = T16[4 + q]; ++n; }
        launch_cvt_tables(sp, dp, n, G, s);
      }
    }
  }
  const float* hin = h_V;
  for (int l = 0; l < w->n_dec; ++l) {
    const NampDecLayerW* D = &w->dec[l];
    const bool last = (l + 1 == w->n_dec);
    float* out = hv[l & 1];
    REQUIRE_PTR(D->W1e_simg); REQUIRE_PTR(D->W2_simg);
    {
      EdgeArgs a = {};
      a.hE16 = h16; a.E_idx = E_idx; a.rank = rank; a.Pa16 = T16[0]; a.Pj016 = T16[1]; a.Pj116 = T16[4 + l];
      a.W1_img = D->W1e_simg; a.W2_img = D->W2_simg; a.b2 = D->b2; a.b3 = D->b3;
      a.partial = partial; a.G = a.G_enc = G; a.N = N; a.K = K;
      ProfScope prof_(NAMP_KIND_DEC_MESSAGE, s);
      if ((rc = launch_edge_bf16s<MODE_DEC_MSG>(a, s))) return rc;
    }
    NampProj pn[2] = {{}, {}};
    int np = 0;
    if (!last) {
      const NampDecLayerW* Dn = &w->dec[l + 1];
      pn[0] = {Dn->W1a_img, Dn->b1, nullptr, P[0]};
      pn[1] = {Dn->W1v_img, nullptr, Dn->tok, P[1]};
      np = 2;
    }
    const float* pnx[2] = {last ? nullptr : w->dec[l + 1].W1a_ximg, last ? nullptr : w->dec[l + 1].W1v_ximg};
    {
      Out16Req rq = {{T16[0], T16[1]}, np, false};
      g_out16 = np ? &rq : nullptr;
      rc = node_update_auto(D->flags, D->Win_ximg, D->Wout_ximg, pnx, D->ln1_g, D->ln1_b, D->Win_img, D->b_in, D->Wout_img, D->b_out,
                            D->ln2_g, D->ln2_b, hin, partial, D->W3_img, D->W3_ximg, D->b3, mask, out, pn, np, S, G, K, stream);
      g_out16 = nullptr;
      if (rc) return rc;
      if (!last && !rq.honoured) cvt({P[0], P[1]}, {T16[0], T16[1]});
    }
    hin = out;
  }
  CHECK_LAUNCH();
  return namp_logits_log_softmax(w->Wout_w, w->Wout_b, hin, log_probs, logits, G, w->vocab, stream);
}

// ProteinMPNN.score's device path in one call: encoder + parallel decoder.  While the batch takes the fused residue
// tail (fp32, B*N <= NAMP_FUSED_TAIL_MAX_RESIDUES) the encoder/decoder boundary is fused too: the last EncLayer's
// message launch also projects the decoder's layer-0 and encoder-context tables, and the last edge update rides in
// front of DecLayer 0's message phase — 2 + n_enc + n_dec launches instead of 3 + 2 n_enc + n_dec.
int namp_encdec_fwd(const NampModelW* w, const float* V, const float* E, const int32_t* E_idx, const int32_t* mask,
                    const int32_t* S, const int32_t* rank, float* h_V, float* h_E, float* log_probs, float* logits,
                    void* ws, size_t ws_bytes, int B, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_encdec_fwd: null weights");
  REQUIRE(w->n_enc >= 1 && w->n_enc <= NAMP_MAX_LAYERS && w->n_dec >= 1 && w->n_dec <= NAMP_MAX_LAYERS,
          "namp_encdec_fwd: n_enc=%d / n_dec=%d out of range", w->n_enc, w->n_dec);
  REQUIRE_PTR(V); OPTIONAL_PTR(E); REQUIRE_PTR(h_V); REQUIRE_PTR(h_E); REQUIRE_PTR(ws);
  if (!E_idx || !S || !rank || !log_probs) return fail(NAMP_EINVAL, "namp_encdec_fwd: null E_idx / S / rank / log_probs");
  int rc = check_dims(__func__, B, N, K);
  if (rc) return rc;
  const int G = B * N;
  const size_t half = namp_workspace_bytes(B, B, N, K);
  REQUIRE(ws_bytes >= 2 * half, "namp_encdec_fwd: workspace too small (%zu bytes, need 2 x namp_workspace_bytes = %zu)", ws_bytes, 2 * half);
  const bool bf = (w->enc[0].flags & NAMP_FLAG_BF16) != 0 || (w->dec[0].flags & NAMP_FLAG_BF16) != 0;
  const int prec = prec_of(w->enc[0].flags);
  REQUIRE(prec == prec_of(w->dec[0].flags), "namp_encdec_fwd: encoder and decoder layers must use the same precision");
  // bf16 storage needs the 32x32x16-order images of every per-edge block (optional fields of the weight structs)
  bool simg = true;
  for (int l = 0; l < w->n_enc; ++l)
    simg = simg && w->enc[l].W1b_simg && w->enc[l].W2_simg && w->enc[l].W11b_simg && w->enc[l].W12_simg && w->enc[l].W13_simg;
  for (int l = 0; l < w->n_dec; ++l) simg = simg && w->dec[l].W1e_simg && w->dec[l].W2_simg;
  if (bf && simg && E && G > NAMP_FUSED_TAIL_MAX_RESIDUES && edge_geom(G, K).grid > 2 * device_cus())
    return encdec_bf16_storage(w, V, E, E_idx, mask, S, rank, h_V, h_E, log_probs, logits, ws, ws_bytes, B, N, K, stream);
  if (G > NAMP_FUSED_TAIL_MAX_RESIDUES || bf || w->n_dec + 4 > 8) {
    if ((rc = namp_encoder_fwd(w, V, E, E_idx, mask, h_V, h_E, ws, half, B, N, K, stream))) return rc;
    return namp_decoder_fwd(w, h_V, h_E, E_idx, S, mask, rank, log_probs, logits, nullptr, (char*)ws + half, half, B, B, N, K, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  Carver c(ws, ws_bytes);
  unsigned* sync = (unsigned*)c.take(NAMP_SYNC_WORDS);                      // first block of the workspace (namp_persistent_status)
  float* hv[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* P[8];
  for (int i = 0; i < 8; ++i) P[i] = c.take((size_t)G * NAMP_HIDDEN);
  float* dhv[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* PA[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* PB[2] = {c.take((size_t)G * NAMP_HIDDEN), c.take((size_t)G * NAMP_HIDDEN)};
  float* Pfw[NAMP_MAX_LAYERS];
  for (int l = 0; l < w->n_dec; ++l) Pfw[l] = c.take((size_t)G * NAMP_HIDDEN);
  if (!Pfw[w->n_dec - 1]) return fail(NAMP_EWORKSPACE, "namp_encdec_fwd: workspace too small (%zu bytes)", ws_bytes);

  // one persistent launch for the whole pass when every workgroup of it is resident at once (see encdec_persistent_kernel)
  const EdgeGeom eg = edge_geom(G, K);
  int pdev = 0;
  const bool persistent = w->n_enc == 3 && w->n_dec == 3 && eg.grid <= device_cus() && eg.npw <= 4 && persist_acquire(s, &pdev);
  if (rc) return rc;

  const NampEncLayerW* L0 = &w->enc[0];
  NampProj pre = {w->Wv_img, w->Wv_b, nullptr, hv[0]};
  NampProj p0[2] = {{L0->W1a_img, L0->b1, nullptr, P[0]}, {L0->W1c_img, nullptr, nullptr, P[1]}};
  if (prec_of(L0->flags) == PREC_X3 && w->Wv_ximg && L0->W1a_ximg && L0->W1c_ximg) {
    // split-bf16 form of the first launch (two chained fp32 MFMA GEMMs per unit otherwise: 15 us of pure latency at cfg2)
    OPTIONAL_PTR(w->Wv_ximg); OPTIONAL_PTR(L0->W1a_ximg); OPTIONAL_PTR(L0->W1c_ximg);
    REQUIRE_PTR(V); REQUIRE_PTR(w->Wv_b); REQUIRE_PTR(L0->b1);
    const NampProj prex = {w->Wv_ximg, w->Wv_b, nullptr, hv[0]};
    const NampProj px[2] = {{L0->W1a_ximg, L0->b1, nullptr, P[0]}, {L0->W1c_ximg, nullptr, nullptr, P[1]}};
    ProfScope prof_(NAMP_KIND_NODE_LINEAR, s);
    launch_node_linear(V, nullptr, B * N, B * N, N, px, 2, &prex, s, true, persistent ? sync : nullptr);
    CHECK_LAUNCH();
  } else {
    REQUIRE_PTR(w->Wv_img); REQUIRE_PTR(w->Wv_b); REQUIRE_PTR(L0->W1a_img); REQUIRE_PTR(L0->W1c_img); REQUIRE_PTR(L0->b1);
    ProfScope prof_(NAMP_KIND_NODE_LINEAR, s);
    launch_node_linear(V, nullptr, B * N, B * N, N, p0, 2, &pre, s, false, persistent ? sync : nullptr);
    CHECK_LAUNCH();
  }
  if (E) { REQUIRE_PTR(w->We_img); REQUIRE_PTR(w->We_b); }    // embedded inside the first message launch
  if (persistent) {
    PersistArgs PA_ = {};
    PA_.sync = sync; PA_.embed = E ? 1 : 0;
    REQUIRE_PTR(w->Wout_w); REQUIRE_PTR(w->Wout_b);
    auto common = [&](StageArgs& st) {
      st.hE = E ? E : h_E; st.E_idx = E_idx; st.mask = mask; st.rank = rank;
      st.G = st.G_enc = G; st.N = N; st.K = K; st.TPN = eg.tpn;
    };
    int cur_ = 0;
    for (int l = 0; l < 3; ++l) {
      const NampEncLayerW* L = &w->enc[l];
      const bool last = (l == 2);
      StageArgs& st = PA_.st[l];
      common(st);
      const int tm = (l & 1) ? 4 : 0, tn = (l & 1) ? 0 : 4, te = (l & 1) ? 6 : 2, tp = (l & 1) ? 2 : 6;
      st.Pa = P[tm]; st.Pj0 = P[tm + 1]; st.Pj1 = nullptr;
      st.W1_img = pick_img(prec, L->W1b_img, nullptr, L->W1b_ximg); st.W2_img = pick_img(prec, L->W2_img, nullptr, L->W2_ximg);
      st.W3_img = pick_img(prec, L->W3_img, nullptr, L->W3_ximg); st.b2 = L->b2; st.b3 = L->b3;
      REQUIRE_PTR(st.W1_img); REQUIRE_PTR(st.W2_img); REQUIRE_PTR(st.W3_img); REQUIRE_PTR(st.b2); REQUIRE_PTR(st.b3);
      if (l == 0) {
        if (E) { st.eW1_img = pick_img(prec, w->We_img, nullptr, w->We_ximg); st.eb2 = w->We_b; REQUIRE_PTR(st.eW1_img); REQUIRE_PTR(st.eb2); }
      } else {
        const NampEncLayerW* Lp = &w->enc[l - 1];
        st.ePa = P[tp]; st.ePj = P[tp + 1];
        st.eW1_img = pick_img(prec, Lp->W11b_img, nullptr, Lp->W11b_ximg); st.eW2_img = pick_img(prec, Lp->W12_img, nullptr, Lp->W12_ximg);
        st.eW3_img = pick_img(prec, Lp->W13_img, nullptr, Lp->W13_ximg); st.eb2 = Lp->b12; st.eb3 = Lp->b13;
        st.ln_g = Lp->ln3_g; st.ln_b = Lp->ln3_b;
        REQUIRE_PTR(st.eW1_img); REQUIRE_PTR(st.eW2_img); REQUIRE_PTR(st.eW3_img); REQUIRE_PTR(st.ln_g); REQUIRE_PTR(st.ln_b);
      }
      st.hE_out = nullptr;                                     // rows stay in registers
      NampProj pe[8];
      int np = 0;
      pe[np++] = {L->W11a_img, L->b11, nullptr, P[te]};
      pe[np++] = {L->W11c_img, nullptr, nullptr, P[te + 1]};
      if (!last) {
        const NampEncLayerW* Ln = &w->enc[l + 1];
        pe[np++] = {Ln->W1a_img, Ln->b1, nullptr, P[tn]};
        pe[np++] = {Ln->W1c_img, nullptr, nullptr, P[tn + 1]};
      } else {
        const NampDecLayerW* D0 = &w->dec[0];
        for (int d = 0; d < 3; ++d) pe[np++] = {w->dec[d].W1v_img, nullptr, nullptr, Pfw[d]};
        pe[np++] = {D0->W1a_img, D0->b1, nullptr, PA[0]};
        pe[np++] = {D0->W1v_img, nullptr, D0->tok, PB[0]};
      }
      if ((rc = check_proj(__func__, pe, np, S))) return rc;
      fill_tail(st.tail, L->ln1_g, L->ln1_b, L->Win_img, L->b_in, L->Wout_img, L->b_out, L->ln2_g, L->ln2_b, hv[cur_], mask,
                last ? h_V : hv[cur_ ^ 1], pe, np, S);
      st.tail.m3_img = L->W3_img; st.tail.m3_b = L->b3;
      cur_ ^= 1;
    }
    const float* hin_ = h_V;
    for (int l = 0; l < 3; ++l) {
      const NampDecLayerW* D = &w->dec[l];
      StageArgs& st = PA_.st[3 + l];
      common(st);
      st.Pa = PA[l & 1]; st.Pj0 = PB[l & 1]; st.Pj1 = Pfw[l];
      st.W1_img = pick_img(prec, D->W1e_img, nullptr, D->W1e_ximg); st.W2_img = pick_img(prec, D->W2_img, nullptr, D->W2_ximg);
      st.W3_img = pick_img(prec, D->W3_img, nullptr, D->W3_ximg); st.b2 = D->b2; st.b3 = D->b3;
      REQUIRE_PTR(st.W1_img); REQUIRE_PTR(st.W2_img); REQUIRE_PTR(st.W3_img); REQUIRE_PTR(st.b2); REQUIRE_PTR(st.b3);
      if (l == 0) {
        const NampEncLayerW* Lp = &w->enc[2];
        st.ePa = P[2]; st.ePj = P[3];                          // te of the last encoder layer (l = 2)
        st.eW1_img = pick_img(prec, Lp->W11b_img, nullptr, Lp->W11b_ximg); st.eW2_img = pick_img(prec, Lp->W12_img, nullptr, Lp->W12_ximg);
        st.eW3_img = pick_img(prec, Lp->W13_img, nullptr, Lp->W13_ximg); st.eb2 = Lp->b12; st.eb3 = Lp->b13;
        st.ln_g = Lp->ln3_g; st.ln_b = Lp->ln3_b;
        REQUIRE_PTR(st.eW1_img); REQUIRE_PTR(st.eW2_img); REQUIRE_PTR(st.eW3_img); REQUIRE_PTR(st.ln_g); REQUIRE_PTR(st.ln_b);
        st.hE_out = h_E;                                       // the encoder's h_E leaves the registers once, here
      }
      NampProj pn[2] = {{}, {}};
      int np = 0;
      if (l < 2) {
        const NampDecLayerW* Dn = &w->dec[l + 1];
        pn[np++] = {Dn->W1a_img, Dn->b1, nullptr, PA[(l + 1) & 1]};
        pn[np++] = {Dn->W1v_img, nullptr, Dn->tok, PB[(l + 1) & 1]};
      }
      if ((rc = check_proj(__func__, pn, np, S))) return rc;
      fill_tail(st.tail, D->ln1_g, D->ln1_b, D->Win_img, D->b_in, D->Wout_img, D->b_out, D->ln2_g, D->ln2_b, hin_, mask, dhv[l & 1],
                pn, np, S);
      st.tail.m3_img = D->W3_img; st.tail.m3_b = D->b3;
      if (l == 2) { st.tail.head_w = w->Wout_w; st.tail.head_b = w->Wout_b; st.tail.log_probs = log_probs; st.tail.logits = logits; st.tail.vocab = w->vocab; }
      hin_ = dhv[l & 1];
    }
    {
      ProfScope prof_(NAMP_KIND_ENCDEC_PERSISTENT, s);
      if ((rc = namp_internal_launch_persistent(&PA_, prec == PREC_X3 ? 1 : 0, eg.grid, eg.nwaves * 64, s)))
        return fail(NAMP_ELAUNCH, "namp_encdec_fwd: persistent launch setup failed (%d)", rc);
    }
    CHECK_LAUNCH();
    persist_mark(s, pdev);
    return NAMP_OK;
  }
  int cur = 0;
  for (int l = 0; l < w->n_enc; ++l) {
    const NampEncLayerW* L = &w->enc[l];
    const bool last = (l + 1 == w->n_enc);
    float* out = last ? h_V : hv[cur ^ 1];
    const int tm = (l & 1) ? 4 : 0, tn = (l & 1) ? 0 : 4, te = (l & 1) ? 6 : 2, tp = (l & 1) ? 2 : 6;
    NampProj pe[8];
    int np = 0;
    pe[np++] = {L->W11a_img, L->b11, nullptr, P[te]};
    pe[np++] = {L->W11c_img, nullptr, nullptr, P[te + 1]};
    if (!last) {
      const NampEncLayerW* Ln = &w->enc[l + 1];
      pe[np++] = {Ln->W1a_img, Ln->b1, nullptr, P[tn]};
      pe[np++] = {Ln->W1c_img, nullptr, nullptr, P[tn + 1]};
    } else {          // decoder tables from h_V_enc (model_utils.py:406-413): Pfw_l, and layer 0's Pa / Pbw
      const NampDecLayerW* D0 = &w->dec[0];
      for (int d = 0; d < w->n_dec; ++d) pe[np++] = {w->dec[d].W1v_img, nullptr, nullptr, Pfw[d]};
      pe[np++] = {D0->W1a_img, D0->b1, nullptr, PA[0]};
      pe[np++] = {D0->W1v_img, nullptr, D0->tok, PB[0]};
    }
    if ((rc = check_proj(__func__, pe, np, S))) return rc;
    EdgeArgs a = {};
    a.hE = h_E; a.E_idx = E_idx; a.mask = mask; a.Pa = P[tm]; a.Pj0 = P[tm + 1];
    a.W1_img = pick_img(prec, L->W1b_img, nullptr, L->W1b_ximg); a.W2_img = pick_img(prec, L->W2_img, nullptr, L->W2_ximg);
    a.W3_img = pick_img(prec, L->W3_img, nullptr, L->W3_ximg); a.b2 = L->b2; a.b3 = L->b3;
    REQUIRE_PTR(a.W1_img); REQUIRE_PTR(a.W2_img); REQUIRE_PTR(a.W3_img);
    a.G = a.G_enc = G; a.N = N; a.K = K;
    fill_tail(a.tail, L->ln1_g, L->ln1_b, L->Win_img, L->b_in, L->Wout_img, L->b_out, L->ln2_g, L->ln2_b, hv[cur], mask, out,
              pe, np, S);
    a.tail.m3_img = L->W3_img; a.tail.m3_b = L->b3;
    if (l == 0 && E) {      // h_E = W_e.E + b_e (model_utils.py:89) in front of the first message phase
      a.hE = E; a.hE_out = h_E; a.eW1_img = pick_img(prec, w->We_img, nullptr, w->We_ximg); a.eb2 = w->We_b;
      REQUIRE_PTR(a.eW1_img);
      ProfScope prof_(NAMP_KIND_ENC_MESSAGE, s);
      rc = launch_edge_tail_fused<MODE_ENC_MSG, PRE_EMBED>(a, prec, s);
    } else if (l == 0) {
      ProfScope prof_(NAMP_KIND_ENC_MESSAGE, s);
      rc = launch_edge_tail<MODE_ENC_MSG>(a, prec, s);
    } else {
      const NampEncLayerW* Lp = &w->enc[l - 1];
      a.hE_out = h_E; a.ePa = P[tp]; a.ePj = P[tp + 1];
      a.eW1_img = pick_img(prec, Lp->W11b_img, nullptr, Lp->W11b_ximg); a.eW2_img = pick_img(prec, Lp->W12_img, nullptr, Lp->W12_ximg);
      a.eW3_img = pick_img(prec, Lp->W13_img, nullptr, Lp->W13_ximg); a.eb2 = Lp->b12; a.eb3 = Lp->b13;
      REQUIRE_PTR(a.eW1_img); REQUIRE_PTR(a.eW2_img); REQUIRE_PTR(a.eW3_img);
      a.ln_g = Lp->ln3_g; a.ln_b = Lp->ln3_b;
      ProfScope prof_(NAMP_KIND_ENC_EDGE_MESSAGE, s);
      rc = launch_edge_tail_fused<MODE_ENC_MSG>(a, prec, s);
    }
    if (rc) return rc;
    CHECK_LAUNCH();
    cur ^= 1;
  }
  const int te_last = ((w->n_enc - 1) & 1) ? 6 : 2;
  const float* hin = h_V;
  for (int l = 0; l < w->n_dec; ++l) {
    const NampDecLayerW* D = &w->dec[l];
    const bool last = (l + 1 == w->n_dec);
    float* out = dhv[l & 1];
    NampProj pn[2] = {{}, {}};
    int np = 0;
    if (!last) {
      const NampDecLayerW* Dn = &w->dec[l + 1];
      pn[0] = {Dn->W1a_img, Dn->b1, nullptr, PA[(l + 1) & 1]};
      pn[1] = {Dn->W1v_img, nullptr, Dn->tok, PB[(l + 1) & 1]};
      np = 2;
    }
    if (l > 0) {
      if ((rc = namp_dec_message_update(D, h_E, E_idx, rank, PA[l & 1], PB[l & 1], Pfw[l], hin, mask, out, pn, np, S,
                                        last ? w->Wout_w : nullptr, w->Wout_b, log_probs, logits, w->vocab, B, B, N, K, stream)))
        return rc;
    } else {          // last EncLayer's edge update + DecLayer 0 message + tail
      const NampEncLayerW* Lp = &w->enc[w->n_enc - 1];
      EdgeArgs a = {};
      a.hE = h_E; a.hE_out = h_E; a.E_idx = E_idx; a.rank = rank; a.Pa = PA[0]; a.Pj0 = PB[0]; a.Pj1 = Pfw[0];
      a.ePa = P[te_last]; a.ePj = P[te_last + 1];
      a.eW1_img = pick_img(prec, Lp->W11b_img, nullptr, Lp->W11b_ximg); a.eW2_img = pick_img(prec, Lp->W12_img, nullptr, Lp->W12_ximg);
      a.eW3_img = pick_img(prec, Lp->W13_img, nullptr, Lp->W13_ximg); a.eb2 = Lp->b12; a.eb3 = Lp->b13;
      a.ln_g = Lp->ln3_g; a.ln_b = Lp->ln3_b;
      a.W1_img = pick_img(prec, D->W1e_img, nullptr, D->W1e_ximg); a.W2_img = pick_img(prec, D->W2_img, nullptr, D->W2_ximg);
      a.W3_img = pick_img(prec, D->W3_img, nullptr, D->W3_ximg); a.b2 = D->b2; a.b3 = D->b3;
      REQUIRE_PTR(a.eW1_img); REQUIRE_PTR(a.eW2_img); REQUIRE_PTR(a.eW3_img); REQUIRE_PTR(a.W1_img); REQUIRE_PTR(a.W2_img); REQUIRE_PTR(a.W3_img);
      a.G = a.G_enc = G; a.N = N; a.K = K;
      if ((rc = check_proj(__func__, pn, np, S))) return rc;
      fill_tail(a.tail, D->ln1_g, D->ln1_b, D->Win_img, D->b_in, D->Wout_img, D->b_out, D->ln2_g, D->ln2_b, hin, mask, out,
                pn, np, S);
      a.tail.m3_img = D->W3_img; a.tail.m3_b = D->b3;
      if (last) { a.tail.head_w = w->Wout_w; a.tail.head_b = w->Wout_b; a.tail.log_probs = log_probs; a.tail.logits = logits; a.tail.vocab = w->vocab; }
      ProfScope prof_(NAMP_KIND_ENC_EDGE_DEC_MESSAGE, s);
      rc = launch_edge_tail_fused<MODE_DEC_MSG>(a, prec, s);
      if (rc) return rc;
      CHECK_LAUNCH();
    }
    hin = out;
  }
  return NAMP_OK;
}

int namp_decoder_fwd(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                     const int32_t* S, const int32_t* mask, const int32_t* rank, float* log_probs, float* logits,
                     float* h_V_dec, void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream) {
  REQUIRE(w != nullptr, "namp_decoder_fwd: null weights");
  REQUIRE(w->n_dec >= 1 && w->n_dec <= NAMP_MAX_LAYERS, "namp_decoder_fwd: n_dec=%d out of range", w->n_dec);
  REQUIRE_PTR(h_V_enc); REQUIRE_PTR(h_E); REQUIRE_PTR(ws); OPTIONAL_PTR(h_V_dec);
  if (!E_idx || !S || !rank || !log_probs) return fail(NAMP_EINVAL, "namp_decoder_fwd: null E_idx / S / rank / log_probs");
  int rc = check_dims(__func__, B_dec, N, K);
  if (rc) return rc;
  REQUIRE(B_enc >= 1 && B_dec % B_enc == 0, "namp_decoder_fwd: B_dec=%d must be a multiple of B_enc=%d", B_dec, B_enc);
  const int Gd = B_dec * N, Ge = B_enc * N, tpn = (K + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  Carver c(ws, ws_bytes);
  float* hv[2] = {c.take((size_t)Gd * NAMP_HIDDEN), c.take((size_t)Gd * NAMP_HIDDEN)};
  float* PA[2] = {c.take((size_t)Gd * NAMP_HIDDEN), c.take((size_t)Gd * NAMP_HIDDEN)};   // ping-pong per layer
  float* PB[2] = {c.take((size_t)Gd * NAMP_HIDDEN), c.take((size_t)Gd * NAMP_HIDDEN)};
  float* Pa = PA[0];
  float* Pbw = PB[0];
  float* partial = c.take((size_t)Gd * tpn * (NAMP_HIDDEN + 1) + 3);      // K-sums [G][tpn][128] + weight sums [G][tpn]
  const bool fused = Gd <= NAMP_FUSED_TAIL_MAX_RESIDUES;
  float* Pfw[NAMP_MAX_LAYERS];
  for (int l = 0; l < w->n_dec; ++l) Pfw[l] = c.take((size_t)Ge * NAMP_HIDDEN);
  if (!Pfw[w->n_dec - 1] || !partial) return fail(NAMP_EWORKSPACE, "namp_decoder_fwd: workspace too small (%zu bytes)", ws_bytes);

  // encoder-context tables Pfw_l = W1v_l . h_V_enc (h_EXV_encoder of model_utils.py:410-413)
  // layer-0 residue tables from h_V^(0) = h_V_enc broadcast over decoder batches, and the encoder-context
  // tables; one launch when decoder and encoder batches coincide (<= 8 blocks), two otherwise
  const NampDecLayerW* D0 = &w->dec[0];
  NampProj pf[NAMP_MAX_LAYERS + 2];
  int nf = 0;
  for (int l = 0; l < w->n_dec; ++l) pf[nf++] = {w->dec[l].W1v_img, nullptr, nullptr, Pfw[l]};
  NampProj p0[2] = {{D0->W1a_img, D0->b1, nullptr, Pa}, {D0->W1v_img, nullptr, D0->tok, Pbw}};
  if (B_dec == B_enc && nf + 2 <= 8) {
    pf[nf++] = p0[0]; pf[nf++] = p0[1];
    bool xok = !fused && prec_of(D0->flags) == PREC_X3 && D0->W1a_ximg && D0->W1v_ximg && aligned16(D0->W1a_ximg) && aligned16(D0->W1v_ximg);
    for (int l = 0; l < w->n_dec && xok; ++l) xok = w->dec[l].W1v_ximg && aligned16(w->dec[l].W1v_ximg);
    if (xok) {                                             // large batch, split-bf16 mode: see namp_encoder_fwd
      NampProj pfx[8];
      for (int i = 0; i < nf; ++i) pfx[i] = pf[i];
      for (int l = 0; l < w->n_dec; ++l) pfx[l].img = w->dec[l].W1v_ximg;
      pfx[nf - 2].img = D0->W1a_ximg; pfx[nf - 1].img = D0->W1v_ximg;
      ProfScope prof_(NAMP_KIND_NODE_LINEAR, s);
      if ((rc = check_proj(__func__, pf, nf, S))) return rc;
      if ((rc = launch_node_linear(h_V_enc, S, Gd, Ge, N, pfx, nf, nullptr, s, true))) return rc;
      CHECK_LAUNCH();
    } else if ((rc = namp_node_linear(h_V_enc, S, B_dec, B_enc, N, pf, nf, nullptr, stream))) return rc;
  } else {
    if ((rc = namp_node_linear(h_V_enc, nullptr, B_enc, B_enc, N, pf, nf, nullptr, stream))) return rc;
    if ((rc = namp_node_linear(h_V_enc, S, B_dec, B_enc, N, p0, 2, nullptr, stream))) return rc;
  }
  const float* hin = h_V_enc;
  if (B_dec != B_enc) {
    for (int b = 0; b < B_dec; b += B_enc) {
      hipError_t e = hipMemcpyAsync(hv[0] + (size_t)b * N * NAMP_HIDDEN, h_V_enc, (size_t)Ge * NAMP_HIDDEN * 4,
                                    hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return fail(NAMP_ELAUNCH, "namp_decoder_fwd: hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    hin = hv[0];
  }
  int cur = 0;
  for (int l = 0; l < w->n_dec; ++l) {
    const NampDecLayerW* D = &w->dec[l];
    const bool last = (l + 1 == w->n_dec);
    float* out = (last && h_V_dec) ? h_V_dec : hv[cur ^ 1];
    NampProj pn[2] = {{}, {}};
    int np = 0;
    if (!last) {     // next layer's Pa / Pbw go to the other ping-pong slot
      const NampDecLayerW* Dn = &w->dec[l + 1];
      pn[0] = {Dn->W1a_img, Dn->b1, nullptr, PA[(l + 1) & 1]};
      pn[1] = {Dn->W1v_img, nullptr, Dn->tok, PB[(l + 1) & 1]};
      np = 2;
    }
    if (fused) {
      // the last layer's launch also evaluates the output head on the residues it has just updated
      if ((rc = namp_dec_message_update(D, h_E, E_idx, rank, PA[l & 1], PB[l & 1], Pfw[l], hin, mask, out, pn, np, S,
                                        last ? w->Wout_w : nullptr, w->Wout_b, log_probs, logits, w->vocab,
                                        B_dec, B_enc, N, K, stream)))
        return rc;
    } else {
      if ((rc = namp_dec_message(D, h_E, E_idx, rank, PA[l & 1], PB[l & 1], Pfw[l], partial, B_dec, B_enc, N, K, stream)))
        return rc;
      const float* pnx[2] = {last ? nullptr : w->dec[l + 1].W1a_ximg, last ? nullptr : w->dec[l + 1].W1v_ximg};
      if ((rc = node_update_auto(D->flags, D->Win_ximg, D->Wout_ximg, pnx, D->ln1_g, D->ln1_b, D->Win_img, D->b_in, D->Wout_img,
                                 D->b_out, D->ln2_g, D->ln2_b, hin, partial, D->W3_img, D->W3_ximg, D->b3, mask, out, pn, np, S, Gd,
                                 K, stream)))
        return rc;
    }
    hin = out;
    cur ^= 1;
  }
  if (fused) return NAMP_OK;
  return namp_logits_log_softmax(w->Wout_w, w->Wout_b, hin, log_probs, logits, Gd, w->vocab, stream);
}

}  // extern "C"
