// Kernels of the NA-MPNN encoder/decoder path (gfx950).  See namp_device.h for the register
// layout conventions and DESIGN.md for the per-kernel roofline accounting.
#pragma once
#include "namp_device.h"
#ifndef NAMP_MAX_LAYERS
#define NAMP_MAX_LAYERS 8           // == include/namp.h (this header is also compiled without it: namp_persist.hip)
#endif

// ------------------------------------------------------------------------------------------
// pack_image_kernel: nn.Linear weight block -> MFMA fragment image (see namp_device.h).
//   W: [OUT x *] row-major with leading dimension ld; the block is columns [col0, col0+IN).
//   img[tk][tn][lane][r], tk < IN/16, tn < OUT/16.
// ------------------------------------------------------------------------------------------
static __global__ void pack_image_kernel(const float* __restrict__ W, int ld, int col0, int OUT, int IN,
                                  float* __restrict__ img) {
  const int total = OUT * IN;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int r = e & 3, lane = (e >> 2) & 63, t = e >> 8;
    const int ntn = OUT >> 4;
    const int tn = t % ntn, tk = t / ntn;
    const int n = 16 * tn + (lane & 15), k = 16 * tk + 4 * (lane >> 4) + r;
    img[e] = W[(size_t)n * ld + col0 + k];
  }
}

// bf16 fragment image of a [128 x 128] block (namp_device.h, bf16 throughput mode)
static __global__ void pack_image_bf16_kernel(const float* __restrict__ W, int ld, int col0, __bf16* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 128 * 128) return;
  const int j = e & 7, lane = (e >> 3) & 63, tn = (e >> 9) & 7, s = e >> 12;
  const int n = 16 * tn + (lane & 15), k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  img[e] = (__bf16)W[(size_t)n * ld + col0 + k];
}

// x3 image (namp_device.h, chain_gemm_x3): bf16 fragment image of W_hi = bf16(W), then the one of W_mid = bf16(W - W_hi)
static __global__ void pack_image_x3_kernel(const float* __restrict__ W, int ld, int col0, __bf16* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 128 * 128) return;
  const int j = e & 7, lane = (e >> 3) & 63, tn = (e >> 9) & 7, s = e >> 12;
  const int n = 16 * tn + (lane & 15), k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  const float v = W[(size_t)n * ld + col0 + k];
  const __bf16 hi = (__bf16)v;
  img[e] = hi;
  img[128 * 128 + e] = (__bf16)(v - (float)hi);
}

// x3 image of a general block [OUT x IN] (multiples of 16 / 32): hi[s = IN/32][tn = OUT/16][lane][8 bf16] followed by mid in
// the same order — pack_image_x3_kernel's layout for OUT = IN = 128.  Used by the residue-level kernels (W_in, W_out).
static __global__ void pack_image_x3_general_kernel(const float* __restrict__ W, int ld, int col0, int OUT, int IN, __bf16* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= OUT * IN) return;
  const int ntn = OUT >> 4;
  const int j = e & 7, lane = (e >> 3) & 63, t = e >> 9;
  const int tn = t % ntn, s = t / ntn;
  const int n = 16 * tn + (lane & 15), k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  const float v = W[(size_t)n * ld + col0 + k];
  const __bf16 hi = (__bf16)v;
  img[e] = hi;
  img[(size_t)OUT * IN + e] = (__bf16)(v - (float)hi);
}

// All of a training step's weight images in ONE launch (round 5: 116 pack launches per cfg5 step before).  Descriptor table in device memory;
// kind 1 = x3 image of a [128 x 128] block, 2 = bf16 image of a [128 x 128] block, 3 = x3 image of a general [OUT x IN] block; `transposed`
// packs the image of the block's transpose straight from the untransposed storage (element (n, k) = W[k * ld + n]).
struct PackDesc { const float* W; void* img; int ld, out_f, in_f, kind, transposed, first_block; };
static __global__ __launch_bounds__(256) void pack_multi_kernel(const PackDesc* __restrict__ tab, int ndesc) {
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {                                              // last descriptor whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = tab[lo];
  const int e = ((int)blockIdx.x - d.first_block) * 256 + threadIdx.x;
  const int total = d.out_f * d.in_f;
  if (e >= total) return;
  const int ntn = d.out_f >> 4;
  const int j = e & 7, lane = (e >> 3) & 63, t = e >> 9;
  const int tn = t % ntn, s = t / ntn;
  const int n = 16 * tn + (lane & 15), k = 32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  const float v = d.transposed ? d.W[(size_t)k * d.ld + n] : d.W[(size_t)n * d.ld + k];
  __bf16* img = (__bf16*)d.img;
  const __bf16 hi16 = (__bf16)v;
  img[e] = hi16;
  if (d.kind != 2) img[(size_t)total + e] = (__bf16)(v - (float)hi16);
}

// Split-bf16 image of edge_embedding.weight [128 x 5200] for edge_features_kernel<true>: the positional k-tile stays an
// fp32 fragment tile (2048 floats), then one 48 KiB block per RBF chunk c = 3a + bg (6 atom pairs = 3 bf16 K-steps of two
// pairs): [hi: 3 steps x 8 tn x 64 lanes x 8 bf16][mid: same].  Slot j of lane (m, g) in step s is RBF 4g + (j&3) of pair
// 2s + (j>>2) — each lane feeds the RBFs it generates for two consecutive atom pairs, no cross-lane traffic.
static __global__ void pack_feat_x3_kernel(const float* __restrict__ W, int ld, float* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < 2048) {
    const int r = e & 3, lane = (e >> 2) & 63, tn = e >> 8;
    img[e] = W[(size_t)(16 * tn + (lane & 15)) * ld + 4 * (lane >> 4) + r];
  }
  if (e >= 54 * 12288) return;
  const int c = e / 12288, q = e - c * 12288;
  const int j = q & 7, lane = (q >> 3) & 63, tn = (q >> 9) & 7, st = q >> 12;
  const int col = 16 * (1 + 6 * c + 2 * st + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
  const float v = W[(size_t)(16 * tn + (lane & 15)) * ld + col];
  __bf16* blk = (__bf16*)(img + 2048 + (size_t)c * 12288);
  const __bf16 hi = (__bf16)v;
  blk[q] = hi;
  blk[12288 + q] = (__bf16)(v - (float)hi);
}

// ------------------------------------------------------------------------------------------
// gather_cat_kernel — a1 + a3: out[row] = [ nbrs[row][0:C1] | nodes[b*N + idx[row]][0:C2] ]
// (reference gather_nodes / cat_neighbors_nodes, inference/model_utils.py:713-732).
// One float4 per thread, flat over the output so stores are perfectly coalesced; with
// C1 = C2 = 128 a wave moves exactly one output row (512 B streamed + 512 B gathered).
// C1 may be 0 (pure gather_nodes).  C1, C2 multiples of 4.
// ------------------------------------------------------------------------------------------
template <int UNROLL>
__global__ void gather_cat_kernel(const float* __restrict__ nodes, const float* __restrict__ nbrs,
                                  const int32_t* __restrict__ idx, float* __restrict__ out,
                                  long rows, int NK /* N*K rows per batch */, int N, int C1, int C2) {
  const int s1 = C1 >> 2, slots = (C1 + C2) >> 2;
  // XCD-aware row ranges: block b runs on XCD b % 8 (observed dispatch order; only speed depends on it), and each XCD has
  // its own 4 MiB L2.  Giving every XCD one contiguous eighth of the (batch-major) rows keeps the node tables it gathers
  // from — 512 KB per complex — resident in ITS L2; with rows dealt round-robin every L2 saw every table (at B = 64 the
  // gathered rows then came from the fabric: 2.9 GB fetched per launch instead of 1.6).  gridDim.x is a multiple of 8.
  const int xcd = blockIdx.x & 7, slot_b = blockIdx.x >> 3;
  const long rows_per = (rows + 7) >> 3;
  const long r0 = xcd * rows_per;
  const long r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
  const long base = r0 * slots;
  const long total = r1 > r0 ? (r1 - r0) * slots : 0;
  const long stride = (long)(gridDim.x >> 3) * blockDim.x;
  long e0 = (long)slot_b * blockDim.x + threadIdx.x;
  for (; e0 < total; e0 += stride * UNROLL) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long e = base + e0 + u * stride;
      if (e0 + u * stride < total) {
        const long row = e / slots;
        const int s = (int)(e - row * slots);
        if (s < s1) {
          v[u] = *(const f4*)(nbrs + row * C1 + 4 * s);
        } else {
          const long b = row / NK;
          const long j = b * N + idx[row];
          v[u] = *(const f4*)(nodes + j * C2 + 4 * (s - s1));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      // the output is streamed (3 GB at the cfg3 shape, never re-read here): non-temporal stores keep it from evicting the
      // node tables out of L2 (+2-3 %)
      if (e0 + u * stride < total) __builtin_nontemporal_store(v[u], (f4*)(out + 4 * (base + e0 + u * stride)));
    }
  }
}

// scalar fallback for channel counts that are not multiples of 4 (e.g. the C=1 mask gather)
static __global__ void gather_cat_scalar_kernel(const float* __restrict__ nodes, const float* __restrict__ nbrs,
                                         const int32_t* __restrict__ idx, float* __restrict__ out,
                                         long rows, int NK, int N, int C1, int C2) {
  const int ct = C1 + C2;
  const long total = rows * ct;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / ct;
    const int c = (int)(e - row * ct);
    out[e] = (c < C1) ? nbrs[row * C1 + c] : nodes[((row / NK) * N + idx[row]) * C2 + (c - C1)];
  }
}

// ------------------------------------------------------------------------------------------
// Residue-level tail of EncLayer / DecLayer (model_utils.py:690-697,646-656) and the projections the
// next edge kernels gather, as a workgroup-cooperative device function over ONE 16-row tile:
//      x    = LayerNorm1(pre)            pre = h_V + sum_k messages / 30  (caller supplies it, T layout)
//      y    = LayerNorm2(x + W_out . gelu(W_in . x + b_in) + b_out)
//      h_V' = mask * y
//      out_p = W_p . h_V' + bias_p (+ tok_p[S])               p < nproj <= 8
// Waves 0..7 each own 64 of the 512 hidden units (W_in slice (T) -> gelu -> W_out slice (T)); their
// partial [16 x 128] outputs are reduced through LDS.  Projection work is dealt out in (block p,
// channel tile tn) units over all waves.  Used by node_update_kernel (16 residues per workgroup) and
// as the fused tail of the message kernels (the workgroup's own <= 12 residues).
// ------------------------------------------------------------------------------------------
#ifdef NAMP_ABL_STAMPS
// phase attribution of the sampler step (tools/build_variants.sh stamps:-DNAMP_ABL_STAMPS): wall-clock ticks (100 MHz) of workgroup 0,
// summed per phase over the whole launch; read back by namp_debug_stamps()
__device__ long long namp_stamp_acc[16];
__device__ long long namp_stamp_prev;
#define NAMP_STAMP(slot)                                                                          \
  do {                                                                                            \
    __syncthreads();                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                                    \
      const long long now_ = wall_clock64();                                                      \
      namp_stamp_acc[slot] += now_ - namp_stamp_prev;                                             \
      namp_stamp_prev = now_;                                                                     \
    }                                                                                             \
  } while (0)
#define NAMP_STAMP_BEGIN()                                                                         \
  do {                                                                                            \
    __syncthreads();                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) namp_stamp_prev = wall_clock64();                    \
  } while (0)
#elif defined(NAMP_ABL_WSTAMPS)
// Per-WAVE event log of workgroup 0 (tools/build_variants.sh wstamps:-DNAMP_ABL_WSTAMPS; tools/sample_wstamps.py): lane 0 of every wave appends
// (slot, s_memtime) — no barrier, one 16-byte store per event — so the phases' true durations AND the waits at the barriers between them can be
// read off per wave.  (The barrier-per-stamp form above costs ~1 us per stamp and synchronises the waves: good for sums, not for a time line.)
#define NAMP_WS_EVENTS 8192
__device__ long long namp_wstamp_log[8][NAMP_WS_EVENTS][2];
__device__ int namp_wstamp_n[8];                      // events logged by the LAST launch (written at its end)
__shared__ int namp_ws_cur[8];                        // the running event count lives in LDS: a global counter would put a dependent load into every stamp
#define NAMP_STAMP(slot)                                                                          \
  do {                                                                                            \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) {                    \
      const int w_ = threadIdx.x >> 6;                                                            \
      const int c_ = namp_ws_cur[w_];                                                             \
      if ((unsigned)c_ < (unsigned)NAMP_WS_EVENTS) {                                              \
        namp_wstamp_log[w_][c_][0] = (slot);                                                      \
        namp_wstamp_log[w_][c_][1] = (long long)__builtin_amdgcn_s_memtime();                     \
        namp_ws_cur[w_] = c_ + 1;                                                                 \
      }                                                                                           \
    }                                                                                             \
  } while (0)
#define NAMP_WSTAMP_INIT() do { if (threadIdx.x < 8) namp_ws_cur[threadIdx.x] = 0; __syncthreads(); } while (0)
#define NAMP_WSTAMP_FINI() do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x < 8) namp_wstamp_n[threadIdx.x] = namp_ws_cur[threadIdx.x]; } while (0)
#define NAMP_STAMP_BEGIN() do {} while (0)
#else
#define NAMP_STAMP(slot) do {} while (0)
#define NAMP_STAMP_BEGIN() do {} while (0)
#endif
#ifndef NAMP_WSTAMP_INIT
#define NAMP_WSTAMP_INIT() do {} while (0)
#define NAMP_WSTAMP_FINI() do {} while (0)
#endif
struct ProjDesc {
  const float* img;    // 64 KiB image of the [128x128] block
  const float* bias;   // [128] or null
  const float* tok;    // [vocab][128] table added per residue by token S[n], or null
  float* out;          // [G_out][128]
};

struct NodeTail {
  const float* hV;        // [G][128] layer input
  const int32_t* mask;    // [G] or null
  const float* ln1_g; const float* ln1_b;
  const float* Win_img;   // image of W_in  [512 x 128]: img[tk 8][tn 32][lane]
  const float* b_in;      // [512]
  const float* Wout_img;  // image of W_out [128 x 512]: img[tk 32][tn 8][lane]
  const float* b_out;     // [128]
  const float* ln2_g; const float* ln2_b;
  float* hV_out;          // [G][128]
  const int32_t* S;       // [G] tokens for tok tables (or null)
  // Layer 3 of the message MLP, hoisted behind the K-sum (it is linear: sum_k w_k (W3 a_k + b3) = W3 (sum_k w_k a_k) + b3 sum_k w_k):
  // when m3_img is set, the rows handed to the tail are the K-sums of the layer-2 activations (with their weight sums) and the
  // tail starts with x = h_V + W3 . rows + b3 * wsum — one 128 x 128 product per RESIDUE instead of one per edge.
  // null: the rows already are h_V + message (sampler, dec_layer operator).
  const float* m3_img;    // fp32 fragment image of W3 (node_update_multi_kernel<T, true>: its x3 image)
  const float* m3_b;      // b3 [128]
  // optional output head on h_V' (last decoder layer): log_softmax(W_out . h_V' + b), model_utils.py:420-421
  const float* head_w;    // [vocab][128] plain layout, or null
  const float* head_b;    // [vocab]
  float* log_probs;       // [G][vocab]
  float* logits;          // [G][vocab] or null
  int vocab;
  int nproj;
  ProjDesc p[8];
};

// log_softmax head for one residue: y[c] = ybase[c * sc] lives in LDS; lane t < vocab owns logit t.
__device__ __forceinline__ void tail_head_row(const NodeTail& a, const float* ybase, const int sc, const int orow,
                                              const int lane) {
  float z = -INFINITY;
  if (lane < a.vocab) {
    const float* w = a.head_w + (long)lane * NAMP_H;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int c = 0; c < NAMP_H; c += 4) {
      const f4 wv = *(const f4*)(w + c);
      s0 = fmaf(wv.x, ybase[(c + 0) * sc], s0); s1 = fmaf(wv.y, ybase[(c + 1) * sc], s1);
      s2 = fmaf(wv.z, ybase[(c + 2) * sc], s2); s3 = fmaf(wv.w, ybase[(c + 3) * sc], s3);
    }
    z = (s0 + s1) + (s2 + s3) + a.head_b[lane];
  }
  float mx = z;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e = (lane < a.vocab) ? expf(z - mx) : 0.f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) e += __shfl_xor(e, o);
  if (lane < a.vocab) {
    a.log_probs[(long)orow * a.vocab + lane] = (z - mx) - logf(e);
    if (a.logits) a.logits[(long)orow * a.vocab + lane] = z;
  }
}

#define FFN_LD 132   // padded row stride (floats) of the LDS tiles: conflict-free ds_write_b128
#define NODE_TAIL_LDS ((16 + 16 + 8 * 16) * FFN_LD * 4)

// x: pre-LayerNorm1 rows in the T layout (lane (m,g) holds channels 16t+4g+r of tile row m).
// row0: first residue of the tile, nrows: rows of the tile that are real residues, G: total residues.
// lds: NODE_TAIL_LDS bytes.  DEEP: request every weight fragment of a phase up front (256-VGPR budget).
template <bool DEEP>
__device__ __forceinline__ void node_tail(const NodeTail& a, f4 (&x)[8], const float wsum_m, const int row0, const int nrows, const int G,
                                          float* lds, const int tid, const int wave, const int nwaves, const int lane) {
  float* xs = lds;                          // [16][FFN_LD]    x = LN1(...)
  float* ys = xs + 16 * FFN_LD;             // [16][FFN_LD]    h_V' tile for the projections
  float* ps = ys + 16 * FFN_LD;             // [8][16][FFN_LD] per-wave partial FFN outputs
  const int m = lane & 15, g = lane >> 4;

  if (a.m3_img) {
    // hoisted message layer 3: x holds the K-sums (every wave the whole tile); wave w evaluates channel tile w
    if (wave < 8) {
      f4 o[1] = {*(const f4*)(a.m3_b + 16 * wave + 4 * g) * wsum_m};
      chain_gemm_global<8, 1, false>(o, x, (const f4*)a.m3_img + wave * 64 + lane, 8);
      *(f4*)(ys + m * FFN_LD + 16 * wave + 4 * g) = o[0];
    }
    __syncthreads();
    const int hr = (m < nrows && row0 + m < G) ? row0 + m : row0;
    const float* hsrc = a.hV + (long)hr * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(hsrc + 16 * t) + *(const f4*)(ys + m * FFN_LD + 16 * t + 4 * g);
  }

  f4 win[DEEP ? 8 : 1][4];
  if (DEEP && wave < 8) {
    const f4* w = (const f4*)a.Win_img + (4 * wave) * 64 + lane;
#pragma unroll
    for (int tk = 0; tk < 8; ++tk)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) win[DEEP ? tk : 0][tn] = w[(tk * 32 + tn) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  layernorm_row_T(x, a.ln1_g, a.ln1_b, g);
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(xs + m * FFN_LD + 16 * t + 4 * g) = x[t];
  }
  if (wave < 8) {
    f4 hacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) hacc[t] = *(const f4*)(a.b_in + 64 * wave + 16 * t + 4 * g);
    f4 oacc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) oacc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    if (DEEP) {
#pragma unroll
      for (int tk = 0; tk < 8; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) hacc[tn] = mfma4(win[DEEP ? tk : 0][tn][r], x[tk][r], hacc[tn]);
      f4 wo[4][8];
      const f4* w = (const f4*)a.Wout_img + (4 * wave) * 8 * 64 + lane;
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) wo[tk][tn] = w[(tk * 8 + tn) * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) hacc[t] = gelu4(hacc[t]);
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tn = 0; tn < 8; ++tn) oacc[tn] = mfma4(wo[tk][tn][r], hacc[tk][r], oacc[tn]);
    } else {
      chain_gemm_global<8, 4, false>(hacc, x, (const f4*)a.Win_img + (4 * wave) * 64 + lane, 32);
#pragma unroll
      for (int t = 0; t < 4; ++t) hacc[t] = gelu4(hacc[t]);
      chain_gemm_global<4, 8, false>(oacc, hacc, (const f4*)a.Wout_img + (4 * wave) * 8 * 64 + lane, 8);
    }
    float* dst = ps + (wave * 16 + m) * FFN_LD + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = oacc[t];
  }
  __syncthreads();
  // LN2: thread -> (row = tid/32, 4 channels)
  if (tid < 512) {
    const int r = tid >> 5, c = (tid & 31) * 4;
    f4 v = *(const f4*)(xs + r * FFN_LD + c) + *(const f4*)(a.b_out + c);
#pragma unroll
    for (int w = 0; w < 8; ++w) v += *(const f4*)(ps + (w * 16 + r) * FFN_LD + c);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / 128.0f);
    v -= mean;
    float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / 128.0f) + 1e-5f);
    const int orow = row0 + r;
    const bool ok = (r < nrows) && (orow < G);
    const float mk = (a.mask && ok) ? (float)a.mask[orow] : 1.0f;
    const f4 y = (v * rstd * *(const f4*)(a.ln2_g + c) + *(const f4*)(a.ln2_b + c)) * mk;
    *(f4*)(ys + r * FFN_LD + c) = y;
    if (ok) *(f4*)(a.hV_out + (long)orow * NAMP_H + c) = y;
  }
  if (a.nproj == 0 && !a.head_w) return;
  __syncthreads();
  if (a.head_w) {
    for (int n = wave; n < nrows; n += nwaves)
      if (row0 + n < G) tail_head_row(a, ys + n * FFN_LD, 1, row0 + n, lane);
  }
  // projections of h_V': unit u = (block p, channel tile tn)
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(ys + m * FFN_LD + 16 * t + 4 * g);
  const int row = row0 + m;
  const bool valid = (m < nrows) && (row < G);
  const int rr = valid ? row : row0;
  // unit u = 8*pi + tn goes to wave u % nwaves.  The block loop is unrolled so that a.p[pi] is a static
  // index (a dynamic index into the kernarg struct makes hipcc spill the whole struct to scratch).
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) {
    if (pi >= a.nproj) break;
    const ProjDesc d = a.p[pi];
    for (int tn = ((wave - pi * 8) % nwaves + nwaves) % nwaves; tn < 8; tn += nwaves) {
      f4 wf[8];
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wf[tk] = ((const f4*)d.img)[(tk * 8 + tn) * 64 + lane];
      f4 acc = d.bias ? *(const f4*)(d.bias + 16 * tn + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
      if (d.tok) acc += *(const f4*)(d.tok + (long)a.S[rr] * NAMP_H + 16 * tn + 4 * g);
#pragma unroll
      for (int tk = 0; tk < 8; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(wf[tk][r], x[tk][r], acc);
      if (valid) *(f4*)(d.out + (long)row * NAMP_H + 16 * tn + 4 * g) = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// node_tail_rows<R> — the same residue tail for a tile of <= R (4 or 8) real residues: the fused message
// kernel owns 12/TPN residues per workgroup (4 at K=48, 6 at K=32), and the sampler's workgroup owns one
// residue per sample stream.  A 16-row MFMA tile would spend most of its issue slots on padding, which
// made the tail MFMA-bound (12 us of a 21 us tail at 4 rows); with few rows the work is bound by
// streaming the 768 KiB of weights through the CU anyway, so it runs on the VALU instead: lane
// (n_local, g) of fragment (tk, tn) holds W[16tn + n_local][16tk + 4g + r], multiplies it with the R
// residues' activations x[.][16tk + 4g + r] (broadcast LDS reads of a [k][R] transposed tile) and the four
// g-lanes of a channel are summed at the end — the existing MFMA weight images are reused as is.
// Tile row n is residue orow[n] (global row, < 0 = padding); rows need not be consecutive.
// ------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

// acc[n] += sum_k w[k] * x[k][n] for the R residues of the tile: two residues per v_pk_fma_f32 (the weight broadcast to
// both halves) — this loop is instruction-issue bound, and the packed form halves its instruction count.
// BULK: request the unit's LDS operands in two halves of 16 reads instead of k-tile by k-tile (one wait per half instead of per k-tile;
// 64 more live registers: the sampler's 256-register workgroups only)
template <int R, bool BULK = false>
__device__ __forceinline__ void rows_fma(float (&acc)[R], const f4 (&wf)[8], const float* xT, const int g) {
  if constexpr (BULK && R == 4) {
    f2 a0 = (f2){acc[0], acc[1]}, a1 = (f2){acc[2], acc[3]};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f4 xv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = *(const f4*)(xT + (16 * (4 * half + (q >> 2)) + 4 * g + (q & 3)) * 4);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float w = wf[4 * half + (q >> 2)][q & 3];
        const f2 w2 = (f2){w, w};
        a0 = __builtin_elementwise_fma(w2, (f2){xv[q].x, xv[q].y}, a0);
        a1 = __builtin_elementwise_fma(w2, (f2){xv[q].z, xv[q].w}, a1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
    return;
  }
  f2 a2[R / 2];
#pragma unroll
  for (int q = 0; q < R / 2; ++q) a2[q] = (f2){acc[2 * q], acc[2 * q + 1]};
#pragma unroll
  for (int tk = 0; tk < 8; ++tk) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xp = xT + (16 * tk + 4 * g + r) * R;                 // R residues at this k
      const f2 w2 = (f2){wf[tk][r], wf[tk][r]};
#pragma unroll
      for (int q = 0; q < R; q += 4) {
        const f4 xv = *(const f4*)(xp + q);
        a2[q / 2] = __builtin_elementwise_fma(w2, (f2){xv.x, xv.y}, a2[q / 2]);
        a2[q / 2 + 1] = __builtin_elementwise_fma(w2, (f2){xv.z, xv.w}, a2[q / 2 + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later k-tiles from being hoisted (register pressure)
  }
#pragma unroll
  for (int q = 0; q < R / 2; ++q) { acc[2 * q] = a2[q].x; acc[2 * q + 1] = a2[q].y; }
}

#define ROWS_TAIL_LDS_FLOATS(R) ((128 + 512 + 4 * 128 + 128) * (R) + 64 + 128 * (R))

// RowFn: int operator()(int n) -> global row of tile row n (any runtime n in [0,R)), < 0 for padding.
struct ConsecutiveRows {
  int row0, nrows, G;
  __device__ __forceinline__ int operator()(int n) const { return (n < nrows && row0 + n < G) ? row0 + n : -1; }
};

// SC1: outputs are stored WRITE-THROUGH (sc1, agent scope) — what another workgroup of the same launch gathers after a grid
// barrier must not sit dirty in this XCD's L2 (persistent forward; no release fence is needed then, Guideline 16 R1).
template <bool SC1>
__device__ __forceinline__ void st_out(float* p, const float v) {
  if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// The units of one phase (a 16-channel output tile = 8 weight fragments of 1 KiB per wave) dealt over the waves: unit u, u + nwaves, ...
// PF = false: load the unit's fragments, then compute (the fused edge launches: 250 workgroups stream the same weights, the phase
// is bound by L2 bandwidth chip-wide and a deeper queue measured nothing, profiles/r02k).  PF = true (the sampler: ONE workgroup per
// <= 4 residues with the chip idle around it — the phase is bound by the latency of each unit's 8 KiB): the next unit's fragments
// are requested before the current unit's FMAs, so two units are in flight per wave.  frag(u, tk) -> address of fragment tk.
// rot (PF = false; total a power of two): the units are visited as (u + rot) % total.  Every workgroup of a fused launch streams the SAME
// weights in the same order, so at any moment the whole chip asks a few L2 channels for the same ~100 KiB; rot = the workgroup's index
// within its XCD spreads the 32 CUs of an XCD over the whole matrix (profiles/r04g).  The result does not depend on the order.
template <bool PF, class FragFn, class BodyFn>
__device__ __forceinline__ void tail_units(const int u0, const int total, const int step, const int lane, FragFn frag, BodyFn body,
                                           const int rot = 0) {
  if constexpr (!PF) {
#pragma unroll 1
    for (int u_ = u0; u_ < total; u_ += step) {
      const int u = (u_ + rot) & (total - 1);
      f4 wf[8];
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wf[tk] = frag(u, tk)[lane];
      body(u, wf);
    }
  } else {
    f4 wa[8], wb[8];
    int u = u0;
    if (u < total) {
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wa[tk] = frag(u, tk)[lane];
    }
#pragma unroll 1
    while (u < total) {
      int un = u + step;
      if (un < total) {
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) wb[tk] = frag(un, tk)[lane];
      }
      body(u, wa);
      u = un;
      if (u >= total) break;
      un = u + step;
      if (un < total) {
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) wa[tk] = frag(un, tk)[lane];
      }
      body(u, wb);
      u = un;
    }
  }
}

// ------------------------------------------------------------------------------------------
// node_tail_mfma4<R> — the residue tail of <= R (4 or 8) rows on v_mfma_f32_4x4x1_16b_f32 (round 4).  The VALU form below reads the
// rows' activations from LDS for every weight element (one broadcast ds_read_b128 per two packed FMAs): at 4 rows the tail of the fused
// cfg2 launches was bound by LDS bandwidth (~10 us of its 18 us; an 8-row tail on half of the workgroups — half the weight bytes — took 27 us,
// profiles/r04g).  The 4x4x1 MFMA is 16 independent 4 x 4 outer products per instruction (D[b][i][j] += A[b][i] B[b][j], lane = 4b + i for
// A, 4b + j for B and D; exact fp32 FMAs at ~23 MAC / clk / SIMD, the rate of the packed VALU FMAs it replaces) — exactly the shape of "4
// channels x 4 rows at one k".  With block b = (kb, cb): 4 k-slots x 4 channel quads,
//     A: lane (kb, cb, i) <- W[16u + 4cb + i][16S + 4kb + r]      = register r of fragment S of the EXISTING fp32 MFMA image, same lane
//     B: lane (kb, cb, j) <- x[16S + 4kb + r][row j]              = 32 registers for a 128-long reduction, loaded once per phase
// one unit (16 channels x 128 k) is 32 MFMAs on registers only, then one sum over the four kb lane groups.  Row tiles are kept in LDS in
// "B order" ([S][kb][row][r] floats) so that operand loads and the 16-channel results are 16-byte accesses.  Units, weight images,
// LDS regions 0-3 and the arguments are those of node_tail_rows (which forwards here unless NAMP_TAIL_VALU is defined); results differ from
// the VALU form in the order of the fp32 additions only.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int tail4_off(const int k, const int j) { return (((k >> 4) * 4 + ((k >> 2) & 3)) * 4 + j) * 4 + (k & 3); }

template <int NB>
__device__ __forceinline__ void tail4_x(f4 (&xr)[NB][8], const float* xB, const int lane) {
  const float* p = xB + ((lane >> 4) * 4 + (lane & 3)) * 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int S = 0; S < 8; ++S) xr[nb][S] = *(const f4*)(p + nb * 512 + S * 64);
}

// o[nb][i] = sum_k W[16u + 4cb + i][k] x[k][4nb + j] on every lane (kb, cb, j)
template <int NB>
__device__ __forceinline__ void tail4_unit(f4 (&o)[NB], const f4 (&wf)[8], const f4 (&xr)[NB][8]) {
  f4 c0[NB], c1[NB];
#ifdef NAMP_ABL_T4_NOMFMA
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    o[nb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < 8; ++S) o[nb] += wf[S] * xr[nb][S];
  }
  return;
#endif
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { c0[nb] = (f4){0.f, 0.f, 0.f, 0.f}; c1[nb] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int S = 0; S < 8; ++S) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      c0[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[S].x, xr[nb][S].x, c0[nb], 0, 0, 0);
      c1[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[S].y, xr[nb][S].y, c1[nb], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      c0[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[S].z, xr[nb][S].z, c0[nb], 0, 0, 0);
      c1[nb] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[S].w, xr[nb][S].w, c1[nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f4 v = c0[nb] + c1[nb];
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] += __shfl_xor(v[c], 16); v[c] += __shfl_xor(v[c], 32); }
    o[nb] = v;
  }
}

// The fragments of a wave's first (up to) three units of a phase, requested together and AHEAD of the barrier that precedes the phase — the
// weights do not depend on the rows.  With the arithmetic on the matrix pipe a unit costs ~0.15 us; requesting its 8 KiB when the unit starts
// left three exposed L2 round trips per phase (1.4-1.6 us each with 250 workgroups asking at once, profiles/r04g).
#define TAIL4_AHEAD 3
template <class FragFn>
__device__ __forceinline__ void tail4_request1(f4 (&wf)[8], const int u, const int lane, FragFn frag) {
#ifdef NAMP_ABL_T4_NOLOAD
#pragma unroll
  for (int tk = 0; tk < 8; ++tk) wf[tk] = (f4){1.f * u, 2.f * lane, 3.f, 4.f * tk};
  return;
#endif
#pragma unroll
  for (int tk = 0; tk < 8; ++tk) wf[tk] = frag(u, tk)[lane];
}
template <class FragFn>
__device__ __forceinline__ void tail4_request(f4 (&wf)[TAIL4_AHEAD][8], const int u0, const int total, const int step, const int lane, FragFn frag) {
#pragma unroll
  for (int t = 0; t < TAIL4_AHEAD; ++t) {
    const int u = u0 + t * step;
    tail4_request1(wf[t], u < total ? u : u0, lane, frag);           // (no branch around the loads; a repeated unit is an L1 hit)
  }
}
// the phase itself: the requested units — slot t is refilled by next(t) as soon as its unit is done, so that the NEXT phase's fragments travel
// while this phase's other units and the barrier run — then (workgroups of fewer than 11 waves) the remaining units on demand
template <class FragFn, class BodyFn, class NextFn>
__device__ __forceinline__ void tail4_phase(f4 (&wf)[TAIL4_AHEAD][8], const int u0, const int total, const int step, const int lane, FragFn frag,
                                            BodyFn body, NextFn next) {
#pragma unroll
  for (int t = 0; t < TAIL4_AHEAD; ++t) {
    const int u = u0 + t * step;
    if (u < total) body(u, wf[t]);
    next(t);
  }
  tail_units<false>(u0 + TAIL4_AHEAD * step, total, step, lane, frag, body);
}

template <int R, typename RowFn, bool SC1, bool M3, bool PF, bool AH>
__device__ __forceinline__ void node_tail_mfma4(const NodeTail& a, f4 (&x)[8], const float wsum_m, const RowFn orow, float* lds,
                                                const int tid, const int wave, const int nwaves, const int lane) {
  static_assert(R == 4 || R == 8, "node_tail_mfma4: 4 or 8 rows");
  constexpr int NB = R / 4;
  // AH (the fused launches at 4 rows): a phase's fragments requested ahead of it.  PF (the sampler's lone workgroups): two units in flight
  // per wave, as node_tail_rows.  Neither: one unit at a time.
  constexpr bool AHEAD = AH && !PF && R == 4;
  float* xB = lds;                    // [NB][512]      x = LN1(...) in B order                      (region 0 of node_tail_rows)
  float* hB = xB + 128 * R;           // [4][NB][512]   gelu(W_in x + b_in), per 128-wide quarter of the hidden layer, B order
  float* oP = hB + 512 * R;           // [4][R][128]    W_out partials over the quarters, row-major; before that: [R][128] layer-3 rows
  float* yT = oP + 4 * 128 * R;       // [128][R]       h_V' (k-major, as node_tail_rows leaves it: output head, sampler)
  float* red = yT + 128 * R;          // [2][32]
  float* yB = red + 64;               // [NB][512]      h_V' in B order for the projections
  const int m = lane & 15, g = lane >> 4;
  const int cb = (lane >> 2) & 3, j = lane & 3;        // as an operand / result lane (kb = g)
  const bool st_lane = lane < 16;                      // the kb = 0 group stores a unit's results
  const auto win_frag = [&](const int u, const int tk) { return (const f4*)a.Win_img + (tk * 32 + u) * 64; };
  const auto wout_frag = [&](const int u, const int tk) { return (const f4*)a.Wout_img + ((8 * (u >> 3) + tk) * 8 + (u & 7)) * 64; };

  NAMP_STAMP_BEGIN();
  f4 wa[AHEAD ? TAIL4_AHEAD : 1][8];
  float bin_a[TAIL4_AHEAD] = {0.f, 0.f, 0.f};
  if constexpr (AHEAD) {
    tail4_request(wa, wave, 32, nwaves, lane, win_frag);
#pragma unroll
    for (int t = 0; t < TAIL4_AHEAD; ++t) {
      const int u = wave + t * nwaves;
      bin_a[t] = a.b_in[16 * (u < 32 ? u : wave) + 4 * cb + g];
    }
  }
  if (M3 && a.m3_img) {
    if (wave == 0 && m < R) {
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(xB + (m >> 2) * 512 + t * 64 + g * 16 + (m & 3) * 4) = x[t];
      if (g == 0) red[m] = wsum_m;
    }
    __syncthreads();
    {
      f4 xr[NB][8];
      tail4_x<NB>(xr, xB, lane);
      tail_units<PF>(wave, 8, nwaves, lane,
                     [&](const int u, const int tk) { return (const f4*)a.m3_img + (tk * 8 + u) * 64; },
                     [&](const int u, const f4 (&wf)[8]) {
                       f4 o[NB];
                       tail4_unit<NB>(o, wf, xr);
                       if (st_lane) {
                         const f4 b = *(const f4*)(a.m3_b + 16 * u + 4 * cb);
#pragma unroll
                         for (int nb = 0; nb < NB; ++nb) *(f4*)(oP + (4 * nb + j) * NAMP_H + 16 * u + 4 * cb) = o[nb] + b * red[4 * nb + j];
                       }
                     });
    }
    __syncthreads();
    const int mr = m < R ? m : 0;
    int hr = orow(mr);
    if (hr < 0) hr = orow(0) < 0 ? 0 : orow(0);
    const float* hsrc = a.hV + (long)hr * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(hsrc + 16 * t) + *(const f4*)(oP + mr * NAMP_H + 16 * t + 4 * g);
  }

  NAMP_STAMP(7);                       // hoisted layer 3 (when done here)
  layernorm_row_T(x, a.ln1_g, a.ln1_b, g);
  if (wave == 0 && m < R) {
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(xB + (m >> 2) * 512 + t * 64 + g * 16 + (m & 3) * 4) = x[t];
  }
  __syncthreads();
  NAMP_STAMP(8);                       // LayerNorm 1
  // ---- hidden = gelu(W_in x + b_in): 32 units; every lane holds a unit's sums after the kb reduction, lane group kb takes component kb
  {
    f4 xr[NB][8];
    tail4_x<NB>(xr, xB, lane);
    const auto body = [&](const int u, const f4 (&wf)[8]) {
      f4 o[NB];
      tail4_unit<NB>(o, wf, xr);
      const int t_ = AHEAD ? (u - wave) / nwaves : 0;
      const float b = (AHEAD && t_ < TAIL4_AHEAD) ? (t_ == 0 ? bin_a[0] : t_ == 1 ? bin_a[1] : bin_a[2]) : a.b_in[16 * u + 4 * cb + g];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float v = g == 0 ? o[nb].x : g == 1 ? o[nb].y : g == 2 ? o[nb].z : o[nb].w;
        hB[((u >> 3) * NB + nb) * 512 + (u & 7) * 64 + cb * 16 + j * 4 + g] = gelu_erf(v + b);
      }
    };
    if constexpr (AHEAD)
      tail4_phase(wa, wave, 32, nwaves, lane, win_frag, body, [&](const int t) {
        const int u = wave + t * nwaves;
        tail4_request1(wa[t], u < 32 ? u : wave, lane, wout_frag);
      });
    else tail_units<PF>(wave, 32, nwaves, lane, win_frag, body);
  }
  __syncthreads();
  NAMP_STAMP(9);                       // W_in + GELU
  // ---- W_out: units (channel tile tn, hidden quarter q); partials summed in the LayerNorm2 pass
  {
    const auto body = [&](const int u, const f4 (&wf)[8]) {
      const int tn = u & 7, q = u >> 3;
      f4 xr[NB][8];
      tail4_x<NB>(xr, hB + q * NB * 512, lane);
      f4 o[NB];
      tail4_unit<NB>(o, wf, xr);
      if (st_lane) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(oP + (q * R + 4 * nb + j) * NAMP_H + 16 * tn + 4 * cb) = o[nb];
      }
    };
    // (refill: the wave's unit of projection block pi = t — at most one per block with >= 8 waves: channel tile (wave - 8 pi) mod nwaves, if < 8)
    if constexpr (AHEAD)
      tail4_phase(wa, wave, 32, nwaves, lane, wout_frag, body, [&](const int pi) {
        if (pi < a.nproj) {
          const int tn = ((wave - pi * 8) % nwaves + nwaves) % nwaves;
          const f4* img = (const f4*)(pi == 0 ? a.p[0].img : pi == 1 ? a.p[1].img : a.p[2].img) + ((tn < 8 ? tn : 0)) * 64 + lane;
#pragma unroll
          for (int tk = 0; tk < 8; ++tk) wa[pi][tk] = img[tk * 8 * 64];
        }
      });
    else tail_units<PF>(wave, 32, nwaves, lane, wout_frag, body);
  }
  __syncthreads();
  NAMP_STAMP(10);                      // W_out partials
  // ---- LayerNorm2 over channels: thread -> (row n = tid / 128, channel c = tid % 128), 128 * R threads
  const int n_ = tid >> 7, c_ = tid & 127;
  const bool ln_thr = tid < 128 * R;
  const int boff = ln_thr ? (n_ >> 2) * 512 + tail4_off(c_, n_ & 3) : 0;
  float v = 0.f;
  if (ln_thr) {
    v = xB[boff] + a.b_out[c_];
#pragma unroll
    for (int q = 0; q < 4; ++q) v += oP[(q * R + n_) * NAMP_H + c_];
    float s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
  }
  __syncthreads();
  float d = 0.f;
  if (ln_thr) {
    const float mean = (red[2 * n_] + red[2 * n_ + 1]) * (1.0f / 128.0f);
    d = v - mean;
    float q = d * d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[32 + wave] = q;
  }
  __syncthreads();
  if (ln_thr) {
    const float rstd = rsqrtf((red[32 + 2 * n_] + red[32 + 2 * n_ + 1]) * (1.0f / 128.0f) + 1e-5f);
    const int orw = orow(n_);
    const float mk = (a.mask && orw >= 0) ? (float)a.mask[orw] : 1.0f;
    const float y = (d * rstd * a.ln2_g[c_] + a.ln2_b[c_]) * mk;
    yT[c_ * R + n_] = y;
    yB[boff] = y;
    if (orw >= 0) st_out<SC1>(a.hV_out + (long)orw * NAMP_H + c_, y);
  }
  NAMP_STAMP(11);                      // partial sums + LayerNorm 2 + h_V' out
  if (a.nproj == 0 && !a.head_w) return;
  __syncthreads();
  if (a.head_w) {
    for (int n = wave; n < R; n += nwaves) {
      const int orw = orow(n);
      if (orw >= 0) tail_head_row(a, yT + n, R, orw, lane);
    }
  }
  // ---- projections of h_V': unit v = 8 pi + tn (block pi, channel tile tn) -> wave v % nwaves
  f4 xr[NB][8];
  tail4_x<NB>(xr, yB, lane);
  int orw[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) orw[nb] = orow(4 * nb + j);
  auto proj_out = [&](const f4 (&o)[NB], const float* bias, const float* tok, float* out, const int tn) {
    if (!st_lane) return;
    const int c = 16 * tn + 4 * cb;
    const f4 b = bias ? *(const f4*)(bias + c) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (orw[nb] < 0) continue;
      f4 r = o[nb] + b;
      if (tok) r += *(const f4*)(tok + (long)a.S[orw[nb]] * NAMP_H + c);
      float* dst = out + (long)orw[nb] * NAMP_H + c;
      if (SC1) { st_out<true>(dst, r.x); st_out<true>(dst + 1, r.y); st_out<true>(dst + 2, r.z); st_out<true>(dst + 3, r.w); }
      else *(f4*)dst = r;
    }
  };
  if constexpr (PF) {
    // (the sampler's tails have at most two blocks: a runtime index into a.p[] would copy the descriptors to scratch)
    tail_units<true>(wave, 8 * (a.nproj < 2 ? a.nproj : 2), nwaves, lane,
                     [&](const int v, const int tk) { return (const f4*)((v >> 3) ? a.p[1].img : a.p[0].img) + (tk * 8 + (v & 7)) * 64; },
                     [&](const int v, const f4 (&wf)[8]) {
                       const bool second = (v >> 3) != 0;
                       f4 o[NB];
                       tail4_unit<NB>(o, wf, xr);
                       proj_out(o, second ? a.p[1].bias : a.p[0].bias, second ? a.p[1].tok : a.p[0].tok, second ? a.p[1].out : a.p[0].out, v & 7);
                     });
    NAMP_STAMP(12);
    return;
  }
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) {
    if (pi >= a.nproj) break;
    const ProjDesc pd = a.p[pi];
    bool first = AHEAD && pi < TAIL4_AHEAD;
#pragma unroll 1
    for (int tn = ((wave - pi * 8) % nwaves + nwaves) % nwaves; tn < 8; tn += nwaves) {
      f4 o[NB];
      if (first) {
        tail4_unit<NB>(o, wa[pi < TAIL4_AHEAD ? pi : 0], xr);
      } else {
        f4 wf[8];
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) wf[tk] = ((const f4*)pd.img)[(tk * 8 + tn) * 64 + lane];
        tail4_unit<NB>(o, wf, xr);
      }
      first = false;
      proj_out(o, pd.bias, pd.tok, pd.out, tn);
    }
  }
  NAMP_STAMP(12);                      // output head + projections
}

// M3 = false: the caller has already applied the hoisted message layer 3 (x = h_V + message), whatever a.m3_img says
template <int R, typename RowFn, bool SC1 = false, bool M3 = true, bool PF = false, bool AH = false>
__device__ __forceinline__ void node_tail_rows(const NodeTail& a, f4 (&x)[8], const float wsum_m, const RowFn orow, float* lds,
                                               const int tid, const int wave, const int nwaves, const int lane) {
#ifndef NAMP_TAIL_VALU
  node_tail_mfma4<R, RowFn, SC1, M3, PF, AH>(a, x, wsum_m, orow, lds, tid, wave, nwaves, lane);
  return;
#endif
  float* xT = lds;                    // [128][R]    x = LN1(...)          (k-major, residue-minor)
  float* hT = xT + 128 * R;           // [512][R]    gelu(W_in x + b_in)
  float* oP = hT + 512 * R;           // [4][128][R] W_out partials over k-quarters
  float* yT = oP + 4 * 128 * R;       // [128][R]    h_V'
  float* red = yT + 128 * R;          // [2][32]     LayerNorm2 cross-wave sums
  const int m = lane & 15, g = lane >> 4;

  NAMP_STAMP_BEGIN();
  if (M3 && a.m3_img) {
    // hoisted message layer 3 (see NodeTail): x = K-sums of the layer-2 activations -> x = h_V + W3 . x + b3 * wsum
    if (wave == 0 && m < R) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        xT[(16 * t + 4 * g + 0) * R + m] = x[t].x; xT[(16 * t + 4 * g + 1) * R + m] = x[t].y;
        xT[(16 * t + 4 * g + 2) * R + m] = x[t].z; xT[(16 * t + 4 * g + 3) * R + m] = x[t].w;
      }
      if (g == 0) red[m] = wsum_m;
    }
    __syncthreads();
#pragma unroll 1
    for (int tn = wave; tn < 8; tn += nwaves) {
      f4 wf[8];
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wf[tk] = ((const f4*)a.m3_img)[(tk * 8 + tn) * 64 + lane];
      float acc[R];
#pragma unroll
      for (int n = 0; n < R; ++n) acc[n] = 0.f;
      rows_fma<R>(acc, wf, xT, g);
#pragma unroll
      for (int n = 0; n < R; ++n) acc[n] = xg_sum(acc[n]);
      if (g == 0) {
        const float b = a.m3_b[16 * tn + m];
#pragma unroll
        for (int n = 0; n < R; ++n) yT[(16 * tn + m) * R + n] = acc[n] + b * red[n];
      }
    }
    __syncthreads();
    const int mr = m < R ? m : 0;
    int hr = orow(mr);
    if (hr < 0) hr = orow(0) < 0 ? 0 : orow(0);
    const float* hsrc = a.hV + (long)hr * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float* yp = yT + (16 * t + 4 * g) * R + mr;
      x[t] = *(const f4*)(hsrc + 16 * t) + (f4){yp[0], yp[R], yp[2 * R], yp[3 * R]};
    }
  }

  NAMP_STAMP(7);
  layernorm_row_T(x, a.ln1_g, a.ln1_b, g);
  if (wave == 0 && m < R) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      xT[(16 * t + 4 * g + 0) * R + m] = x[t].x; xT[(16 * t + 4 * g + 1) * R + m] = x[t].y;
      xT[(16 * t + 4 * g + 2) * R + m] = x[t].z; xT[(16 * t + 4 * g + 3) * R + m] = x[t].w;
    }
  }
  __syncthreads();
  NAMP_STAMP(8);
  // ---- hidden = gelu(W_in x + b_in): 32 channel tiles dealt over the waves
#ifdef NAMP_ABL_NOROT
  const int rot = 0;
#else
  const int rot = PF ? 0 : (int)((blockIdx.x >> 3) & 31);       // the workgroup's index within its XCD (see tail_units)
#endif
#ifdef NAMP_ABL_TAILPF
  constexpr bool PFU = true;
#else
  constexpr bool PFU = PF;
#endif
  tail_units<PFU>(wave, 32, nwaves, lane,
                 [&](const int tn, const int tk) { return (const f4*)a.Win_img + (tk * 32 + tn) * 64; },
                 [&](const int tn, const f4 (&wf)[8]) {
                   float acc[R];
#pragma unroll
                   for (int n = 0; n < R; ++n) acc[n] = 0.f;
                   rows_fma<R, PF>(acc, wf, xT, g);
#pragma unroll
                   for (int n = 0; n < R; ++n) acc[n] = xg_sum(acc[n]);
                   if (g == 0) {
                     const float b = a.b_in[16 * tn + m];
#pragma unroll
                     for (int n = 0; n < R; ++n) hT[(16 * tn + m) * R + n] = gelu_erf(acc[n] + b);
                   }
                 }, rot);
  __syncthreads();
  NAMP_STAMP(9);
  // ---- W_out: units (channel tile tn, k-quarter kq); partials reduced in the LayerNorm2 pass
  tail_units<PFU>(wave, 32, nwaves, lane,
                 [&](const int u, const int tk) { return (const f4*)a.Wout_img + ((8 * (u >> 3) + tk) * 8 + (u & 7)) * 64; },
                 [&](const int u, const f4 (&wf)[8]) {
                   const int tn = u & 7, kq = u >> 3;
                   float acc[R];
#pragma unroll
                   for (int n = 0; n < R; ++n) acc[n] = 0.f;
                   rows_fma<R, PF>(acc, wf, hT + 128 * kq * R, g);
#pragma unroll
                   for (int n = 0; n < R; ++n) acc[n] = xg_sum(acc[n]);
                   if (g == 0) {
#pragma unroll
                     for (int n = 0; n < R; ++n) oP[(kq * 128 + 16 * tn + m) * R + n] = acc[n];
                   }
                 }, rot);
  __syncthreads();
  NAMP_STAMP(10);
  // ---- LayerNorm2 over channels: thread -> (residue n = tid / 128, channel c = tid % 128), 128*R threads
  const int n_ = tid >> 7, c_ = tid & 127;
  const bool ln_thr = tid < 128 * R;
  float v = 0.f;
  if (ln_thr) {
    v = xT[c_ * R + n_] + a.b_out[c_];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) v += oP[(kq * 128 + c_) * R + n_];
    float s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
  }
  __syncthreads();
  float d = 0.f;
  if (ln_thr) {
    const float mean = (red[2 * n_] + red[2 * n_ + 1]) * (1.0f / 128.0f);
    d = v - mean;
    float q = d * d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[32 + wave] = q;
  }
  __syncthreads();
  if (ln_thr) {
    const float rstd = rsqrtf((red[32 + 2 * n_] + red[32 + 2 * n_ + 1]) * (1.0f / 128.0f) + 1e-5f);
    const int orw = orow(n_);
    const float mk = (a.mask && orw >= 0) ? (float)a.mask[orw] : 1.0f;
    const float y = (d * rstd * a.ln2_g[c_] + a.ln2_b[c_]) * mk;
    yT[c_ * R + n_] = y;
    if (orw >= 0) st_out<SC1>(a.hV_out + (long)orw * NAMP_H + c_, y);
  }
  NAMP_STAMP(11);
  if (a.nproj == 0 && !a.head_w) return;
  __syncthreads();
  if (a.head_w) {
    for (int n = wave; n < R; n += nwaves) {
      const int orw = orow(n);
      if (orw >= 0) tail_head_row(a, yT + n, R, orw, lane);
    }
  }
  // ---- projections of h_V': unit v = 8 pi + tn (block pi, channel tile tn) -> wave v % nwaves
  if constexpr (PF) {
    // (the sampler's tails have at most two blocks: a runtime index into a.p[] would copy the descriptors to scratch)
    tail_units<true>(wave, 8 * (a.nproj < 2 ? a.nproj : 2), nwaves, lane,
                     [&](const int v, const int tk) { return (const f4*)((v >> 3) ? a.p[1].img : a.p[0].img) + (tk * 8 + (v & 7)) * 64; },
                     [&](const int v, const f4 (&wf)[8]) {
                       const bool second = (v >> 3) != 0;
                       const int tn = v & 7;
                       float acc[R];
#pragma unroll
                       for (int n = 0; n < R; ++n) acc[n] = 0.f;
                       rows_fma<R, true>(acc, wf, yT, g);
#pragma unroll
                       for (int n = 0; n < R; ++n) acc[n] = xg_sum(acc[n]);
                       if (g == 0) {
                         const int c = 16 * tn + m;
                         const float* bias = second ? a.p[1].bias : a.p[0].bias;
                         const float* tok = second ? a.p[1].tok : a.p[0].tok;
                         float* out = second ? a.p[1].out : a.p[0].out;
                         const float b = bias ? bias[c] : 0.f;
#pragma unroll
                         for (int n = 0; n < R; ++n) {
                           const int orw = orow(n);
                           if (orw >= 0) {
                             float o = acc[n] + b;
                             if (tok) o += tok[(long)a.S[orw] * NAMP_H + c];
                             st_out<SC1>(out + (long)orw * NAMP_H + c, o);
                           }
                         }
                       }
                     });
    return;
  }
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) {
    if (pi >= a.nproj) break;
    const ProjDesc pd = a.p[pi];
#pragma unroll 1
    for (int tn_ = ((wave - pi * 8) % nwaves + nwaves) % nwaves; tn_ < 8; tn_ += nwaves) {
      const int tn = (tn_ + rot) & 7;
      f4 wf[8];
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wf[tk] = ((const f4*)pd.img)[(tk * 8 + tn) * 64 + lane];
      float acc[R];
#pragma unroll
      for (int n = 0; n < R; ++n) acc[n] = 0.f;
      rows_fma<R>(acc, wf, yT, g);
#pragma unroll
      for (int n = 0; n < R; ++n) acc[n] = xg_sum(acc[n]);
      if (g == 0) {
        const int c = 16 * tn + m;
        const float b = pd.bias ? pd.bias[c] : 0.f;
#pragma unroll
        for (int n = 0; n < R; ++n) {
          const int orw = orow(n);
          if (orw >= 0) {
            float o = acc[n] + b;
            if (pd.tok) o += pd.tok[(long)a.S[orw] * NAMP_H + c];
            st_out<SC1>(pd.out + (long)orw * NAMP_H + c, o);
          }
        }
      }
    }
  }
  NAMP_STAMP(12);
}

// ------------------------------------------------------------------------------------------
// node_tail_x3_rows<R> — the residue tail of <= R (<= 16) scattered rows as ONE 16-row MFMA tile of split-bf16 products, for
// the sampler's 8-wave workgroups (round 3).  The VALU form above is right where 250 workgroups stream the same weights (bound by
// L2 bandwidth chip-wide); in the sampler ONE workgroup serves a whole dependency level with the chip idle around it, and the
// VALU form's 640 dependent-LDS-read + packed-FMA groups per wave were 15 us of a 29 us layer (profiles/r03c).  Here (the schedule
// of node_update_multi_kernel): every wave holds the LayerNorm-1 rows in the register-chain layout (lane (m, g): channels
// 16t + 4g + r of row m), wave w owns hidden units 64w .. 64w+63 through BOTH FFN products — W_in slice (4 channel tiles x 4
// K-steps), exact-erf GELU in registers, W_out slice (8 channel tiles x its 2 K-steps) — the eight [R x 128] partial outputs are
// summed through LDS, every wave normalises the rows again (LayerNorm-2 in the chain layout) and takes its share of the projection
// tiles.  120 x 16x16x32 bf16 MFMAs and 80 KiB of x3 fragments (straight from L2, one tile ahead) per wave; padding rows (m >= R)
// ride along for free.  a.Win_img / a.Wout_img / a.p[].img are x3 images here (namp_pack_image_x3_general / namp_pack_image_x3).
// nwaves must be 8.  lds: >= (2048 + 8 * R * 128 + 128 * R + 64) floats; leaves yT[c * R + n] = h_V' where node_tail_rows does.
// ------------------------------------------------------------------------------------------
// SC1: the rows other workgroups of the SAME launch gather later (persistent level walk) are stored write-through
template <bool SC1>
__device__ __forceinline__ void st_f4(float* p, const f4 v) {
  if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else *(f4*)p = v;
}

template <int R, typename RowFn, bool SC1 = false>
__device__ __forceinline__ void node_tail_x3_rows(const NodeTail& a, f4 (&x)[8], const RowFn orow, float* lds,
                                                  const int tid, const int wave, const int lane) {
  float* part = lds + 128 * R;                       // [8 waves][R][128] partial W_out products (hT | oP of the VALU form)
  float* yT = lds + (128 + 512 + 4 * 128) * R;       // [128][R] h_V' for the caller's output head
  const int m = lane & 15, g = lane >> 4;
  const int mr = m < R ? m : 0;
  const float* win_p = a.Win_img; const float* wout_p = a.Wout_img; const float* p0_p = a.p[0].img; const float* p1_p = a.p[1].img;
  const float* bin_p = a.b_in; const float* bout_p = a.b_out; const float* l1g = a.ln1_g; const float* l1b = a.ln1_b;
  const float* l2g = a.ln2_g; const float* l2b = a.ln2_b;
  NAMP_STAMP(7);                       // (entry: images for the next layer copied, K-sums read)
  layernorm_row_T(x, l1g, l1b, g);
  NAMP_STAMP(8);                       // LayerNorm 1
  bf8 xh[4], xm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) split_x3(x[2 * s], x[2 * s + 1], xh[s], xm[s]);
  // ---- hidden = gelu(W_in x + b_in), this wave's four channel tiles 4w .. 4w+3
  const bf8* wi = (const bf8*)win_p + lane;
  constexpr int WI_MID = 512 * 128 / 8;              // bf8 units between the hi and the mid image
  f4 h[4];
  bf8 fa[8], fb[8];
#pragma unroll
  for (int s = 0; s < 4; ++s) { fa[2 * s] = wi[(s * 32 + 4 * wave) * 64]; fa[2 * s + 1] = wi[WI_MID + (s * 32 + 4 * wave) * 64]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tn = 4 * wave + q;
    bf8 (&cur)[8] = (q & 1) ? fb : fa;
    bf8 (&nxt)[8] = (q & 1) ? fa : fb;
    if (q + 1 < 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { nxt[2 * s] = wi[(s * 32 + tn + 1) * 64]; nxt[2 * s + 1] = wi[WI_MID + (s * 32 + tn + 1) * 64]; }
    }
    f4 acc = *(const f4*)(bin_p + 16 * tn + 4 * g);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma_x3(cur[2 * s], cur[2 * s + 1], xh[s], xm[s], acc);
    h[q] = gelu4_scalar(acc);
  }
  NAMP_STAMP(9);                       // W_in + GELU
  // ---- W_out restricted to this wave's 64 hidden units: K-steps 2w, 2w+1 of the [128 x 512] image; partial rows to LDS
  const bf8* wo = (const bf8*)wout_p + lane;
  constexpr int WO_MID = 128 * 512 / 8;
  bf8 hh[2], hm[2];
  split_x3(h[0], h[1], hh[0], hm[0]);
  split_x3(h[2], h[3], hh[1], hm[1]);
  {
    // four channel tiles per group: 16 fragments in flight (hi | mid of 2 K-steps x 4 tiles), the next group requested first
    bf8 ga[16], gb[16];
    auto fetch = [&](bf8 (&dst)[16], const int t0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dst[4 * q + 2 * s2] = wo[((2 * wave + s2) * 8 + t0 + q) * 64];
          dst[4 * q + 2 * s2 + 1] = wo[WO_MID + ((2 * wave + s2) * 8 + t0 + q) * 64];
        }
    };
    fetch(ga, 0);
    fetch(gb, 4);
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      bf8 (&cur)[16] = grp ? gb : ga;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 po = (f4){0.f, 0.f, 0.f, 0.f};
        po = mfma_x3(cur[4 * q], cur[4 * q + 1], hh[0], hm[0], po);
        po = mfma_x3(cur[4 * q + 2], cur[4 * q + 3], hh[1], hm[1], po);
        // (channel tile t of row m sits at tile position t ^ m: the R rows of one tile then fall into different LDS banks — with plain rows, 512 B
        // apart, every read of the sums below was a 4-way conflict, and with 64 of them per lane the phase took 4.5 us per layer: profiles/r05b)
        if (m < R) *(f4*)(part + ((wave * R + m) * 128 + 16 * ((4 * grp + q) ^ m) + 4 * g)) = po;
      }
    }
  }
  NAMP_STAMP(22);                      // (W_out partials written; barrier wait follows)
  __syncthreads();
  NAMP_STAMP(10);                      // W_out partials
  // ---- the eight partial products, reduce-scatter: wave w adds up channel tile w of the R rows (8 reads per lane instead of 64) and leaves the
  // sums in lds[0 .. 128 R) (same tile swizzle); then every wave picks up its rows' full sums
  static_assert(R <= 8, "the tile swizzle t ^ m needs m < 8");
  {
    f4 sv = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) sv += *(const f4*)(part + ((w * R + mr) * 128 + 16 * (wave ^ mr) + 4 * g));
    if (m < R) *(f4*)(lds + m * 128 + 16 * (wave ^ m) + 4 * g) = sv;
  }
  __syncthreads();
  // ---- y = LayerNorm2(x + W_out h + b_out) * mask, in the chain layout, by every wave
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = (x[t] + *(const f4*)(bout_p + 16 * t + 4 * g)) + *(const f4*)(lds + mr * 128 + 16 * (t ^ mr) + 4 * g);
  NAMP_STAMP(13);                      // (partial sums read)
  layernorm_row_T(x, l2g, l2b, g);
  NAMP_STAMP(14);                      // (LayerNorm 2)
  const int orw = (m < R) ? orow(m) : -1;
  {
    const float mk = (a.mask && orw >= 0) ? (float)a.mask[orw] : 1.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] *= mk;
  }
  if (wave == 0 && m < R) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      yT[(16 * t + 4 * g + 0) * R + m] = x[t].x; yT[(16 * t + 4 * g + 1) * R + m] = x[t].y;
      yT[(16 * t + 4 * g + 2) * R + m] = x[t].z; yT[(16 * t + 4 * g + 3) * R + m] = x[t].w;
      if (orw >= 0) st_f4<SC1>(a.hV_out + (long)orw * NAMP_H + 16 * t + 4 * g, x[t]);
    }
  }
  NAMP_STAMP(11);                      // partial sums + LayerNorm 2 + h_V' out
  // ---- projections of h_V' (<= 2 blocks: the next layer's Pa / Pv): unit v = 8 pi + tn -> wave v % 8
  if (a.nproj > 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) split_x3(x[2 * s], x[2 * s + 1], xh[s], xm[s]);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      if (pi >= a.nproj) break;
      const bf8* wp = (const bf8*)(pi ? p1_p : p0_p) + lane;
      const int tn = wave;
      bf8 f[8];
#pragma unroll
      for (int s = 0; s < 4; ++s) { f[2 * s] = wp[(s * 8 + tn) * 64]; f[2 * s + 1] = wp[NAMP_BIMG_BYTES / 16 + (s * 8 + tn) * 64]; }
      f4 acc = a.p[pi].bias ? *(const f4*)(a.p[pi].bias + 16 * tn + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma_x3(f[2 * s], f[2 * s + 1], xh[s], xm[s], acc);
      if (orw >= 0) {
        if (a.p[pi].tok) acc += *(const f4*)(a.p[pi].tok + (long)a.S[orw] * NAMP_H + 16 * tn + 4 * g);
        st_f4<SC1>(a.p[pi].out + (long)orw * NAMP_H + 16 * tn + 4 * g, acc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// edge_mlp_kernel — the hot kernel.  One wave = one 16-row tile = 16 of the K neighbours of
// one residue; a workgroup = (blockDim/64)/TPN residues, TPN = ceil(K/16) tiles per residue.
// Weights: the three 128x128 images go through a 2 x 64 KiB LDS ring shared by all waves, filled by
// LDS-DMA (global_load_lds): W1 and W2 up front, W3 into W1's slot while layer 2 computes.  (Streaming
// an image per wave from L2 instead was measured 10 us slower per launch: 12 waves x 64 KiB per CU and
// layer exceeds what the CU's L1 path sustains — profiles/r01_ablation.md.)  Activations stay in
// registers (namp_device.h).
//
// With W1 = [W1a | W1b | W1c] applied to [h_V_i | h_E_ik | h_V_j]  (EncLayer,
// model_utils.py:684-687,699-702) the first layer is evaluated as
//      W1b . h_E_ik  +  (W1a . h_V_i + b1)  +  W1c . h_V_j
//      \_ MFMA here _/   \__ Pa[i] table __/    \_ Pj[j] table _/
// i.e. the two node-only products are hoisted into per-residue tables computed once by
// node_linear_kernel and *gathered* here; no [N,K,384] concatenation is ever materialised.
// The decoder (DecLayer on h_ESV, model_utils.py:416-418,636-646) is the same with
//      first layer = W1e . h_E_ik + Pa[i] + (bw ? Pbw[j] : Pfw[j]),
//      Pbw[j] = W1s . W_s[S_j] + W1v . h_V_j ,  Pfw[j] = W1v . h_V^enc_j ,
//      bw = [rank(j) < rank(i)]   (the order mask of model_utils.py:391-396 as a rank compare).
//
//   MODE_ENC_MSG : partial[i][kt][:] = sum_{k in tile} mask_i mask_j / 30 * MLP(...)
//   MODE_DEC_MSG : partial[i][kt][:] = sum_{k in tile}           1 / 30 * MLP(...)
//   MODE_ENC_EDGE: h_E'[i,k,:] = LayerNorm3(h_E[i,k,:] + MLP(...))           (in place allowed)
//   MODE_EMBED   : h_E[i,k,:]  = W_e . E[i,k,:] + b_e   (single layer; model_utils.py:89)
// ------------------------------------------------------------------------------------------
enum { MODE_ENC_MSG = 0, MODE_DEC_MSG = 1, MODE_ENC_EDGE = 2, MODE_EMBED = 3 };

// bf16 row storage ("fragment order B", the operand order of v_mfma_f32_32x32x16_bf16, see namp_bf16s32.h): channel c of a 128-wide row
// sits at element 16(c>>4) + 8((c>>2)&1) + 4((c>>3)&1) + (c&3).  A T-layout accumulator lane (row m, g) holds channels 16t + 4g .. +3
// of tile t: those four go to elements frag_off(t, g) .. +3 (8 bytes).
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int frag_off(const int t, const int g) { return 16 * t + 8 * (g & 1) + 4 * (g >> 1); }
__device__ __forceinline__ void st_frag4(__bf16* row, const int t, const int g, const f4 v) {
  *(bf4*)(row + frag_off(t, g)) = (bf4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

struct EdgeArgs {
  const float* hE;             // [G_enc*K][128] input edge rows
  float* hE_out;               // ENC_EDGE / EMBED output rows
  const int32_t* E_idx;        // [G_enc][K] neighbour ids, local to the complex
  const int32_t* mask;         // [G] residue mask (may be null = all ones)
  const int32_t* mask_attend;  // optional explicit [G_enc][K] (ENC_MSG); null -> mask_i*mask_j
  const int32_t* rank;         // DEC_MSG: [G] decoding rank of every residue
  const float* Pa;             // [G][128] per-residue first-layer term (bias folded in)
  const float* Pj0;            // ENC: Pc [G][128];   DEC: Pbw [G][128]
  const float* Pj1;            // DEC: Pfw [G_enc][128]
  const float* W1_img;         // 64 KiB images (fp32) or 32 KiB bf16 images (BF16 = true)
  const float* W2_img;
  const float* W3_img;
  const float* b1;             // EMBED only (otherwise folded into Pa)
  const float* b2;
  const float* b3;
  const float* ln_g;           // ENC_EDGE
  const float* ln_b;
  // FUSE (message modes with the fused tail): the PREVIOUS layer's edge update runs first on the same rows —
  // h_E <- LN3(h_E + MLP'(..)) with these tables / images, stored to hE_out — and the message phase consumes it
  // from registers (one launch, one read of h_E, for EncLayer l's edge update + layer l+1's message)
  const float* ePa; const float* ePj;
  const float* eW1_img; const float* eW2_img; const float* eW3_img;
  const float* eb2; const float* eb3;
  uint32_t drop_thresh, drop_seed; float drop_scale;   // ENC_EDGE, training only: dropout on the message (thresh 0 = off)
  // bf16 STORAGE (throughput mode on large batches, edge_mlp_bf16s32_kernel): rows of 128 bf16 in fragment order B
  // [s][g][j] <-> channel 32s + 16(j>>2) + 4g + (j&3): 16-byte piece (s, g) is what lane (m, g) feeds into MFMA step s, and the four
  // lanes of a row cover one contiguous 64-byte segment per load / store instruction
  const __bf16* hE16; __bf16* hE16_out;
  const __bf16* Pa16; const __bf16* Pj016; const __bf16* Pj116;
  float* partial;              // MSG modes without the fused tail: [G][TPN][128]
  NodeTail tail;               // MSG modes with the fused tail (TAIL = true)
  int G;                       // residues processed by this launch (decoder: B_dec*N)
  int G_enc;                   // residues on the encoder side (B_enc*N); decoder batch b maps to b % B_enc
  int N, K, TPN;
};

// TAIL (message modes only): the workgroup goes on to update its own residues (node_tail) instead of
// writing partial sums — one launch per layer half instead of two, worth it while the whole batch is
// a single wave of workgroups (every workgroup re-streams the 768 KiB of FFN / projection weights).
#define EDGE_TAIL_LDS (2 * NAMP_IMG_BYTES + 12 * NAMP_H * 4 + 64 + 2048)   // ring + per-wave K-sums + their weight sums + the fused
                                                                          // edge update's constant vectors (eb2 | eb3 | LN3 weight | bias)

// TAIL: 0 = write partial sums; 4 / 8 = node_tail_rows<4/8> (the workgroup owns <= 4 / <= 6 residues);
// 16 = node_tail (16-row MFMA tile).  Chosen by the host from 12/TPN so that only one variant is inlined.
// BF16: message / edge GEMMs on v_mfma_f32_16x16x32_bf16 with all three 32 KiB images resident in LDS
// (throughput mode, namp_device.h); everything else — tables, K-sum, LayerNorms, residue tail — stays fp32.
// PRE: a stage run on the same rows in front of the message phase (fp32 message modes with the fused tail):
//   PRE_EDGE  — the previous layer's edge update (h_E <- LN3(h_E + MLP'), see EdgeArgs);
//   PRE_EMBED — h_E = W_e . E + b_e (eW1_img / eb2), model_utils.py:89, for the first EncLayer.
enum { PRE_NONE = 0, PRE_EMBED = 1, PRE_EDGE = 3 };
// PREC: how the per-edge 128 x 128 GEMMs are evaluated — PREC_F32 exact fp32 MFMA; PREC_X3 split-bf16 (fp32-equivalent
// to ~2^-16, the default parity mode: namp_device.h chain_gemm_x3); PREC_BF16 plain bf16 (throughput mode).
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_X3 = 2 };
// The body of edge_mlp_kernel as a device function, so that the persistent forward (encdec_persistent_kernel) can run the
// stages of a whole encoder + decoder pass back to back on the SAME rows with h_E held in registers:
//   PERSIST 0: the stand-alone launch (x is a scratch array);
//   PERSIST 1: a stage of the persistent kernel that loads its rows from a.hE on entry;
//   PERSIST 2: a stage that finds its rows in x (left there by the previous stage).
// With PERSIST != 0 x is preserved (holds the — possibly updated — h_E row on exit), rows are stored only if a.hE_out is
// given, and the kernel's LDS is re-used stage after stage (the caller separates stages by a grid barrier).
template <int MODE, int TAIL, int PREC, int PRE, int PERSIST, class Args>
__device__ __forceinline__ void edge_stage(const Args& a, f4 (&x)[8], char* smem) {
#ifndef NAMP_ABL_NOL2PF
  f4 l2pf0 = (f4){0.f, 0.f, 0.f, 0.f}, l2pf1 = l2pf0;
#endif
  constexpr bool BF16 = (PREC == PREC_BF16);
  constexpr bool X3 = (PREC == PREC_X3);
  static_assert(PRE == PRE_NONE || (!BF16 && TAIL != 0 && (MODE == MODE_ENC_MSG || MODE == MODE_DEC_MSG)), "PRE: fp32-class message + tail only");
  constexpr bool FUSE = (PRE == PRE_EDGE);
  constexpr bool MSG = (MODE == MODE_ENC_MSG || MODE == MODE_DEC_MSG);    // layer 3 is hoisted behind the K-sum (NodeTail.m3_img)
  // fused tail of <= 8 residues, fp32-class GEMMs: W3 still streams through the ring (under layer 2) and the per-residue
  // product runs as ONE MFMA tile out of LDS, channel tiles dealt over the waves — fetching the fragments from L2 after the K-sum
  // instead put a full L2 round trip on the launch's critical path
  constexpr bool M3_LDS = MSG && !BF16 && (TAIL == 4 || TAIL == 8);
  char* buf0 = smem;
  char* buf1 = smem + NAMP_IMG_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;

  const int m = lane & 15, g = lane >> 4;
  const int npw = nwaves / a.TPN;                      // residues per workgroup
  const int node_l = wave / a.TPN, kt = wave - node_l * a.TPN;
  // XCD-aware residue ranges (round 5): workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it), each XCD has its own
  // L2, and a residue's neighbours are mostly near it in the chain — so XCD x takes the x-th contiguous eighth of the workgroups' residue
  // blocks and its L2 holds (about) an eighth of every gathered table instead of all of it.  With blocks dealt round-robin every L2 fetched
  // every table: 73 MB of HBM-side traffic per fused launch against 52 MB algorithmic, unchanged for four rounds.
  const int bid = xcd_block_index(blockIdx.x, gridDim.x);
  int node = bid * npw + node_l;
  const bool wave_active = (node_l < npw) && (node < a.G);
  if (!wave_active) node = 0;
  // encoder-side residue (decoder batches are replicas of encoder batches: h_V.repeat(B_decoder,..))
  const int b_dec = node / a.N;
  const int i_loc = node - b_dec * a.N;
  const int node_enc = (MODE == MODE_DEC_MSG) ? ((b_dec % (a.G_enc / a.N)) * a.N + i_loc) : node;
  const int k = 16 * kt + m;
  const bool valid = wave_active && (k < a.K);
  const long erow = (long)node_enc * a.K + (valid ? k : 0);

  // ---- per-row operands: the h_E row (B operand of layer 1) and the hoisted first-layer terms
  f4 acc[8];
  f4 ybuf[8];
  f4 (&y)[8] = PERSIST ? ybuf : x;     // layer-2 pre-activations / residue-tail rows: x itself unless x must survive the stage
  f4 pjv[8];                       // gathered neighbour term, added after layer 1 (its latency hides under the MFMAs)
  float w_row = 0.f;
#ifdef NAMP_ABL_NOPROLOG
#pragma unroll
  for (int t = 0; t < 8; ++t) { x[t] = (f4){0.1f * lane, 0.2f, 0.3f, 0.4f * t}; acc[t] = x[t]; pjv[t] = x[t]; }
  w_row = valid ? 1.f : 0.f;
  if (true) {
  } else if (MODE == MODE_EMBED) {
#else
  if (PERSIST != 2) {
    const float* src = a.hE + erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
  }
  if (MODE == MODE_EMBED) {
#endif
    // training (namp_edge_embed_ln): the rows are PRE-LayerNorm (norm_edges, na_model_utils.py:509) — normalised here, so that the
    // normalised rows never exist in memory (the backward launch and the weight-gradient contraction re-derive them from the same rows)
    if (a.ln_g) layernorm_row_T(x, a.ln_g, a.ln_b, g);
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(a.b1 + 16 * t + 4 * g); pjv[t] = (f4){0.f, 0.f, 0.f, 0.f}; }
  } else if (PRE == PRE_EMBED) {
    // tables are loaded after the embedding GEMM
  } else if (FUSE) {
    // edge-update tables first (EncLayer addressing: j in the same complex); the message tables follow after LayerNorm3
    const int j_loc = a.E_idx[erow];
    const int j = node_enc - i_loc + j_loc;
    const float* pa = a.ePa + (long)node_enc * NAMP_H + 4 * g;
    const float* pj = a.ePj + (long)j * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
  } else {
    const int j_loc = a.E_idx[erow];
    const float* pj;
    if (MODE == MODE_DEC_MSG) {
      const int j_dec = b_dec * a.N + j_loc;
      const bool bw = a.rank[j_dec] < a.rank[node];
      pj = bw ? (a.Pj0 + (long)j_dec * NAMP_H) : (a.Pj1 + (long)(node_enc - i_loc + j_loc) * NAMP_H);
      w_row = valid ? (1.0f / 30.0f) : 0.f;
    } else {
      const int j = node - i_loc + j_loc;
      pj = a.Pj0 + (long)j * NAMP_H;
      if (MODE == MODE_ENC_MSG) {
        int ma;
        if (a.mask_attend) ma = a.mask_attend[erow];
        else ma = a.mask ? (a.mask[node] * a.mask[j]) : 1;
        w_row = valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
      }
    }
    const float* pa = a.Pa + (long)node * NAMP_H + 4 * g;
    pj += 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
  }

  const f4* w0 = (const f4*)buf0 + lane;
  const f4* w1 = (const f4*)buf1 + lane;

  if (BF16) {
    const bf8* bw = (const bf8*)smem + lane;               // image l at smem + l * 32 KiB
    dma_to_lds(smem, a.W1_img, 32, wave, nwaves, lane);
    if (MODE != MODE_EMBED) dma_to_lds(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, nwaves, lane);
    if (MODE == MODE_ENC_EDGE) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.W3_img, 32, wave, nwaves, lane);
    wait_dma_and_sync();                                   // the only barrier of the MLP
    chain_gemm_bf16<false, false>(acc, x, bw);
    if (MODE != MODE_EMBED) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] += pjv[t];
      if (MSG) {                                           // layer 2 in the F orientation: rows 4g+r of channel 16t + m
#pragma unroll
        for (int t = 0; t < 8; ++t) { const float b = a.b2[16 * t + m]; y[t] = (f4){b, b, b, b}; }
        chain_gemm_bf16<true, true>(y, acc, bw + (NAMP_BIMG_BYTES / 16));
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(a.b2 + 16 * t + 4 * g);
        chain_gemm_bf16<false, true>(y, acc, bw + (NAMP_BIMG_BYTES / 16));
      }
    }
  } else {
  float* cstf = (float*)(smem + 2 * NAMP_IMG_BYTES + 12 * NAMP_H * 4 + 64);
  if (FUSE) {
    // ---- fused edge update of the previous layer: images eW1 -> buf1, eW2 -> buf0, eW3 -> buf1, so that the
    // message images land in their usual slots (W1 buf0, W2 buf1, W3 buf0) one GEMM ahead of their use
    dma_to_lds(buf1, a.eW1_img, 64, wave, nwaves, lane);
    dma_to_lds(buf0, a.eW2_img, 64, wave, nwaves, lane);
    // eb2 | eb3 | LayerNorm-3 weight | bias -> LDS: as 32 16-byte loads per wave they were 32 KiB x 12 waves through the CU's
    // 64 B/clk vector-memory path per launch (profiles/r02k_bf16s_ablation.md, last table)
    for (int i = tid; i < 512; i += (int)blockDim.x) {      // (a workgroup has 7 .. 12 waves)
      const float* srcv = i < 128 ? a.eb2 : i < 256 ? a.eb3 : i < 384 ? a.ln_g : a.ln_b;
      cstf[i] = srcv[i & 127];
    }
    wait_dma_and_sync();
    gemm128<X3, false, false>(acc, x, w1);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] += pjv[t];
    __syncthreads();                                      // every wave is done with buf1 (eW1)
    dma_to_lds(buf1, a.eW3_img, 64, wave, nwaves, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) pjv[t] = *(const f4*)(cstf + 16 * t + 4 * g);
    gemm128<X3, false, true>(pjv, acc, w0);       // pjv = edge-MLP layer-2 pre-activations
    wait_dma_and_sync();                                  // eW3 landed; buf0 (eW2) is free
    dma_to_lds(buf0, a.W1_img, 64, wave, nwaves, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(cstf + 128 + 16 * t + 4 * g);
    gemm128<X3, false, true>(acc, pjv, w1);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] += acc[t];           // residual
    layernorm_row_T(x, cstf + 256, cstf + 384, g);        // x = updated h_E row: stored, and the message input
#ifdef NAMP_ABL_NOSTORE
    if (valid && a.G < 0) {
#else
    if (valid && (PERSIST == 0 || a.hE_out != nullptr)) {
#endif
      float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = x[t];
    }
  }
  if (PRE != PRE_NONE) {
    if (PRE == PRE_EMBED) {
      // ---- fused edge embedding: W_e -> buf1 and W1 -> buf0 up front; x <- W_e . x + b_e, stored as h_E
      dma_to_lds(buf1, a.eW1_img, 64, wave, nwaves, lane);
      dma_to_lds(buf0, a.W1_img, 64, wave, nwaves, lane);
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(a.eb2 + 16 * t + 4 * g);
      wait_dma_and_sync();
      gemm128<X3, false, false>(acc, x, w1);
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = acc[t];
      if (valid && (PERSIST == 0 || a.hE_out != nullptr)) {
        float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = x[t];
      }
    }
    // message tables of this layer
#ifdef NAMP_ABL_NOTABLE2
    {
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc[t] = x[t]; pjv[t] = x[t]; }
      w_row = valid ? (1.0f / 30.0f) : 0.f;
    }
    if (false)
#endif
    {
      const int j_loc = a.E_idx[erow];
      const float* pj;
      if (MODE == MODE_DEC_MSG) {
        const int j_dec = b_dec * a.N + j_loc;
        const bool bw = a.rank[j_dec] < a.rank[node];
        pj = bw ? (a.Pj0 + (long)j_dec * NAMP_H) : (a.Pj1 + (long)(node_enc - i_loc + j_loc) * NAMP_H);
        w_row = valid ? (1.0f / 30.0f) : 0.f;
      } else {
        const int j = node - i_loc + j_loc;
        pj = a.Pj0 + (long)j * NAMP_H;
        int ma;
        if (a.mask_attend) ma = a.mask_attend[erow];
        else ma = a.mask ? (a.mask[node] * a.mask[j]) : 1;
        w_row = valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
      }
      const float* pa = a.Pa + (long)node * NAMP_H + 4 * g;
      pj += 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
    }
    if (PRE == PRE_EMBED) __syncthreads();                // W1 landed with W_e; every wave is done with buf1 (W_e)
    else wait_dma_and_sync();                             // W1 landed in buf0; buf1 (eW3) is free
    dma_to_lds(buf1, a.W2_img, 64, wave, nwaves, lane);
  } else {
  // ---- weight staging: W1 -> buf0 and W2 -> buf1 by LDS-DMA.  Issued AFTER the per-row operand loads
  // above (the VM counter retires in order: loads queued behind a bulk DMA could not be consumed before it).
  dma_to_lds(buf0, a.W1_img, 64, wave, nwaves, lane);
  if (MODE != MODE_EMBED) dma_to_lds(buf1, a.W2_img, 64, wave, nwaves, lane);
  wait_dma_and_sync();
  }
  // ---- layer 1 (T): acc = Pa + W1b . h_E (+ Pj afterwards)
  gemm128<X3, false, false>(acc, x, w0);

  if (MODE != MODE_EMBED) {
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] += pjv[t];           // acc = layer-1 pre-activations
  if (MSG) {
    // ---- layer 2 (F): lane (m, g) gets rows 4g..4g+3 of channel 16*tn + m — what the K-sum needs; there is no per-edge layer 3
    if (PRE != PRE_NONE) wait_dma_and_sync();             // W2 (issued one GEMM ago) landed; buf0 (W1) is free
    else if (M3_LDS) __syncthreads();                     // every wave is done with buf0 (W1)
    if (M3_LDS) dma_to_lds(buf0, a.W3_img, 64, wave, nwaves, lane);   // for the per-residue layer 3 behind the K-sum
#ifndef NAMP_ABL_NOL2PF
    if (TAIL == 4 || TAIL == 8) {
      // The residue tail's weights (W_in, W_out, the projections: 0.6-0.8 MB that every workgroup reads) are pulled into this XCD's L2 one
      // GEMM ahead, each workgroup asking for the 1/32 slice of its index within the XCD (blockIdx / 8: a speed choice only); the rows of
      // h_E streamed since the previous launch's tail have evicted them.  Two registers per lane until the K-sum (cfg2 0.496 -> 0.490 ms).
      const int slice = (blockIdx.x >> 3) & 31;
      const f4* p0; const f4* p1;
      if (wave < 8) {
        p0 = (const f4*)a.tail.Win_img + (slice * 8 + wave) * 64; p1 = (const f4*)a.tail.Wout_img + (slice * 8 + wave) * 64;
      } else {
        const int q = wave - 8;
        const float* img = q == 0 ? a.tail.p[0].img : q == 1 ? a.tail.p[1].img : q == 2 ? a.tail.p[2].img : a.tail.p[3].img;
        if (q >= a.tail.nproj) img = a.tail.Win_img;
        p0 = (const f4*)img + (slice * 2) * 64; p1 = p0 + 64;
      }
      l2pf0 = p0[lane]; l2pf1 = p1[lane];
    }
#endif
#pragma unroll
    for (int t = 0; t < 8; ++t) { const float b = a.b2[16 * t + m]; y[t] = (f4){b, b, b, b}; }
    gemm128<X3, true, true>(y, acc, w1);
  } else {
  if (PRE != PRE_NONE) wait_dma_and_sync();               // W2 (issued one GEMM ago) landed; buf0 (W1) is free
  else __syncthreads();                                   // every wave is done with buf0 (W1)
  dma_to_lds(buf0, a.W3_img, 64, wave, nwaves, lane);     // lands while layer 2 runs out of buf1

  // ---- layer 2 (T); GELU of layer 1 is applied k-tile by k-tile inside the MFMA loop
#pragma unroll
  for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(a.b2 + 16 * t + 4 * g);
  gemm128<X3, false, true>(y, acc, w1);           // y = layer-2 pre-activations
  wait_dma_and_sync();                                    // W3 has landed in buf0
  }
  }
  }
  if (MODE == MODE_EMBED) {
    if (valid) {
      if (a.hE16_out) {                 // bf16 storage in fragment order B
        __bf16* dst = a.hE16_out + erow * NAMP_H;
#pragma unroll
        for (int t = 0; t < 8; ++t) st_frag4(dst, t, g, acc[t]);
      } else {
        float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
      }
    }
    return;
  }

  if (MODE == MODE_ENC_EDGE) {
    // ---- layer 3 (T) + residual + LayerNorm3, written back row-wise
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(a.b3 + 16 * t + 4 * g);
    if (BF16) chain_gemm_bf16<false, true>(acc, y, (const bf8*)smem + lane + 2 * (NAMP_BIMG_BYTES / 16));
    else      gemm128<X3, false, true>(acc, y, w0);
    if (a.drop_thresh) {                                           // training forward: dropout3 on the message
      const uint32_t key = drop_row_key(a.drop_seed, erow);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] *= drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, a.drop_scale);
    }
    if (a.ln_g) {                                                  // null: write the bare message
      const float* src = a.hE + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] += *(const f4*)(src + 16 * t);
      layernorm_row_T(acc, a.ln_g, a.ln_b, g);
    }
    if (valid) {
      float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
    }
  } else {
    // ---- K-sum of the layer-2 activations (layer 3 follows per residue: NodeTail.m3_img).  y: rows 4g+r of channel 16t + m;
    // weights of rows 4g+r live in lanes with (lane&15) == 4g+r
    float wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = __shfl(w_row, 4 * g + r);
    float wsum = w_row;                                         // sum of the tile's 16 row weights (lanes m = 0..15 of any g)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) wsum += __shfl_xor(wsum, o);
    if (!TAIL) {
      float* dst = a.partial + ((long)node * a.TPN + kt) * NAMP_H + m;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f4 v = gelu_prec<PREC>(y[t]);
        float s = (v.x * wr[0] + v.y * wr[1]) + (v.z * wr[2] + v.w * wr[3]);
        s = xg_sum(s);
        if (wave_active && g == 0) dst[16 * t] = s;
      }
      if (wave_active && lane == 0) a.partial[(long)a.G * a.TPN * NAMP_H + (long)node * a.TPN + kt] = wsum;
    } else {
      // per-tile sums -> LDS (beyond the weight ring, which slower waves may still be reading)
      float* dpart = (float*)(smem + 2 * NAMP_IMG_BYTES);       // [nwaves][128], then [nwaves] weight sums
      float* dws = dpart + 12 * NAMP_H;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f4 v = gelu_prec<PREC>(y[t]);
        float s = (v.x * wr[0] + v.y * wr[1]) + (v.z * wr[2] + v.w * wr[3]);
        s = xg_sum(s);
        if (g == 0) dpart[wave * NAMP_H + 16 * t + m] = s;
      }
      if (lane == 0) dws[wave] = wsum;
#ifndef NAMP_ABL_NOL2PF
      asm volatile("" :: "v"(l2pf0), "v"(l2pf1));               // (the prefetch above has landed)
#endif
      if (M3_LDS) wait_dma_and_sync();                          // all tiles summed; W3 landed in buf0
      else __syncthreads();                                     // all tiles summed; weight ring is free
      // tile rows = this workgroup's residues: row m -> residue row0 + m
      const int row0 = bid * npw;
      const int trow = row0 + m;
      const bool tvalid = (m < npw) && (trow < a.G);
      const int mm = tvalid ? m : 0;
      float wsum_m = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) y[t] = (f4){0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < a.TPN; ++q) {
        const float* dp = dpart + (mm * a.TPN + q) * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) y[t] += *(const f4*)(dp + 16 * t);
        wsum_m += dws[mm * a.TPN + q];
      }
#ifdef NAMP_ABL_NOTAIL
      if (tvalid) {
        float* dst = a.tail.hV_out + (long)trow * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = y[t];
      }
      return;
#endif
#ifdef NAMP_ABL_PAIRTAIL
      // TIMING PROBE ONLY (wrong results): every other workgroup runs an 8-row tail (its rows + its neighbour's), the others none
      if (blockIdx.x & 1) return;
#endif
      if (TAIL == 4 || TAIL == 8) {
#ifdef NAMP_ABL_PAIRTAIL
        constexpr int R = 8;
        const ConsecutiveRows orow = {row0, 2 * npw, a.G};
#else
        constexpr int R = (TAIL == 4 || TAIL == 8) ? TAIL : 4;
        const ConsecutiveRows orow = {row0, npw, a.G};
#endif
        if (M3_LDS) {
          // x = h_V + W3 . (K-sums) + b3 * wsum: wave w < 8 evaluates channel tile w of the (<= 8-row) tile, exchanged through
          // the K-sum area (8 rows x 128 floats <= 12 x 128)
          __syncthreads();                                      // every wave has read dpart / dws
          for (int tn = wave; tn < 8; tn += nwaves) {          // (a workgroup may have fewer than 8 waves: K > 96)
            const f4 o = tile_gemm1<X3>(*(const f4*)(a.tail.m3_b + 16 * tn + 4 * g) * wsum_m, y, w0, tn);
            if (m < 8) *(f4*)(dpart + m * NAMP_H + 16 * tn + 4 * g) = o;
          }
          __syncthreads();
          const float* hsrc = a.tail.hV + (long)(row0 + mm) * NAMP_H + 4 * g;
          const float* dsrc = dpart + (mm & 7) * NAMP_H + 4 * g;
#pragma unroll
          for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(hsrc + 16 * t) + *(const f4*)(dsrc + 16 * t);
          node_tail_rows<R, ConsecutiveRows, (PERSIST != 0), false, false, !BF16>(a.tail, y, 0.f, orow, (float*)smem, tid, wave, nwaves, lane);
        } else {
          node_tail_rows<R, ConsecutiveRows, (PERSIST != 0), true, false, !BF16>(a.tail, y, wsum_m, orow, (float*)smem, tid, wave, nwaves, lane);
        }
      } else {
        node_tail<false>(a.tail, y, wsum_m, row0, npw, a.G, (float*)smem, tid, wave, nwaves, lane);
      }
    }
  }
}

template <int MODE, int TAIL, int PREC = PREC_F32, int PRE = PRE_NONE>
__global__ __launch_bounds__(768) void edge_mlp_kernel(const EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4 x[8];
  edge_stage<MODE, TAIL, PREC, PRE, 0>(a, x, smem);
}

// ------------------------------------------------------------------------------------------
// encdec_persistent_kernel — the WHOLE encoder + decoder forward of a small batch (one wave of workgroups: at most one
// workgroup per CU, every workgroup resident) as ONE launch.  A workgroup keeps the rows it owns — the K neighbours of its
// <= 12/TPN residues — in registers from the edge embedding to the last DecLayer: h_E is read once (as E) and written once
// (the encoder's output), instead of 5 reads + 4 writes over the seven launches of the fused chain.  Stages
//   S0  h_E = W_e.E + b_e ; EncLayer 0 message + residue tail          (tables for S1)
//   S1  EncLayer 0 edge update ; EncLayer 1 message + tail
//   ..  (n_enc stages)
//   Sd0 last edge update (h_E stored) ; DecLayer 0 message + tail
//   Sd1.. DecLayer l message + tail ; the last one evaluates the output head
// are the bodies of the fused launches (edge_stage).  What a residue tail writes — h_V' and the first-layer tables the
// NEXT stage gathers from OTHER workgroups' residues — crosses workgroups, so stages are separated by a grid barrier:
// the tail stores them write-through (sc1), every wave drains its stores, one lane arrives on a counter (two-level: per
// group of workgroups, then across groups), polls relaxed, and one agent-scope acquire drops the CU's stale L1 lines
// (cdna_hip_programming.md, Guideline 16).  Residency comes from the grid size alone (<= number of CUs, one 133 KB-LDS
// workgroup per CU); every spin is bounded and reports through sync[SYNC_TIMEOUT] instead of hanging.
// The counters are zeroed by the launch that precedes this one in the stream (node_linear_kernel, NodeLinearArgs.zero).
// ------------------------------------------------------------------------------------------
#define NAMP_PERSIST_MAX_STAGES 16
#define NAMP_SYNC_GROUPS 8
#define NAMP_SYNC_WORDS 64          // [0..7] group arrival counters, [8] top counter, [16..23] group generations, [32] timeout code
#define NAMP_SYNC_TOP 8
#define NAMP_SYNC_GEN 16
#define NAMP_SYNC_TIMEOUT 32
#define NAMP_SYNC_DEFER 40          // [40..45] three 64-bit device pointers of the sampler's deferred group draw (dec_sample_kernel, MODE 2)
#ifndef NAMP_SPIN_LIMIT
#define NAMP_SPIN_LIMIT (1u << 22)  // polls (each followed by s_sleep): several seconds — a hang becomes a reported failure
#endif

typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ bool spin_until_ge(gu32* word, unsigned target, gu32* tmo, unsigned code) {
  unsigned spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > NAMP_SPIN_LIMIT) {
      __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

// epoch = 1, 2, ... within one launch.  Group g = blockIdx % 8 (the observed XCD of the block: a speed choice only).
__device__ __forceinline__ void grid_barrier(unsigned* sync, const unsigned epoch, const int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every wave: its own (write-through) stores have completed
  __syncthreads();
#ifdef NAMP_ABL_NOGRIDBARRIER
  return;
#endif
  if (tid == 0) {
    gu32* sy = (gu32*)sync;
    const int grp = blockIdx.x % NAMP_SYNC_GROUPS;
    const unsigned ngroups = gridDim.x < NAMP_SYNC_GROUPS ? gridDim.x : NAMP_SYNC_GROUPS;
    const unsigned in_group = (gridDim.x - grp + NAMP_SYNC_GROUPS - 1) / NAMP_SYNC_GROUPS;
#ifdef NAMP_PERSIST_RELEASE_FENCE                                 // plain table stores + release fence instead of sc1 stores (A/B)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the write-back has completed before the arrival (G16 pitfall 12)
#endif
    const unsigned old = __hip_atomic_fetch_add(sy + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == in_group * epoch) {                           // last arriver of the group: arrive on the top counter,
      __hip_atomic_fetch_add(sy + NAMP_SYNC_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until_ge(sy + NAMP_SYNC_TOP, ngroups * epoch, sy + NAMP_SYNC_TIMEOUT, 0x10000000u | epoch);
      __hip_atomic_store(sy + NAMP_SYNC_GEN + grp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... release the group
    } else {
      spin_until_ge(sy + NAMP_SYNC_GEN + grp, epoch, sy + NAMP_SYNC_TIMEOUT, 0x20000000u | epoch);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// One stage's arguments: the members of EdgeArgs the fused fp32-class message stages read, under the same names (edge_stage is
// templated on the argument type), without the members of other modes — six of these must fit the 4 KiB kernel-argument
// segment.  The stage functions read them straight from that segment (scalar loads at the point of use); copying them
// into registers up front cost 540 spilled SGPRs.
struct StageArgs {
  const float* hE; float* hE_out; const int32_t* E_idx; const int32_t* mask; const int32_t* rank;
  const float* Pa; const float* Pj0; const float* Pj1;
  const float* W1_img; const float* W2_img; const float* W3_img; const float* b2; const float* b3;
  const float* ln_g; const float* ln_b;
  const float* ePa; const float* ePj; const float* eW1_img; const float* eW2_img; const float* eW3_img;
  const float* eb2; const float* eb3;
  NodeTail tail;
  int G, G_enc, N, K, TPN;
  // members of EdgeArgs that no persistent stage uses
  static constexpr const int32_t* mask_attend = nullptr;
  static constexpr const float* b1 = nullptr;
  static constexpr uint32_t drop_thresh = 0, drop_seed = 0;
  static constexpr float drop_scale = 1.0f;
  static constexpr const __bf16* hE16 = nullptr;
  static constexpr __bf16* hE16_out = nullptr;
  static constexpr float* partial = nullptr;
};

struct PersistArgs {
  StageArgs st[6];                      // 3 EncLayer + 3 DecLayer stages
  unsigned* sync;                       // NAMP_SYNC_WORDS zeroed words
  int embed;                            // stage 0 starts from E (h_E = W_e.E + b_e) rather than from an embedded h_E
};

// The stage arguments are read from the kernel-argument segment itself at the point of use (through the segment pointer,
// not through the by-value parameter, whose address cannot be taken without the compiler copying the 3.6 KB into scratch).
__device__ __forceinline__ const PersistArgs* persist_kernargs() {
  return (const PersistArgs*)__builtin_amdgcn_kernarg_segment_ptr();
}

// TAIL as in edge_mlp_kernel (4 = up to 4 residues per workgroup, K in 33..48).  The stage sequence is written out for
// 3 + 3 layers (the reference's only configuration, design_model.json; other depths take the launch chain).
template <int TAIL, int PREC>
__global__ __launch_bounds__(768) void encdec_persistent_kernel(const PersistArgs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const PersistArgs* P = persist_kernargs();
  const int tid = threadIdx.x;
  f4 x[8];
  if (P->embed) edge_stage<MODE_ENC_MSG, TAIL, PREC, PRE_EMBED, 1>(P->st[0], x, smem);
  else          edge_stage<MODE_ENC_MSG, TAIL, PREC, PRE_NONE, 1>(P->st[0], x, smem);
  grid_barrier(P->sync, 1, tid);
  edge_stage<MODE_ENC_MSG, TAIL, PREC, PRE_EDGE, 2>(P->st[1], x, smem);
  grid_barrier(P->sync, 2, tid);
  edge_stage<MODE_ENC_MSG, TAIL, PREC, PRE_EDGE, 2>(P->st[2], x, smem);
  grid_barrier(P->sync, 3, tid);
  edge_stage<MODE_DEC_MSG, TAIL, PREC, PRE_EDGE, 2>(P->st[3], x, smem);
  grid_barrier(P->sync, 4, tid);
  edge_stage<MODE_DEC_MSG, TAIL, PREC, PRE_NONE, 2>(P->st[4], x, smem);
  grid_barrier(P->sync, 5, tid);
  edge_stage<MODE_DEC_MSG, TAIL, PREC, PRE_NONE, 2>(P->st[5], x, smem);
  // A bounded spin that gave up (another kernel occupying CUs on a different stream can starve a non-resident workgroup: co-residency is
  // only guarded against other persistent launches) let its workgroup run past the barrier on stale tables.  The call has long
  // returned NAMP_OK by then, so the failure must travel with the data: every workgroup that sees the timeout word set when it
  // finishes — the one that timed out always does — overwrites the log-probabilities of its residues with NaN
  // (namp_persistent_status still reports which barrier gave up).
  __syncthreads();
  if (__hip_atomic_load((gu32*)P->sync + NAMP_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    const StageArgs& L = P->st[5];
    const int npw = (int)(blockDim.x >> 6) / L.TPN;
    const int vocab = L.tail.vocab;
    for (int e = tid; e < npw * vocab; e += blockDim.x) {
      const int node = blockIdx.x * npw + e / vocab;
      if (node < L.G && L.tail.log_probs) L.tail.log_probs[(long)node * vocab + e % vocab] = __builtin_nanf("");
    }
  }
}

// ------------------------------------------------------------------------------------------
// edge_mlp_bf16_persistent_kernel — the bf16 throughput mode for LARGE batches (more tiles than one wave of
// workgroups).  All three 32 KiB bf16 images stay in LDS for the life of the workgroup (one DMA, one barrier), and
// every wave walks its own strided list of 16-row tiles with NO further synchronisation: waves drift apart, so one
// wave's row loads / GELUs / stores overlap the other waves' MFMAs instead of every workgroup paying
// launch + weight staging + prologue + drain in lock step (measured 16.5 us per 12-tile pass at cfg3, of which
// the MFMAs are ~2 us).  Same arithmetic, in the same order per row, as edge_mlp_kernel<MODE, 0, true>.
// ------------------------------------------------------------------------------------------
// per-row operands of one tile that do not depend on its h_E row: resolved one tile ahead so that the dependent
// E_idx -> mask / rank -> table-address chain of tile n+1 completes under tile n's GEMMs
struct TileMeta {
  long pa_row, pj_row; bool pj_from1;      // table rows: Pa[pa_row]; Pj0[pj_row] or (pj_from1) Pj1[pj_row]
  long erow; float w_row; int node, kt; bool valid;
};

// The edge row (index into E_idx / h_E) of lane m of a tile: what tile_meta reads its neighbour id from.
template <int MODE>
__device__ __forceinline__ long tile_erow(const EdgeArgs& a, long tile, int m) {
  const int node = (int)((unsigned)tile / (unsigned)a.TPN);
  const int kt = (int)tile - node * a.TPN;
  const int b_dec = node / a.N;
  const int i_loc = node - b_dec * a.N;
  const int node_enc = (MODE == MODE_DEC_MSG) ? ((b_dec % (a.G_enc / a.N)) * a.N + i_loc) : node;
  const int k = 16 * kt + m;
  return (long)node_enc * a.K + (k < a.K ? k : 0);
}

template <int MODE>
__device__ __forceinline__ TileMeta tile_meta(const EdgeArgs& a, long tile, int m, int g) {
  TileMeta t;
  t.node = (int)((unsigned)tile / (unsigned)a.TPN);       // G * TPN < 2^31 (checked by the launchers): no 64-bit division
  t.kt = (int)tile - t.node * a.TPN;
  const int b_dec = t.node / a.N;
  const int i_loc = t.node - b_dec * a.N;
  const int node_enc = (MODE == MODE_DEC_MSG) ? ((b_dec % (a.G_enc / a.N)) * a.N + i_loc) : t.node;
  const int k = 16 * t.kt + m;
  t.valid = k < a.K;
  t.erow = (long)node_enc * a.K + (t.valid ? k : 0);
  t.w_row = 0.f;
  const int j_loc = a.E_idx[t.erow];
  if (MODE == MODE_DEC_MSG) {
    const int j_dec = b_dec * a.N + j_loc;
    const bool bwd = a.rank[j_dec] < a.rank[t.node];
    t.pj_from1 = !bwd;
    t.pj_row = bwd ? (long)j_dec : (long)(node_enc - i_loc + j_loc);
    t.w_row = t.valid ? (1.0f / 30.0f) : 0.f;
  } else {
    const int j = t.node - i_loc + j_loc;
    t.pj_from1 = false;
    t.pj_row = j;
    if (MODE == MODE_ENC_MSG) {
      int ma;
      if (a.mask_attend) ma = a.mask_attend[t.erow];
      else ma = a.mask ? (a.mask[t.node] * a.mask[j]) : 1;
      t.w_row = t.valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
    }
  }
  t.pa_row = t.node;
  return t;
}

// tile_meta for a step loop that must not wait on its own requests: the neighbour id j_loc = E_idx[tile_erow] was requested by the caller
// earlier, and the mask / rank reads are issued without control flow around them (a load behind a branch makes the compiler drain the
// in-order memory counter at the join): absent mask arrays read the constant 1.
__device__ const int32_t g_meta_one = 1;
template <int MODE>
__device__ __forceinline__ TileMeta tile_meta_pre(const EdgeArgs& a, long tile, int m, int j_loc) {
  TileMeta t;
  t.node = (int)((unsigned)tile / (unsigned)a.TPN);
  t.kt = (int)tile - t.node * a.TPN;
  const int b_dec = t.node / a.N;
  const int i_loc = t.node - b_dec * a.N;
  const int node_enc = (MODE == MODE_DEC_MSG) ? ((b_dec % (a.G_enc / a.N)) * a.N + i_loc) : t.node;
  const int k = 16 * t.kt + m;
  t.valid = k < a.K;
  t.erow = (long)node_enc * a.K + (t.valid ? k : 0);
  t.w_row = 0.f;
  if (MODE == MODE_DEC_MSG) {
    const int j_dec = b_dec * a.N + j_loc;
    const bool bwd = a.rank[j_dec] < a.rank[t.node];
    t.pj_from1 = !bwd;
    t.pj_row = bwd ? (long)j_dec : (long)(node_enc - i_loc + j_loc);
    t.w_row = t.valid ? (1.0f / 30.0f) : 0.f;
  } else {
    const int j = t.node - i_loc + j_loc;
    t.pj_from1 = false;
    t.pj_row = j;
    if (MODE == MODE_ENC_MSG) {
      // (global address space stated: a select of generic pointers would become flat loads, which retire out of order with the rest)
      typedef const __attribute__((address_space(1))) int32_t* gptr;
      const gptr one = (gptr)&g_meta_one;
      const gptr p0 = a.mask_attend ? (gptr)(a.mask_attend + t.erow) : a.mask ? (gptr)(a.mask + t.node) : one;
      const gptr p1 = (!a.mask_attend && a.mask) ? (gptr)(a.mask + j) : one;
      const int ma = *p0 * *p1;
      t.w_row = t.valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
    }
  }
  t.pa_row = t.node;
  return t;
}

template <int MODE>
__global__ __launch_bounds__(768) void edge_mlp_bf16_persistent_kernel(const EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long ntiles = (long)a.G * a.TPN;
  const long stride = (long)gridDim.x * nwaves;
  long tile = (long)blockIdx.x * nwaves + wave;
  // first tile's operands go out before the weight DMA (the VM counter retires in order)
  TileMeta cur = tile_meta<MODE>(a, tile < ntiles ? tile : 0, m, g);
  f4 xn[8];
  {
    const float* src = a.hE + cur.erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) xn[t] = *(const f4*)(src + 16 * t);
  }
  // constant vectors and the tile's single Pa row through LDS instead of the vector-memory path (see edge_mlp_bf16s32_kernel)
  float* cst = (float*)(smem + 3 * NAMP_BIMG_BYTES);
  if (MODE == MODE_ENC_EDGE && tid < 512) {
    const float* srcv = tid < 128 ? a.b2 : tid < 256 ? a.b3 : tid < 384 ? a.ln_g : a.ln_b;
    cst[tid] = srcv ? srcv[tid & 127] : 0.f;
  }
  if (MODE != MODE_ENC_EDGE && tid < 128) cst[tid] = a.b2[tid];
  const float* pa_slot = (const float*)(smem + 3 * NAMP_BIMG_BYTES + 2048 + wave * 512);
  auto pa_fetch = [&](const long row) {
    if (lane < 32)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.Pa + row * NAMP_H + 4 * lane),
                                       (__attribute__((address_space(3))) void*)pa_slot, 16, 0, 0);
  };
  pa_fetch(cur.pa_row);
  dma_to_lds(smem, a.W1_img, 32, wave, nwaves, lane);
  dma_to_lds(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, nwaves, lane);
  if (MODE == MODE_ENC_EDGE) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.W3_img, 32, wave, nwaves, lane);
  wait_dma_and_sync();
  const bf8* bw = (const bf8*)smem + lane;
  for (; tile < ntiles; tile += stride) {
    asm volatile("" ::: "memory");        // the weight fragments are loop-invariant LDS reads: keep them out of registers
    f4 x[8], acc[8], pjv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = xn[t];
    const TileMeta me = cur;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this tile's Pa row (DMA'd one tile ahead) and h_E row have landed
    {
      const float* pj = (me.pj_from1 ? a.Pj1 : a.Pj0) + me.pj_row * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(pa_slot + 16 * t + 4 * g); pjv[t] = *(const f4*)(pj + 16 * t); }
    }
    // next tile: metadata chain + h_E row, in flight under this tile's GEMMs
    const long nt = tile + stride;
    cur = tile_meta<MODE>(a, nt < ntiles ? nt : tile, m, g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot has been read before the next row overwrites it
    pa_fetch(cur.pa_row);
    {
      const float* src = a.hE + cur.erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) xn[t] = *(const f4*)(src + 16 * t);
    }
    chain_gemm_bf16<false, false>(acc, x, bw);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] += pjv[t];
    f4 (&y)[8] = pjv;
    if (MODE == MODE_ENC_EDGE) {
#pragma unroll
      for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(cst + 16 * t + 4 * g);
      chain_gemm_bf16<false, true>(y, acc, bw + (NAMP_BIMG_BYTES / 16));
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(cst + 128 + 16 * t + 4 * g);
      chain_gemm_bf16<false, true>(acc, y, bw + 2 * (NAMP_BIMG_BYTES / 16));
      if (a.drop_thresh) {                                        // training forward: dropout3 on the message
        const uint32_t key = drop_row_key(a.drop_seed, me.erow);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] *= drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, a.drop_scale);
      }
      if (a.ln_g) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] += x[t];               // residual: the row is still in registers
        layernorm_row_T(acc, cst + 256, cst + 384, g);
      }
      if (me.valid) {
        float* dst = a.hE_out + me.erow * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
      }
    } else {
      // layer 2 in the F orientation: rows 4g+r of channel 16t + m
#pragma unroll
      for (int t = 0; t < 8; ++t) { const float b = cst[16 * t + m]; y[t] = (f4){b, b, b, b}; }
      chain_gemm_bf16<true, true>(y, acc, bw + (NAMP_BIMG_BYTES / 16));
      // K-sum of the layer-2 activations (layer 3 is applied per residue by the residue kernel: NodeTail.m3_img)
      float wr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) wr[r] = __shfl(me.w_row, 4 * g + r);
      float wsum = me.w_row;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) wsum += __shfl_xor(wsum, o);
      float* dst = a.partial + ((long)me.node * a.TPN + me.kt) * NAMP_H + m;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f4 v = gelu_prec<PREC_BF16>(y[t]);
        float s_ = (v.x * wr[0] + v.y * wr[1]) + (v.z * wr[2] + v.w * wr[3]);
        s_ = xg_sum(s_);
        if (g == 0) dst[16 * t] = s_;
      }
      if (lane == 0) a.partial[(long)a.G * a.TPN * NAMP_H + (long)me.node * a.TPN + me.kt] = wsum;
    }
  }
}

// ------------------------------------------------------------------------------------------
// edge_mlp_x3_persistent_kernel — the split-bf16 (parity) mode for LARGE batches.  The three 64 KiB x3 images cannot all
// stay in LDS, so the 2-slot ring keeps turning — but across the workgroup's rounds of 12 tiles instead of once per
// launch: W1 of round r+1 streams in under GEMM 3 of round r, the next round's h_E rows and table addresses are
// requested one round ahead (edge_mlp_bf16_persistent_kernel's TileMeta chain), and launch ramp / first-image wait are
// paid once per workgroup instead of once per 12 tiles.  Same arithmetic per row as edge_mlp_kernel<MODE, 0, PREC_X3>.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(768) void edge_mlp_x3_persistent_kernel(const EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long ntiles = (long)a.G * a.TPN;
  const long stride = (long)gridDim.x * nwaves;
  long base = (long)blockIdx.x * nwaves;                   // workgroup-uniform: every wave takes part in every barrier
  char* slotA = smem;                                      // holds (or receives) W1 of the current round
  char* slotB = smem + NAMP_IMG_BYTES;
  TileMeta cur = tile_meta<MODE>(a, base + wave < ntiles ? base + wave : 0, m, g);
  f4 xn[8];
  {
    const float* src = a.hE + cur.erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) xn[t] = *(const f4*)(src + 16 * t);
  }
  // edge update: b2 | b3 | LayerNorm-3 weight | bias staged behind the ring (see edge_mlp_bf16s32_kernel): 32 of the 64 16-byte
  // loads per tile were these four constant vectors
  float* cst = (float*)(smem + 2 * NAMP_IMG_BYTES);
  if (MODE == MODE_ENC_EDGE && tid < 512) {
    const float* srcv = tid < 128 ? a.b2 : tid < 256 ? a.b3 : tid < 384 ? a.ln_g : a.ln_b;
    cst[tid] = srcv ? srcv[tid & 127] : 0.f;              // (ln_g / ln_b are null when the bare message is requested)
  }
  if (MODE != MODE_ENC_EDGE && tid < 128) cst[tid] = a.b2[tid];
  // the tile's single Pa row through a 512-byte LDS slot per wave, DMA'd one round ahead (see edge_mlp_bf16s32_kernel)
  const float* pa_slot = (const float*)(smem + 2 * NAMP_IMG_BYTES + 2048 + wave * 512);
  auto pa_fetch = [&](const long row) {
    if (lane < 32)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.Pa + row * NAMP_H + 4 * lane),
                                       (__attribute__((address_space(3))) void*)pa_slot, 16, 0, 0);
  };
  pa_fetch(cur.pa_row);
  dma_to_lds(slotA, a.W1_img, 64, wave, nwaves, lane);
  for (; base < ntiles; base += stride) {
    const bool active = base + wave < ntiles;
    f4 x[8], acc[8], pjv[8];
    const TileMeta me = cur;
    wait_dma_and_sync();                                   // W1 landed; everyone is done with the previous round's GEMM 3 (slotB)
    dma_to_lds(slotB, a.W2_img, 64, wave, nwaves, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = xn[t];
    {
      // (the wait at the top of the round drained the vector-memory counter: this round's Pa row has landed in the slot)
      const float* pj = (me.pj_from1 ? a.Pj1 : a.Pj0) + me.pj_row * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc[t] = *(const f4*)(pa_slot + 16 * t + 4 * g); pjv[t] = *(const f4*)(pj + 16 * t); }
    }
    // next round: metadata chain + h_E row, in flight under this round's GEMMs
    const long nb = base + stride;
    const bool more = nb < ntiles;                         // workgroup-uniform
    cur = tile_meta<MODE>(a, nb + wave < ntiles ? nb + wave : (active ? base + wave : 0), m, g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot has been read before the next row overwrites it
    pa_fetch(cur.pa_row);
    {
      const float* src = a.hE + cur.erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) xn[t] = *(const f4*)(src + 16 * t);
    }
    gemm128<true, false, false>(acc, x, (const f4*)slotA + lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] += pjv[t];
    wait_dma_and_sync();                                   // W2 landed; everyone is done with slotA (W1)
    f4 (&y)[8] = pjv;
    if (MODE != MODE_ENC_EDGE) {
      // message modes: two images per round (layer 3 is hoisted behind the K-sum) — W1 of the next round streams into slotA
      // under GEMM 2, the slots keep their roles
      if (more) dma_to_lds(slotA, a.W1_img, 64, wave, nwaves, lane);
#pragma unroll
      for (int t = 0; t < 8; ++t) { const float b = cst[16 * t + m]; y[t] = (f4){b, b, b, b}; }
      gemm128<true, true, true>(y, acc, (const f4*)slotB + lane);       // F orientation: rows 4g+r of channel 16t + m
      // K-sum of the layer-2 activations (layer 3 is applied per residue by the residue kernel: NodeTail.m3_img)
      float wr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) wr[r] = __shfl(me.w_row, 4 * g + r);
      float wsum = me.w_row;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) wsum += __shfl_xor(wsum, o);
      float* dst = a.partial + ((long)me.node * a.TPN + me.kt) * NAMP_H + m;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f4 v = gelu_prec<PREC_X3>(y[t]);
        float s_ = (v.x * wr[0] + v.y * wr[1]) + (v.z * wr[2] + v.w * wr[3]);
        s_ = xg_sum(s_);
        if (active && g == 0) dst[16 * t] = s_;
      }
      if (active && lane == 0) a.partial[(long)a.G * a.TPN * NAMP_H + (long)me.node * a.TPN + me.kt] = wsum;
      continue;
    }
    dma_to_lds(slotA, a.W3_img, 64, wave, nwaves, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(cst + 16 * t + 4 * g);
    gemm128<true, false, true>(y, acc, (const f4*)slotB + lane);
    wait_dma_and_sync();                                   // W3 landed; everyone is done with slotB (W2)
    if (more) dma_to_lds(slotB, a.W1_img, 64, wave, nwaves, lane);   // next round's W1, under GEMM 3
    if (MODE == MODE_ENC_EDGE) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(cst + 128 + 16 * t + 4 * g);
      gemm128<true, false, true>(acc, y, (const f4*)slotA + lane);
      if (a.drop_thresh) {
        const uint32_t key = drop_row_key(a.drop_seed, me.erow);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] *= drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, a.drop_scale);
      }
      if (a.ln_g) {
        const float* src = a.hE + me.erow * NAMP_H + 4 * g;           // residual: re-read (L2), 32 registers less than keeping x
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] += *(const f4*)(src + 16 * t);
        layernorm_row_T(acc, cst + 256, cst + 384, g);
      }
      if (active && me.valid) {
        float* dst = a.hE_out + me.erow * NAMP_H + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
      }
    }
    char* tmp = slotA; slotA = slotB; slotB = tmp;         // next round's W1 sits in this round's slotB
  }
}

// ------------------------------------------------------------------------------------------
// dec_sample_kernel — the autoregressive sampler (ProteinMPNN.sample, non-symmetric branch,
// inference/model_utils.py:126-218) as ONE persistent launch.  Sample streams are independent, so
// there is no inter-workgroup traffic: a workgroup owns up to 4 streams (tile rows of the residue tail)
// and walks their decoding orders; at step t stream b visits residue i = order[b][t] and runs, per
// decoder layer l, exactly the message MLP + residue tail of the parallel decoder restricted to that
// residue (the three-term first layer of edge_mlp_kernel, with Pbw[j] = Pv_l[j] + tok_l[S[j]] gathered for
// neighbours already decoded in this stream and the static encoder-context table Pfw_l[j] for the rest),
// then the W_out head, temperature softmax with the special tokens removed (model_utils.py:194-205),
// an inverse-CDF draw from a caller-supplied uniform (torch.multinomial's stream is not reproducible
// off-device; the caller seeds torch.rand instead) and the table updates for the sampled token.
// State written by a stream is only ever read by the same workgroup (barrier + workgroup fence); words
// that share cache lines with not-yet-written neighbours (S) are read with L1-bypassing loads.
// ------------------------------------------------------------------------------------------
#define NAMP_SAMPLE_SLOTS 4
#define NAMP_WALK_MAX_GRID 128      // workgroups of the persistent level walk (all must be resident: one 155 KiB-LDS workgroup per CU)
// LDS of the sampler: the 2 x 64 KiB weight ring | the residue tail's scratch (its own region: the next layer's images stream into the ring
// while the tail runs) | per-wave partial K-sums | node / visit ids
#define SAMPLE_TAIL_BYTES (ROWS_TAIL_LDS_FLOATS(NAMP_SAMPLE_SLOTS) * 4)
#define SAMPLE_LDS (2 * NAMP_IMG_BYTES + SAMPLE_TAIL_BYTES + 12 * NAMP_H * 4 + 64)

// generic -> global: a pointer loaded from memory has no address space; every pointer of this ABI is device global memory
template <class T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(__attribute__((address_space(1))) T*)p; }

// Whole-wave (64-lane) reductions without LDS round trips: a butterfly inside each 16-lane DPP row (quad_perm pairs, then the row's mirrors as
// pairings), then the four row totals through v_readlane.  __shfl_xor compiles to ds_bpermute_b32 — an LDS round trip per step, six per reduction,
// and the sampler's head + draw is a chain of five reductions and a scan on the critical path of every level.
template <int CTRL>
__device__ __forceinline__ float wv_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wv_lane(const float v, const int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float wave_sum64(float v) {
  v += wv_dpp<0xB1>(v); v += wv_dpp<0x4E>(v); v += wv_dpp<0x141>(v); v += wv_dpp<0x140>(v);      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
  return (wv_lane(v, 0) + wv_lane(v, 16)) + (wv_lane(v, 32) + wv_lane(v, 48));
}
__device__ __forceinline__ float wave_max64(float v) {
  v = fmaxf(v, wv_dpp<0xB1>(v)); v = fmaxf(v, wv_dpp<0x4E>(v)); v = fmaxf(v, wv_dpp<0x141>(v)); v = fmaxf(v, wv_dpp<0x140>(v));
  return fmaxf(fmaxf(wv_lane(v, 0), wv_lane(v, 16)), fmaxf(wv_lane(v, 32), wv_lane(v, 48)));
}
// inclusive prefix sum over the 64 lanes: row_shr scans inside the rows (zeros shifted in), then the totals of the rows below
__device__ __forceinline__ float wave_scan64(float v, const int lane) {
  v += wv_dpp<0x111>(v); v += wv_dpp<0x112>(v); v += wv_dpp<0x114>(v); v += wv_dpp<0x118>(v);
  const float r0 = wv_lane(v, 15), r1 = wv_lane(v, 31), r2 = wv_lane(v, 47);
  const int row = lane >> 4;
  return v + (row == 0 ? 0.f : row == 1 ? r0 : row == 2 ? (r0 + r1) : ((r0 + r1) + r2));
}

struct SampleLayer {
  const float* Z1;         // [G_enc * K][128]  W1e_l . h_E[i,k]: static, so it is computed for every edge and layer BEFORE the walk (round 5)
  const float* W2_img; const float* W3_img; const float* b2; const float* b3;
  const float* tok;        // [vocab][128]  W1s . W_s
  const float* Pfw;        // [G_enc][128]  W1v_l . h_V^enc            (static)
  float* Pa;               // layer 0: [G_enc][128] static; else [G_dec][128], written at the residue's step
  float* Pv;               // layer 0: == Pfw;                 else [G_dec][128]  W1v_l . h_V^(l)
  NodeTail tail;           // LN1/FFN/LN2 of this layer; hV = h^(l), hV_out = h^(l+1); projections -> next layer's Pa / Pv
};

struct SampleArgs {
  const float* hE; const int32_t* E_idx;
  const int32_t* mask_true;    // [G_enc]  the residue mask itself: gates the whole context of a residue (mask_bw / mask_fw,
                               //          model_utils.py:135-137) — NOT the same as tail.mask when the stream-0 quirk is on
  const int32_t* chain_mask;   // [G_enc]  mask * chain_mask
  const int32_t* S_true;       // [G_enc]
  const float* bias;           // [G_enc][vocab]
  const int32_t* order;        // [B_dec][N]
  const int32_t* rank;         // [B_dec][N]
  const float* uniform;        // [B_dec][N] indexed by step
  const int32_t* S_forced;     // optional [B_dec][N] indexed by residue (teacher forcing)
  // symmetry-tied sampling (model_utils.py:219-326): consecutive visits of `order` form groups that share one draw
  const int32_t* group_first;  // optional [B_dec][N]: visit index of the first member of visit v's group (null: v itself)
  const int32_t* group_last;   // optional [B_dec][N]: 1 if visit v closes its group                       (null: always)
  const float* sym_w;          // optional [G_enc]: weight of a residue's logits in its group's sum        (null: 1)
  const float* pair_bias;      // optional [G_enc][vocab][N][vocab] (model_utils.py:116,170-172)
  const float* head_wT; const float* head_b;   // head_wT [128][64]: W_out transposed, token t of channel c at c * 64 + t (zero beyond vocab): coalesced reads
  int32_t* S_out;              // [B_dec][N]   (also the running sequence read back for decoded neighbours)
  float* probs_out;            // [B_dec][N][vocab]
  float* logp_out;             // [B_dec][N][vocab]
  unsigned long long special;  // bit t set -> token t can never be drawn
  float inv_T;
  int B_dec, B_enc, N, K, TPN, n_layers, vocab, slots;
  SampleLayer l[NAMP_MAX_LAYERS];      // 8 x 480 B: with the scalars and the launch's own arguments 4,072 of the 4,096 kernel-argument bytes
};
static_assert(sizeof(SampleArgs) + 32 <= 4096, "SampleArgs + the sampler launch's scalar arguments must fit the kernel-argument segment");

struct SampleRows {             // tile row n -> residue of stream (b0 + n) at this step, or -1
  const int* node_lds;
  __device__ __forceinline__ int operator()(int n) const { return node_lds[n]; }
};

// ------------------------------------------------------------------------------------------
// bf16 STORAGE of the throughput mode's large batches (namp_bf16s32.h): with bf16 MFMA those launches are bound by the bytes of h_E
// and of the gathered table rows, so those are kept in bf16 too, in fragment order B (frag_off above) — the order in which
// v_mfma_f32_32x32x16_bf16 takes a row as an operand, so GEMM-1 operands need no conversion and the edge update writes rows in the order
// the next launch reads them.  fp32 accumulation, fp32 LayerNorm / K-sum / residue tail.
// (Tried: fp32 tables — no bf16 -> fp32 conversions, no conversion launches — 6.93 -> 7.13 ms per cfg3 step: the gathers' L2 bytes count.)
// ------------------------------------------------------------------------------------------
// fp32 [rows][128] (plain channel order) -> bf16 fragment order B, for up to 4 tables per launch
struct CvtTables { const float* src[4]; __bf16* dst[4]; int n; long rows; };
static __global__ void cvt_tables_bf16_kernel(const CvtTables c) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte output piece: (row, s', hk)
  if (e >= c.rows * 16) return;
  const long row = e >> 4;
  const int sp = (int)(e >> 1) & 7, hk = (int)e & 1;                 // consecutive threads write consecutive pieces
  for (int q = 0; q < c.n; ++q) {
    const float* s0 = c.src[q] + row * NAMP_H + 16 * sp + 4 * hk;
    const f4 lo = *(const f4*)s0, hi = *(const f4*)(s0 + 8);
    *(bf8*)(c.dst[q] + row * NAMP_H + 8 * (2 * sp + hk)) = pack_bf16<false>(lo, hi);
  }
}

// W_out [vocab][128] -> [128][64] (token-minor, zero beyond vocab) for the sampler's head
static __global__ void head_transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int vocab) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;              // c * 64 + t
  if (e >= NAMP_H * 64) return;
  const int c = e >> 6, t = e & 63;
  WT[e] = t < vocab ? W[t * NAMP_H + c] : 0.f;
}

// sample_levels_kernel: dependency level of every visit of the plain sampling branch.  One wave per stream walks its
// decoding order once: level(i) = 1 + max level of the neighbours visited before i (0 if none) — lanes cover the K
// neighbours, levels live in LDS.  level_out[b][t] is indexed by VISIT t.
// dep_idx (optional, [B_enc][N][D], -1 = none): further residues a step depends on besides the graph neighbours — with `pair_bias` the
// bias of residue i reads the token of every residue j whose block pair_bias[i, :, j, :] is not all zero (model_utils.py:169-172).
// group_first / group_last (optional, [B_dec][N] by visit; symmetry-tied sampling): the visits of a group share ONE level — 1 + the
// highest level among the members' dependencies in EARLIER groups (a member's neighbours inside its own group are handled by the
// group's own work item, which runs its members one after the other).
static __global__ __launch_bounds__(64) void sample_levels_kernel(const int32_t* __restrict__ E_idx, const int32_t* __restrict__ order,
                                                          const int32_t* __restrict__ rank, const int32_t* __restrict__ dep_idx, int D,
                                                          const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_last,
                                                          int32_t* __restrict__ level_out, int B_enc, int N, int K) {
  extern __shared__ int lv[];                                    // [N]
  const int b = blockIdx.x, lane = threadIdx.x;
  const int b_enc = b % B_enc;
  const int32_t* rk = rank + (long)b * N;
  const int32_t* ord = order + (long)b * N;
  int dg = -1;                                                   // running maximum over the open group's members
  // The visit's residue and its first 64 neighbours are requested one visit AHEAD (they do not depend on the levels): what is left on the
  // chain of a visit is the LDS look-up of the neighbours' levels and one wave maximum (DPP + readlane, no LDS round trips) — the walk is
  // serial over the N visits of a stream, and at ~0.8 us per visit it was 78 us of a 97-residue design call.
  // (the residues of 64 consecutive visits sit one per lane and are handed out by v_readlane: no load on the way to a visit's residue)
  int ord_blk = ord[lane < N ? lane : 0];
  int i_n = __builtin_amdgcn_readlane(ord_blk, 0);
  int j_n = lane < K ? E_idx[((long)b_enc * N + i_n) * K + lane] : -1;
  int rj_n = j_n >= 0 ? rk[j_n] : 0x7fffffff;
  for (int t = 0; t < N; ++t) {
    const int i = i_n, j0 = j_n, rj0 = rj_n;
    if (t + 1 < N) {
      if (((t + 1) & 63) == 0) ord_blk = ord[t + 1 + lane < N ? t + 1 + lane : 0];
      i_n = __builtin_amdgcn_readlane(ord_blk, (t + 1) & 63);
      j_n = lane < K ? E_idx[((long)b_enc * N + i_n) * K + lane] : -1;
      rj_n = j_n >= 0 ? rk[j_n] : 0x7fffffff;
    }
    const int vf = group_first ? group_first[(long)b * N + t] : t;       // rank[i] == t; the group's first visit
    int d = -1;
    if (rj0 < vf) d = lv[j0];                                    // earlier groups already have their level
    for (int k = lane + 64; k < K; k += 64) {
      const int j = E_idx[((long)b_enc * N + i) * K + k];
      if (rk[j] < vf) d = max(d, lv[j]);
    }
    if (dep_idx)
      for (int k = lane; k < D; k += 64) {
        const int j = dep_idx[((long)b_enc * N + i) * D + k];
        if (j >= 0 && rk[j] < vf) d = max(d, lv[j]);
      }
    d = (int)wave_max64((float)d);                               // (levels < 2^24: exact in fp32)
    dg = (vf == t) ? d : max(dg, d);
    const bool closes = group_last ? (group_last[(long)b * N + t] != 0) : true;
    if (closes)
      for (int v = vf + lane; v <= t; v += 64) { lv[ord[v]] = dg + 1; level_out[(long)b * N + v] = dg + 1; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // LDS write visible to the wave's next iteration (the requests made ahead stay in flight)
    __builtin_amdgcn_wave_barrier();
  }
}

// LEVEL = false: the sequential walk, one workgroup per <= 4 streams (a dense pair_bias, whose steps depend on every earlier one;
// K > 128).  LEVEL = true: ONE step for an arbitrary list of (stream, visit) pairs — the plain branch's
// step for residue i depends only on the neighbours decoded before it, so the host groups the visits of all streams into
// dependency levels (sample_levels_kernel: level = 1 + max level of the earlier neighbours) and launches one grid per level:
// ~64 launches instead of 1000 sequential steps at N = 1000, K = 48, with every workgroup of the chip busy.  Same arithmetic
// per residue, same uniform per visit, hence the same draws.
// MAXW = 8 waves per workgroup (K <= 128: 8 / TPN streams per workgroup) leaves 256 VGPRs per lane — the three row operands, the
// residue tail's two fragment sets in flight and the head fit without scratch (the 12-wave form spilled 165 registers per lane); MAXW = 12
// only serves K > 128.
// MODE 0: the sequential walk.  MODE 1: one dependency level per launch (work = the level's (stream, visit) pairs).  MODE 2 (round 3):
// ALL levels in one persistent launch — work = every (stream, visit) pair sorted by level, level_off[l] = first pair of level l
// (level_off[l] >= nwork ends the walk), a grid barrier between levels.  A kernel boundary per level costs more than its ~2 us here:
// the level's one or two workgroups re-fetch the 2.5 MB of decoder weights through a cold L2 (they came at ~45 GB/s, half of a level's
// 90 us at B = 1), and the host has to read the level histogram back before it can launch.  What later levels gather from other
// workgroups — the next layer's Pv rows — is stored write-through (sc1); tokens travel through agent-scope atomics as before; no
// workgroup ever reads a row before the level that wrote it (rank order), so no reader holds a stale line.
#ifndef NAMP_SAMPLE_AHEAD
#define NAMP_SAMPLE_AHEAD 1
#endif
// the sampler's three tile GEMMs: split-bf16 with the fragments requested ahead (chain_gemm_x3_ahead) in the 8-wave forms
#define SGEMM(FLIP, ACT) sample_gemm<X3, MAXW == 8 && NAMP_SAMPLE_AHEAD, FLIP, ACT>
template <bool X3, bool AHEAD, bool FLIP, bool ACT>
__device__ __forceinline__ void sample_gemm(f4 (&acc)[8], const f4 (&x)[8], const f4* w) {
  if constexpr (X3 && AHEAD) chain_gemm_x3_ahead<FLIP, ACT>(acc, x, (const bf8*)w);
  else gemm128<X3, FLIP, ACT>(acc, x, w);
}
template <int MODE, bool X3 = false, int MAXW = 8>
__global__ __launch_bounds__(MAXW * 64) void dec_sample_kernel(const SampleArgs a, const int32_t* __restrict__ work,
                                                               const int32_t* __restrict__ work_n, int nwork,
                                                               const int32_t* __restrict__ level_off, unsigned* sync) {
  constexpr bool LEVEL = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;
  char* buf1 = smem + NAMP_IMG_BYTES;
  float* tail_lds = (float*)(smem + 2 * NAMP_IMG_BYTES);
  float* dpart = (float*)(smem + 2 * NAMP_IMG_BYTES + SAMPLE_TAIL_BYTES);            // [12 waves][128]
  int* node_lds = (int*)(dpart + 12 * NAMP_H);                   // [4] residue (dec-global) per slot, -1 idle
  int* t_lds = node_lds + NAMP_SAMPLE_SLOTS;                     // [4] visit index per slot (LEVEL)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int slot = wave / a.TPN, kt = wave - slot * a.TPN;
  const f4* w0 = (const f4*)buf0 + lane;
  const f4* w1 = (const f4*)buf1 + lane;
  const SampleRows rows = {node_lds};
  NAMP_WSTAMP_INIT();
  float tot = 0.f;                                  // running symmetry-group logit sum (head waves)
  // every token starts "not drawn" (-1): the reference's h_S is all-zero until a residue is assigned (:264)
  // (LEVEL: the host fills S_out with -1 before the first level)
  if (!LEVEL) {
    for (int s_ = 0; s_ < a.slots; ++s_) {
      const int bs = blockIdx.x * a.slots + s_;
      if (bs < a.B_dec)
        for (int q = tid; q < a.N; q += blockDim.x) __hip_atomic_store(a.S_out + (long)bs * a.N + q, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }

  // The first layer's images (through registers: see copy_to_lds).  Round 5: the first-layer product W1e . h_E does not depend on the decoded state
  // — it is a table (SampleLayer.Z1) made before the walk —, so a layer needs W2 and W3 only: both are resident when the layer starts, there is
  // no copy and no barrier between the layer's products, and a step has one tile product per layer less.
  copy_to_lds<8>(buf0, a.l[0].W3_img, 64, wave, nwaves, lane);
  copy_to_lds<8>(buf1, a.l[0].W2_img, 64, wave, nwaves, lane);
  __syncthreads();
  // One step: the workgroup's <= a.slots items starting at item0 (MODE 0: streams item0 .. at visit t_seq; else work-list entries
  // item0 .. < nitems).  `more`: another step follows in this launch (its first layer's images are requested under this step's last tail).
#ifdef NAMP_ABL_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) namp_stamp_prev = wall_clock64();
#endif
  // MODE 2 with symmetry groups split over several work items (members that are not graph neighbours of each other run in parallel):
  // every member's logits go to zbuf [B_dec][N][vocab]; after the level's grid barrier one wave per group sums them in visit order — the
  // same fma chain as the running sum of the walk — and draws (close = (stream, last visit) per group sorted by level, close_off per level)
  float* zbuf = nullptr; const int32_t* close_list = nullptr; const int32_t* close_off = nullptr;
  if constexpr (MODE == 2) {
    const unsigned long long* dp = (const unsigned long long*)(sync + NAMP_SYNC_DEFER);
    zbuf = as_global((float*)dp[0]); close_list = as_global((const int32_t*)dp[1]); close_off = as_global((const int32_t*)dp[2]);
  }
  // softmax((total + bias_t [+ pair_bias_t]) / T) with bias of the group's last residue (visit t of stream bq), special tokens removed,
  // renormalised (model_utils.py:194-205 / :300-312); inverse-CDF draw; the token goes to every member v_first .. t.  One wave.
  auto draw = [&](const int bq, const int t, const int v_first, const float total) __attribute__((always_inline)) {
    const int iq = a.order[(long)bq * a.N + t];
    const int ne = (bq % a.B_enc) * a.N + iq;
    const long vis = (long)bq * a.N + t;
    float add = (lane < a.vocab) ? a.bias[(long)ne * a.vocab + lane] : 0.f;
    if (a.pair_bias && lane < a.vocab) {
      const float* pb = a.pair_bias + ((long)ne * a.vocab + lane) * a.N * a.vocab;
      float acc_pb = 0.f;
      const int rk_i = LEVEL ? a.rank[(long)bq * a.N + iq] : 0;
      for (int j2 = 0; j2 < a.N; ++j2) {
        int Sj2 = __hip_atomic_load(a.S_out + (long)bq * a.N + j2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // decoded by dependency level, a residue that comes LATER in the decoding order may already hold its token (it sits in
        // an earlier or in this level when nothing ties it to residue i): the sequential walk sees PAD there
        if (LEVEL && a.rank[(long)bq * a.N + j2] >= rk_i) Sj2 = -1;
        if (Sj2 < 0) Sj2 = a.vocab - 1;            // not decoded yet: the reference's initial S is PAD (:157)
        acc_pb += pb[(long)j2 * a.vocab + Sj2];
      }
      add += acc_pb;
    }
    const float u = a.uniform[vis];                                 // (requested ahead of the reductions)
    float zt = (lane < a.vocab) ? (total + add) * a.inv_T : -INFINITY;
    const float mt = wave_max64(zt);
    float p = (lane < a.vocab) ? expf(zt - mt) : 0.f;
    p = p / wave_sum64(p);
    if ((a.special >> lane) & 1ull) p = 0.f;
    p = p / wave_sum64(p);
    // inverse CDF: first token whose inclusive prefix sum exceeds u
    const float cdf = wave_scan64(p, lane);
    const unsigned long long hit = __ballot(p > 0.f && cdf > u);
    const unsigned long long any = __ballot(p > 0.f);
    int S_t = hit ? (int)__builtin_ctzll(hit) : (any ? 63 - (int)__builtin_clzll(any) : 0);
    // assign the draw to every member of the group, in visit order; a fixed member (chain_mask 0) replaces the
    // running token by its own and passes THAT on — the reference's behaviour (model_utils.py:318-324)
    for (int v = v_first; v <= t; ++v) {
      const int im = a.order[(long)bq * a.N + v];
      const int nem = (bq % a.B_enc) * a.N + im;
      const long ndm = (long)bq * a.N + im;
      if (a.S_forced) S_t = a.S_forced[ndm];
      const int cm = a.chain_mask[nem];
      if (!cm) S_t = a.S_true[nem];
      if (lane < a.vocab) a.probs_out[ndm * a.vocab + lane] = cm ? p : 0.f;
      if (lane == 0) __hip_atomic_store(a.S_out + ndm, S_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto step = [&](const int item0, const int nitems, const int t_seq, const bool more) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");      // keep the step's (loop-invariant) vector loads inside the step, not in registers across steps
    NAMP_STAMP(0);                      // (time since the previous stamp: barrier / launch-side)
    const int item = item0 + slot;                                // LEVEL: index into the work list; else: the stream
    const bool in_list = (slot < a.slots) && (item < nitems);
    const int wi = in_list ? item : item0;
    // LEVEL: a work item is (stream, first visit) of a symmetry group of work_n[item] consecutive visits (null: single visits); t_seq counts
    // the members — the slots of a workgroup step through their groups together, a slot whose group is shorter idles for the rest
    const bool wave_active = in_list && (!LEVEL || !work_n || t_seq < work_n[wi]);
    const int bb = LEVEL ? work[2 * wi] : (wave_active ? item : 0);
    const int t_level = LEVEL ? work[2 * wi + 1] + (wave_active ? t_seq : 0) : 0;
    const int b_enc = bb % a.B_enc;
    const int t = LEVEL ? t_level : t_seq;
    const int i_loc = a.order[(long)bb * a.N + t];
    const int node = bb * a.N + i_loc;                   // stream-global residue
    const int node_enc = b_enc * a.N + i_loc;
    if (tid < NAMP_SAMPLE_SLOTS) {
      const int it = item0 + tid;
      if (LEVEL) {
        const bool ok = tid < a.slots && it < nitems && (!work_n || t_seq < work_n[it]);
        const int bs = ok ? work[2 * it] : 0, ts = ok ? work[2 * it + 1] + t_seq : 0;
        node_lds[tid] = ok ? bs * a.N + a.order[(long)bs * a.N + ts] : -1;
        t_lds[tid] = ts;
      } else {
        node_lds[tid] = (tid < a.slots && it < a.B_dec) ? it * a.N + a.order[(long)it * a.N + t] : -1;
        t_lds[tid] = t;
      }
    }
    const int k = 16 * kt + m;
    const bool valid = wave_active && (k < a.K);
    const long erow = (long)node_enc * a.K + (valid ? k : 0);
    const int j_loc = a.E_idx[erow];
    const bool bw = a.rank[(long)bb * a.N + j_loc] < a.rank[(long)bb * a.N + i_loc];
    const int j_dec = bb * a.N + j_loc, j_enc = b_enc * a.N + j_loc;
    int S_j = -1;          // -1: neighbour visited but its token not drawn yet (only inside a symmetry group): h_S = 0
    if (bw) S_j = __hip_atomic_load(a.S_out + j_dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bypass L1
    const bool has_tok = bw && S_j >= 0;
    const float w_row = valid ? (1.0f / 30.0f) : 0.f;
    // mask_bw and mask_fw both carry mask_i: a masked residue sees an all-zero context [0 | 0 | 0] next to its own h_V.  (The
    // parallel decoder never notices — its output mask is the same mask_i — but here the output mask may be stream 0's.)
    const float ctx = a.mask_true[node_enc] ? 1.0f : 0.0f;

    // The layer loop is NOT unrolled (round 3).  A step executes every instruction once, so an unrolled step streams ~100 KB of
    // code through the 64 KiB instruction cache per level and runs at the speed of its instruction fetches — ~7 ns per
    // instruction, 90 us per level whatever the instructions do (profiles/r03c: neither an MFMA tail, nor faster weight staging,
    // nor one persistent launch moved it).  Rolled, the ~30 KB layer body is fetched once and re-used by the other layers — and,
    // in the persistent level walk, by every later level.  The per-layer arguments are read from the kernel-argument segment
    // with a run-time index (a by-value struct indexed at run time would be copied to scratch).
    const SampleArgs* A = (const SampleArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      // pointers read through the segment pointer are generic: tell the compiler they are global (flat loads otherwise)
      SampleLayer L = A->l[l];
      L.Z1 = as_global(L.Z1); L.W2_img = as_global(L.W2_img); L.W3_img = as_global(L.W3_img); L.b2 = as_global(L.b2);
      L.b3 = as_global(L.b3); L.tok = as_global(L.tok); L.Pfw = as_global(L.Pfw); L.Pa = as_global(L.Pa); L.Pv = as_global(L.Pv);
      {
        NodeTail& T = L.tail;
        T.hV = as_global(T.hV); T.mask = as_global(T.mask); T.ln1_g = as_global(T.ln1_g); T.ln1_b = as_global(T.ln1_b);
        T.Win_img = as_global(T.Win_img); T.b_in = as_global(T.b_in); T.Wout_img = as_global(T.Wout_img); T.b_out = as_global(T.b_out);
        T.ln2_g = as_global(T.ln2_g); T.ln2_b = as_global(T.ln2_b); T.hV_out = as_global(T.hV_out); T.S = as_global(T.S);
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          T.p[pi].img = as_global(T.p[pi].img); T.p[pi].bias = as_global(T.p[pi].bias); T.p[pi].tok = as_global(T.p[pi].tok);
          T.p[pi].out = as_global(T.p[pi].out);
        }
      }
      const float* b2p = L.b2; const float* b3p = L.b3;
      f4 x[8], acc[8];
      NAMP_STAMP(2);                    // (the layer's argument block read)
      {
        // z1 = ctx * (W1e . h_E)[i,k] + Pa[i] + ctx * (Pv | Pfw)[j] (+ tok[S_j])
        const float* src = L.Z1 + erow * NAMP_H + 4 * g;
        const float* pa = L.Pa + (long)(l == 0 ? node_enc : node) * NAMP_H + 4 * g;
        const float* pj = (bw ? L.Pv + (long)(l == 0 ? j_enc : j_dec) * NAMP_H : L.Pfw + (long)j_enc * NAMP_H) + 4 * g;
        const float* tk = L.tok + (long)(has_tok ? S_j : 0) * NAMP_H + 4 * g;
        const float tokf = has_tok ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f4 z = *(const f4*)(src + 16 * q) * ctx;
          const f4 pav = *(const f4*)(pa + 16 * q);
          f4 pjv = *(const f4*)(pj + 16 * q);
          pjv = (pjv + *(const f4*)(tk + 16 * q) * tokf) * ctx;            // (no branch: a conditional load here cost one memory round trip per q)
          acc[q] = (z + pav) + pjv;
        }
      }
      NAMP_STAMP(1);                    // index chain + row gather
      // A wave whose slot holds no residue (a level of B = 1 has ~2.5 of 4) skips the tile products — it would only take matrix-pipe
      // and issue time from the wave it shares a SIMD with — but keeps every barrier and its share of the image copies.
      if (wave_active) {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = *(const f4*)(b2p + 16 * q + 4 * g);
        SGEMM(false, true)(x, acc, w1);
      }
      NAMP_STAMP(3);                    // product 2 (W2 . gelu(z1))
      if (wave_active) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float bq = b3p[16 * q + m];
          acc[q] = (f4){bq, bq, bq, bq};
        }
        SGEMM(true, true)(acc, x, w0);
        float wr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r] = __shfl(w_row, 4 * g + r);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float sres = (acc[q].x * wr[0] + acc[q].y * wr[1]) + (acc[q].z * wr[2] + acc[q].w * wr[3]);
          sres = xg_sum(sres);
          if (g == 0) dpart[wave * NAMP_H + 16 * (q ^ slot) + m] = sres;      // (tile q at position q ^ slot: the slots' rows in different banks)
        }
      } else if (lane < 32) {
        *(f4*)(dpart + wave * NAMP_H + 4 * lane) = (f4){0.f, 0.f, 0.f, 0.f};      // (read by the tail's padding rows: keep it finite)
      }
      NAMP_STAMP(20);                   // (product 3 + K-sum done; the difference to the next stamp is this wave's barrier wait)
      __syncthreads();
      NAMP_STAMP(4);                    // product 3 + K-sum
      // both ring slots are free: the next layer's W3 / W2 (the next step's first layer in the sequential walk) are copied in
      // ahead of the residue tail, whose scratch lives behind the ring
      if (l + 1 < a.n_layers) {
        copy_to_lds<8>(buf0, as_global(A->l[l + 1].W3_img), 64, wave, nwaves, lane);
        copy_to_lds<8>(buf1, as_global(A->l[l + 1].W2_img), 64, wave, nwaves, lane);
      } else if (more) {
        copy_to_lds<8>(buf0, a.l[0].W3_img, 64, wave, nwaves, lane);
        copy_to_lds<8>(buf1, a.l[0].W2_img, 64, wave, nwaves, lane);
      }
      NAMP_STAMP(21);                   // (next images copied)
      // residue tail over the workgroup's <= 4 streams: tile row m -> stream slot m
      {
        const int nd = (m < NAMP_SAMPLE_SLOTS) ? node_lds[m] : -1;
        const int ms = nd >= 0 ? m : 0;
        const int ndc = nd >= 0 ? nd : node_lds[0];
        // h^(l) of the residue: layer 0 reads the (stream-shared) encoder output
        long hrow = ndc;
        if (l == 0) { const int bq = ndc / a.N; hrow = (long)(bq % a.B_enc) * a.N + (ndc - bq * a.N); }
        const float* hsrc = L.tail.hV + hrow * NAMP_H + 4 * g;
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = *(const f4*)(hsrc + 16 * q);
        for (int q2 = 0; q2 < a.TPN; ++q2) {
          const float* dp = dpart + (ms * a.TPN + q2) * NAMP_H + 4 * g;
#pragma unroll
          for (int q = 0; q < 8; ++q) x[q] += *(const f4*)(dp + 16 * (q ^ ms));
        }
#ifndef NAMP_ABL_SAMPLE_NOTAIL
        // split-bf16 mode, 8-wave workgroups: the tail as one MFMA tile on x3 images (both walks: level == sequential bit for bit);
        // otherwise the VALU form on fp32 images (prefetching in the level kernel, whose 256 registers hold two units in flight)
        if constexpr (X3 && MAXW == 8) node_tail_x3_rows<NAMP_SAMPLE_SLOTS, SampleRows, MODE == 2>(L.tail, x, rows, tail_lds, tid, wave, lane);
        else node_tail_rows<NAMP_SAMPLE_SLOTS, SampleRows, MODE == 2, true, LEVEL && MAXW == 8>(L.tail, x, 0.f, rows, tail_lds, tid, wave, nwaves, lane);
#endif
      }
      NAMP_STAMP(23);                   // (tail done; barrier wait follows)
      __syncthreads();            // tail outputs (h^(l+1), next layer's Pa / Pv) visible to every wave; LDS reusable
      NAMP_STAMP(5);                    // next images requested + residue tail
    }

    // ---- output head + draw, one wave per stream slot (wave n owns slot n for the whole walk, so the running
    // logit sum of a symmetry group lives in its registers).  h^(n_layers) row of slot n is yT[c * R + n].
    {
      const float* yT = tail_lds + (128 + 512 + 4 * 128) * NAMP_SAMPLE_SLOTS;
      const float* head_wT = a.head_wT; const float* head_b = a.head_b;
      for (int n = wave; n < NAMP_SAMPLE_SLOTS; n += nwaves) {
        const int nd = node_lds[n];
        if (nd < 0) continue;
        const int bq = nd / a.N, iq = nd - bq * a.N;
        const int ne = (bq % a.B_enc) * a.N + iq;
        float z = -INFINITY;
        {
          // logits: token = lane, the transposed table read 256 contiguous bytes per channel (the [vocab][128] layout made every lane walk its own
          // 512-byte row: 33 cache lines per load instruction); h_V' of the slot is broadcast from LDS.  Same four partial sums as before.
          const float* w = head_wT + lane;
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
          for (int c = 0; c < NAMP_H; c += 4) {
            s0 = fmaf(w[(c + 0) * 64], yT[(c + 0) * NAMP_SAMPLE_SLOTS + n], s0); s1 = fmaf(w[(c + 1) * 64], yT[(c + 1) * NAMP_SAMPLE_SLOTS + n], s1);
            s2 = fmaf(w[(c + 2) * 64], yT[(c + 2) * NAMP_SAMPLE_SLOTS + n], s2); s3 = fmaf(w[(c + 3) * 64], yT[(c + 3) * NAMP_SAMPLE_SLOTS + n], s3);
          }
          if (lane < a.vocab) z = (s0 + s1) + (s2 + s3) + head_b[lane];
        }
        // log_softmax(logits) of this residue                          (model_utils.py:190 / :296-297)
        const float mx = wave_max64(z);
        const float e = wave_sum64((lane < a.vocab) ? expf(z - mx) : 0.f);
        const float logp = (z - mx) - logf(e);
        if (lane < a.vocab) a.logp_out[(long)nd * a.vocab + lane] = a.chain_mask[ne] ? logp : 0.f;
        // group logit sum: total += symmetry_weight[i] * logits          (model_utils.py:298)
        const int t = t_lds[n];                          // this slot's visit (== the walk's step unless LEVEL)
        if (MODE == 2 && zbuf) {                         // deferred group draw: the member's logits wait for the level's closing pass
          if (lane < a.vocab) __hip_atomic_store(zbuf + (long)nd * a.vocab + lane, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          continue;
        }
        const long vis = (long)bq * a.N + t;
        const int v_first = a.group_first ? a.group_first[vis] : t;
        const float wsym = a.sym_w ? a.sym_w[ne] : 1.0f;
        const float zz = (lane < a.vocab) ? z : 0.f;
        tot = (v_first == t) ? wsym * zz : fmaf(wsym, zz, tot);
        const bool closes = a.group_last ? (a.group_last[vis] != 0) : true;
        if (!closes) continue;
        draw(bq, t, v_first, tot);
      }
    }
    NAMP_STAMP(24);                     // (head + draw done; barrier wait follows)
    __syncthreads();              // S of this step is published before the next step's neighbours read it
    NAMP_STAMP(6);                      // output head + draw
  };

  // LEVEL: members of the largest symmetry group among the workgroup's items item0 .. (1 without groups)
  auto members = [&](const int item0, const int nitems) {
    int n = 1;
    if (work_n)
      for (int s_ = 0; s_ < a.slots; ++s_)
        if (item0 + s_ < nitems) n = max(n, work_n[item0 + s_]);
    return n;
  };
  if constexpr (MODE == 0) {
#pragma unroll 1
    for (int t_seq = 0; t_seq < a.N; ++t_seq) step(blockIdx.x * a.slots, a.B_dec, t_seq, t_seq + 1 < a.N);
  } else if constexpr (MODE == 1) {
    if (!work_n) {
      step(blockIdx.x * a.slots, nwork, 0, false);
    } else {
      const int nmem = members(blockIdx.x * a.slots, nwork);
#pragma unroll 1
      for (int q = 0; q < nmem; ++q) step(blockIdx.x * a.slots, nwork, q, q + 1 < nmem);
    }
  } else {
    unsigned epoch = 0;
#pragma unroll 1
    for (int lvl = 0;; ++lvl) {
      const int off = level_off[lvl];
      if (off >= nwork) break;
      const int end = level_off[lvl + 1];
#pragma unroll 1
      for (int base = off + blockIdx.x * a.slots; base < end; base += gridDim.x * a.slots) {
        const int nmem = members(base, end);
#pragma unroll 1
        for (int q = 0; q < nmem; ++q) step(base, end, q, true);
      }
      grid_barrier(sync, ++epoch, tid);
      if (zbuf) {
        // the level's groups: one wave each sums its members' logits in visit order and draws
        const int c1 = close_off[lvl + 1];
#pragma unroll 1
        for (int gi = close_off[lvl] + blockIdx.x * nwaves + wave; gi < c1; gi += gridDim.x * nwaves) {
          const int bq = close_list[2 * gi], tl = close_list[2 * gi + 1];
          const int vf = a.group_first ? a.group_first[(long)bq * a.N + tl] : tl;
          float tt = 0.f;
          for (int v = vf; v <= tl; ++v) {
            const int im = a.order[(long)bq * a.N + v];
            const float zv = (lane < a.vocab) ? __hip_atomic_load(zbuf + ((long)bq * a.N + im) * a.vocab + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            const float wsym = a.sym_w ? a.sym_w[(bq % a.B_enc) * a.N + im] : 1.0f;
            tt = (v == vf) ? wsym * zv : fmaf(wsym, zv, tt);
          }
          draw(bq, tl, vf, tt);
        }
        grid_barrier(sync, ++epoch, tid);
      }
    }
    // a barrier that gave up let its workgroup run ahead of data it needed: the failure travels with the outputs (cf. encdec_persistent_kernel)
    if (__hip_atomic_load((gu32*)sync + NAMP_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      for (long e = (long)blockIdx.x * blockDim.x + tid; e < (long)a.B_dec * a.N * a.vocab; e += (long)gridDim.x * blockDim.x)
        a.logp_out[e] = __builtin_nanf("");
    }
  }
  NAMP_WSTAMP_FINI();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA of a prefetched image is in flight when the workgroup's LDS is released
}

// ------------------------------------------------------------------------------------------
// node_linear_sum_kernel — out[n] = sum_q W_q . X_q[n]: the data gradient of several residue-level linear maps of one input (dL/dx = sum_q g_q W_q,
// train._NodeLinears.backward) in ONE launch — one wave per 16-row tile runs the n products into the same accumulators; as n node_linear launches
// the sum cost n - 1 stock additions of [G,128] tensors besides.  Images as node_linear_kernel's (x3 images for X3 = 1 / 2).
// ------------------------------------------------------------------------------------------
struct NodeLinearSumArgs { const float* X[8]; const float* img[8]; float* out; int G, n; };
template <int X3>
__global__ __launch_bounds__(256) void node_linear_sum_kernel(const NodeLinearSumArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 16 >= a.G) return;
  const int m = lane & 15, g = lane >> 4;
  const int row = tile * 16 + m;
  const bool valid = row < a.G;
  const long off = (long)(valid ? row : 0) * NAMP_H + 4 * g;
  f4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q) {                                // static indices into the argument block
    if (q >= a.n) break;
    f4 x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(a.X[q] + off + 16 * t);
    if constexpr (X3) chain_gemm_global_x3<X3 == 2>(acc, x, (const bf8*)a.img[q] + lane);
    else chain_gemm_global<8, 8, false>(acc, x, (const f4*)a.img[q] + lane, 8);
  }
  if (valid) {
    float* dst = a.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
  }
}

// ------------------------------------------------------------------------------------------
// node_linear_kernel — residue-level projections  out_p[n] = W_p . X[src(n)] + bias_p (+ tok_p[S[n]])
// for up to 8 weight blocks p at once.  One wave per (16-residue tile, p); weights stream
// straight from L2 as coalesced fragment loads (each element is used once per wave, so LDS
// staging would buy nothing).  Produces the Pa / Pc / Pbw / Pfw tables, h_V = W_v.V + b
// (model_utils.py:88) and the per-token table W1s . W_s (33 rows).
// ------------------------------------------------------------------------------------------
struct NodeLinearArgs {
  const float* X;      // [G_src][128]
  const int32_t* S;    // [G_out] tokens (only if some tok != null)
  int G_out, G_src, N; // output row n reads source row ((n / N) % (G_src / N)) * N + n % N
  int nproj;
  ProjDesc pre;        // optional first stage: h = pre.img . X + pre.bias (stored to pre.out), the
                       // projections then apply to h (h_V = W_v.V + b feeding enc0's tables in one launch)
  ProjDesc p[8];
  unsigned* zero;      // optional: 64 words cleared by this launch (the grid-barrier state of the persistent launch that follows)
  __bf16* out16[8];    // optional per block: the projection ALSO (out == null: only) goes out as bf16 rows in fragment order — the tables
                       // the bf16-storage edge launches gather (edge_mlp_bf16s32_kernel); replaces a separate conversion launch
};

template <int X3>        // X3 = 1: every image is an x3 image (namp_pack_image_x3), the GEMMs run as split-bf16 products; 2: hi . hi products only (bf16 mode)
__global__ __launch_bounds__(256) void node_linear_kernel(const NodeLinearArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int unit = blockIdx.x * 4 + wave;
  const int ntiles = (a.G_out + 15) >> 4;
  if (a.zero && blockIdx.x == 0 && threadIdx.x < 64) a.zero[threadIdx.x] = 0u;
  if (unit >= ntiles * a.nproj) return;
  const int tile = unit / a.nproj, pi = unit - tile * a.nproj;
  const int m = lane & 15, g = lane >> 4;
  const int row = tile * 16 + m;
  const bool valid = row < a.G_out;
  const int rr = valid ? row : 0;
  const int b = rr / a.N;
  const int src_row = (b % (a.G_src / a.N)) * a.N + (rr - b * a.N);
  // select the descriptor without dynamic indexing of the kernarg struct
  ProjDesc d = a.p[0];
  __bf16* o16 = a.out16[0];
#pragma unroll
  for (int q = 1; q < 8; ++q) if (pi == q) { d = a.p[q]; o16 = a.out16[q]; }

  f4 x[8], acc[8];
  const float* src = a.X + (long)src_row * NAMP_H + 4 * g;
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
  if (a.pre.img) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = a.pre.bias ? *(const f4*)(a.pre.bias + 16 * t + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) chain_gemm_global_x3<X3 == 2>(acc, x, (const bf8*)a.pre.img + lane);
    else chain_gemm_global<8, 8, false>(acc, x, (const f4*)a.pre.img + lane, 8);
    if (pi == 0 && valid && a.pre.out) {
      float* dst = a.pre.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = acc[t];
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = d.bias ? *(const f4*)(d.bias + 16 * t + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
  if constexpr (X3) chain_gemm_global_x3<X3 == 2>(acc, x, (const bf8*)d.img + lane);
  else chain_gemm_global<8, 8, false>(acc, x, (const f4*)d.img + lane, 8);
  if (d.tok) {
    const float* tk = d.tok + (long)a.S[rr] * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] += *(const f4*)(tk + 16 * t);
  }
  if (valid && d.out) {
    float* dst = d.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
  }
  if (valid && o16) {                                  // fragment order B
    __bf16* dst = o16 + (long)row * NAMP_H;
#pragma unroll
    for (int t = 0; t < 8; ++t) st_frag4(dst, t, g, acc[t]);
  }
}

// ------------------------------------------------------------------------------------------
// node_update_kernel — node_tail over 16 residues per workgroup (8 waves, 2 per SIMD, up to 256 VGPRs
// so that every weight fragment of a phase is in flight at once).  The unfused path: used for large
// batches, where 16 real rows per weight pass beat the message kernels' <= 12.
//      pre = h_V + sum_t partial[n][t]      (partial already carries mask / 30)
// ------------------------------------------------------------------------------------------
struct NodeUpdateArgs {
  NodeTail t;
  const float* partial;   // [G][TPN][128] (+ [G][TPN] weight sums behind it when t.m3_img is set) or null (no message term)
  int G, TPN;
  __bf16* out16[8];       // node_update_multi_kernel only, optional per projection: bf16 fragment-order copy (out == null: only that)
};

static __global__ __launch_bounds__(512) void node_update_kernel(const NodeUpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int rr = (row0 + m < a.G) ? row0 + m : row0;
  f4 x[8];
  float wsum_m = 0.f;
  if (a.t.m3_img) {                                       // partial = K-sums of layer-2 activations + their weight sums (see NodeTail)
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < a.TPN; ++p) wsum_m += a.partial[(long)a.G * a.TPN * NAMP_H + (long)rr * a.TPN + p];
  } else {
    const float* src = a.t.hV + (long)rr * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
  }
  if (a.partial) {
    for (int p = 0; p < a.TPN; ++p) {
      const float* ps_ = a.partial + ((long)rr * a.TPN + p) * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] += *(const f4*)(ps_ + 16 * t);
    }
  }
  node_tail<true>(a.t, x, wsum_m, row0, 16, a.G, (float*)smem, tid, wave, 8, lane);
}

// node_update_multi_kernel<T> — the same residue update for T 16-row tiles per workgroup (large batches).  With one tile
// per workgroup every 16 residues re-stream the 768 KiB of FFN + projection weights from L2 (3 GB per launch at
// B*N = 64,000: the launch is L2-bandwidth-bound).  Here each wave's weight fragments are requested once per phase and
// applied to all T tiles: W_in (its 64 hidden units) against the T LayerNorm-1 tiles kept in LDS, then W_out, then each
// projection fragment against the T output tiles.  Arithmetic per row is that of node_tail<true>.
#define NODE_MULTI_LDS(T) (((2 * (T) * 16) + 8 * 16) * FFN_LD * 4)
// X3 form: two more [T][16] row sets hold the LayerNorm-1 / h_V' rows already SPLIT into bf16 hi | mid fragment pieces (256 + 256 B
// per row, the row pitch of the fp32 tiles) — every wave used to re-split the same fp32 rows for each of its GEMM phases (and for
// each projection block): 576 of ~1,000 VALU instructions per tile and wave in a kernel that is instruction-issue bound.
#define NODE_MULTI_LDS_X3(T) (((4 * (T) * 16) + 8 * 16) * FFN_LD * 4)

// X3: Win_img / Wout_img / every projection image are x3 images (pack_image_x3_general_kernel / pack_image_x3_kernel) and
// the three GEMM phases run as split-bf16 products: 144 bf16 MFMAs per tile instead of 384 fp32 MFMAs (5.3x fewer cycles).
template <int T, int X3 = 0>       // X3: 0 exact fp32 MFMA, 1 split-bf16 products, 2 hi . hi products of the same images (bf16 mode)
__global__ __launch_bounds__(512) void node_update_multi_kernel(const NodeUpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = (float*)smem;                       // [T][16][FFN_LD]  LN1 outputs
  float* ys = xs + T * 16 * FFN_LD;               // [T][16][FFN_LD]  h_V' tiles
  float* ps = ys + T * 16 * FFN_LD;               // [8][16][FFN_LD]  per-wave partial FFN outputs of the tile in flight
  // X3: split copies of the xs / ys rows: piece (s, g) of a row = the 8 bf16 lane (m, g) feeds into MFMA step s; hi at byte 0, mid at 256
  char* xsp = (char*)(ps + 8 * 16 * FFN_LD);      // [T][16][FFN_LD * 4 bytes]
  char* ysp = xsp + T * 16 * FFN_LD * 4;
  auto put_split = [&](char* base, const int row, const f4 (&v)[8], const int g_) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      bf8 hi, mid;
      split_x3(v[2 * s_], v[2 * s_ + 1], hi, mid);
      *(bf8*)(base + row * (FFN_LD * 4) + 16 * (4 * s_ + g_)) = hi;
      if constexpr (X3 != 2) *(bf8*)(base + row * (FFN_LD * 4) + 256 + 16 * (4 * s_ + g_)) = mid;
    }
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16 * T;
  const NodeTail& t = a.t;
  // ---- hoisted message layer 3 (see NodeTail): K-sums -> xs, W3 . sums + b3 * wsum -> ys (wave w: channel tile w of every tile)
  if (t.m3_img) {
    for (int q = wave; q < T; q += 8) {
      const int row = row0 + 16 * q + m;
      const int rr = row < a.G ? row : (a.G - 1);
      f4 x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = (f4){0.f, 0.f, 0.f, 0.f};
      float ws = 0.f;
      for (int p = 0; p < a.TPN; ++p) {
        const float* ps_ = a.partial + ((long)rr * a.TPN + p) * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] += *(const f4*)(ps_ + 16 * c);
        ws += a.partial[(long)a.G * a.TPN * NAMP_H + (long)rr * a.TPN + p];
      }
      if constexpr (X3) put_split(xsp, q * 16 + m, x, g);
      else {
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f4*)(xs + (q * 16 + m) * FFN_LD + 16 * c + 4 * g) = x[c];
      }
      if (g == 0) ps[q * 16 + m] = ws;
    }
    __syncthreads();
    const f4 b3v = *(const f4*)(t.m3_b + 16 * wave + 4 * g);
    if constexpr (X3) {
      bf8 wh[4], wm[4];
      const bf8* w3 = (const bf8*)t.m3_img + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s) { wh[s] = w3[(s * 8 + wave) * 64]; wm[s] = w3[NAMP_BIMG_BYTES / 16 + (s * 8 + wave) * 64]; }
#pragma unroll
      for (int q = 0; q < T; ++q) {
        f4 o = b3v * ps[q * 16 + m];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const char* xr = xsp + (q * 16 + m) * (FFN_LD * 4) + 16 * (4 * s + g);
          o = mfma_xs<X3 == 2>(wh[s], wm[s], *(const bf8*)xr, *(const bf8*)(xr + 256), o);
        }
        *(f4*)(ys + (q * 16 + m) * FFN_LD + 16 * wave + 4 * g) = o;
      }
    } else {
      f4 wf[8];
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) wf[tk] = ((const f4*)t.m3_img)[(tk * 8 + wave) * 64 + lane];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        f4 o = b3v * ps[q * 16 + m];
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) {
          const f4 xv = *(const f4*)(xs + (q * 16 + m) * FFN_LD + 16 * tk + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) o = mfma4(wf[tk][r], xv[r], o);
        }
        *(f4*)(ys + (q * 16 + m) * FFN_LD + 16 * wave + 4 * g) = o;
      }
    }
    __syncthreads();
  }
  // ---- phase 0: pre-activation + LayerNorm1, one wave per tile
  for (int q = wave; q < T; q += 8) {
    const int row = row0 + 16 * q + m;
    const int rr = row < a.G ? row : (a.G - 1);
    f4 x[8];
    const float* src = t.hV + (long)rr * NAMP_H + 4 * g;
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = *(const f4*)(src + 16 * c);
    if (t.m3_img) {
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] += *(const f4*)(ys + (q * 16 + m) * FFN_LD + 16 * c + 4 * g);
    } else if (a.partial) {
      for (int p = 0; p < a.TPN; ++p) {
        const float* ps_ = a.partial + ((long)rr * a.TPN + p) * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] += *(const f4*)(ps_ + 16 * c);
      }
    }
    layernorm_row_T(x, t.ln1_g, t.ln1_b, g);
#pragma unroll
    for (int c = 0; c < 8; ++c) *(f4*)(xs + (q * 16 + m) * FFN_LD + 16 * c + 4 * g) = x[c];
    if constexpr (X3) put_split(xsp, q * 16 + m, x, g);
  }
  __syncthreads();
  // ---- phase A: hidden = gelu(W_in x + b_in); wave w owns hidden units 64w .. 64w+63
  f4 hacc[T][4];
  if constexpr (X3) {
    bf8 wh[4][4], wm[4][4];
    const bf8* w = (const bf8*)t.Win_img + (4 * wave) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) { wh[s][tn] = w[(s * 32 + tn) * 64]; wm[s][tn] = w[512 * 128 / 8 + (s * 32 + tn) * 64]; }
#pragma unroll
    for (int q = 0; q < T; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) hacc[q][c] = *(const f4*)(t.b_in + 64 * wave + 16 * c + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const char* xr = xsp + (q * 16 + m) * (FFN_LD * 4) + 16 * (4 * s + g);
        const bf8 hi = *(const bf8*)xr, mid = *(const bf8*)(xr + 256);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) hacc[q][tn] = mfma_xs<X3 == 2>(wh[s][tn], wm[s][tn], hi, mid, hacc[q][tn]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) hacc[q][c] = gelu4(hacc[q][c]);
    }
  } else {
    f4 win[8][4];
    const f4* w = (const f4*)t.Win_img + (4 * wave) * 64 + lane;
#pragma unroll
    for (int tk = 0; tk < 8; ++tk)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) win[tk][tn] = w[(tk * 32 + tn) * 64];
#pragma unroll
    for (int q = 0; q < T; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) hacc[q][c] = *(const f4*)(t.b_in + 64 * wave + 16 * c + 4 * g);
#pragma unroll
      for (int tk = 0; tk < 8; ++tk) {
        const f4 xv = *(const f4*)(xs + (q * 16 + m) * FFN_LD + 16 * tk + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) hacc[q][tn] = mfma4(win[tk][tn][r], xv[r], hacc[q][tn]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) hacc[q][c] = gelu4(hacc[q][c]);
    }
  }
  // ---- phase B: W_out, then LayerNorm2 per tile
  {
    f4 wo[X3 ? 1 : 4][8];
    bf8 woh[X3 ? 2 : 1][8], wom[X3 ? 2 : 1][8];
    if constexpr (X3) {
      const bf8* w = (const bf8*)t.Wout_img + (2 * wave) * 8 * 64 + lane;       // K-steps 2*wave, 2*wave+1: this wave's 64 hidden units
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) { woh[s][tn] = w[(s * 8 + tn) * 64]; wom[s][tn] = w[128 * 512 / 8 + (s * 8 + tn) * 64]; }
    } else {
      const f4* w = (const f4*)t.Wout_img + (4 * wave) * 8 * 64 + lane;
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) wo[X3 ? 0 : tk][tn] = w[(tk * 8 + tn) * 64];
    }
#pragma unroll
    for (int q = 0; q < T; ++q) {
      f4 oacc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) oacc[c] = (f4){0.f, 0.f, 0.f, 0.f};
      if constexpr (X3) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf8 hi, mid;
          split_x3(hacc[q][2 * s], hacc[q][2 * s + 1], hi, mid);
#pragma unroll
          for (int tn = 0; tn < 8; ++tn) oacc[tn] = mfma_xs<X3 == 2>(woh[s][tn], wom[s][tn], hi, mid, oacc[tn]);
        }
      } else {
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tn = 0; tn < 8; ++tn) oacc[tn] = mfma4(wo[X3 ? 0 : tk][tn][r], hacc[q][tk][r], oacc[tn]);
      }
      float* dst = ps + (wave * 16 + m) * FFN_LD + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = oacc[c];
      __syncthreads();
      {   // LN2: thread -> (row = tid/32, 4 channels)
        const int r = tid >> 5, c = (tid & 31) * 4;
        f4 v = *(const f4*)(xs + (q * 16 + r) * FFN_LD + c) + *(const f4*)(t.b_out + c);
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) v += *(const f4*)(ps + (w2 * 16 + r) * FFN_LD + c);
        float s_ = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s_ += __shfl_xor(s_, o);
        const float mean = s_ * (1.0f / 128.0f);
        v -= mean;
        float qq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) qq += __shfl_xor(qq, o);
        const float rstd = rsqrtf(qq * (1.0f / 128.0f) + 1e-5f);
        const int orow = row0 + 16 * q + r;
        const bool ok = orow < a.G;
        const float mk = (t.mask && ok) ? (float)t.mask[orow] : 1.0f;
        const f4 y = (v * rstd * *(const f4*)(t.ln2_g + c) + *(const f4*)(t.ln2_b + c)) * mk;
        *(f4*)(ys + (q * 16 + r) * FFN_LD + c) = y;
        if constexpr (X3) {
          // channels c..c+3 = elements 4 (tile & 1) .. + 3 of piece (s = tile / 2, g = (c % 16) / 4): half a piece per thread
          typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
          const int tile_ = c >> 4, g_ = (c & 15) >> 2;
          const bf4 hi = (bf4){(__bf16)y.x, (__bf16)y.y, (__bf16)y.z, (__bf16)y.w};
          const bf4 mid = (bf4){(__bf16)(y.x - (float)hi[0]), (__bf16)(y.y - (float)hi[1]), (__bf16)(y.z - (float)hi[2]),
                                (__bf16)(y.w - (float)hi[3])};
          char* yp = ysp + (q * 16 + r) * (FFN_LD * 4) + 16 * (4 * (tile_ >> 1) + g_) + 8 * (tile_ & 1);
          *(bf4*)yp = hi;
          if constexpr (X3 != 2) *(bf4*)(yp + 256) = mid;
        }
        if (ok) *(f4*)(t.hV_out + (long)orow * NAMP_H + c) = y;
      }
      __syncthreads();                                        // ps is free for the next tile; ys[q] is complete
    }
  }
  if (t.nproj == 0 && !t.head_w) return;
  if (t.head_w) {
    for (int n = wave; n < 16 * T; n += 8)
      if (row0 + n < a.G) tail_head_row(t, ys + n * FFN_LD, 1, row0 + n, lane);
  }
  // ---- projections of h_V': unit (block pi, channel tile tn) -> one wave; its fragment serves all T tiles
#pragma unroll
  for (int pi = 0; pi < 8; ++pi) {
    if (pi >= t.nproj) break;
    const ProjDesc d = t.p[pi];
    {
      const int tn = wave;                                    // 8 waves <-> 8 channel tiles
      f4 wf[X3 ? 1 : 8];
      bf8 wfh[X3 ? 4 : 1], wfm[X3 ? 4 : 1];
      if constexpr (X3) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wfh[s] = ((const bf8*)d.img)[(s * 8 + tn) * 64 + lane];
          wfm[s] = ((const bf8*)d.img)[NAMP_BIMG_BYTES / 16 + (s * 8 + tn) * 64 + lane];
        }
      } else {
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) wf[X3 ? 0 : tk] = ((const f4*)d.img)[(tk * 8 + tn) * 64 + lane];
      }
      const f4 bias = d.bias ? *(const f4*)(d.bias + 16 * tn + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int row = row0 + 16 * q + m;
        const bool valid = row < a.G;
        const int rr = valid ? row : (a.G - 1);
        f4 acc = bias;
        if (d.tok) acc += *(const f4*)(d.tok + (long)t.S[rr] * NAMP_H + 16 * tn + 4 * g);
        if constexpr (X3) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const char* yr = ysp + (q * 16 + m) * (FFN_LD * 4) + 16 * (4 * s + g);
            acc = mfma_xs<X3 == 2>(wfh[s], wfm[s], *(const bf8*)yr, *(const bf8*)(yr + 256), acc);
          }
        } else {
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) {
          const f4 xv = *(const f4*)(ys + (q * 16 + m) * FFN_LD + 16 * tk + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma4(wf[X3 ? 0 : tk][r], xv[r], acc);
        }
        }
        if (valid && d.out) *(f4*)(d.out + (long)row * NAMP_H + 16 * tn + 4 * g) = acc;
        if (valid && a.out16[pi]) st_frag4(a.out16[pi] + (long)row * NAMP_H, tn, g, acc);   // fragment order B
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// dec_ctx_message_kernel — DecLayer.forward's message on a MATERIALISED context (model_utils.py:636-646): the
// operator form the reference's sampler and score() call, h_ESV [G][K][384] = [h_E | h_S_j | h_V_j] given by the
// caller.  First layer = Pa[i] (W1a . h_V_i + b1, hoisted) + the three 128-wide blocks of the row against
// W1e / W1s / W1v; exact fp32 MFMA, weights streamed from L2 (this is the drop-in operator, not the hot path: the
// model never materialises h_ESV).  One wave per 16-row tile; writes per-tile partial sums for node_update.
// ------------------------------------------------------------------------------------------
struct DecCtxArgs {
  const float* ctx;            // [G][K][384]
  const float* mask_attend;    // optional [G][K] (float, like the reference's argument)
  const float* Pa;             // [G][128]
  const float* W1e_img; const float* W1s_img; const float* W1v_img; const float* W2_img; const float* W3_img;
  const float* b2; const float* b3;
  float* partial;              // [G][TPN][128]
  int G, K, TPN;
};

static __global__ __launch_bounds__(256) void dec_ctx_message_kernel(const DecCtxArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile >= (long)a.G * a.TPN) return;
  const int node = (int)(tile / a.TPN), kt = (int)(tile - (long)node * a.TPN);
  const int m = lane & 15, g = lane >> 4;
  const int k = 16 * kt + m;
  const bool valid = k < a.K;
  const long erow = (long)node * a.K + (valid ? k : 0);
  f4 x[8], acc[8];
  {
    const float* pa = a.Pa + (long)node * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(pa + 16 * t);
  }
#pragma unroll 1
  for (int blk = 0; blk < 3; ++blk) {
    const float* src = a.ctx + erow * 384 + 128 * blk + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
    const float* img = blk == 0 ? a.W1e_img : blk == 1 ? a.W1s_img : a.W1v_img;
    chain_gemm_global<8, 8, false>(acc, x, (const f4*)img + lane, 8);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) { acc[t] = gelu4(acc[t]); x[t] = *(const f4*)(a.b2 + 16 * t + 4 * g); }
  chain_gemm_global<8, 8, false>(x, acc, (const f4*)a.W2_img + lane, 8);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    x[t] = gelu4(x[t]);
    const float b = a.b3[16 * t + m];
    acc[t] = (f4){b, b, b, b};
  }
  chain_gemm_global<8, 8, true>(acc, x, (const f4*)a.W3_img + lane, 8);
  const float w_row = valid ? (a.mask_attend ? a.mask_attend[erow] : 1.0f) * (1.0f / 30.0f) : 0.f;
  float wr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wr[r] = __shfl(w_row, 4 * g + r);
  float* dst = a.partial + ((long)node * a.TPN + kt) * NAMP_H + m;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float s_ = (acc[t].x * wr[0] + acc[t].y * wr[1]) + (acc[t].z * wr[2] + acc[t].w * wr[3]);
    s_ = xg_sum(s_);
    if (g == 0) dst[16 * t] = s_;
  }
}

// ------------------------------------------------------------------------------------------
// logits_kernel — log_softmax(W_out . h_V + b) over the 33-letter vocabulary
// (model_utils.py:420-421).  One wave per residue; lane t < V owns logit t.  W_out is staged once per workgroup in LDS as
// [channel / 4][token] 16-byte pieces, so the 32 reads of a dot product are conflict-free ds_read_b128 (read straight from
// global memory, lane t walks row t: 33 cache lines per load instruction — 195 us for 64,000 residues, now 1/8 of that);
// a workgroup serves `per_wg` residues.  Same summation order as before: results unchanged.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ hV, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ log_probs,
                                                     float* __restrict__ logits_out, int G, int V, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* wl = (f4*)smem;                                                      // [32][V]
  const int lane = threadIdx.x & 63;
  for (int idx = threadIdx.x; idx < 32 * V; idx += 256) {
    const int c4 = idx / V, t = idx - c4 * V;
    wl[idx] = *(const f4*)(W + (long)t * NAMP_H + 4 * c4);
  }
  __syncthreads();
  const int first = blockIdx.x * per_wg, last = min(G, first + per_wg);
  const float b = lane < V ? bias[lane] : 0.f;
  const f4* wt = wl + (lane < V ? lane : 0);
  // four residues per wave and pass: one W_out fragment read serves four dot products (per-residue summation order unchanged)
  for (int n0 = first + 4 * (threadIdx.x >> 6); n0 < last; n0 += 16) {
    const float* h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = hV + (long)min(n0 + r, last - 1) * NAMP_H;
    f4 s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c4 = 0; c4 < NAMP_H / 4; ++c4) {
      const f4 wv = wt[c4 * V];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f4 hv = *(const f4*)(h[r] + 4 * c4);
        s[r].x = fmaf(wv.x, hv.x, s[r].x); s[r].y = fmaf(wv.y, hv.y, s[r].y);
        s[r].z = fmaf(wv.z, hv.z, s[r].z); s[r].w = fmaf(wv.w, hv.w, s[r].w);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int node = n0 + r;
      const float z = lane < V ? (s[r].x + s[r].y) + (s[r].z + s[r].w) + b : -INFINITY;
      float mx = z;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float e = (lane < V) ? expf(z - mx) : 0.f;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) e += __shfl_xor(e, o);
      if (lane < V && node < last) {
        log_probs[(long)node * V + lane] = (z - mx) - logf(e);
        if (logits_out) logits_out[(long)node * V + lane] = z;
      }
    }
  }
}

// ==========================================================================================
// a11 — graph construction + edge featurisation (ProteinFeaturesNA.forward, model_utils.py:528-593)
// ==========================================================================================

// prep_atoms_kernel: per residue, the 18-atom frame (16 real + virtual Cb + virtual N_na, model_utils.py:548-569),
// its 0/1 atom mask as a bit field, and the kNN reference point P = CA + C1' (model_utils.py:573).
// Atom order is the reference's atom_dict (run.py:15-19): N, CA, C, O, OP1, OP2, P, O5', C5', C4', O4', C3', O3', C2', O2', C1'.
static __global__ void prep_atoms_kernel(const float* __restrict__ X, const int32_t* __restrict__ X_m,
                                  const int32_t* __restrict__ protein_mask, const int32_t* __restrict__ dna_mask,
                                  const int32_t* __restrict__ rna_mask, float* __restrict__ X18,
                                  uint32_t* __restrict__ M18, float* __restrict__ P, int G, int ref_atom) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= G) return;
  const float* x = X + (long)n * 48;
  float* o = X18 + (long)n * 54;
  uint32_t bits = 0;
  for (int a = 0; a < 16; ++a) {
    o[3 * a] = x[3 * a]; o[3 * a + 1] = x[3 * a + 1]; o[3 * a + 2] = x[3 * a + 2];
    if (X_m[(long)n * 16 + a]) bits |= 1u << a;
  }
  auto virt = [&](int i0, int i1, int i2, float wa, float wb, float wc, float* dst) {      // get_Cb, model_utils.py:521-526
    const float bx = x[3 * i1] - x[3 * i0], by = x[3 * i1 + 1] - x[3 * i0 + 1], bz = x[3 * i1 + 2] - x[3 * i0 + 2];
    const float cx = x[3 * i2] - x[3 * i1], cy = x[3 * i2 + 1] - x[3 * i1 + 1], cz = x[3 * i2 + 2] - x[3 * i1 + 2];
    const float ax = by * cz - bz * cy, ay = bz * cx - bx * cz, az = bx * cy - by * cx;
    dst[0] = wa * ax + wb * bx + wc * cx + x[3 * i1];
    dst[1] = wa * ay + wb * by + wc * cy + x[3 * i1 + 1];
    dst[2] = wa * az + wb * bz + wc * cz + x[3 * i1 + 2];
  };
  virt(0, 1, 2, -0.58273431f, 0.56802827f, -0.54067466f, o + 48);        // Cb from N, CA, C
  virt(10, 15, 13, -0.56967352f, 0.51055973f, -0.53122153f, o + 51);     // N_na from O4', C1', C2'
  if (protein_mask[n]) bits |= 1u << 16;
  if (rna_mask[n] + dna_mask[n]) bits |= 1u << 17;
  M18[n] = bits;
  P[3 * n] = x[3] + x[3 * ref_atom]; P[3 * n + 1] = x[4] + x[3 * ref_atom + 1]; P[3 * n + 2] = x[5] + x[3 * ref_atom + 2];
}

// Round 6 — the edge-feature launch of one complex in parts (edge_features_kernel below): its residue blocks (npw consecutive residues, one
// workgroup's worth; at most 256 of them) ranked longest first by the number of the 18 atoms present in their residues, ties by index.
// order[r] = block of rank r, order[256] = number of long blocks (FEAT_LONG_ATOMS or more atoms: a protein residue has 5, a nucleotide
// 12-13).  Runs as one extra workgroup of the neighbour-search launch (blockIdx.x == gridDim.x - 1), beside it.
#define FEAT_LONG_ATOMS 8
struct FeatRank { const uint32_t* M18; int32_t* order; int G, npw, nblk; };

static __device__ __forceinline__ void feat_rank_blocks(const FeatRank r, char* smem) {
  const int tid = threadIdx.x;                                       // 256 threads
  uint16_t* ck = (uint16_t*)smem;                                    // composite = atoms << 8 | 255 - block: unique, 0 for no block
  int* cnt = (int*)(smem + 512);
  if (tid == 0) cnt[0] = 0;
  uint32_t u = 0;
  if (tid < r.nblk)
    for (int q = 0; q < r.npw; ++q) { const int nd = tid * r.npw + q; if (nd < r.G) u |= r.M18[nd]; }
  const uint32_t cb = tid < r.nblk ? (uint32_t)((__popc(u) << 8) | (255 - tid)) : 0u;
  ck[tid] = (uint16_t)cb;
  __syncthreads();
  const unsigned long long longs = __ballot(cb >= (uint32_t)(FEAT_LONG_ATOMS << 8));
  if ((tid & 63) == 0) atomicAdd(cnt, __popcll(longs));
  int rank = 0;
  for (int i = 0; i < 32; ++i) {
    const uint4 q = ((const uint4*)smem)[i];
    const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int z = 0; z < 4; ++z) rank += ((wd[z] & 0xffffu) > cb ? 1 : 0) + ((wd[z] >> 16) > cb ? 1 : 0);
  }
  if (tid < r.nblk) r.order[rank] = tid;
  __syncthreads();
  if (tid == 0) r.order[256] = cnt[0];
}

// knn_kernel: _dist (model_utils.py:489-497).  One workgroup per residue row: masked distances to all L
// residues, row maximum, then a bitonic sort of 64-bit keys (distance bits << 32 | index) in LDS and the K
// smallest are written in ascending order (ties: unmasked before masked, then by index; torch.topk leaves them unspecified).
// The distance uses the reference's operation order with contraction off so that near-ties round identically.
static __global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ P, const int32_t* __restrict__ mask,
                                                  int32_t* __restrict__ E_idx, int L, int Lp2, int K, const FeatRank fr) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (fr.order && blockIdx.x == gridDim.x - 1) { feat_rank_blocks(fr, smem); return; }
  unsigned long long* keys = (unsigned long long*)smem;            // [Lp2]
  float* red = (float*)(keys + Lp2);                               // [4]
  const int row = blockIdx.x, tid = threadIdx.x;
  const int b = row / L;
  const float* Pb = P + (long)b * L * 3;
  const int32_t* mb = mask + (long)b * L;
  const float px = P[3 * (long)row], py = P[3 * (long)row + 1], pz = P[3 * (long)row + 2];
  const float mi = (float)mask[row];
  float dmax = 0.f;
  for (int j = tid; j < L; j += 256) {
    const float dx = px - Pb[3 * j], dy = py - Pb[3 * j + 1], dz = pz - Pb[3 * j + 2];
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    const float d = (mi * (float)mb[j]) * sqrtf(s + 1e-6f);
    dmax = fmaxf(dmax, d);
    keys[j] = (unsigned long long)__float_as_uint(d) << 32;       // provisional: distance only
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
  if ((tid & 63) == 0) red[tid >> 6] = dmax;
  __syncthreads();
  dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int j = tid; j < Lp2; j += 256) {
    if (j < L) {
      const float m2 = mi * (float)mb[j];
      const float d = __uint_as_float((unsigned)(keys[j] >> 32)) + (1.0f - m2) * dmax;
      // ties: lower index first, but a real distance beats the "masked -> row maximum" substitute it equals (the farthest
      // unmasked residue always ties with every masked / padded one; torch.topk leaves the choice open — keeping the real
      // residue is the deterministic, padding-independent answer)
      keys[j] = ((unsigned long long)__float_as_uint(d) << 32) | (m2 == 0.f ? 0x80000000ull : 0ull) | (unsigned)j;
    } else {
      keys[j] = ~0ull;
    }
  }
  __syncthreads();
  for (int k = 2; k <= Lp2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int idx = tid; idx < Lp2; idx += 256) {
        const int ixj = idx ^ j;
        if (ixj > idx) {
          const unsigned long long a = keys[idx], c = keys[ixj];
          const bool up = (idx & k) == 0;
          if ((a > c) == up) { keys[idx] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int k = tid; k < K; k += 256) E_idx[(long)row * K + k] = (int32_t)(keys[k] & 0x7fffffffu);
}

// knn_select_kernel: the same neighbour lists without sorting the whole row.  The K smallest of L unique 64-bit keys are
// found by a most-significant-digit radix SELECT (8 bits per pass: histogram of the digit over the keys that still match
// the prefix, scan, descend into the bin that holds the K-th key) which stops as soon as "keys below the bin + keys in the
// bin" fit the final sort (Kp2 = K rounded up to a power of two >= 64: 2-3 passes on real distance distributions, 8 at
// most); those <= Kp2 candidates are compacted, sorted (one wave with cross-lane exchanges when Kp2 = 64, the bitonic
// network in LDS otherwise) and the first K written.  Keys are unique (index in the low bits), so the result is the
// sorted row's prefix exactly.  O(L) per pass instead of O(L log^2 L): 13x fewer LDS passes at L = 3000.
static __global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ P, const int32_t* __restrict__ mask,
                                                         int32_t* __restrict__ E_idx, int L, int K, int Kp2, const FeatRank fr) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (fr.order && blockIdx.x == gridDim.x - 1) { feat_rank_blocks(fr, smem); return; }
  unsigned long long* keys = (unsigned long long*)smem;            // [L]
  unsigned long long* sel = keys + L;                              // [Kp2]
  uint32_t* hist = (uint32_t*)(sel + Kp2);                         // [256]
  uint32_t* misc = hist + 256;                                     // [0..3] wave totals, [4] bin, [5] below, [6] bin count, [7] cursor
  float* red = (float*)(misc + 8);                                 // [4]
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = row / L;
  const float* Pb = P + (long)b * L * 3;
  const int32_t* mb = mask + (long)b * L;
  const float px = P[3 * (long)row], py = P[3 * (long)row + 1], pz = P[3 * (long)row + 2];
  const float mi = (float)mask[row];
  float dmax = 0.f;
  for (int j = tid; j < L; j += 256) {
    const float dx = px - Pb[3 * j], dy = py - Pb[3 * j + 1], dz = pz - Pb[3 * j + 2];
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    const float d = (mi * (float)mb[j]) * sqrtf(s + 1e-6f);
    dmax = fmaxf(dmax, d);
    keys[j] = (unsigned long long)__float_as_uint(d) << 32;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
  if (lane == 0) red[wave] = dmax;
  __syncthreads();
  dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int j = tid; j < L; j += 256) {
    const float m2 = mi * (float)mb[j];
    const float d = __uint_as_float((unsigned)(keys[j] >> 32)) + (1.0f - m2) * dmax;
    keys[j] = ((unsigned long long)__float_as_uint(d) << 32) | (m2 == 0.f ? 0x80000000ull : 0ull) | (unsigned)j;   // ties: see knn_kernel
  }
  // ---- radix select
  unsigned long long prefix = 0;          // digits fixed so far (bits above `shift + 8` after the pass at `shift`)
  int remaining = K;                      // rank of the K-th key among the keys matching the prefix
  int below = 0;                          // keys strictly below the prefix: all selected
  int shift = 64;
  while (shift > 0) {
    shift -= 8;
    hist[tid] = 0;
    __syncthreads();
    for (int j = tid; j < L; j += 256) {
      const unsigned long long key = keys[j];
      if (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    const uint32_t c = hist[tid];
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) misc[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += misc[w];
    const uint32_t excl = incl - c;
    if (excl < (uint32_t)remaining && (uint32_t)remaining <= incl) { misc[4] = tid; misc[5] = excl; misc[6] = c; }
    __syncthreads();
    const uint32_t bin = misc[4], ex = misc[5], cnt = misc[6];
    prefix |= (unsigned long long)bin << shift;
    below += (int)ex;
    remaining -= (int)ex;
    if (below + (int)cnt <= Kp2) break;   // everything below the bin plus the whole bin fits the final sort
  }
  // ---- compact the candidates: keys whose digits down to `shift` are <= the prefix's  (below + bin count of them, >= K)
  if (tid == 0) misc[7] = 0;
  for (int j = tid; j < Kp2; j += 256) sel[j] = ~0ull;
  __syncthreads();
  for (int j = tid; j < L; j += 256) {
    const unsigned long long key = keys[j];
    if ((key >> shift) <= (prefix >> shift)) sel[atomicAdd(&misc[7], 1u)] = key;
  }
  __syncthreads();
  if (Kp2 == 64) {
    if (wave == 0) {
      unsigned long long v = sel[lane];
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const unsigned lo = __shfl_xor((unsigned)v, j), hi = __shfl_xor((unsigned)(v >> 32), j);
          const unsigned long long o = ((unsigned long long)hi << 32) | lo;
          const bool up = (lane & k) == 0, lower = (lane & j) == 0;
          const bool keep_min = (up == lower);
          v = keep_min ? (v < o ? v : o) : (v > o ? v : o);
        }
      }
      if (lane < K) E_idx[(long)row * K + lane] = (int32_t)(v & 0x7fffffffu);
    }
    return;
  }
  for (int k = 2; k <= Kp2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int idx = tid; idx < Kp2; idx += 256) {
        const int ixj = idx ^ j;
        if (ixj > idx) {
          const unsigned long long a = sel[idx], c2 = sel[ixj];
          const bool up = (idx & k) == 0;
          if ((a > c2) == up) { sel[idx] = c2; sel[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int k = tid; k < K; k += 256) E_idx[(long)row * K + k] = (int32_t)(sel[k] & 0x7fffffffu);
}

// ---- decoding order on the device (round 6; namp_order.h's kernel and, folded into the featuriser's edge-feature launch, its first workgroups) ----
// order[b] = argsort((mask * chain_mask + 1e-4) * |randn|) and its inverse permutation (model_utils.py:389-390; na_model_utils.py:623): one workgroup
// per stream, bitonic sort of (key, index) pairs in LDS (ascending key, ties by index; a NaN key sorts last).  The keys are the same fp32 operations
// torch runs.  Any workgroup size that is a multiple of 64; P2 * 8 bytes of LDS.
struct OrderJob {
  const float* mask; const float* chain_mask; const float* randn;    // [B_mask][L], [B_mask][L] or null, [B][L]
  int64_t* order64; int32_t* order32; int32_t* rank32;               // [B][L]; either order may be null
  int B, B_mask, L, P2;                                              // B == 0: no job
};
static __device__ __forceinline__ void decoding_order_body(const OrderJob o, const int b, char* smem) {
  float* key = (float*)smem;
  int* idx = (int*)(smem + (size_t)o.P2 * 4);
  const int bm = b % o.B_mask, L = o.L, P2 = o.P2;
  for (int i = threadIdx.x; i < P2; i += blockDim.x) {
    float k = __builtin_inff();
    if (i < L) {
      const float cm = o.mask[(long)bm * L + i] * (o.chain_mask ? o.chain_mask[(long)bm * L + i] : 1.0f);
      k = (cm + 0.0001f) * fabsf(o.randn[(long)b * L + i]);
    }
    key[i] = k; idx[i] = i;
  }
  __syncthreads();
  // a before b: smaller key; NaN after everything; ties (and NaN pairs) by index; padding entries (index >= L) carry +inf and larger indices
  auto before = [](const float ka, const int ia, const float kb, const int ib) {
    const bool na = ka != ka, nb = kb != kb;
    if (na != nb) return nb;
    if (!na && ka != kb) return ka < kb;
    return ia < ib;
  };
  // Compare-exchanges of stride <= 64 stay inside one wave's 128-entry block (thread t of a wave handles pair t of ITS block), and a wave's LDS
  // accesses execute in program order: those sub-stages need no workgroup barrier, only the larger strides do (15 barriers instead of 78 at 4,096
  // entries; none below 256) — the barriers were half of the launch (20 us at L = 1,000, 46 us at a 13 x 2,400 batch).
  for (int size = 2; size <= P2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (P2 >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));            // index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const float ka = key[lo], kb = key[hi];
        const int ia = idx[lo], ib = idx[hi];
        if (before(kb, ib, ka, ia) == up) { key[lo] = kb; key[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
      }
      if (stride > 64 || (stride == 1 && size < P2 && size >= 64)) __syncthreads();     // next sub-stage crosses waves (or the next stage opens with a stride > 64)
      else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const int v = idx[i];
    if (o.order64) o.order64[(long)b * L + i] = v;
    if (o.order32) o.order32[(long)b * L + i] = v;
    o.rank32[(long)b * L + v] = i;
  }
}

#ifdef FEAT_STAMPS
// -DFEAT_STAMPS (tools/feat_stamps.py): per workgroup of the last edge_features launch: s_memrealtime (100 MHz, one clock for the chip) at entry, behind the set-up, at the end; residue
// block, part, chunks walked, HW_ID
#define GETREG_IMMED(SZ, OFF, REG) (((SZ) << 11) | ((OFF) << 6) | (REG))
__device__ unsigned long long g_feat_stamps[1024][8];
extern "C" int namp_debug_feat_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_feat_stamps), sizeof(g_feat_stamps), 0, hipMemcpyDeviceToHost);
}
#endif

// edge_features_kernel: RBF + positional features -> edge_embedding (5200 -> 128, no bias) -> LayerNorm
// (model_utils.py:499-519, 577-585), optionally followed by W_e (model_utils.py:89).  Same tiling as
// edge_mlp_kernel (one wave = 16 neighbours of one residue, activations in registers); the GEMM's 5200-long
// reduction never exists in memory: k-tile 0 is the 16 positional features, k-tile 1 + 18a + b is the 16 RBFs of
// atom pair (a of residue i, b of neighbour j), generated in registers right before their MFMAs.  The 2.66 MB
// weight image streams through a 2 x 48 KiB LDS ring in chunks of 6 k-tiles (one (a, b-group)); 63.9 MFLOP per
// residue, MFMA-bound.
struct FeatArgs {
  const float* X18; const uint32_t* M18;          // [G][54], [G]
  const int32_t* E_idx;                           // [G][K]
  const int32_t* R_idx; const int32_t* chain;     // [G]
  const float* Wedge_img;                         // image of edge_embedding.weight [128 x 5200]: 325 k-tiles
  const float* pos_w; const float* pos_b;         // embeddings.linear [16 x 66], [16]
  const float* ln_g; const float* ln_b;           // norm_edges
  const float* We_img; const float* We_b;         // optional fused W_e (x3 image for edge_features_kernel<true>)
  float* E_out;                                   // [G][K][128] or null
  float* hE_out;                                  // [G][K][128] or null (needs We_img)
  int G, L, K, TPN;
  // nparts = 2..4 (round 6; gridDim.x = nparts x residue blocks): workgroup (block, part) walks every nparts-th of the atom-pair chunks its
  // residues need and stores its pre-LayerNorm partial rows to pbuf[part]; feat_finish_kernel adds the parts, normalises and embeds.
  // 0 / 1: the whole row here.
  int nparts;
  float* pbuf[4];
  const int32_t* order;                           // feat_rank_blocks' output (nparts > 1)
  OrderJob ord;                                   // ord.B > 0: the launch's first ord.B workgroups sort the decoding orders (independent of the features;
                                                  // a launch of its own cost score() ~12 us of cross-stream hand-over at 1,000 residues)
};

#define FEAT_CHUNK_BYTES (6 * 8 * 64 * 16)        // 6 k-tiles x 8 tn x 1 KiB
#define FEAT_XJ_BYTES (54 * 16 * 4)               // one wave's 16 neighbour frames
#define FEAT_DBUF_BYTES (6 * 16 * 4)               // one wave's distances to the six atoms of a b-group
#define FEAT_LDS (2 * NAMP_IMG_BYTES + 4 * FEAT_XJ_BYTES + 256 + 12 * FEAT_DBUF_BYTES)

// X3: 0 exact fp32 MFMA; 1 split-bf16 products; 2 plain bf16 products on the x3 image's hi half (mixed-precision training)
template <int X3>
__global__ __launch_bounds__(768) void edge_features_kernel(const FeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int npw = nwaves / a.TPN;
  const int node_l = wave / a.TPN, kt = wave - node_l * a.TPN;
  if ((int)blockIdx.x < a.ord.B) { decoding_order_body(a.ord, blockIdx.x, smem); return; }      // (workgroup-uniform; first in dispatch order)
  const int bx0 = (int)blockIdx.x - a.ord.B, ngrid = (int)gridDim.x - a.ord.B;
#ifdef FEAT_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
  int blk = bx0, part = 0, myparts = 1;
  if (a.nparts > 1) {
    // One complex, at most a round of the chip unsplit: the launch lasts as long as its longest workgroup — a block of residues that holds a
    // nucleotide walks 39-54 atom-pair chunks, a protein block ~10.  Long blocks (feat_rank_blocks) are taken by nparts workgroups, each
    // walking every nparts-th chunk and leaving partial rows for feat_finish_kernel; the others by one workgroup that finishes its rows
    // itself.  Dispatch order is longest first, parts of a block next to each other: a long workgroup must not start behind short ones.
    // The grid is sized for every block being long; the surplus exits here.  (Tried, profiles/r06f: one workgroup per CU popping the same
    // items from a counter — no surplus, greedy packing: the same 151 us per call at 1,000 residues, with scalar spills around the item loop;
    // ranking by the exact chunk counts, neighbours' atoms included, by the neighbour-search workgroup that finishes last: launch 113 -> 111
    // us, but +22 us in the neighbour search for its per-row device-scope fences; three / four parts: 116-120 us.  The launch's sum of
    // workgroup durations / 256 CUs is 70 us unsplit and 79 us in two parts: second-round workgroups of ~35 us behind 86 us parts.)  (Tried: one workgroup per CU popping the same items from a
    // counter — no surplus, greedy packing: the same 151 us per call at 1,000 residues, with spilled scalars around the item loop.)
    const int P = a.nparts, nblk = ngrid / P, bx = bx0;
    const int n_long = a.order[256];
    int slot_i = bx / P;
    if (bx < n_long * P) { part = bx - slot_i * P; myparts = P; }
    else { slot_i = n_long + bx - n_long * P; if (slot_i >= nblk) return; }
    blk = a.order[slot_i];
  }
  // RBF centres of this lane: mu = 2 + (4g + r) * 20/15, sigma = 1.25  (model_utils.py:499-507)
  const float mu0 = 2.0f + (4 * g + 0) * (20.0f / 15.0f), mu1 = 2.0f + (4 * g + 1) * (20.0f / 15.0f);
  const float mu2 = 2.0f + (4 * g + 2) * (20.0f / 15.0f), mu3 = 2.0f + (4 * g + 3) * (20.0f / 15.0f);
  const float* img1 = a.Wedge_img + 8 * 64 * 4;                    // k-tile 1 onwards
  const int chunk_kb = (X3 == 2 ? FEAT_CHUNK_BYTES / 2 : FEAT_CHUNK_BYTES) / 1024;     // plain bf16 products read the hi half of a chunk only
  uint32_t* vote = (uint32_t*)(smem + 2 * NAMP_IMG_BYTES + 4 * FEAT_XJ_BYTES);   // [2][16]
  int node = blk * npw + node_l;
  const bool wave_active = (node_l < npw) && (node < a.G);
  if (!wave_active) node = 0;
  const int bq = node / a.L;
  const int k = 16 * kt + m;
  const bool valid = wave_active && (k < a.K);
  const long erow = (long)node * a.K + (valid ? k : 0);
  const int j = bq * a.L + a.E_idx[erow];

  // neighbour frames of the wave's 16 rows in LDS, [coordinate q][row m] (54 registers per lane otherwise — the kernel
  // spilled); own frame read per atom from L1/L2.  Written and read by this wave only: no barrier.  Waves 0-3 / 4-7 use
  // the free 16 KiB tails of the two ring slots, waves 8-11 the space behind the ring.
  float* xj = (float*)(smem + (wave < 4 ? FEAT_CHUNK_BYTES + wave * FEAT_XJ_BYTES
                                : wave < 8 ? NAMP_IMG_BYTES + FEAT_CHUNK_BYTES + (wave - 4) * FEAT_XJ_BYTES
                                           : 2 * NAMP_IMG_BYTES + (wave - 8) * FEAT_XJ_BYTES)) + m;
  {
    const float* src = a.X18 + (long)j * 54;
#pragma unroll
    for (int q = 0; q < 14; ++q)
      if (4 * q + g < 54) xj[(4 * q + g) * 16] = src[4 * q + g];
  }
  const uint32_t mj = a.M18[j], mi = a.M18[node];
  const float* xi_base = a.X18 + (long)node * 54;

  f4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- RBF chunks: c = 3a + bg, 6 k-tiles each, through the LDS ring.  An RBF feature is exactly zero when either atom is
  // absent (M18), and absence is structured: a protein residue has 5 of the 18 atoms (N, CA, C, O, Cb), a nucleotide the
  // other 13.  Zero k-tiles are skipped — for the whole workgroup (no DMA, no barrier) when no row of the workgroup needs
  // the chunk, per wave (no distance / exp / MFMA work) when none of the wave's 16 neighbours has atom b or its own residue
  // lacks atom a.  Adding exact zeros changes nothing, so the result is bit-identical to the dense evaluation; on
  // protein-protein neighbourhoods 25 of the 324 atom pairs remain.
  const uint32_t mi_s = __builtin_amdgcn_readfirstlane(wave_active ? mi : 0u);
  uint32_t mj_w = valid ? mj : 0u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mj_w |= __shfl_xor(mj_w, o);
  const uint32_t mj_s = __builtin_amdgcn_readfirstlane(mj_w);
  if (lane == 0) { vote[wave] = mi_s; vote[16 + wave] = mj_s; }
  __syncthreads();
  uint32_t wg_mi = 0, wg_mj = 0;
  for (int q = 0; q < nwaves; ++q) { wg_mi |= vote[q]; wg_mj |= vote[16 + q]; }
  unsigned long long need = 0;
  for (int aa = 0; aa < 18; ++aa)
    if ((wg_mi >> aa) & 1u)
      for (int bg = 0; bg < 3; ++bg)
        if ((wg_mj >> (6 * bg)) & 63u) need |= 1ull << (3 * aa + bg);
  need = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(need >> 32)) << 32) |
         __builtin_amdgcn_readfirstlane((uint32_t)need);
  if (myparts > 1) {                                                // this part's share: every myparts-th needed chunk
    unsigned long long mine = 0, rest = need;
    int ord = 0;
    while (rest) {
      const unsigned long long low = rest & (0ull - rest);
      if (ord == part) mine |= low;
      rest ^= low; ord = ord + 1 == myparts ? 0 : ord + 1;
    }
    need = mine;
  }
  __syncthreads();                                                  // votes consumed before the ring overwrites... (slot tail is not DMA'd, but keep order simple)
#ifdef FEAT_STAMPS
  const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
  const int st_chunks = __popcll(need);
#endif
  int slot = 0;
  if (need) dma_to_lds(smem, img1 + (long)__builtin_ctzll(need) * (FEAT_CHUNK_BYTES / 4), chunk_kb, wave, nwaves, lane);
  // ---- chunk 0: positional k-tile (weights: first 8 KiB of the image, read straight from L2), under the first chunk's DMA
  if (part == 0) {
    const int off = a.R_idx[node] - a.R_idx[j];
    const int same = (a.chain[node] == a.chain[j]) ? 1 : 0;
    int d = off + 32; d = d < 0 ? 0 : (d > 64 ? 64 : d);
    d = d * same + (1 - same) * 65;                                  // PositionalEncodings, model_utils.py:613-616
    f4 xk;
    xk.x = a.pos_w[(4 * g + 0) * 66 + d] + a.pos_b[4 * g + 0];
    xk.y = a.pos_w[(4 * g + 1) * 66 + d] + a.pos_b[4 * g + 1];
    xk.z = a.pos_w[(4 * g + 2) * 66 + d] + a.pos_b[4 * g + 2];
    xk.w = a.pos_w[(4 * g + 3) * 66 + d] + a.pos_b[4 * g + 3];
    const f4* w = (const f4*)a.Wedge_img + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) acc[tn] = mfma4(w[tn * 64][r], xk[r], acc[tn]);
  }
#ifndef FEAT_NOPIPE
  if constexpr (X3 != 0) {
    // Round 6 — split-bf16 / bf16 products: a step (K = 32: the RBFs of two atom pairs) as compact run-time loops over (b-group, step) instead
    // of nine unrolled copies with a branch per atom; the distance through the bare v_sqrt_f32 (1 ulp; sqrtf expands to ~13 more instructions
    // per value for the last half ulp and denormal inputs — D^2 >= 1e-6 here); the distances of a chunk computed once per tile instead of by
    // each of the four lane groups, the atom masks folded into them.  2.75 -> 2.31 ms at a 31,100-token batch, 149 -> 130 us per featurize
    // call of one 1,000-residue complex.  Generating a step's operands between the previous step's MFMAs (also across the chunk's barrier)
    // measured the same as this plain order (profiles/r06g).  Rows equal to the round-5 form's to fp32 rounding (9e-6 after the LayerNorm;
    // both 2.5e-5 from the exact-fp32 instantiation, which is unchanged).
    // the tile's distances to the six atoms of the chunk's b-group, once per chunk: lane (m, g) takes atoms g and g + 4 of row m (the four
    // lane groups used to repeat every distance); an absent atom's distance is the cap — every RBF of D = 40 A is exactly 0 in fp32, which
    // is what the mask multiplication produced.  Written and read by this wave only (LDS operations of a wave complete in order).
    float* dbuf = (float*)(smem + 2 * NAMP_IMG_BYTES + 4 * FEAT_XJ_BYTES + 256 + wave * FEAT_DBUF_BYTES) + m;
    auto dist = [&](const float xi0, const float xi1, const float xi2, const int bg, const uint32_t pres) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (!((pres >> (4 * r)) & (r ? 3u : 15u))) continue;         // wave-uniform: none of these atoms among the tile's neighbours
        const int sl = g + 4 * r;
        if (sl < 6) {
          const int bb = 6 * bg + sl;
          const float* xb = xj + (3 * bb) * 16;
          const float dx = xi0 - xb[0], dy = xi1 - xb[16], dz = xi2 - xb[32];
#ifdef FEAT_PRECISE_SQRT
          const float D = fminf(sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f), 40.0f);
#else
          const float D = fminf(__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f), 40.0f);
#endif
          dbuf[sl * 16] = ((mj >> bb) & 1u) ? D : 40.0f;
        }
      }
    };
    auto gen = [&](const int st, const uint32_t pres, bf8& hi, bf8& mid) {
      f4 xk[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // this lane's four RBFs have equidistant centres mu0 + r * 4/3: with v_r = (D - mu_r) * 0.8 * sqrt(log2 e) = v_0 - r d,
        //   G_r = exp(-((D - mu_r)/1.25)^2) = 2^(-v_r^2),   G_{r+1} / G_r = 2^(2 d v_r - d^2) =: q_r,   q_{r+1} = q_r * 2^(-2 d^2)
        // — two v_exp_f32 and five multiplications instead of four exponentials with their argument arithmetic.  D is capped at 40 A (every
        // RBF is exactly 0 in fp32 beyond 34 A) so that q_0 stays finite; relative error of G_3 ~1e-6, far below the split-bf16 products.
        const float D = ((pres >> (2 * st + h)) & 1u) ? dbuf[(2 * st + h) * 16] : 40.0f;    // (wave-uniform: the slot was not written)
        const float v0 = (D - mu0) * 0.9608979270291599f;
        const float G0 = __builtin_amdgcn_exp2f(-(v0 * v0));
        const float q0 = __builtin_amdgcn_exp2f(fmaf(v0, 2.5623944720777594f, -1.6414663576336648f));
        const float G1 = G0 * q0, q1 = q0 * 0.10273981490249438f;
        const float G2 = G1 * q1, q2 = q1 * 0.10273981490249438f;
        xk[h].x = G0; xk[h].y = G1; xk[h].z = G2; xk[h].w = G2 * q2;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hi[4 * h + r] = (__bf16)xk[h][r];
          mid[4 * h + r] = (__bf16)(xk[h][r] - (float)hi[4 * h + r]);
        }
    };
    auto mul = [&](const bf8& hi, const bf8& mid, const bf8* wb) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {                                  // product-major over four tiles, as chain_gemm_x3
        bf8 wh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wh[q] = wb[(4 * h + q) * 64];
        if (X3 == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], mid, acc[4 * h + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bf8 wm = wb[(FEAT_CHUNK_BYTES / 32) + (4 * h + q) * 64];
            acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hi, acc[4 * h + q], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], hi, acc[4 * h + q], 0, 0, 0);
      }
    };
    // (Measured at this point, profiles/r06g: a chunk takes 2.1 us against 1.4 us of MFMAs; the distances shared across the lane groups —
    // 40 % fewer VALU instructions per step — moved it 0.7 %; the next chunk through registers instead of LDS-DMA +4 %; 16 waves -1 % at a
    // batch and +17 % at one complex.  What is left is the chain barrier -> fragment reads -> products of a step with three waves per SIMD.)
#pragma unroll 1
    for (int aa = 0; aa < 18; ++aa) {
      if (!((need >> (3 * aa)) & 7ull)) continue;                    // workgroup-uniform: no chunk of atom a is needed
      const float xi0 = xi_base[3 * aa], xi1 = xi_base[3 * aa + 1], xi2 = xi_base[3 * aa + 2];
      const bool wave_a = (mi_s >> aa) & 1u;                         // (the wave's rows share their residue: its mask is wave-uniform)
#pragma unroll 1
      for (int bg = 0; bg < 3; ++bg) {
        const int c = 3 * aa + bg;
        if (!((need >> c) & 1ull)) continue;                         // workgroup-uniform
        // this wave's steps of the chunk: bit st set when a neighbour of the tile has one of the step's two atoms and the residue has atom a
        const uint32_t pres = wave_a ? ((mj_s >> (6 * bg)) & 63u) : 0u;
        uint32_t steps = ((pres & 3u) ? 1u : 0u) | ((pres & 12u) ? 2u : 0u) | ((pres & 48u) ? 4u : 0u);
        const bf8* wc = (const bf8*)(smem + slot * NAMP_IMG_BYTES) + lane;
#ifndef FEAT_ABL_NOGEN
        if (steps) dist(xi0, xi1, xi2, bg, pres);                    // (in front of the barrier: under the chunk's DMA)
#endif
#ifndef FEAT_ABL_NOBARRIER
        wait_dma_and_sync();                                         // chunk c has landed; everyone is done with the previous one
#endif
        {
          const unsigned long long rest = (c + 1 < 64) ? (need >> (c + 1)) : 0ull;
          if (rest) dma_to_lds(smem + (slot ^ 1) * NAMP_IMG_BYTES, img1 + (long)(c + 1 + __builtin_ctzll(rest)) * (FEAT_CHUNK_BYTES / 4),
                               chunk_kb, wave, nwaves, lane);
        }
        slot ^= 1;
#pragma unroll 1
        while (steps) {
          const int st = __builtin_ctz(steps);
          bf8 hi, mid;
#ifdef FEAT_ABL_NOGEN
          hi = (bf8){(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f}; mid = hi;
#else
          gen(st, pres, hi, mid);
#endif
#ifdef FEAT_ABL_NOMUL
          acc[st & 7][0] += (float)hi[0] + (float)mid[3];
#else
          __builtin_amdgcn_s_setprio(1);                             // the products ahead of the other waves' VALU work: -0.8 % (the same hint in
          mul(hi, mid, wc + st * 8 * 64);                            // chain_gemm_x3 cost the split-bf16 edge launches 4.7 %: profiles/r06g)
          __builtin_amdgcn_s_setprio(0);
#endif
          steps &= steps - 1u;
        }
      }
    }
  } else
#endif
#pragma unroll 1
  for (int aa = 0; aa < 18; ++aa) {
    if (!((need >> (3 * aa)) & 7ull)) continue;                      // workgroup-uniform: no chunk of atom a is needed
    const float xi0 = xi_base[3 * aa], xi1 = xi_base[3 * aa + 1], xi2 = xi_base[3 * aa + 2];
    const float mia = (float)((mi >> aa) & 1u);
    const bool wave_a = (mi_s >> aa) & 1u;
#pragma unroll
    for (int bg = 0; bg < 3; ++bg) {
      const int c = 3 * aa + bg;
      if (!((need >> c) & 1ull)) continue;                           // workgroup-uniform
      wait_dma_and_sync();                                           // chunk c has landed; everyone is done with the previous one
      {
        const unsigned long long rest = (c + 1 < 64) ? (need >> (c + 1)) : 0ull;
        if (rest) dma_to_lds(smem + (slot ^ 1) * NAMP_IMG_BYTES, img1 + (long)(c + 1 + __builtin_ctzll(rest)) * (FEAT_CHUNK_BYTES / 4),
                             chunk_kb, wave, nwaves, lane);
      }
      const f4* w = (const f4*)(smem + slot * NAMP_IMG_BYTES) + lane;
      slot ^= 1;
      if (!wave_a) continue;                                         // wave-uniform: this residue lacks atom a
      if constexpr (X3) {
        // split-bf16 form: one K = 32 step covers the RBFs of two atom pairs; x = hi + mid, W = hi + mid (pack_feat_x3_kernel)
        const bf8* wb = (const bf8*)w;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
          const int b0 = 6 * bg + 2 * st;
          if (!((mj_s >> b0) & 3u)) continue;                        // wave-uniform: neither atom of the step is present
          f4 xk[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int bb = b0 + h;
            if ((mj_s >> bb) & 1u) {
              const float dx = xi0 - xj[(3 * bb) * 16], dy = xi1 - xj[(3 * bb + 1) * 16], dz = xi2 - xj[(3 * bb + 2) * 16];
              // this lane's four RBFs have equidistant centres mu0 + r * 4/3: with v_r = (D - mu_r) * 0.8 * sqrt(log2 e) = v_0 - r d,
              //   G_r = exp(-((D - mu_r)/1.25)^2) = 2^(-v_r^2),   G_{r+1} / G_r = 2^(2 d v_r - d^2) =: q_r,   q_{r+1} = q_r * 2^(-2 d^2)
              // — two v_exp_f32 and six multiplications instead of four exponentials with their argument arithmetic (the RBF generation
              // is more than half of this kernel's issue time beside bf16 MFMAs).  D is capped at 40 A (every RBF is exactly 0 in fp32
              // beyond 34 A) so that q_0 stays finite; relative error of G_3 ~1e-6, far below the split-bf16 products that consume it.
              const float D = fminf(sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f), 40.0f);
              const float mk = mia * (float)((mj >> bb) & 1u);
              const float v0 = (D - mu0) * 0.9608979270291599f;
              const float G0 = __builtin_amdgcn_exp2f(-(v0 * v0)) * mk;
              const float q0 = __builtin_amdgcn_exp2f(fmaf(v0, 2.5623944720777594f, -1.6414663576336648f));
              const float G1 = G0 * q0, q1 = q0 * 0.10273981490249438f;
              const float G2 = G1 * q1, q2 = q1 * 0.10273981490249438f;
              xk[h].x = G0; xk[h].y = G1; xk[h].z = G2; xk[h].w = G2 * q2;
            } else {
              xk[h] = (f4){0.f, 0.f, 0.f, 0.f};
            }
          }
          bf8 hi, mid;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              hi[4 * h + r] = (__bf16)xk[h][r];
              mid[4 * h + r] = (__bf16)(xk[h][r] - (float)hi[4 * h + r]);
            }
#pragma unroll
          for (int h = 0; h < 2; ++h) {                              // product-major over four tiles, as chain_gemm_x3
            bf8 wh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wh[q] = wb[(st * 8 + 4 * h + q) * 64];
            if (X3 == 1) {
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], mid, acc[4 * h + q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const bf8 wm = wb[(FEAT_CHUNK_BYTES / 32) + (st * 8 + 4 * h + q) * 64];
                acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hi, acc[4 * h + q], 0, 0, 0);
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], hi, acc[4 * h + q], 0, 0, 0);
          }
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int bb = 6 * bg + q;
        if (!((mj_s >> bb) & 1u)) continue;                          // wave-uniform: no neighbour of this tile has atom b
        const float dx = xi0 - xj[(3 * bb) * 16], dy = xi1 - xj[(3 * bb + 1) * 16], dz = xi2 - xj[(3 * bb + 2) * 16];
        const float D = sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f);
        const float mk = mia * (float)((mj >> bb) & 1u);
        const float t0 = (D - mu0) * 0.8f, t1 = (D - mu1) * 0.8f, t2 = (D - mu2) * 0.8f, t3 = (D - mu3) * 0.8f;
        f4 xk;
        xk.x = __expf(-(t0 * t0)) * mk; xk.y = __expf(-(t1 * t1)) * mk;
        xk.z = __expf(-(t2 * t2)) * mk; xk.w = __expf(-(t3 * t3)) * mk;
        f4 wf[8];
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) wf[tn] = w[(q * 8 + tn) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tn = 0; tn < 8; ++tn) acc[tn] = mfma4(wf[tn][r], xk[r], acc[tn]);
      }
    }
  }
#ifdef FEAT_STAMPS
  if (tid == 0 && bx0 < 1024) {
    unsigned long long* o = g_feat_stamps[bx0];
    o[0] = st0; o[1] = st1; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = blk; o[4] = part | (myparts << 8); o[5] = st_chunks;
    o[6] = __builtin_amdgcn_s_getreg(GETREG_IMMED(32 - 1, 0, 4)); o[7] = __builtin_amdgcn_s_getreg(GETREG_IMMED(4 - 1, 0, 20));
  }
#endif
  if (myparts > 1) {                                               // partial rows out; feat_finish_kernel does the rest
    if (valid) {
      float* dst = a.pbuf[part] + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
    }
    return;
  }
  // ---- LayerNorm (norm_edges) -> E; optional W_e embed -> h_E
  if (a.ln_g) layernorm_row_T(acc, a.ln_g, a.ln_b, g);             // null: E_out receives the pre-LayerNorm rows
  if (a.E_out && valid) {
    float* dst = a.E_out + erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
  }
  if (a.hE_out) {
    __syncthreads();                                                 // ring free
    dma_to_lds(smem, a.We_img, 64, wave, nwaves, lane);
    f4 out[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) out[t] = *(const f4*)(a.We_b + 16 * t + 4 * g);
    wait_dma_and_sync();
    gemm128<X3 != 0, false, false>(out, acc, (const f4*)smem + lane);
    if (valid) {
      float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = out[t];
    }
  }
}

// feat_finish_kernel — behind an edge_features launch in parts: row = part 0 + part 1 (+ ...), LayerNorm (norm_edges) -> E, optional W_e embed -> h_E
// (the epilogue of edge_features_kernel).  One wave per 16-row tile; pbuf[0] / pbuf[1] may be the output buffers themselves (a wave reads its
// tile's rows before it writes them).
template <int X3>
__global__ __launch_bounds__(768) void feat_finish_kernel(const FeatArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int npw = nwaves / a.TPN;
  const int node_l = wave / a.TPN, kt = wave - node_l * a.TPN;
  int node = blockIdx.x * npw + node_l;
  const bool wave_active = (node_l < npw) && (node < a.G);
  if (!wave_active) node = 0;
  const int k = 16 * kt + m;
  const bool valid = wave_active && (k < a.K);
  const long erow = (long)node * a.K + (valid ? k : 0);
  {                                                                  // blocks edge_features_kernel did not split are finished already
    uint32_t u = 0;
    for (int q = 0; q < npw; ++q) { const int nd = (int)blockIdx.x * npw + q; if (nd < a.G) u |= a.M18[nd]; }
    if (__popc(u) < FEAT_LONG_ATOMS) return;
  }
  if (a.hE_out) dma_to_lds(smem, a.We_img, 64, wave, nwaves, lane);
  f4 acc[8];
  {
    const float* p0 = a.pbuf[0] + erow * NAMP_H + 4 * g;
    const float* p1 = a.pbuf[1] + erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(p0 + 16 * t) + *(const f4*)(p1 + 16 * t);
    for (int q = 2; q < a.nparts; ++q) {
      const float* pq = a.pbuf[q] + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = acc[t] + *(const f4*)(pq + 16 * t);
    }
  }
  if (a.ln_g) layernorm_row_T(acc, a.ln_g, a.ln_b, g);
  if (a.E_out && valid) {
    float* dst = a.E_out + erow * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = acc[t];
  }
  if (a.hE_out) {
    f4 out[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) out[t] = *(const f4*)(a.We_b + 16 * t + 4 * g);
    wait_dma_and_sync();
    gemm128<X3 != 0, false, false>(out, acc, (const f4*)smem + lane);
    if (valid) {
      float* dst = a.hE_out + erow * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(dst + 16 * t) = out[t];
    }
  }
}
