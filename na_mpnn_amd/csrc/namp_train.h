// Backward (training) kernels of the NA-MPNN message path on gfx950 — the gradient side of
// EncLayer / DecLayer (na_model_utils.py:196-283) and of the edge featuriser's 5200 -> 128 embedding
// (na_model_utils.py:499-507).  The reference wraps every layer in torch.utils.checkpoint
// (na_model_utils.py:606,637): activations are recomputed in the backward pass.  The same policy is
// applied here at kernel granularity — edge_chain_bwd_kernel recomputes the per-edge MLP from its inputs
// and walks it backwards in registers; nothing but the layer inputs is kept from the forward pass.
//
//   forward chain (per edge row e = (i, k), j = E_idx[i,k]):
//       z1 = W1b . h_E[e] + Pa[i] + Pj[j]        a1 = gelu(z1)
//       z2 = W2 . a1 + b2                        a2 = gelu(z2)
//       z3 = W3 . a2 + b3
//   backward, given g3 = dL/dz3 per row:
//       g2 = (W3^T g3) * gelu'(z2)     g1 = (W2^T g2) * gelu'(z1)     dL/dh_E[e] = W1b^T g1
//   and the row tensors A1 = a1, A2 = a2, G1, G2, G3 go to HBM, from which
//       dW3 = G3^T A2, dW2 = G2^T A1, dW1b = G1^T h_E                 (wgrad_kernel: contraction over edges)
//       dL/dPa[i] = sum_k G1[i,k],  dL/dPj[j] += G1[e]               (residue-level, done by the caller)
// The transposed products reuse the register chain of namp_device.h: "W^T . g" is the T-orientation
// GEMM with the fragment image of W^T, so gradients flow lane-locally exactly like activations do.
#pragma once
#include "namp_device.h"

// gelu(x) and d/dx gelu(x) = Phi(x) + x phi(x), sharing exp(-x^2/2) and the A-S 7.1.26 erf of gelu_erf().
__device__ __forceinline__ void gelu_val_grad(float x, float& val, float& grad) {
  const float v = x * 0.84932180028801904f;
  const float e = __builtin_amdgcn_exp2f(-(v * v));                     // exp(-x^2 / 2)
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(v), 0.27273943f, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float y = fmaf(-(q * t), e, 1.0f);                              // |erf(x / sqrt 2)|
  const float phi_big = fmaf(copysignf(0.5f, x), y, 0.5f);              // Phi(x)
  val = x * phi_big;
  grad = fmaf(x * e, 0.3989422804014327f, phi_big);
}

// in: pre-activations z; out: z <- gelu'(z), returns gelu(z) — one exp / rcp pair serves both
__device__ __forceinline__ f4 gelu_split4(f4& z) {
#ifdef DW_EXP_NOGELU
  { const f4 v_ = z; z = z * 0.5f; return v_; }
#endif
  float v0, v1, v2, v3, d0, d1, d2, d3;
  gelu_val_grad(z.x, v0, d0); gelu_val_grad(z.y, v1, d1); gelu_val_grad(z.z, v2, d2); gelu_val_grad(z.w, v3, d3);
  z = (f4){d0, d1, d2, d3};
  return (f4){v0, v1, v2, v3};
}

// Row tensors of the MIXED-PRECISION mode (precision code 2): A1, A2, G1, G2, G3 are [E][128] bf16 in plain channel order —
// half the bytes for the launch's stores, for the row contractions (namp_train_wgrad) and for the table-gradient gather.
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf4 to_bf4(const f4 v) { bf4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w; return o; }
__device__ __forceinline__ f4 from_bf4(const bf4 v) { return (f4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
template <bool BF> __device__ __forceinline__ void st_row4(float* base, const long off, const f4 v) {
  if (BF) *(bf4*)((__bf16*)base + off) = to_bf4(v); else *(f4*)(base + off) = v;
}
template <bool BF> __device__ __forceinline__ f4 ld_row4(const float* base, const long off) {
  if (BF) return from_bf4(*(const bf4*)((const __bf16*)base + off));
  return *(const f4*)(base + off);
}

// A wave's 16-row tile (register-chain layout: lane (m, g) holds channels 16t + 4g + r of row m) as bf16 rows: the 8-byte piece (t, g) of a row sits at
// piece index 4t + g, so a lane's own pieces are 32 bytes apart and a store instruction moves 16 x 32 bytes.  v_permlane16_swap_b32 (gfx950: the odd
// 16-lane rows of one register against the even rows of another) hands lane group g the piece of g ^ 1: even groups then store pieces (4t + g, 4t + g + 1),
// odd ones (4(t+1) + g - 1, 4(t+1) + g) — four 16-byte stores per lane and tile instead of eight 8-byte ones.
#ifndef NAMP_ABL_NOPAIRST
__device__ __forceinline__ void st_tile_bf16(float* base, const long row_off, const f4 (&v)[8], const int g) {
  __bf16* dst = (__bf16*)base + row_off;
  typedef int i4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    const bf4 x = to_bf4(v[t]), y = to_bf4(v[t + 1]);
    typedef int i2v __attribute__((ext_vector_type(2)));
    const i2v xi = __builtin_bit_cast(i2v, x), yi = __builtin_bit_cast(i2v, y);
    const auto s0 = __builtin_amdgcn_permlane16_swap(xi[0], yi[0], false, false);      // [0]: x with its odd rows replaced by y's even rows; [1]: y with its
    const auto s1 = __builtin_amdgcn_permlane16_swap(xi[1], yi[1], false, false);      //      even rows replaced by x's odd rows
    const i4v o = (i4v){(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
    const int col = (g & 1) ? 16 * (t + 1) + 4 * (g - 1) : 16 * t + 4 * g;
    *(i4v*)(dst + col) = o;
  }
}
#else
__device__ __forceinline__ void st_tile_bf16(float* base, const long row_off, const f4 (&v)[8], const int g) {
#pragma unroll
  for (int t = 0; t < 8; ++t) st_row4<true>(base, row_off + 4 * g + 16 * t, v[t]);
}
#endif

enum { BWD_ENC_MSG = 0, BWD_DEC_MSG = 1, BWD_ROWS = 2, BWD_EDGE_LN = 3 };

struct EdgeBwdArgs {
  const float* hE;             // [E][128] edge rows the forward pass consumed
  const int32_t* E_idx;        // [G][K]
  const int32_t* mask;         // ENC_MSG: [G] residue mask (null = ones)
  const int32_t* mask_attend;  // ENC_MSG: optional explicit [G][K]
  const int32_t* rank;         // DEC_MSG: [G]
  const float* Pa;             // [G][128]
  const float* Pj0;            // ENC: Pc;  DEC: Pbw
  const float* Pj1;            // DEC: Pfw
  const float* W1_img;         // forward images of W1b (or W1e), W2
  const float* W2_img;
  const float* W3t_img;        // images of W3^T, W2^T, W1b^T
  const float* W2t_img;
  const float* W1t_img;
  const float* b2;
  const float* g_rows;         // BWD_ROWS: dL/dz3 per edge row [E][128]
  const float* g_node;         // MSG modes: dL/d(K-sum of a2) per residue [G][128]; dL/da2[e] = w_e * g_node[i]
  float* A1; float* A2;        // [E][128] activations gelu(z1), gelu(z2) (for wgrad)
  float* G1; float* G2; float* G3;   // [E][128] (G3 only written in MSG modes)
  // (unused since layer 3 moved behind the K-sum: kept so that the argument block keeps its layout)
  float* S3; float* w3;
  float* g_hE;                 // [E][128]
  const float* g_hE_in;        // message modes with acc_hE: the other consumer's dL/dh_E rows (may alias g_hE)
  float* g_Pa;                 // optional [G][128], ZEROED by the caller: += sum_k G1[i,k]   (fp32 atomics)
  float* g_Pj0; float* g_Pj1;  // optional [G][128], zeroed: += G1[e] at the row's gathered table (Pj0 / Pj1 like the forward)
  // BWD_EDGE_LN: the whole edge update h_E' = LN3(h_E + dropout(z3)) is differentiated here; g_rows = dL/dh_E'
  const float* W3_img; const float* b3;      // forward image of W3 (z3 is recomputed)
  const float* ln_g;                          // LayerNorm3 weight
  uint32_t drop_thresh, drop_seed; float drop_scale;
  long drop_row0;              // the dropout mask is a hash of (seed, drop_row0 + row): a launch over a slice of the batch passes the slice's first row
  float* dgb_part;             // [gridDim.x][2][128]: per-workgroup sums of g*xhat (-> d ln weight) and g (-> d ln bias)
  long E;                      // G * K rows
  int G, N, K;
  int gpa_tiles;               // g_Pa is [E/16][128] per-tile sums (K % 16 == 0: a tile = one residue), plain stores — deterministic
  int acc_hE;                  // message modes: g_hE already holds another consumer's dL/dh_E — add to it instead of overwriting
};

// One wave = 16 consecutive edge rows of the flat [G*K] edge list (tiles may straddle residues: every
// per-row operand is gathered per lane anyway).  8 waves per workgroup share the 2 x 64 KiB LDS weight ring;
// the five images W1, W2, W3^T, W2^T, W1^T stream through it by LDS-DMA one GEMM ahead of their use.
// X3: the six / five GEMMs as split-bf16 products (chain_gemm_x3; the images are then x3 images), like the forward kernels
// PREC: 0 exact fp32 MFMA, 1 split-bf16 products (X3), 2 plain bf16 products (mixed-precision mode; 32 KiB images)
template <int MODE, int PREC>
__global__ __launch_bounds__(512) void edge_chain_bwd_kernel(const EdgeBwdArgs a) {
  constexpr int IMG_KB = (PREC == 2) ? 32 : 64;
  constexpr bool RB = (PREC == 2);            // bf16 row tensors (A1, A2, G1, G2, G3)
  // Message modes: layer 3 sits BEHIND the K-sum (forward: NodeTail.m3_img), so the upstream gradient g_node is dL/d(sum_k w_k a2_k)
  // and dL/da2 of row k is just w_k * g_node — no W3^T product, no A2 / G3 rows, four GEMMs (W1, W2, W2^T, W1^T + dh_E).
  constexpr bool MSG = (MODE == BWD_ENC_MSG || MODE == BWD_DEC_MSG);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // BWD_EDGE_LN: the waves' column sums for d(ln weight), d(ln bias) — one slot per wave, added in wave order at the end (LDS float atomics here
  // made the sum's order, hence its last bits, vary from run to run)
  __shared__ float colsum[MODE == BWD_EDGE_LN ? 8 : 1][2 * NAMP_H];
  // b2 (| b3 | LayerNorm-3 weight) staged once per workgroup: as 16-byte loads per lane they are 8 KiB per wave and vector through the
  // CU's 64 B/clk vector-memory path (profiles/r02k_bf16s_ablation.md, last table)
  __shared__ __attribute__((aligned(16))) float cstb[3 * NAMP_H];
  if (threadIdx.x < NAMP_H) cstb[threadIdx.x] = a.b2[threadIdx.x];
  else if (MODE == BWD_EDGE_LN && threadIdx.x < 3 * NAMP_H) cstb[threadIdx.x] = (threadIdx.x < 2 * NAMP_H ? a.b3 : a.ln_g)[threadIdx.x & (NAMP_H - 1)];
  char* buf0 = smem;
  char* buf1 = smem + NAMP_IMG_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const long e_raw = ((long)blockIdx.x * nwaves + wave) * 16 + m;
  const bool valid = e_raw < a.E;
  const long e = valid ? e_raw : (a.E - 1);
  const int node = (int)(e / a.K);
  const int i_loc = node % a.N;
  const int j = node - i_loc + a.E_idx[e];

  f4 x[8], z1[8], z2[8], pjv[8];
  float w_row = 0.f;
  float* gpj = a.g_Pj0;        // table-gradient buffer this row scatters into
  {
    const float* src = a.hE + e * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
    const float* pj;
    if (MODE == BWD_DEC_MSG) {
      const bool bwd = a.rank[j] < a.rank[node];
      pj = bwd ? (a.Pj0 + (long)j * NAMP_H) : (a.Pj1 + (long)j * NAMP_H);
      gpj = bwd ? a.g_Pj0 : a.g_Pj1;
      w_row = valid ? (1.0f / 30.0f) : 0.f;
    } else {
      pj = a.Pj0 + (long)j * NAMP_H;
      if (MODE == BWD_ENC_MSG) {
        int ma;
        if (a.mask_attend) ma = a.mask_attend[e];
        else ma = a.mask ? (a.mask[node] * a.mask[j]) : 1;
        w_row = valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
      }
    }
    const float* pa = a.Pa + (long)node * NAMP_H + 4 * g;
    pj += 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
  }
  const f4* w0 = (const f4*)buf0 + lane;
  const f4* w1 = (const f4*)buf1 + lane;

  dma_to_lds(buf0, a.W1_img, IMG_KB, wave, nwaves, lane);
  dma_to_lds(buf1, a.W2_img, IMG_KB, wave, nwaves, lane);
  wait_dma_and_sync();
  // ---- recompute: z1, z2
  gemm128p<PREC, false>(z1, x, w0);
#pragma unroll
  for (int t = 0; t < 8; ++t) z1[t] += pjv[t];
  __syncthreads();                                            // everyone is done with W1
  dma_to_lds(buf0, MODE == BWD_EDGE_LN ? a.W3_img : (MSG ? a.W2t_img : a.W3t_img), IMG_KB, wave, nwaves, lane);
  // activations and their derivatives from ONE evaluation each: x <- a1 = gelu(z1), z1 <- gelu'(z1)
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = gelu_split4(z1[t]);
  if (valid) {
#pragma unroll
    for (int t = 0; t < 8; ++t) st_row4<RB>(a.A1, e * NAMP_H + 4 * g + 16 * t, x[t]);
    if (MODE == BWD_EDGE_LN) {       // 6-GEMM variant: gelu'(z1) waits in its own G1 row (L2) instead of 32 VGPRs
#pragma unroll
      for (int t = 0; t < 8; ++t) st_row4<RB>(a.G1, e * NAMP_H + 4 * g + 16 * t, z1[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) z2[t] = *(const f4*)(cstb + 16 * t + 4 * g);
  gemm128p<PREC, false>(z2, x, w1);
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = gelu_split4(z2[t]);      // x <- a2 (only stored, for dW3), z2 <- gelu'(z2)
  // Row stores are issued right AFTER a ring barrier, never right before one: s_waitcnt vmcnt(0) also waits for store
  // acknowledgements (gfx9 has one counter), and a store issued after the barrier has a whole GEMM to retire.
  auto store_rows = [&](float* base, const f4 (&v)[8]) {
    if (valid) {
#pragma unroll
      for (int t = 0; t < 8; ++t) st_row4<RB>(base, e * NAMP_H + 4 * g + 16 * t, v[t]);
    }
  };
  // ---- upstream gradient rows
  f4 gr[8];
  if (MODE == BWD_EDGE_LN) {
    // z3 = W3 a2 + b3 (a2 is still in x), then backwards through LayerNorm3 and the dropout mask
    wait_dma_and_sync();                                      // W3 landed in buf0; W2 (buf1) is free
    dma_to_lds(buf1, a.W3t_img, IMG_KB, wave, nwaves, lane);
    store_rows(a.A2, x);
    f4 z3[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) z3[t] = *(const f4*)(cstb + NAMP_H + 16 * t + 4 * g);
    gemm128p<PREC, false>(z3, x, w0);
    asm volatile("" ::: "memory");       // the row loads below have kernel-constant addresses: do not hoist them (96 VGPRs) over the GEMMs
    const uint32_t key = drop_row_key(a.drop_seed, a.drop_row0 + e);
    const float* hsrc = a.hE + e * NAMP_H + 4 * g;
    const float* gsrc = a.g_rows + e * NAMP_H + 4 * g;
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (a.drop_thresh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) z3[t][r] *= drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, a.drop_scale);
      }
      z3[t] += *(const f4*)(hsrc + 16 * t);                    // x = h_E + dropout(z3)
      s1 += (z3[t].x + z3[t].y) + (z3[t].z + z3[t].w);
    }
    const float mean = xg_sum(s1) * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { z3[t] -= mean; s2 += (z3[t].x * z3[t].x + z3[t].y * z3[t].y) + (z3[t].z * z3[t].z + z3[t].w * z3[t].w); }
    const float rstd = rsqrtf(xg_sum(s2) * (1.0f / 128.0f) + 1e-5f);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      z3[t] *= rstd;                                           // xhat
      const f4 gy = valid ? *(const f4*)(gsrc + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
      const f4 gg = gy * *(const f4*)(cstb + 2 * NAMP_H + 16 * t + 4 * g);
      gr[t] = gg;
      m1 += (gg.x + gg.y) + (gg.z + gg.w);
      m2 += (gg.x * z3[t].x + gg.y * z3[t].y) + (gg.z * z3[t].z + gg.w * z3[t].w);
      // column sums over the tile's rows for d(ln weight) = sum g*xhat and d(ln bias) = sum g  ->  LDS
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float cw = gy[r] * z3[t][r], cb = gy[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { cw += __shfl_xor(cw, o); cb += __shfl_xor(cb, o); }
        if (m == 0) { colsum[wave][16 * t + 4 * g + r] = cw; colsum[wave][NAMP_H + 16 * t + 4 * g + r] = cb; }
      }
    }
    m1 = xg_sum(m1) * (1.0f / 128.0f);
    m2 = xg_sum(m2) * (1.0f / 128.0f);
    float* gres = a.g_hE + e * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      gr[t] = (gr[t] - m1 - z3[t] * m2) * rstd;               // dL/dx of LayerNorm = the residual part of dL/dh_E:
      if (valid) *(f4*)(gres + 16 * t) = gr[t];                // parked in the output row (L2), picked up by the last GEMM
      if (a.drop_thresh) {                                     // through the (regenerated) dropout mask: g3 = dL/dz3
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[t][r] *= drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, a.drop_scale);
      }
    }
  } else if (MODE == BWD_ROWS) {
    const float* src = a.g_rows + e * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = valid ? *(const f4*)(src + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
  } else {
    const float* src = a.g_node + (long)node * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = *(const f4*)(src + 16 * t) * w_row;
  }
  // images from here on: W3^T, W2^T, W1b^T in slots (A, B, A) with A = buf0 — or buf1 in BWD_EDGE_LN, whose extra z3 GEMM
  // shifted the ring by one.  Message modes: W2^T (already requested into buf0), W1b^T in slots (B, A) with A = buf1.
  const f4* wA = (MODE == BWD_EDGE_LN || MSG) ? w1 : w0;
  const f4* wB = (MODE == BWD_EDGE_LN || MSG) ? w0 : w1;
  char* bufA = (MODE == BWD_EDGE_LN || MSG) ? buf1 : buf0;
  char* bufB = (MODE == BWD_EDGE_LN || MSG) ? buf0 : buf1;
  f4 acc[8];
  if (MSG) {
    wait_dma_and_sync();                                      // W2^T landed in buf0; buf1 (W2) is free
    dma_to_lds(bufA, a.W1t_img, IMG_KB, wave, nwaves, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = gr[t] * z2[t];        // g2 = w_k * g_node * gelu'(z2)
  } else {
  wait_dma_and_sync();                                        // W3^T landed in slot A; slot B is free
  dma_to_lds(bufB, a.W2t_img, IMG_KB, wave, nwaves, lane);
  if (MODE != BWD_EDGE_LN) store_rows(a.A2, x);
  if (MODE != BWD_ROWS) store_rows(a.G3, gr);
  // ---- g2 = (W3^T g3) * gelu'(z2)
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
  gemm128p<PREC, false>(acc, gr, wA);
#pragma unroll
  for (int t = 0; t < 8; ++t) gr[t] = acc[t] * z2[t];
  wait_dma_and_sync();                                        // W2^T landed in slot B; slot A is free
  dma_to_lds(bufA, a.W1t_img, IMG_KB, wave, nwaves, lane);
  }
  store_rows(a.G2, gr);
  // ---- g1 = (W2^T g2) * gelu'(z1)
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
  gemm128p<PREC, false>(acc, gr, wB);
  if (MODE == BWD_EDGE_LN) {
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = valid ? acc[t] * ld_row4<RB>(a.G1, e * NAMP_H + 4 * g + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = acc[t] * z1[t];
  }
  wait_dma_and_sync();                                        // W1b^T landed in slot A
  store_rows(a.G1, gr);
  // gradients of the hoisted first-layer tables, accumulated here instead of re-reading G1 (sum over k / scatter over j)
  if (a.g_Pj0) {
    if (valid) {
      float* d = gpj + (long)j * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w);
      }
    }
  }
  if (a.g_Pa && a.gpa_tiles) {
    // tiles aligned to residues: the tile's 16 rows belong to ONE residue — sum them (butterfly over the 16 lanes of a
    // channel group) and store the tile's row; the caller adds a residue's K/16 tiles.  No atomics: the summation order is fixed.
    const long tile = (long)blockIdx.x * nwaves + wave;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f4 v = valid ? gr[t] : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
      }
      if (m == 0 && tile * 16 < a.E) *(f4*)(a.g_Pa + tile * NAMP_H + 16 * t + 4 * g) = v;
    }
  } else if (a.g_Pa) {
    // rows of a tile are consecutive edges: residues are non-decreasing, usually one or two per tile -> reduce over the
    // rows of the first / the last residue with two masked butterflies and add once per residue; other shapes (K < 8) go row by row
    const int n_first = __shfl(node, g << 4), n_last = __shfl(node, (g << 4) | 15);
    const bool two = (node == n_first) || (node == n_last);
    const bool simple = __ballot(valid && !two) == 0ull;
    if (simple) {
      const float sel_a = (valid && node == n_first) ? 1.f : 0.f;
      const float sel_b = (valid && node == n_last && n_last != n_first) ? 1.f : 0.f;
      float* da = a.g_Pa + (long)n_first * NAMP_H + 4 * g;
      float* db = a.g_Pa + (long)n_last * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sa = gr[t][r] * sel_a, sb = gr[t][r] * sel_b;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
          if (m == 0) {
            unsafeAtomicAdd(da + 16 * t + r, sa);
            if (n_last != n_first) unsafeAtomicAdd(db + 16 * t + r, sb);
          }
        }
      }
    } else if (valid) {
      float* d = a.g_Pa + (long)node * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w);
      }
    }
  }
  // ---- dL/dh_E = W1b^T g1 (+ the residual path of the edge update)
#pragma unroll
  for (int t = 0; t < 8; ++t)
    acc[t] = ((MODE == BWD_EDGE_LN || a.acc_hE) && valid) ? *(const f4*)((MODE == BWD_EDGE_LN ? a.g_hE : a.g_hE_in) + e * NAMP_H + 4 * g + 16 * t)
                                                            : (f4){0.f, 0.f, 0.f, 0.f};
  gemm128p<PREC, false>(acc, gr, wA);
  if (valid) {
    float* d = a.g_hE + e * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(f4*)(d + 16 * t) = acc[t];
  }
  if (MODE == BWD_EDGE_LN) {
    __syncthreads();
    if (threadIdx.x < 2 * NAMP_H) {
      float s_ = colsum[0][threadIdx.x];
#pragma unroll
      for (int w2 = 1; w2 < 8; ++w2) s_ += colsum[w2][threadIdx.x];
      a.dgb_part[(long)blockIdx.x * 2 * NAMP_H + threadIdx.x] = s_;
    }
  }
}

#ifndef NAMP_TRAIN_EDGE_ONLY      // (namp_train_eu.hip takes the per-edge chain above and nothing below)
// ------------------------------------------------------------------------------------------
// wgrad_kernel: dW[o][c] = sum_rows G[row][o] * act(A[row][c]),  db[o] = sum_rows G[row][o]   (128 x 128, rows ~ 10^6).
// The contraction index is the row, 4 rows per v_mfma_f32_16x16x4_f32: with k-slot g of MFMA r standing for
// row 4g + r of a 16-row step, lane (n = l&15, g) feeds G[row][16to + n] as the A operand and A[row][16tc + n]
// as the B operand — both are plain strided reads of the row-major tensors, no transposes.  A workgroup
// (4 waves, wave w owns output tiles to in {2w, 2w+1} x all tc: 64 accumulator VGPRs) reduces one chunk of rows
// and writes its partial [128][128] (+ [128]); the caller sums the chunks (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
template <bool ACT>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ G, const float* __restrict__ A, long rows,
                                                    long rows_per_chunk, float* __restrict__ dW_part,
                                                    float* __restrict__ db_part) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  long r_end = r_begin + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  f4 acc[2][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[q][t] = (f4){0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  float gv[2][4], av[8][4], gn[2][4], an[8][4];
  // operands of the 16-row step starting at r0 (rows past the chunk end contribute zeros through G)
  auto load = [&](long r0, float (&go)[2][4], float (&ao)[8][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = r0 + 4 * g + r;
      const bool ok = row < r_end;
      const long rr = ok ? row : r_begin;
      const float* gp = G + rr * NAMP_H + n;
      const float* ap = A + rr * NAMP_H + n;
#pragma unroll
      for (int q = 0; q < 2; ++q) { const float v = gp[16 * (2 * wave + q)]; go[q][r] = ok ? v : 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) ao[t][r] = ap[16 * t];
    }
  };
  if (r_begin < r_end) load(r_begin, gv, av);
  for (long r0 = r_begin; r0 < r_end; r0 += 16) {
    const bool more = r0 + 16 < r_end;
    if (more) load(r0 + 16, gn, an);                          // next step's loads fly under this step's 64 MFMAs
    if (ACT) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) av[t][r] = gelu_erf(av[t][r]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) bsum[q] += (gv[q][0] + gv[q][1]) + (gv[q][2] + gv[q][3]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[q][t] = mfma4(gv[q][r], av[t][r], acc[q][t]);
    if (more) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 2; ++q) gv[q][r] = gn[q][r];
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t][r] = an[t][r];
      }
    }
  }
  // D[i = 4g + r][j = n]  ->  dW[16 to + 4g + r][16 tc + n]
  float* out = dW_part + (long)blockIdx.x * NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * (2 * wave + q) + 4 * g + r) * NAMP_H + 16 * t + n] = acc[q][t][r];
  if (db_part) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float s = xg_sum(bsum[q]);
      if (g == 0) db_part[(long)blockIdx.x * NAMP_H + 16 * (2 * wave + q) + n] = s;
    }
  }
}

// wgrad_x3_kernel: the same row contraction as split-bf16 products (namp_device.h, chain_gemm_x3): k-slot (g, j) of
// v_mfma_f32_16x16x32_bf16 stands for row 8g + j of a 32-row step, so lane (n, g) reads column n of its 8 rows of G (A
// operand) and of A (B operand), splits each value into bf16 hi + mid on the fly and issues hi*hi + hi*mid + mid*hi:
// 48 bf16 MFMAs (768 pipe cycles) per 32 rows and wave instead of 128 fp32 MFMAs (4,096).  Waves form a 2 x 2 grid over the
// 128 x 128 output (4 x 4 tiles each), which also halves the operand loads per wave.
__device__ __forceinline__ void split8(const float (&v)[8], bf8& hi, bf8& mid) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { hi[j] = (__bf16)v[j]; mid[j] = (__bf16)(v[j] - (float)hi[j]); }
}

// MID = false: plain bf16 products (hi . hi only) — the mixed-precision mode
// LN: the A rows are PRE-LayerNorm rows y and the operand is LayerNorm(y) = (y - mean) * rstd * gamma + beta, re-derived per loaded value from
// the row statistics the backward launch left in ln_stats [rows][2] (embed_ln_bwd_kernel) — the normalised rows are never stored.
// KS = 2: eight waves — two groups of four, each taking every other 32-row step of the chunk; the second group hands its accumulators to the first
// through LDS at the end.  A 24,000-row contraction is 47 chunks of 16 dependent steps (load, split, 16-48 MFMAs): its duration is one workgroup's
// latency chain, and halving the chain halves the launch.
template <bool MID, bool LN = false, int KS = 1>
__device__ __forceinline__ void wgrad_x3_body(const float* __restrict__ G, const float* __restrict__ A, long rows,
                                              long rows_per_chunk, float* __restrict__ dW_part, float* __restrict__ db_part,
                                              const bool accumulate = false, const float* __restrict__ ln_stats = nullptr,
                                              const float* __restrict__ ln_g = nullptr, const float* __restrict__ ln_b = nullptr) {
  __shared__ __attribute__((aligned(16))) float ks_red[KS == 2 ? 4 * 8 * 64 * 4 + 4 * 4 * 64 : 4];
  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave8 & 3, grp = wave8 >> 2;
  const int n = lane & 15, g = lane >> 4;
  const int to0 = 4 * (wave >> 1), tc0 = 4 * (wave & 1);
  const long r_chunk = (long)blockIdx.x * rows_per_chunk;
  const long r_begin = r_chunk + 32 * grp;
  long r_end = r_chunk + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  f4 acc[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[q][t] = (f4){0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  float gv[4][8], av[4][8], gn[4][8], an[4][8];
  f4 ln_ga = (f4){1.f, 1.f, 1.f, 1.f}, ln_be = (f4){0.f, 0.f, 0.f, 0.f};
  if (LN) { ln_ga = *(const f4*)(ln_g + 16 * tc0 + 4 * n); ln_be = *(const f4*)(ln_b + 16 * tc0 + 4 * n); }
  auto load = [&](long r0, float (&go)[4][8], float (&ao)[4][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = r0 + 8 * g + j;
      const bool ok = row < r_end;
      const long rr = ok ? row : r_chunk;
      // one 16-byte load per row and operand: lane n takes columns 4n .. 4n+3 of the wave's 64-column half, i.e. MFMA tile q
      // holds column 4n + q (not 16q + n) — a permutation of the output channels that the final store undoes.  256 contiguous
      // bytes per row and instruction instead of four 64-byte pieces.
      const f4 gq = *(const f4*)(G + rr * NAMP_H + 16 * to0 + 4 * n);
      f4 aq = *(const f4*)(A + rr * NAMP_H + 16 * tc0 + 4 * n);
      if (LN) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 st = *(const f2*)(ln_stats + 2 * rr);
        aq = (aq - st.x) * st.y * ln_ga + ln_be;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) go[q][j] = ok ? gq[q] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) ao[t][j] = aq[t];
    }
  };
  if (r_begin < r_end) load(r_begin, gv, av);
  for (long r0 = r_begin; r0 < r_end; r0 += 32 * KS) {
    const bool more = r0 + 32 * KS < r_end;
    if (more) load(r0 + 32 * KS, gn, an);
    bf8 gh[4], gm[4], ah[4], am[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      split8(gv[q], gh[q], gm[q]);
      split8(av[q], ah[q], am[q]);
      bsum[q] += ((gv[q][0] + gv[q][1]) + (gv[q][2] + gv[q][3])) + ((gv[q][4] + gv[q][5]) + (gv[q][6] + gv[q][7]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (MID) {
          acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm[q], ah[t], acc[q][t], 0, 0, 0);
          acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], am[t], acc[q][t], 0, 0, 0);
        }
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], ah[t], acc[q][t], 0, 0, 0);
      }
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) { gv[q][j] = gn[q][j]; av[q][j] = an[q][j]; }
    }
  }
  if (KS == 2) {
    f4* rd = (f4*)ks_red + wave * 8 * 64 + lane;                    // [wave][8 tiles][64 lanes] x 16 B = 32 KiB, used twice
    float* rb = ks_red + 4 * 8 * 64 * 4 + wave * 4 * 64 + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (grp == 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) rd[(4 * q + t) * 64] = acc[2 * h + q][t];
        if (h == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rb[q * 64] = bsum[q];
        }
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[2 * h + q][t] += rd[(4 * q + t) * 64];
        if (h == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bsum[q] += rb[q * 64];
        }
      }
      if (h == 0) __syncthreads();
    }
    if (grp == 1) return;
  }
  float* out = dW_part + (long)blockIdx.x * NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // the four column tiles t of one output row are 4 consecutive floats (column 16 tc0 + 4 n + t): one 16-byte store, 256 bytes per 16 lanes
      f4* o = (f4*)(out + (16 * to0 + 4 * (4 * g + r) + q) * NAMP_H + 16 * tc0 + 4 * n);
      f4 v = (f4){acc[q][0][r], acc[q][1][r], acc[q][2][r], acc[q][3][r]};
      if (accumulate) v += *o;                                       // this workgroup's own slot: the sum over launches stays deterministic
      *o = v;
    }
  if (db_part && (wave & 1) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float s = xg_sum(bsum[q]);
      float* o = db_part + (long)blockIdx.x * NAMP_H + 16 * to0 + 4 * n + q;
      if (g == 0) *o = accumulate ? *o + s : s;
    }
  }
}

template <bool MID, int KS = 1>
__global__ __launch_bounds__(256 * KS) void wgrad_x3_kernel(const float* __restrict__ G, const float* __restrict__ A, long rows,
                                                            long rows_per_chunk, float* __restrict__ dW_part,
                                                            float* __restrict__ db_part) {
  wgrad_x3_body<MID, false, KS>(G, A, rows, rows_per_chunk, dW_part, db_part);
}
template <bool MID, int KS = 1>
__global__ __launch_bounds__(256 * KS) void wgrad_x3_ln_kernel(const float* __restrict__ G, const float* __restrict__ Y, const float* __restrict__ ln_stats,
                                                          const float* __restrict__ ln_g, const float* __restrict__ ln_b, long rows,
                                                          long rows_per_chunk, float* __restrict__ dW_part, float* __restrict__ db_part) {
  wgrad_x3_body<MID, true, KS>(G, Y, rows, rows_per_chunk, dW_part, db_part, false, ln_stats, ln_g, ln_b);
}

// Up to 8 contractions over the SAME rows in one launch (blockIdx.y = which): a residue tail's eight [24,000-row] weight-gradient
// blocks would otherwise be eight launches of 47 workgroups each.
struct WgradMulti { const float* G[8]; const float* A[8]; float* dW[8]; float* db[8]; };
template <bool MID, int KS = 1>
__global__ __launch_bounds__(256 * KS) void wgrad_x3_multi_kernel(const WgradMulti m, long rows, long rows_per_chunk, int accumulate) {
  // (accumulate: a launch over another slice of the rows already left its partials in dW / db)
  const float* G = m.G[0]; const float* A = m.A[0]; float* dW = m.dW[0]; float* db = m.db[0];
#pragma unroll
  for (int q = 1; q < 8; ++q)
    if ((int)blockIdx.y == q) { G = m.G[q]; A = m.A[q]; dW = m.dW[q]; db = m.db[q]; }     // static indices: no kernarg spill
  wgrad_x3_body<MID, false, KS>(G, A, rows, rows_per_chunk, dW, db, accumulate != 0);
}

// wgrad_bf16_kernel: the row contraction of the mixed-precision mode on bf16 row tensors (G always bf16; A bf16 — A1 / A2 —
// or fp32 — h_E, residue-level inputs): the loaded values ARE the MFMA operands (no split), 8 bytes per row, lane and operand.
template <bool A_BF16>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const __bf16* __restrict__ G, const void* __restrict__ Av, long rows,
                                                         long rows_per_chunk, float* __restrict__ dW_part, float* __restrict__ db_part) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int to0 = 4 * (wave >> 1), tc0 = 4 * (wave & 1);
  const long r_begin = (long)blockIdx.x * rows_per_chunk;
  long r_end = r_begin + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  f4 acc[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[q][t] = (f4){0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  bf8 gv[4], av[4], gn[4], an[4];
  auto load = [&](long r0, bf8 (&go)[4], bf8 (&ao)[4]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = r0 + 8 * g + j;
      const bool ok = row < r_end;
      const long rr = ok ? row : r_begin;
      const bf4 gq = *(const bf4*)(G + rr * NAMP_H + 16 * to0 + 4 * n);       // columns 4n .. 4n+3 of the wave's 64-column half
      bf4 aq;
      if (A_BF16) aq = *(const bf4*)((const __bf16*)Av + rr * NAMP_H + 16 * tc0 + 4 * n);
      else aq = to_bf4(*(const f4*)((const float*)Av + rr * NAMP_H + 16 * tc0 + 4 * n));
#pragma unroll
      for (int q = 0; q < 4; ++q) go[q][j] = ok ? gq[q] : (__bf16)0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) ao[t][j] = aq[t];
    }
  };
  if (r_begin < r_end) load(r_begin, gv, av);
  for (long r0 = r_begin; r0 < r_end; r0 += 32) {
    const bool more = r0 + 32 < r_end;
    if (more) load(r0 + 32, gn, an);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) sq += (float)gv[q][j];
      bsum[q] += sq;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gv[q], av[t], acc[q][t], 0, 0, 0);
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { gv[q] = gn[q]; av[q] = an[q]; }
    }
  }
  float* out = dW_part + (long)blockIdx.x * NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *(f4*)(out + (16 * to0 + 4 * (4 * g + r) + q) * NAMP_H + 16 * tc0 + 4 * n) = (f4){acc[q][0][r], acc[q][1][r], acc[q][2][r], acc[q][3][r]};
  if (db_part && (wave & 1) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float s_ = xg_sum(bsum[q]);
      if (g == 0) db_part[(long)blockIdx.x * NAMP_H + 16 * to0 + 4 * n + q] = s_;
    }
  }
}

// ------------------------------------------------------------------------------------------
// feat_wgrad_kernel: gradient of edge_embedding.weight [128 x 5200] (na_model_utils.py:499-507):
//     dW[o][c] = sum_e g_pre[e][o] * feat[e][c],   feat = [E_pos (16) | RBF(a, b, r) (18*18*16)]
// with the 5184 RBF features regenerated on the fly like the forward kernel does — they are never stored.
// Column block q: q = 0 the positional features, q = 1 + 18a + b the 16 RBFs of atom pair (a, b): the B operand of
// the row-contraction MFMA (see wgrad_kernel) for block q is exp(-((D_ab(e) - mu_n) / sigma)^2) on lane (n, g) —
// produced directly in operand layout, one exp per MFMA group.  A workgroup = 4 waves x 2 column blocks over one
// chunk of edges; the g_pre tile (64 edges x 128) is staged in LDS once per 8 blocks.  Partials per edge chunk.
// ------------------------------------------------------------------------------------------
#define FEATW_BLOCKS 325
#define FEATW_COLS 5200
#define FEATW_TILE 64
#define FEATW_LD 132            // padded row length of the LDS g_pre tile (floats)

// tile_presence_kernel: for every 64-edge tile, which of the 18 atoms occur on the residue side (word 0) and on the
// neighbour side (word 1) of any of its edges.  An (a, b) block can only be non-zero on a tile that has a on one side and
// b on the other; feat_wgrad_kernel reads these two words instead of voting across the workgroup for every tile.
// (ms: element stride between consecutive atoms' mask values — 1 for M18 [G][18], 4 for the mask lane of the packed [G][18][4] array)
static __global__ __launch_bounds__(256) void tile_presence_kernel(const float* __restrict__ M18, int ms, const int32_t* __restrict__ E_idx, long E,
                                                           int L, int K, int32_t* __restrict__ pres) {
  const int lane = threadIdx.x & 63;
  const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile * FEATW_TILE >= E) return;
  const long el = tile * FEATW_TILE + lane;
  uint32_t bi = 0, bj = 0;
  if (el < E) {
    const int node = (int)(el / K);
    const int j = node - node % L + E_idx[el];
    for (int q = 0; q < 18; ++q) {
      if (M18[((long)node * 18 + q) * ms] != 0.f) bi |= 1u << q;
      if (M18[((long)j * 18 + q) * ms] != 0.f) bj |= 1u << q;
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { bi |= __shfl_xor(bi, o); bj |= __shfl_xor(bj, o); }
  if (lane == 0) { pres[2 * tile] = (int32_t)bi; pres[2 * tile + 1] = (int32_t)bj; }
}

static __global__ __launch_bounds__(256) void feat_wgrad_kernel(const float* __restrict__ X18, const float* __restrict__ M18,
                                                         const int32_t* __restrict__ E_idx, const float* __restrict__ E_pos,
                                                         const float* __restrict__ g_pre, const int32_t* __restrict__ pres,
                                                         long E, long edges_per_chunk, int L, int K,
                                                         float* __restrict__ dW_part) {
  __shared__ float gt[FEATW_TILE * FEATW_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int blk0 = (blockIdx.x * 4 + wave) * 2;               // this wave's two column blocks
  const int wg_blk0 = blockIdx.x * 8;                         // the workgroup's eight
  const long e_begin = (long)blockIdx.y * edges_per_chunk;
  long e_end = e_begin + edges_per_chunk;
  if (e_end > E) e_end = E;
  const float mu = 2.0f + (float)n * (20.0f / 15.0f);
  int pa[2], pb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = blk0 + q - 1;                                // atom pair index, -1 = positional block
    pa[q] = p >= 0 ? p / 18 : 0;
    pb[q] = p >= 0 ? p % 18 : 0;
  }
  f4 acc[2][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[q][t] = (f4){0.f, 0.f, 0.f, 0.f};

  // block `blk` can be non-zero on a tile with presence words (pi, pj)?  (block 0 = positional: always)
  auto block_live = [](int blk, uint32_t pi, uint32_t pj) {
    if (blk >= FEATW_BLOCKS) return false;
    if (blk == 0) return true;
    const int p = blk - 1;
    return (((pi >> (p / 18)) & (pj >> (p % 18))) & 1u) != 0u;
  };
  bool staged = false;                                         // has this workgroup's LDS tile been read since its last staging?
  for (long e0 = e_begin; e0 < e_end; e0 += FEATW_TILE) {
    // Absent atoms make whole (a, b) blocks exactly zero for runs of edges (a protein residue has 5 of the 18 atoms).
    // Liveness comes from the per-tile presence words — workgroup-uniform, no vote: tiles on which none of the
    // workgroup's eight blocks is live cost two scalar loads; otherwise only the live blocks' MFMAs run.
    const uint32_t pi = (uint32_t)__builtin_amdgcn_readfirstlane(pres[2 * (e0 / FEATW_TILE)]);
    const uint32_t pj = (uint32_t)__builtin_amdgcn_readfirstlane(pres[2 * (e0 / FEATW_TILE) + 1]);
    bool any_live = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) any_live = any_live || block_live(wg_blk0 + q, pi, pj);
    if (!any_live) continue;
    const bool live0 = block_live(blk0, pi, pj), live1 = block_live(blk0 + 1, pi, pj);
    if (staged) __syncthreads();                               // the previous staged tile is fully consumed
    staged = true;
    // stage g_pre rows e0 .. e0+63 (zeros past the end)
    for (int idx = tid; idx < FEATW_TILE * 32; idx += 256) {
      const int row = idx >> 5, c4 = idx & 31;
      const long er = e0 + row;
      f4 v = (f4){0.f, 0.f, 0.f, 0.f};
      if (er < e_end) v = *(const f4*)(g_pre + er * NAMP_H + 4 * c4);
      *(f4*)(gt + row * FEATW_LD + 4 * c4) = v;
    }
    // per-lane edge of this tile: distance of the wave's atom pairs (masked pairs -> "infinitely far": RBF = 0)
    float dist[2] = {1e30f, 1e30f};
    if (live0 || live1) {
      const long el = e0 + lane;
      const bool eok = el < e_end;
      const long ec = eok ? el : e_begin;
      const int node = (int)(ec / K);
      const int j = node - node % L + E_idx[ec];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float* xi = X18 + ((long)node * 18 + pa[q]) * 3;
        const float* xj = X18 + ((long)j * 18 + pb[q]) * 3;
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        const float mk = M18[(long)node * 18 + pa[q]] * M18[(long)j * 18 + pb[q]];
        dist[q] = (eok && mk != 0.f) ? sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f) : 1e30f;
      }
    }
    __syncthreads();
    if (live0 || live1) {
#pragma unroll 4
      for (int s = 0; s < FEATW_TILE / 4; ++s) {
        const int row = 4 * s + g;                             // k-slot g of this MFMA group <-> edge e0 + 4s + g
        float av[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] = gt[row * FEATW_LD + 16 * t + n];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float b;
          if (blk0 + q == 0) {
            const long er = e0 + row;
            b = (er < e_end) ? E_pos[er * 16 + n] : 0.f;
          } else {
            const float d = __shfl(dist[q], row);
            const float u = (d - mu) * 0.8f;
            b = __expf(-(u * u));
          }
          if (q == 0 ? live0 : live1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[q][t] = mfma4(av[t], b, acc[q][t]);
          }
        }
      }
    }
  }
  // D[i = 4g + r][j = n] -> dW[16t + 4g + r][16 blk + n]
  float* out = dW_part + (long)blockIdx.y * NAMP_H * FEATW_COLS;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (blk0 + q >= FEATW_BLOCKS) continue;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(16 * t + 4 * g + r) * FEATW_COLS + 16 * (blk0 + q) + n] = acc[q][t][r];
  }
}

// feat_wgrad_x3_kernel: the same contraction as split-bf16 products on v_mfma_f32_16x16x32_bf16 (one MFMA step = 32 edges):
// the g_pre tile is split into bf16 hi / mid ONCE per workgroup while it is staged, transposed to [channel][edge] so that
// lane (n, g) reads its A operand — channel 16t + n, edges 8g .. 8g+7 of the step — as one 16-byte LDS read per plane; the
// RBF operand (8 edges per lane at the lane's centre mu_n) is generated and split in registers.  5.3x fewer MFMA cycles
// than the fp32 form; results differ from it by the split's 2^-16 per product.
#define FEATW_LDT 72            // bf16 elements per channel row of the transposed planes (64 edges + pad: conflict-free b128 writes)

// Round 4 form of the kernel (both precisions): (i) the tile's 64 distances reach the operand lanes through LDS (one write per lane and block, two 16-byte
// broadcast reads per 8 edges, instead of one ds_bpermute per element); (ii) the exponent is one subtraction, one product and a v_exp_f32:
// exp(-((D - mu) / 1.25)^2) = 2^(-(c D - c mu)^2), c = 0.8 sqrt(log2 e), with c D stored once per edge and block; (iii) bf16 mode (MID = false): all 32
// staging requests of a thread go out first, past-the-end rows clamped and zeroed afterwards (a `row < end` test around each load makes the compiler drain
// the in-order memory counter behind each), and a step's eight g_pre fragments are read once with each live block's eight products behind one branch.
// cfg5: 3.43 -> 3.18 ms (split-bf16), 3.03 -> 2.77 ms (bf16) per step.  Measured and NOT kept: (iii) in the split-bf16 instantiation (4.1-6.3 ms: the
// compiler serialises differently and loses a wave per SIMD); four blocks per wave — twice the products per staged tile — at half the occupancy (6.8 ms).
#ifndef FEATW_NBW
#define FEATW_NBW 2
#endif
#define FEATW_WG_BLOCKS (4 * FEATW_NBW)
#define FEATW_GRID_X ((FEATW_BLOCKS + FEATW_WG_BLOCKS - 1) / FEATW_WG_BLOCKS)

// PK: X18 is the PACKED atom array [G][18][4] = (x, y, z, mask) and M18 is unused: an atom of the gathered neighbour is then ONE 16-byte request per
// lane where the separate arrays take four scattered 4-byte ones — with every lane on its own cache line, those requests (16 per lane and tile,
// 1,024 line requests per wave) were what the launch waited for: removing them took 30 % off it, requesting them a tile ahead 2 % (profiles/r05c).
// S16: g_pre arrives as bf16 tiles in the operand order of the staged planes — g16 [tile][128 channels][64 edges] (split-bf16: a second array of
// the remainders behind it, g16_plane elements further) — written by the launch that produced g_pre (embed_ln_bwd_kernel).  Staging a tile is then four
// (eight) 16-byte copies per thread; from the fp32 rows it is 32 four-byte loads, 32 conversions and the transposition, repeated by each of the 41 column
// groups — HALF of this launch once its L2 misses and coordinate gathers were gone (ablation: 2.17 -> 1.06 ms without the staging; profiles/r05g).
template <bool MID, bool PK, bool S16 = false>      // MID = false: hi . hi products only (mixed-precision mode)
__device__ __forceinline__ void feat_wgrad_x3_body(const float* __restrict__ X18, const float* __restrict__ M18,
                                                   const int32_t* __restrict__ E_idx, const float* __restrict__ E_pos,
                                                   const float* __restrict__ g_pre, const int32_t* __restrict__ pres,
                                                   long E, long edges_per_chunk, int L, int K,
                                                   float* __restrict__ dW_part, const __bf16* __restrict__ g16, long g16_plane) {
  __shared__ __attribute__((aligned(16))) __bf16 gh[NAMP_H * FEATW_LDT];
  __shared__ __attribute__((aligned(16))) __bf16 gm[MID ? NAMP_H * FEATW_LDT : 8];
  __shared__ __attribute__((aligned(16))) float dsc[4][FEATW_NBW][FEATW_TILE];      // c * distance per (wave, block, edge of the tile)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // The FEATW_GRID_X column groups of one edge chunk read the same g_pre rows: dealt round-robin to the 8 XCDs (workgroup b -> XCD b % 8) every
  // private L2 fetched every chunk — 7.5 GB measured behind the L2s for 0.94 GB algorithmic.  With the XCD-contiguous order all column groups of
  // a chunk run on ONE XCD, about one and a half chunks at a time.
  const int lin = xcd_block_index((int)(blockIdx.y * FEATW_GRID_X + blockIdx.x), (int)(gridDim.y * FEATW_GRID_X));
  const int bx = lin % FEATW_GRID_X, by = lin / FEATW_GRID_X;
  const int blk0 = (bx * 4 + wave) * FEATW_NBW;
  const int wg_blk0 = bx * FEATW_WG_BLOCKS;
  const long e_begin = (long)by * edges_per_chunk;
  long e_end = e_begin + edges_per_chunk;
  if (e_end > E) e_end = E;
  constexpr float C = 0.9608979270291599f;                     // 0.8 * sqrt(log2 e)
  const float cmu = fmaf((float)n, 20.0f / 15.0f, 2.0f) * C;
  int pa[FEATW_NBW], pb[FEATW_NBW];
#pragma unroll
  for (int q = 0; q < FEATW_NBW; ++q) {
    const int p = blk0 + q - 1;
    pa[q] = p >= 0 ? p / 18 : 0;
    pb[q] = p >= 0 ? p % 18 : 0;
  }
  f4 acc[FEATW_NBW][8];
#pragma unroll
  for (int q = 0; q < FEATW_NBW; ++q)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[q][t] = (f4){0.f, 0.f, 0.f, 0.f};
  auto block_live = [](int blk, uint32_t pi, uint32_t pj) {
    if (blk >= FEATW_BLOCKS) return false;
    if (blk == 0) return true;
    const int p = blk - 1;
    return (((pi >> (p / 18)) & (pj >> (p % 18))) & 1u) != 0u;
  };
  const int sc = tid & 127, sh = tid >> 7;                     // staging: channel, edge half (32 edges) of this thread
  float* dw = &dsc[wave][0][0];
  bool staged = false;
  // Round 5 (profiles/r05c): what bounded this launch was neither its products, nor its exponentials, nor the staging of g_pre (each removed:
  // +-2 %) but two dependent round trips per tile in front of everything else — the tile's presence words (a load + v_readfirstlane: the wave
  // waits for memory before it knows whether the tile is live, 18,000 / 128 times per workgroup, skipped tiles included) and the neighbour index
  // -> coordinates chain of the distances (removed: -30 %).  Now: the presence words of 64 tiles sit one per lane and are handed out by
  // v_readlane; the next live tile's coordinates and the one after's neighbour index are requested a tile ahead.
  const long t_begin = e_begin / FEATW_TILE, t_end = (e_end + FEATW_TILE - 1) / FEATW_TILE;      // (chunks are whole tiles)
  long pw_base = -1;
  int pw_i = 0, pw_j = 0;
  auto presence = [&](const long t, uint32_t& pi, uint32_t& pj) {
    if (t - pw_base >= 64 || pw_base < 0) {
      pw_base = t;
      const long tl = t + lane < t_end ? t + lane : t_end - 1;
      pw_i = pres[2 * tl]; pw_j = pres[2 * tl + 1];
    }
    pi = (uint32_t)__builtin_amdgcn_readlane(pw_i, (int)(t - pw_base));
    pj = (uint32_t)__builtin_amdgcn_readlane(pw_j, (int)(t - pw_base));
  };
  auto next_live = [&](long t, uint32_t& pi, uint32_t& pj) {        // first tile at or behind t with a live block of this workgroup (t_end: none)
    for (; t < t_end; ++t) {
      presence(t, pi, pj);
      bool any_live = false;
#pragma unroll
      for (int q = 0; q < FEATW_WG_BLOCKS; ++q) any_live = any_live || block_live(wg_blk0 + q, pi, pj);
      if (any_live) break;
    }
    return t;
  };
  struct Coords { float xi[FEATW_NBW][3], xj[FEATW_NBW][3], mi[FEATW_NBW], mj[FEATW_NBW]; };
  auto edge_of = [&](const long t) { const long el = t * FEATW_TILE + lane; return el < e_end ? el : e_begin; };
  auto req_coords = [&](Coords& c, const long t, const int jrel) {   // (dead blocks included: no branch around the requests)
    const long ec = edge_of(t);
    const int node = (int)(ec / K);
    const int j = node - node % L + jrel;
#pragma unroll
    for (int q = 0; q < FEATW_NBW; ++q) {
      if constexpr (PK) {
        const f4 vi = *(const f4*)(X18 + ((long)node * 18 + pa[q]) * 4), vj = *(const f4*)(X18 + ((long)j * 18 + pb[q]) * 4);
        c.xi[q][0] = vi.x; c.xi[q][1] = vi.y; c.xi[q][2] = vi.z; c.mi[q] = vi.w;
        c.xj[q][0] = vj.x; c.xj[q][1] = vj.y; c.xj[q][2] = vj.z; c.mj[q] = vj.w;
      } else {
        const float* xi = X18 + ((long)node * 18 + pa[q]) * 3;
        const float* xj = X18 + ((long)j * 18 + pb[q]) * 3;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) { c.xi[q][cc] = xi[cc]; c.xj[q][cc] = xj[cc]; }
        c.mi[q] = M18[(long)node * 18 + pa[q]]; c.mj[q] = M18[(long)j * 18 + pb[q]];
      }
    }
  };
  uint32_t pi0, pj0, pi1, pj1, pi2, pj2;
  long t0 = next_live(t_begin, pi0, pj0);
  long t1 = t0 < t_end ? next_live(t0 + 1, pi1, pj1) : t_end;
  long t2 = t1 < t_end ? next_live(t1 + 1, pi2, pj2) : t_end;
  Coords c0, c1;
  int j1 = 0, j2 = 0;
  if (t0 < t_end) {
    req_coords(c0, t0, E_idx[edge_of(t0)]);
    if (t1 < t_end) j1 = E_idx[edge_of(t1)];
  }
  for (; t0 < t_end; ) {
    const long e0 = t0 * FEATW_TILE;
    const uint32_t pi = pi0, pj = pj0;
    // a tile ahead: the next live tile's coordinates (its neighbour index was requested an iteration ago), the one after's neighbour index
    if (t1 < t_end) req_coords(c1, t1, j1);
    if (t2 < t_end) j2 = E_idx[edge_of(t2)];
    bool live[FEATW_NBW], wave_live = false;
#pragma unroll
    for (int q = 0; q < FEATW_NBW; ++q) { live[q] = block_live(blk0 + q, pi, pj); wave_live = wave_live || live[q]; }
    if (staged) __syncthreads();
    staged = true;
    // stage + split + transpose g_pre rows e0 .. e0+63: this thread's channel, 4 groups of 8 edges
    if constexpr (S16) {
      const bf8* srcp = (const bf8*)(g16 + t0 * (long)(NAMP_H * FEATW_TILE));        // 1,024 pieces of 16 bytes: piece p = (channel p / 8, edges 8 (p % 8) ..)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        *(bf8*)(gh + (p >> 3) * FEATW_LDT + 8 * (p & 7)) = srcp[p];
        if constexpr (MID) *(bf8*)(gm + (p >> 3) * FEATW_LDT + 8 * (p & 7)) = ((const bf8*)(g16 + g16_plane + t0 * (long)(NAMP_H * FEATW_TILE)))[p];
      }
    } else if constexpr (MID) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long er = e0 + 32 * sh + 8 * i + j;
          v[j] = (er < e_end) ? g_pre[er * NAMP_H + sc] : 0.f;
        }
        bf8 hi, mid;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (__bf16)v[j]; mid[j] = (__bf16)(v[j] - (float)hi[j]); }
        *(bf8*)(gh + sc * FEATW_LDT + 32 * sh + 8 * i) = hi;
        *(bf8*)(gm + sc * FEATW_LDT + 32 * sh + 8 * i) = mid;
      }
    } else {
      float v[32];
      const long rb = e0 + 32 * sh;
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = g_pre[(rb + j < e_end ? rb + j : e_end - 1) * NAMP_H + sc];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf8 hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = (__bf16)((rb + 8 * i + j < e_end) ? v[8 * i + j] : 0.f);
        *(bf8*)(gh + sc * FEATW_LDT + 32 * sh + 8 * i) = hi;
      }
    }
    // this lane's edge of the tile: scaled distances of the wave's atom pairs (masked pairs -> "infinitely far": RBF = 0)
    if (wave_live) {
      const bool eok = e0 + lane < e_end;
#pragma unroll
      for (int q = 0; q < FEATW_NBW; ++q) {
        const float dx = c0.xi[q][0] - c0.xj[q][0], dy = c0.xi[q][1] - c0.xj[q][1], dz = c0.xi[q][2] - c0.xj[q][2];
        // (explicit fused chain: the instantiations of this body must agree to the bit, whatever each one's register budget makes of a * b + c)
        dw[q * FEATW_TILE + lane] = (eok && c0.mi[q] * c0.mj[q] != 0.f) ? C * sqrtf(fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, 1e-6f)))) : 1e30f;
      }
    }
    __syncthreads();
    if (wave_live) {
#pragma unroll
      for (int st = 0; st < FEATW_TILE / 32; ++st) {
        const int row0 = 32 * st + 8 * g;                      // this lane's 8 edges of the step
        bf8 bh[FEATW_NBW], bm[FEATW_NBW];
#pragma unroll
        for (int q = 0; q < FEATW_NBW; ++q) {
          if (!live[q]) continue;
          float b[8];
          if (blk0 + q == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const long er = e0 + row0 + j;
              b[j] = (er < e_end) ? E_pos[er * 16 + n] : 0.f;
            }
          } else {
            const f4 d0 = *(const f4*)(dw + q * FEATW_TILE + row0), d1 = *(const f4*)(dw + q * FEATW_TILE + row0 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = (j < 4 ? d0[j & 3] : d1[j & 3]) - cmu;
              b[j] = __builtin_amdgcn_exp2f(-(u * u));
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) { bh[q][j] = (__bf16)b[j]; if (MID) bm[q][j] = (__bf16)(b[j] - (float)bh[q][j]); }
        }
        if constexpr (MID) {
          // split-bf16: fragment by fragment (two 16-byte reads feed up to six products; holding all sixteen fragments costs a wave per SIMD)
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const bf8 ah = *(const bf8*)(gh + (16 * t + n) * FEATW_LDT + row0);
            const bf8 am = *(const bf8*)(gm + (16 * t + n) * FEATW_LDT + row0);
#pragma unroll
            for (int q = 0; q < FEATW_NBW; ++q) {
              if (!live[q]) continue;
              acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[q], acc[q][t], 0, 0, 0);
              acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[q], acc[q][t], 0, 0, 0);
              acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[q], acc[q][t], 0, 0, 0);
            }
          }
        } else {
          // bf16: the step's eight fragments once, then each live block's eight products behind ONE branch (a liveness test per product made
          // the compiler wait for every fragment read on its own)
          bf8 ah[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) ah[t] = *(const bf8*)(gh + (16 * t + n) * FEATW_LDT + row0);
#pragma unroll
          for (int q = 0; q < FEATW_NBW; ++q) {
            if (!live[q]) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[q], acc[q][t], 0, 0, 0);
          }
        }
      }
    }
    // rotate the look-ahead
    c0 = c1; j1 = j2;
    t0 = t1; pi0 = pi1; pj0 = pj1;
    t1 = t2; pi1 = pi2; pj1 = pj2;
    t2 = t1 < t_end ? next_live(t1 + 1, pi2, pj2) : t_end;
  }
  float* out = dW_part + (long)by * NAMP_H * FEATW_COLS;
#pragma unroll
  for (int q = 0; q < FEATW_NBW; ++q) {
    if (blk0 + q >= FEATW_BLOCKS) continue;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(16 * t + 4 * g + r) * FEATW_COLS + 16 * (blk0 + q) + n] = acc[q][t][r];
  }
}

template <bool MID, bool PK, bool S16 = false>
__global__ __launch_bounds__(256) void feat_wgrad_x3_kernel(const float* __restrict__ X18, const float* __restrict__ M18,
                                                            const int32_t* __restrict__ E_idx, const float* __restrict__ E_pos,
                                                            const float* __restrict__ g_pre, const int32_t* __restrict__ pres,
                                                            long E, long edges_per_chunk, int L, int K,
                                                            float* __restrict__ dW_part, const __bf16* __restrict__ g16 = nullptr,
                                                            long g16_plane = 0) {
  feat_wgrad_x3_body<MID, PK, S16>(X18, M18, E_idx, E_pos, g_pre, pres, E, edges_per_chunk, L, K, dW_part, g16, g16_plane);
}
#ifndef FEATW_T16_WAVES
#define FEATW_T16_WAVES 3
#endif
// The mixed-precision launch on bf16 operand tiles at three workgroups per CU (156 registers; at four — 128 registers — it spills 47 and runs 2.5 ms):
// the waves of the other workgroups cover the per-tile round trips (coordinates, LDS, two barriers) that are what is left of a tile visit once the
// staging is four copies.
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FEATW_T16_WAVES, FEATW_T16_WAVES)))
void feat_wgrad_t16_kernel(const float* __restrict__ X18, const int32_t* __restrict__ E_idx, const float* __restrict__ E_pos,
                           const int32_t* __restrict__ pres, long E, long edges_per_chunk, int L, int K, float* __restrict__ dW_part,
                           const __bf16* __restrict__ g16) {
  feat_wgrad_x3_body<false, true, true>(X18, nullptr, E_idx, E_pos, nullptr, pres, E, edges_per_chunk, L, K, dW_part, g16, 0);
}

// ------------------------------------------------------------------------------------------
// ln_rows_*_kernel: LayerNorm over the 128 channels of [rows][128] and its backward (norm_edges on the edge embedding,
// na_model_utils.py:509: the one [E,128]-sized LayerNorm of the training step that is not fused into an edge kernel; the
// stock op runs at ~1 TB/s on these 512-byte rows).  32 lanes x float4 per row, two rows per wave, grid-stride; eps = 1e-5,
// biased variance.  Backward: gx = rstd (g.gamma - mean(g.gamma) - xhat mean(g.gamma.xhat)); d(gamma) = sum g.xhat and
// d(beta) = sum g accumulate per thread over its rows and leave as per-workgroup partials dgb_part [groups][2][128].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

static __global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out, long rows) {
  const int c4 = threadIdx.x & 31;
  const f4 ga = *(const f4*)(gamma + 4 * c4), be = *(const f4*)(beta + 4 * c4);
  for (long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5); r < rows; r += (long)gridDim.x * 8) {
    f4 v = *(const f4*)(x + r * NAMP_H + 4 * c4);
    const float mean = half_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
    v -= mean;
    const float var = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 128.0f);
    const float rstd = rsqrtf(var + 1e-5f);
    *(f4*)(out + r * NAMP_H + 4 * c4) = v * rstd * ga + be;
  }
}

// embed_ln_bwd_kernel: backward of h_E = W_e . LayerNorm(y) + b_e (norm_edges + W_e, na_model_utils.py:509,598) through BOTH steps in one pass over the
// rows: g_E = W_e^T g (tile GEMM at the step's precision), then the LayerNorm backward against the statistics re-derived from the y row —
//     g_pre = rstd * (g_E gamma - mean(g_E gamma) - xhat mean(g_E gamma xhat)),   d gamma = sum g_E xhat,   d beta = sum g_E
// — where two launches (the W_e^T product writing g_E rows, ln_rows_bwd reading them back with y) moved 2.95 GB per cfg5 step, this one moves
// 1.77 GB (g and y in, g_pre out).  It also leaves (mean, rstd) per row for the weight-gradient contraction, which re-derives LayerNorm(y) from y
// (wgrad_x3_ln_kernel): the normalised rows E are never written — 590 MB less traffic in the forward pass and 0.55 GiB less alive through backward.
// Persistent workgroups of 8 waves, the W_e^T image resident in LDS; a wave walks 16-row tiles in the register-chain layout (lane (m, g): channels
// 16 t + 4 g + r of row m) and keeps its d gamma / d beta sums per (row slot, channel) in 64 registers until the end.
struct EmbedLnBwdArgs {
  const float* g;          // [E][128] dL/dh_E rows
  const float* y;          // [E][128] pre-LayerNorm rows
  const float* Wt_img;     // image of W_e^T at the launch's precision
  const float* ln_g;       // LayerNorm weight
  float* g_pre;            // [E][128] dL/dy
  __bf16* g16;             // optional: dL/dy once more as bf16 tiles [ceil(E / 64)][128][64] in the operand order of feat_wgrad_x3_kernel's staged planes
  long g16_plane;          //           (split-bf16: the remainders g - bf16(g) in a second array, g16_plane elements behind the first)
  float* stats;            // [E][2] mean, rstd
  float* dgb_part;         // [gridDim.x][2][128]: sum g_E xhat, sum g_E
  long E;
};
template <int PREC>
__global__ __launch_bounds__(512) void embed_ln_bwd_kernel(const EmbedLnBwdArgs a) {
  constexpr int IMG_KB = (PREC == 2) ? 32 : 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float colsum[8][2 * NAMP_H];           // one slot per wave, added in wave order at the end: deterministic
  __shared__ __attribute__((aligned(16))) float gam[NAMP_H];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  if (tid < NAMP_H) gam[tid] = a.ln_g[tid];
  dma_to_lds(smem, a.Wt_img, IMG_KB, wave, 8, lane);
  wait_dma_and_sync();
  const f4* w = (const f4*)smem + lane;
  f4 dga[8], dbe[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { dga[t] = (f4){0.f, 0.f, 0.f, 0.f}; dbe[t] = dga[t]; }
  const long ntile = (a.E + 15) / 16;
  for (long tile = (long)blockIdx.x * 8 + wave; tile < ntile; tile += (long)gridDim.x * 8) {
    asm volatile("" ::: "memory");                           // the image fragments are loop-invariant LDS reads: keep them out of registers
    const long e_raw = tile * 16 + m;
    const bool valid = e_raw < a.E;
    const long e = valid ? e_raw : (a.E - 1);
    f4 x[8], yv[8], acc[8];
    {
      const float* gs = a.g + e * NAMP_H + 4 * g;
      const float* ys = a.y + e * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { x[t] = *(const f4*)(gs + 16 * t); yv[t] = *(const f4*)(ys + 16 * t); }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { if (!valid) x[t] = (f4){0.f, 0.f, 0.f, 0.f}; acc[t] = (f4){0.f, 0.f, 0.f, 0.f}; }
    gemm128p<PREC, false>(acc, x, w);                        // acc = g_E = W_e^T g
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s1 += (yv[t].x + yv[t].y) + (yv[t].z + yv[t].w);
    const float mean = xg_sum(s1) * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { yv[t] -= mean; s2 += (yv[t].x * yv[t].x + yv[t].y * yv[t].y) + (yv[t].z * yv[t].z + yv[t].w * yv[t].w); }
    const float rstd = rsqrtf(xg_sum(s2) * (1.0f / 128.0f) + 1e-5f);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      yv[t] *= rstd;                                         // xhat
      dga[t] += acc[t] * yv[t];
      dbe[t] += acc[t];
      acc[t] *= *(const f4*)(gam + 16 * t + 4 * g);          // g_E gamma
      m1 += (acc[t].x + acc[t].y) + (acc[t].z + acc[t].w);
      m2 += (acc[t].x * yv[t].x + acc[t].y * yv[t].y) + (acc[t].z * yv[t].z + acc[t].w * yv[t].w);
    }
    m1 = xg_sum(m1) * (1.0f / 128.0f);
    m2 = xg_sum(m2) * (1.0f / 128.0f);
    if (valid) {
      float* dst = a.g_pre + e * NAMP_H + 4 * g;
      // the 16 edges of this wave are a quarter of a 64-edge tile of g16: channel c of edge m at [(tile / 4) * 128 + c][16 (tile % 4) + m]
      __bf16* d16 = a.g16 ? a.g16 + ((tile >> 2) * NAMP_H + 4 * g) * 64 + 16 * (tile & 3) + m : nullptr;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f4 v = (acc[t] - m1 - yv[t] * m2) * rstd;
        *(f4*)(dst + 16 * t) = v;
        if (d16) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const __bf16 hi = (__bf16)v[r];
            d16[(16 * t + r) * 64] = hi;
            if (PREC == 1) d16[a.g16_plane + (16 * t + r) * 64] = (__bf16)(v[r] - (float)hi);
          }
        }
      }
      if (g == 0) { a.stats[2 * e] = mean; a.stats[2 * e + 1] = rstd; }
    }
  }
  // ---- the wave's sums over its row slots m (butterfly), then over the workgroup's waves in wave order
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float cw = dga[t][r], cb = dbe[t][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { cw += __shfl_xor(cw, o); cb += __shfl_xor(cb, o); }
      if (m == 0) { colsum[wave][16 * t + 4 * g + r] = cw; colsum[wave][NAMP_H + 16 * t + 4 * g + r] = cb; }
    }
  __syncthreads();
  if (tid < 2 * NAMP_H) {
    float s_ = colsum[0][tid];
#pragma unroll
    for (int w2 = 1; w2 < 8; ++w2) s_ += colsum[w2][tid];
    a.dgb_part[(long)blockIdx.x * 2 * NAMP_H + tid] = s_;
  }
}


static __global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ gamma, float* __restrict__ gx,
                                                          float* __restrict__ dgb_part, long rows) {
  __shared__ float red[8][2][128];
  const int c4 = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const f4 ga = *(const f4*)(gamma + 4 * c4);
  f4 dg = (f4){0.f, 0.f, 0.f, 0.f}, db = (f4){0.f, 0.f, 0.f, 0.f};
  for (long r = (long)blockIdx.x * 8 + sub; r < rows; r += (long)gridDim.x * 8) {
    f4 v = *(const f4*)(x + r * NAMP_H + 4 * c4);
    const f4 gr = *(const f4*)(g + r * NAMP_H + 4 * c4);
    const float mean = half_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
    v -= mean;
    const float var = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 128.0f);
    const float rstd = rsqrtf(var + 1e-5f);
    const f4 xh = v * rstd;
    const f4 gg = gr * ga;
    const float m1 = half_wave_sum((gg.x + gg.y) + (gg.z + gg.w)) * (1.0f / 128.0f);
    const float m2 = half_wave_sum((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.0f / 128.0f);
    *(f4*)(gx + r * NAMP_H + 4 * c4) = (gg - m1 - xh * m2) * rstd;
    dg += gr * xh;
    db += gr;
  }
  *(f4*)&red[sub][0][4 * c4] = dg;
  *(f4*)&red[sub][1][4 * c4] = db;
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][which][c];
    dgb_part[((long)blockIdx.x * 2 + which) * NAMP_H + c] = s;
  }
}

// ------------------------------------------------------------------------------------------
// scatter_rows_kernel: dL/dPj[j] = sum over the edges e that gathered table row j of G1[e] — the transpose of the
// neighbour gather, evaluated as a GATHER over the reverse adjacency (edges sorted by target once per step, shared by
// all nine per-edge stages): one wave per target row streams its ~K incoming 512-B rows and sums them in registers.
// Deterministic, no atomics (the atomic scatter inside edge_chain_bwd_kernel cost 0.85 ms per launch at cfg5, this
// 0.2 ms).  sel (DecLayer): per edge 1 = the row gathered Pbw -> out0, 0 = Pfw -> out1.
// ------------------------------------------------------------------------------------------
template <bool BF>        // BF: G1 is a bf16 row tensor (mixed-precision mode)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ G1, const int32_t* __restrict__ rev_edge,
                                                           const int32_t* __restrict__ rev_off, const uint8_t* __restrict__ sel,
                                                           float* __restrict__ out0, float* __restrict__ out1, int G) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= G) return;
  const int beg = rev_off[j], end = rev_off[j + 1];
  if constexpr (BF) {
    // bf16 rows are 256 bytes: FOUR edges per step, 16 lanes x 16 bytes (8 channels) each — with 8 bytes per lane (the fp32 form's lane map) a
    // load instruction moved 512 bytes and the launch ran at 3.4 TB/s of its 0.32 GB
    typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
    const int quarter = lane >> 4, c8 = lane & 15;
    float a0[8], a1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const __bf16* Gb = (const __bf16*)G1;
    int p = beg + quarter;
    for (; p + 12 < end; p += 16) {
      int e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = rev_edge[p + 4 * u];
      bf8v v[4];
      bool first[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[u] = *(const bf8v*)(Gb + (long)e[u] * NAMP_H + 8 * c8); first[u] = !sel || sel[e[u]]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 8; ++r) { if (first[u]) a0[r] += (float)v[u][r]; else a1[r] += (float)v[u][r]; }
    }
    for (; p < end; p += 4) {
      const int e = rev_edge[p];
      const bf8v v = *(const bf8v*)(Gb + (long)e * NAMP_H + 8 * c8);
      const bool first = !sel || sel[e];
#pragma unroll
      for (int r = 0; r < 8; ++r) { if (first) a0[r] += (float)v[r]; else a1[r] += (float)v[r]; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      a0[r] += __shfl_xor(a0[r], 16); a0[r] += __shfl_xor(a0[r], 32);
      if (out1) { a1[r] += __shfl_xor(a1[r], 16); a1[r] += __shfl_xor(a1[r], 32); }
    }
    if (quarter == 0) {
      float* o0 = out0 + (long)j * NAMP_H + 8 * c8;
      *(f4*)o0 = (f4){a0[0], a0[1], a0[2], a0[3]}; *(f4*)(o0 + 4) = (f4){a0[4], a0[5], a0[6], a0[7]};
      if (out1) {
        float* o1 = out1 + (long)j * NAMP_H + 8 * c8;
        *(f4*)o1 = (f4){a1[0], a1[1], a1[2], a1[3]}; *(f4*)(o1 + 4) = (f4){a1[4], a1[5], a1[6], a1[7]};
      }
    }
    return;
  }
  const int half = lane >> 5, c4 = lane & 31;                 // two edges per step, 32 lanes x float4 each
  f4 s0 = (f4){0.f, 0.f, 0.f, 0.f}, s1 = (f4){0.f, 0.f, 0.f, 0.f};
  // four steps' edge numbers, then their four rows (and selectors), are requested together: the loop was two dependent round trips per step
  int p = beg + half;
  for (; p + 6 < end; p += 8) {
    int e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = rev_edge[p + 2 * u];
    f4 v[4];
    bool first[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = ld_row4<BF>(G1, (long)e[u] * NAMP_H + 4 * c4); first[u] = !sel || sel[e[u]]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (first[u]) s0 += v[u]; else s1 += v[u]; }
  }
  for (; p < end; p += 2) {
    const int e = rev_edge[p];
    const f4 v = ld_row4<BF>(G1, (long)e * NAMP_H + 4 * c4);
    if (!sel || sel[e]) s0 += v; else s1 += v;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { s0[r] += __shfl_xor(s0[r], 32); s1[r] += __shfl_xor(s1[r], 32); }
  if (half == 0) {
    *(f4*)(out0 + (long)j * NAMP_H + 4 * c4) = s0;
    if (out1) *(f4*)(out1 + (long)j * NAMP_H + 4 * c4) = s1;
  }
}


// ------------------------------------------------------------------------------------------
// Residue tail of EncLayer / DecLayer in TRAINING (na_model_utils.py:236-247, 268-283), fused per workgroup:
//     x1  = LayerNorm1(h_V + dropout1(dh))                    dh = sum_k message / 30 (from the edge kernels)
//     z   = W_in x1 + b_in ;  f = W_out gelu(z) + b_out
//     out = mask * LayerNorm2(x1 + dropout2(f))
// tail_train_fwd_kernel: T 16-row tiles per workgroup, 8 waves — wave w owns hidden units 64w .. 64w+63 through both GEMMs
// (x3 images in registers, applied to all T tiles; partial W_out outputs reduced through LDS), exactly the schedule of
// node_update_multi_kernel<T, true>; both dropouts are counter-based hashes of (seed, row, channel) regenerated by the
// backward launch.  It keeps x1, z (block-major [4][G][128]) and y = x1 + dropout2(f) for the backward launch.
// tail_train_bwd_kernel: the same schedule run backwards — LayerNorm2 backward on the rows, g_h = W_out^T g_f (image of
// W_out^T, shape of W_in's), g_z = g_h * gelu'(z), g_x1 = g_y + W_in^T g_z (image of W_in^T, shape of W_out's, partials through
// LDS), LayerNorm1 backward; writes dL/dh_V, dL/d(dh), and the row tensors g_f, g_z, h = gelu(z) (block-major) from which the
// row-contraction kernel forms dW_in, dW_out, db_in, db_out; LayerNorm weight / bias gradients as per-workgroup partial sums.
// ------------------------------------------------------------------------------------------
struct TailTrainArgs {
  const float* hV; const float* dh; const int32_t* mask;
  const float* ln1_g; const float* ln1_b; const float* b_in; const float* b_out; const float* ln2_g; const float* ln2_b;
  const float* WA_ximg;          // fwd: W_in  [512 x 128];  bwd: W_out^T [512 x 128]
  const float* WB_ximg;          // fwd: W_out [128 x 512];  bwd: W_in^T  [128 x 512]
  float* out; float* x1; float* z; float* y;                       // fwd outputs (bwd: x1, z, y are inputs)
  const float* g_out; float* g_hV; float* g_dh; float* g_f; float* g_z; float* h; float* part;   // bwd
  uint32_t drop_thresh, seed1, seed2; float drop_scale;
  int G;
};

#ifndef FFN_LD
#define FFN_LD 132             // padded row stride (floats) of the LDS tiles (as in namp_kernels.h)
#endif
#ifndef TAIL_T
#define TAIL_T 3             // 48 rows per pass over the 1 MB of FFN images a workgroup pulls from L2 (2: 0.89 ms per cfg5 step in the six fwd + bwd launches; 3: 0.81; 4 spills)
#endif
#define TAIL_LDS (((2 * TAIL_T * 16) + 8 * 16) * FFN_LD * 4)

// per-wave GEMM "A": out[q][4 tiles of this wave's 64 hidden units] = W (512 x 128 image) . rows of tile q (from LDS, fp32)
template <int T>
__device__ __forceinline__ void tail_gemm_A(f4 (&hacc)[T][4], const float* rows, const float* img, const int wave, const int lane) {
  const int m = lane & 15, g = lane >> 4;
  bf8 wh[4][4], wm[4][4];
  const bf8* w = (const bf8*)img + (4 * wave) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) { wh[s][tn] = w[(s * 32 + tn) * 64]; wm[s][tn] = w[512 * 128 / 8 + (s * 32 + tn) * 64]; }
#pragma unroll
  for (int q = 0; q < T; ++q) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* xr = rows + (q * 16 + m) * FFN_LD + 32 * s + 4 * g;
      bf8 hi, mid;
      split_x3(*(const f4*)xr, *(const f4*)(xr + 16), hi, mid);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) hacc[q][tn] = mfma_x3(wh[s][tn], wm[s][tn], hi, mid, hacc[q][tn]);
    }
  }
}

// per-wave GEMM "B" for ONE tile: partial[16 x 128] = W (128 x 512 image, this wave's two K-steps) . v (this wave's 64 units)
__device__ __forceinline__ void tail_gemm_B(f4 (&oacc)[8], const f4 (&v)[4], const bf8 (&woh)[2][8], const bf8 (&wom)[2][8]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf8 hi, mid;
    split_x3(v[2 * s], v[2 * s + 1], hi, mid);
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) oacc[tn] = mfma_x3(woh[s][tn], wom[s][tn], hi, mid, oacc[tn]);
  }
}

__device__ __forceinline__ void tail_load_B(bf8 (&woh)[2][8], bf8 (&wom)[2][8], const float* img, const int wave, const int lane) {
  const bf8* w = (const bf8*)img + (2 * wave) * 8 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) { woh[s][tn] = w[(s * 8 + tn) * 64]; wom[s][tn] = w[128 * 512 / 8 + (s * 8 + tn) * 64]; }
}

static __global__ __launch_bounds__(512) void tail_train_fwd_kernel(const TailTrainArgs a) {
  constexpr int T = TAIL_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = (float*)smem;                       // [T][16][FFN_LD]  x1 = LN1(...)
  float* ys = xs + T * 16 * FFN_LD;               // (unused in forward)
  float* ps = ys + T * 16 * FFN_LD;               // [8][16][FFN_LD]  per-wave partial FFN outputs of the tile in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16 * T;
  // ---- phase 0: pre = h_V + dropout1(dh), LayerNorm1, one wave per tile
  for (int q = wave; q < T; q += 8) {
    const int row = row0 + 16 * q + m;
    const int rr = row < a.G ? row : (a.G - 1);
    f4 x[8];
    const float* src = a.hV + (long)rr * NAMP_H + 4 * g;
    const float* dsrc = a.dh + (long)rr * NAMP_H + 4 * g;
    const uint32_t key = drop_row_key(a.seed1, rr);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f4 d = *(const f4*)(dsrc + 16 * c);
      if (a.drop_thresh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] *= drop_factor(key, 16 * c + 4 * g + r, a.drop_thresh, a.drop_scale);
      }
      x[c] = *(const f4*)(src + 16 * c) + d;
    }
    layernorm_row_T(x, a.ln1_g, a.ln1_b, g);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      *(f4*)(xs + (q * 16 + m) * FFN_LD + 16 * c + 4 * g) = x[c];
      if (row < a.G) *(f4*)(a.x1 + (long)row * NAMP_H + 16 * c + 4 * g) = x[c];
    }
  }
  __syncthreads();
  // ---- phase A: z = W_in x1 + b_in (kept for the backward launch), hidden = gelu(z)
  f4 hacc[T][4];
#pragma unroll
  for (int q = 0; q < T; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) hacc[q][c] = *(const f4*)(a.b_in + 64 * wave + 16 * c + 4 * g);
  tail_gemm_A<T>(hacc, xs, a.WA_ximg, wave, lane);
  const long blk_off = (long)(wave >> 1) * a.G * NAMP_H + 64 * (wave & 1);      // block-major [4][G][128]
#pragma unroll
  for (int q = 0; q < T; ++q) {
    const int row = row0 + 16 * q + m;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (row < a.G) *(f4*)(a.z + blk_off + (long)row * NAMP_H + 16 * c + 4 * g) = hacc[q][c];
      hacc[q][c] = gelu4(hacc[q][c]);
    }
  }
  // ---- phase B: f = W_out hidden + b_out, y = x1 + dropout2(f), LayerNorm2, mask
  {
    bf8 woh[2][8], wom[2][8];
    tail_load_B(woh, wom, a.WB_ximg, wave, lane);
#pragma unroll
    for (int q = 0; q < T; ++q) {
      f4 oacc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) oacc[c] = (f4){0.f, 0.f, 0.f, 0.f};
      tail_gemm_B(oacc, hacc[q], woh, wom);
      float* dst = ps + (wave * 16 + m) * FFN_LD + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = oacc[c];
      __syncthreads();
      {   // thread -> (row = tid/32, 4 channels)
        const int r = tid >> 5, c = (tid & 31) * 4;
        const int orow = row0 + 16 * q + r;
        const bool ok = orow < a.G;
        f4 f = *(const f4*)(a.b_out + c);
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) f += *(const f4*)(ps + (w2 * 16 + r) * FFN_LD + c);
        if (a.drop_thresh) {
          const uint32_t key = drop_row_key(a.seed2, ok ? orow : 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] *= drop_factor(key, c + e, a.drop_thresh, a.drop_scale);
        }
        f4 v = *(const f4*)(xs + (q * 16 + r) * FFN_LD + c) + f;
        if (ok) *(f4*)(a.y + (long)orow * NAMP_H + c) = v;
        float s_ = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s_ += __shfl_xor(s_, o);
        const float mean = s_ * (1.0f / 128.0f);
        v -= mean;
        float qq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) qq += __shfl_xor(qq, o);
        const float rstd = rsqrtf(qq * (1.0f / 128.0f) + 1e-5f);
        const float mk = (a.mask && ok) ? (float)a.mask[orow] : 1.0f;
        const f4 o4 = (v * rstd * *(const f4*)(a.ln2_g + c) + *(const f4*)(a.ln2_b + c)) * mk;
        if (ok) *(f4*)(a.out + (long)orow * NAMP_H + c) = o4;
      }
      __syncthreads();
    }
  }
}

static __global__ __launch_bounds__(512) void tail_train_bwd_kernel(const TailTrainArgs a) {
  constexpr int T = TAIL_T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* gf = (float*)smem;                       // [T][16][FFN_LD]  g_f = dropout2-mask * g_y
  float* gy = gf + T * 16 * FFN_LD;               // [T][16][FFN_LD]  g_y = dL/dy
  float* ps = gy + T * 16 * FFN_LD;               // [8][16][FFN_LD]  per-wave partials of W_in^T g_z
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16 * T;
  const int r = tid >> 5, c = (tid & 31) * 4;     // row layout: thread -> (row, 4 channels)
  f4 dg2 = (f4){0.f, 0.f, 0.f, 0.f}, db2 = dg2, dg1 = dg2, db1 = dg2;
  // ---- phase 0: LayerNorm2 backward on the rows; g_f = dropout2 mask * g_y
#pragma unroll
  for (int q = 0; q < T; ++q) {
    const int orow = row0 + 16 * q + r;
    const bool ok = orow < a.G;
    const long ro = (long)(ok ? orow : 0) * NAMP_H + c;
    f4 v = *(const f4*)(a.y + ro);
    const float mk = a.mask ? (float)a.mask[ok ? orow : 0] : 1.0f;
    f4 go = ok ? *(const f4*)(a.g_out + ro) * mk : (f4){0.f, 0.f, 0.f, 0.f};
    float s_ = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) s_ += __shfl_xor(s_, o);
    const float mean = s_ * (1.0f / 128.0f);
    v -= mean;
    float qq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) qq += __shfl_xor(qq, o);
    const float rstd = rsqrtf(qq * (1.0f / 128.0f) + 1e-5f);
    const f4 xhat = v * rstd;
    dg2 += go * xhat; db2 += go;
    const f4 gg = go * *(const f4*)(a.ln2_g + c);
    float m1 = (gg.x + gg.y) + (gg.z + gg.w), m2 = (gg.x * xhat.x + gg.y * xhat.y) + (gg.z * xhat.z + gg.w * xhat.w);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
    m1 *= (1.0f / 128.0f); m2 *= (1.0f / 128.0f);
    const f4 gyv = (gg - m1 - xhat * m2) * rstd;
    f4 gfv = gyv;
    if (a.drop_thresh) {
      const uint32_t key = drop_row_key(a.seed2, ok ? orow : 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) gfv[e] *= drop_factor(key, c + e, a.drop_thresh, a.drop_scale);
    }
    *(f4*)(gy + (q * 16 + r) * FFN_LD + c) = gyv;
    *(f4*)(gf + (q * 16 + r) * FFN_LD + c) = gfv;
    if (ok) *(f4*)(a.g_f + ro) = gfv;
  }
  __syncthreads();
  // ---- phase A: g_h = W_out^T g_f (this wave's 64 hidden units); g_z = g_h * gelu'(z); h = gelu(z)
  f4 gz[T][4];
#pragma unroll
  for (int q = 0; q < T; ++q)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) gz[q][cc] = (f4){0.f, 0.f, 0.f, 0.f};
  tail_gemm_A<T>(gz, gf, a.WA_ximg, wave, lane);
  const long blk_off = (long)(wave >> 1) * a.G * NAMP_H + 64 * (wave & 1);
#pragma unroll
  for (int q = 0; q < T; ++q) {
    const int row = row0 + 16 * q + m;
    const bool ok = row < a.G;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const long off = blk_off + (long)(ok ? row : 0) * NAMP_H + 16 * cc + 4 * g;
      f4 zz = *(const f4*)(a.z + off);
      const f4 hv = gelu_split4(zz);                 // zz <- gelu'(z)
      gz[q][cc] = ok ? gz[q][cc] * zz : (f4){0.f, 0.f, 0.f, 0.f};
      if (ok) { *(f4*)(a.g_z + off) = gz[q][cc]; *(f4*)(a.h + off) = hv; }
    }
  }
  // ---- phase B: g_x1 = g_y + W_in^T g_z, LayerNorm1 backward
  {
    bf8 woh[2][8], wom[2][8];
    tail_load_B(woh, wom, a.WB_ximg, wave, lane);
#pragma unroll
    for (int q = 0; q < T; ++q) {
      f4 oacc[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) oacc[cc] = (f4){0.f, 0.f, 0.f, 0.f};
      tail_gemm_B(oacc, gz[q], woh, wom);
      float* dst = ps + (wave * 16 + m) * FFN_LD + 4 * g;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) *(f4*)(dst + 16 * cc) = oacc[cc];
      __syncthreads();
      {
        const int orow = row0 + 16 * q + r;
        const bool ok = orow < a.G;
        const long ro = (long)(ok ? orow : 0) * NAMP_H + c;
        f4 gx = *(const f4*)(gy + (q * 16 + r) * FFN_LD + c);
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) gx += *(const f4*)(ps + (w2 * 16 + r) * FFN_LD + c);
        // recompute pre = h_V + dropout1(dh) and its LayerNorm1 statistics
        f4 d = *(const f4*)(a.dh + ro);
        f4 dm = (f4){1.f, 1.f, 1.f, 1.f};
        if (a.drop_thresh) {
          const uint32_t key = drop_row_key(a.seed1, ok ? orow : 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) dm[e] = drop_factor(key, c + e, a.drop_thresh, a.drop_scale);
        }
        f4 v = *(const f4*)(a.hV + ro) + d * dm;
        float s_ = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s_ += __shfl_xor(s_, o);
        const float mean = s_ * (1.0f / 128.0f);
        v -= mean;
        float qq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) qq += __shfl_xor(qq, o);
        const float rstd = rsqrtf(qq * (1.0f / 128.0f) + 1e-5f);
        const f4 xhat = v * rstd;
        if (!ok) gx = (f4){0.f, 0.f, 0.f, 0.f};
        dg1 += gx * xhat; db1 += gx;
        const f4 gg = gx * *(const f4*)(a.ln1_g + c);
        float m1 = (gg.x + gg.y) + (gg.z + gg.w), m2 = (gg.x * xhat.x + gg.y * xhat.y) + (gg.z * xhat.z + gg.w * xhat.w);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        m1 *= (1.0f / 128.0f); m2 *= (1.0f / 128.0f);
        const f4 gp = (gg - m1 - xhat * m2) * rstd;
        if (ok) { *(f4*)(a.g_hV + ro) = gp; *(f4*)(a.g_dh + ro) = gp * dm; }
      }
      __syncthreads();
    }
  }
  // ---- LayerNorm weight / bias gradients: this thread holds the sums over its T rows; add the 16 row-threads of a channel in
  // a fixed order through LDS (ps is free): [kind][row-thread][128] -> thread (kind = tid / 128, channel = tid % 128)
  float* red = ps;
  *(f4*)(red + (0 * 16 + r) * NAMP_H + c) = dg2; *(f4*)(red + (1 * 16 + r) * NAMP_H + c) = db2;
  *(f4*)(red + (2 * 16 + r) * NAMP_H + c) = dg1; *(f4*)(red + (3 * 16 + r) * NAMP_H + c) = db1;
  __syncthreads();
  {
    const int kind = tid >> 7, ch = tid & 127;
    float s_ = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) s_ += red[(kind * 16 + rr) * NAMP_H + ch];
    a.part[(long)blockIdx.x * 4 * NAMP_H + tid] = s_;
  }
}

// ------------------------------------------------------------------------------------------
// Round 3: the loss and the optimiser step as launches of this library.
//
// loss_smoothed_kernel — label-smoothed negative log-likelihood per residue in fp64 (na_model_utils.py:111-146): target =
// one-hot(S) (or the aligned position-probability row where ppm_mask is set, :134), scaled by (1 - weight) on every token that
// belongs to some polymer's residue-type list (:140), plus the per-polymer smoothing mass eps = polymer_mask * restype_mask *
// weight / |list| (:136-139, float32 like the reference's tensors); loss_i = -sum_v target_v * log_probs[i][v].  BWD: the same
// target, d log_probs[i][v] = -target_v * g_i with g_i = dL/d loss_i (the caller's sum(loss * mask) / tokens stays a torch op, so
// autograd supplies g_i = mask_i / tokens * dL).  One thread per residue; V <= 64.
// ------------------------------------------------------------------------------------------
struct LossArgs {
  const int32_t* S; const float* log_probs; const float* pm[3]; const float* rm[3]; float eps_scale[3];   // weight / |list| as float32
  double one_minus_w; const int32_t* ppm_mask; const double* aligned_ppm; long G; int V;
};
template <bool BWD>
__global__ __launch_bounds__(256) void loss_smoothed_kernel(const LossArgs a, double* __restrict__ loss, const double* __restrict__ g_loss,
                                                            float* __restrict__ g_lp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.G) return;
  const int s_i = a.S[i];
  const bool ppm = a.ppm_mask && a.ppm_mask[i] != 0;
  const float p0 = a.pm[0][i], p1 = a.pm[1][i], p2 = a.pm[2][i];
  const double gi = BWD ? g_loss[i] : 0.0;
  double acc = 0.0;
  for (int v = 0; v < a.V; ++v) {
    const float r0 = a.rm[0][v], r1 = a.rm[1][v], r2 = a.rm[2][v];
    double t = ppm ? a.aligned_ppm[i * a.V + v] : (v == s_i ? 1.0 : 0.0);
    if ((r0 + r1 + r2) != 0.f) t *= a.one_minus_w;
    const float eps = ((p0 * r0) * a.eps_scale[0] + (p1 * r1) * a.eps_scale[1]) + (p2 * r2) * a.eps_scale[2];
    t += (double)eps;
    if (BWD) g_lp[i * a.V + v] = (float)(-t * gi);
    else acc += t * (double)a.log_probs[i * a.V + v];
  }
  if (!BWD) loss[i] = -acc;
}

// ------------------------------------------------------------------------------------------
// Multi-tensor gradient clip + Adam (torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step, na_run.py:233-238 / na_model_utils.py:
// 648-686): every parameter tensor in ONE launch.  The plan (static per model): block b works on elements [off[b], off[b] + 2048) of
// tensor t[b]; per step a table of the tensors' param / grad / exp_avg / exp_avg_sq pointers.  adam_sqnorm_kernel: per-block sums of
// grad^2; adam_coef_kernel: total norm and the clip coefficient min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1);
// adam_step_kernel: grad *= coef (written back, like clip_grad_norm_), exp_avg += (grad - exp_avg)(1 - beta1), exp_avg_sq =
// exp_avg_sq beta2 + (1 - beta2) grad^2, param -= step_size * exp_avg / (sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps) — torch's
// operation order in fp32.
// ------------------------------------------------------------------------------------------
#define NAMP_ADAM_CHUNK 2048
struct AdamPlan {
  const int32_t* blk_tensor; const long long* blk_off; const long long* numel;      // [nblocks], [nblocks], [ntensors]
  const unsigned long long* ptrs;                                                      // [4][ntensors]: param, grad, exp_avg, exp_avg_sq
  int ntensors, nblocks;
};
static __global__ __launch_bounds__(256) void adam_sqnorm_kernel(const AdamPlan p, float* __restrict__ partial) {
  const int b = blockIdx.x;
  const int t = p.blk_tensor[b];
  const long long off = p.blk_off[b], n = p.numel[t];
  const float* g = (const float*)p.ptrs[1 * p.ntensors + t];
  float s = 0.f;
  for (int e = threadIdx.x; e < NAMP_ADAM_CHUNK && off + e < n; e += 256) { const float v = g[off + e]; s = fmaf(v, v, s); }
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[b] = (red[0] + red[1]) + (red[2] + red[3]);
}
static __global__ __launch_bounds__(256) void adam_coef_kernel(const float* __restrict__ partial, int nblocks, float max_norm, float* __restrict__ out2) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[b];
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    float coef = 1.0f;
    if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.0f) coef = 1.0f; }
    out2[0] = norm; out2[1] = coef;
  }
}
static __global__ __launch_bounds__(256) void adam_step_kernel(const AdamPlan p, const float* __restrict__ coef2, float one_m_beta1, float beta2,
                                                               float one_m_beta2, float step_size, float bc2_sqrt, float eps) {
  const int b = blockIdx.x;
  const int t = p.blk_tensor[b];
  const long long off = p.blk_off[b], n = p.numel[t];
  float* w = (float*)p.ptrs[0 * p.ntensors + t];
  float* g = (float*)p.ptrs[1 * p.ntensors + t];
  float* m = (float*)p.ptrs[2 * p.ntensors + t];
  float* v = (float*)p.ptrs[3 * p.ntensors + t];
  const float coef = coef2 ? coef2[1] : 1.0f;
  for (int e = threadIdx.x; e < NAMP_ADAM_CHUNK && off + e < n; e += 256) {
    const long long i = off + e;
    float gi = g[i];
    if (coef != 1.0f) { gi *= coef; g[i] = gi; }
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * one_m_beta1;                             // lerp_ (weight = float(1 - beta1), the host's double rounded)
    vi = vi * beta2 + (one_m_beta2 * gi) * gi;                     // mul_ + addcmul_
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    m[i] = mi; v[i] = vi;
    w[i] = w[i] + (-step_size) * (mi / denom);                     // addcdiv_
  }
}

// ------------------------------------------------------------------------------------------
// reduce_sum_kernel — the sums over per-workgroup / per-chunk / per-tile partials that follow nearly every backward launch
// (dW partials [n][128][128], db / d ln partials [n][128], per-tile dL/dPa rows [G][K/16][128], per-tile K-sums of the forward), up to
// 16 of them in ONE launch instead of one stock reduction each (~100 per cfg5 step, 1.3 ms).  Segment s: dst[a * Mb + b] =
// sum_{i < n} src[a * sa + i * sn + b], a < A, b < Mb.  A workgroup owns 256 consecutive outputs of one segment; vector segments (Mb, sa, sn
// multiples of 4): 64 lanes x 16 bytes across the outputs and 4 interleaved slices of i, added in slice order through LDS; scalar segments: one
// output per thread, i ascending.  Fixed summation order: deterministic.
// ------------------------------------------------------------------------------------------
#define NAMP_REDUCE_MAX 16
struct ReduceSeg { const float* src; float* dst; long A, Mb, sa, sn; int n; int vec; };      // vec: 0 scalar path, else the slice count (4 or 16)
struct ReduceArgs { ReduceSeg seg[NAMP_REDUCE_MAX]; int first_block[NAMP_REDUCE_MAX + 1]; int nseg; };

static __global__ __launch_bounds__(256) void reduce_sum_kernel(const ReduceArgs ra) {
  __shared__ f4 red[256];
  int s = 0;
  while (s + 1 < ra.nseg && (int)blockIdx.x >= ra.first_block[s + 1]) ++s;
  const ReduceSeg& g = ra.seg[s];
  const long tile = (long)blockIdx.x - ra.first_block[s];
  const long total = g.A * g.Mb;
  const int tid = threadIdx.x;
  if (g.vec) {
    // S slices of i per workgroup: 4 (64 x 16-byte outputs per workgroup) for short sums, 16 (16 outputs) for long ones — with 4, the [512][128 x 128]
    // partials of a row contraction were 192 workgroups of 128 dependent-free but serial requests per thread: 312 us for 100 MB
    const int S = g.vec, Wd = 256 / S;
    const int c = tid % Wd, sl = tid / Wd;
    const long o = (tile * Wd + c) * 4;
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
    if (o < total) {
      const long a = o / g.Mb, b = o - a * g.Mb;
      const float* p = g.src + a * g.sa + b;
      int i = sl;
      for (; i + 7 * S < g.n; i += 8 * S) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f4*)(p + (long)(i + u * S) * g.sn);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; i < g.n; i += S) acc += *(const f4*)(p + (long)i * g.sn);
    }
    if (sl) red[(sl - 1) * Wd + c] = acc;
    __syncthreads();
    if (sl == 0 && o < total) {
      for (int q = 1; q < S; ++q) acc += red[(q - 1) * Wd + c];
      *(f4*)(g.dst + o) = acc;
    }
  } else {
    const long o = tile * 256 + tid;
    if (o < total) {
      const long a = o / g.Mb, b = o - a * g.Mb;
      const float* p = g.src + a * g.sa + b;
      float acc = 0.f;
      for (int i = 0; i < g.n; ++i) acc += p[(long)i * g.sn];
      g.dst[o] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Positional edge features of the training step (PositionalEncodings, na_model_utils.py:537-541 / 577-582) on two launches (round 5) instead of
// ~25 stock ones (the class index as six int64 tensor expressions, a 10^6-row table gather, the data gradient as a [10^6 x 128] x [128 x 16] library
// GEMM, the table gradient as one-hot batched GEMMs).
//   pos_features_kernel: d[e] = same-chain ? clip(R_i - R_j + 32, 0, 64) : 65;  E_pos[e][k] = W[k][d] + b[k]  (W = embeddings.linear.weight [16][66])
//   pos_grad_kernel:     gp[e][k] = sum_c g[e][c] * Wedge[c][k] (k < 16: the embedding's positional columns);  table[d[e]][k] += gp;  table[66][k] += gp
//     (= d weight^T, d bias).  A wave takes one edge row at a time: lane (k, part) multiplies channels 32 part .. +31 — the row's values are
//     broadcast loads, the 32 weights stay in registers —, the four parts meet through two cross-row shuffles and lanes of part 0 add into the
//     wave's PRIVATE table in LDS (plain read-modify-write, fixed order: deterministic); per-workgroup partials [groups][67][16] leave the chip.
// ------------------------------------------------------------------------------------------
#define POS_CLASSES 66
#define POS_DIM 16
static __global__ __launch_bounds__(256) void pos_features_kernel(const int32_t* __restrict__ R_idx, const int32_t* __restrict__ chain,
                                                           const int32_t* __restrict__ E_idx, const float* __restrict__ W, const float* __restrict__ b,
                                                           int32_t* __restrict__ d_out, float* __restrict__ E_pos, long E, int L, int K) {
  __shared__ float tab[POS_CLASSES * POS_DIM];                    // [d][k] = W[k][d] + b[k]
  for (int q = threadIdx.x; q < POS_CLASSES * POS_DIM; q += blockDim.x) { const int d = q / POS_DIM, k = q % POS_DIM; tab[q] = W[k * POS_CLASSES + d] + b[k]; }
  __syncthreads();
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    const int node = (int)(e / K);
    const int j = node - node % L + E_idx[e];
    int off = R_idx[node] - R_idx[j] + 32;
    off = off < 0 ? 0 : (off > 64 ? 64 : off);
    const int d = (chain[node] == chain[j]) ? off : 65;
    d_out[e] = d;
    const f4* t = (const f4*)(tab + d * POS_DIM);
    f4* o = (f4*)(E_pos + e * POS_DIM);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  }
}

#define POS_GRAD_WAVES 8
// sum over the 16 lanes of a DPP row, total in lane 15 (row_shr 8, 4, 2, 1 with zeros shifted in)
template <int CTRL>
__device__ __forceinline__ float pos_dpp_add(const float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float pos_row_sum15(float v) {
  v = pos_dpp_add<0x118>(v); v = pos_dpp_add<0x114>(v); v = pos_dpp_add<0x112>(v); v = pos_dpp_add<0x111>(v);
  return v;
}
// A wave takes 16 edge rows at a time: the rows in the register-chain layout (coalesced 16-byte loads), gp = g . Wedge[:, :16] as 32 exact-fp32
// v_mfma_f32_16x16x4 against fragments of the 16 positional columns held in registers — lane (m, g) ends up with gp[row m][4g .. 4g+3] —, then
// LDS adds into the wave's PRIVATE table at the row's class (ds_add_f32; lanes of one instruction that meet at an address are served in lane
// order, instructions in program order: deterministic) and one add per tile into the bias row after a DPP sum over the 16 rows.
// (First form, one row per wave-iteration with broadcast loads: 362 us per cfg5 launch, 1.6 TB/s — 16 lanes fetched the same 16 bytes.)
static __global__ __launch_bounds__(64 * POS_GRAD_WAVES) void pos_grad_kernel(const float* __restrict__ g, const float* __restrict__ Wedge, int ld,
                                                                       const int32_t* __restrict__ d, float* __restrict__ part, long E) {
  __shared__ float tab[POS_GRAD_WAVES][(POS_CLASSES + 1) * POS_DIM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, gq = lane >> 4;
  for (int q = lane; q < (POS_CLASSES + 1) * POS_DIM; q += 64) tab[wave][q] = 0.f;
  f4 wf[8];                                                       // A fragments: W^T[n = m][16 tk + 4 gq + r] = Wedge[16 tk + 4 gq + r][m]
#pragma unroll
  for (int tk = 0; tk < 8; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) wf[tk][r] = Wedge[(long)(16 * tk + 4 * gq + r) * ld + m];
  const long ntile = (E + 15) / 16;
  const long nw = (long)gridDim.x * POS_GRAD_WAVES;
  float* tw = tab[wave];
  for (long tile = (long)blockIdx.x * POS_GRAD_WAVES + wave; tile < ntile; tile += nw) {
    const long e_raw = tile * 16 + m;
    const bool valid = e_raw < E;
    const long e = valid ? e_raw : E - 1;
    const float* row = g + e * NAMP_H + 4 * gq;
    f4 x[8];
#pragma unroll
    for (int tk = 0; tk < 8; ++tk) x[tk] = *(const f4*)(row + 16 * tk);
    const int cls = d[e];
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tk = 0; tk < 8; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma4(wf[tk][r], x[tk][r], acc);
    if (!valid) acc = (f4){0.f, 0.f, 0.f, 0.f};
    float* dst = tw + cls * POS_DIM + 4 * gq;
    atomicAdd(dst + 0, acc.x); atomicAdd(dst + 1, acc.y); atomicAdd(dst + 2, acc.z); atomicAdd(dst + 3, acc.w);
    f4 s = acc;
    s.x = pos_row_sum15(s.x); s.y = pos_row_sum15(s.y); s.z = pos_row_sum15(s.z); s.w = pos_row_sum15(s.w);
    if (m == 15) {
      float* db = tw + POS_CLASSES * POS_DIM + 4 * gq;
      db[0] += s.x; db[1] += s.y; db[2] += s.z; db[3] += s.w;
    }
  }
  __syncthreads();
  for (int q = tid; q < (POS_CLASSES + 1) * POS_DIM; q += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < POS_GRAD_WAVES; ++w_) s += tab[w_][q];
    part[(long)blockIdx.x * (POS_CLASSES + 1) * POS_DIM + q] = s;
  }
}

// ------------------------------------------------------------------------------------------
// Reverse adjacency of the neighbour lists (the transpose of the gather: for every table row j the edges e = (i, k) with E_idx[i, k] = j, ascending),
// which scatter_rows_kernel streams.  A counting sort on the target row — keys < B * N — in four small launches (round 5) instead of a 64-bit
// stable radix sort of 10^6 pairs, a bincount and a cumsum on stock kernels (~0.4 ms per step): count, scan, fill (atomic slots: any order
// inside a row), then one wave per row puts its segment in ascending edge order by rank counting — the result is the stable sort's, bit for bit.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void radj_count_kernel(const int32_t* __restrict__ E_idx, int32_t* __restrict__ counts, long E, int N, int K) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    const int node = (int)(e / K);
    atomicAdd(counts + (node - node % N + E_idx[e]), 1);
  }
}
// offsets[0 .. G] = exclusive scan of counts[0 .. G); cursor[j] = offsets[j] (the fill's running slots).  One workgroup.
static __global__ __launch_bounds__(1024) void radj_scan_kernel(const int32_t* __restrict__ counts, int32_t* __restrict__ offsets, int32_t* __restrict__ cursor,
                                                         int G) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < G; base += 1024) {
    const int i = base + tid;
    const int c = i < G ? counts[i] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < G) { offsets[i] = before + incl - c; cursor[i] = before + incl - c; }
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) offsets[G] = carry;
}
static __global__ __launch_bounds__(256) void radj_fill_kernel(const int32_t* __restrict__ E_idx, int32_t* __restrict__ cursor, int32_t* __restrict__ tmp,
                                                        long E, int N, int K) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    const int node = (int)(e / K);
    const int slot = atomicAdd(cursor + (node - node % N + E_idx[e]), 1);
    tmp[slot] = (int32_t)e;
  }
}
// one wave per row: out[off + rank(x)] = x for the row's (distinct) edge numbers x, rank = how many of the row's numbers are smaller
static __global__ __launch_bounds__(256) void radj_sort_kernel(const int32_t* __restrict__ offsets, const int32_t* __restrict__ tmp, int32_t* __restrict__ out, int G) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= G) return;
  const int off = offsets[row], n = offsets[row + 1] - off;
  for (int base = 0; base < n; base += 64) {
    const int mine = base + lane < n ? tmp[off + base + lane] : 0x7fffffff;
    int cnt = 0;
    for (int cb = 0; cb < n; cb += 64) {
      const int other = cb + lane < n ? tmp[off + cb + lane] : 0x7fffffff;
      const int m_ = n - cb < 64 ? n - cb : 64;
      for (int t = 0; t < m_; ++t) cnt += (__builtin_amdgcn_readlane(other, t) < mine) ? 1 : 0;
    }
    if (base + lane < n) out[off + cnt] = mine;
  }
}

// ------------------------------------------------------------------------------------------
// Two small residue-level reductions of the training step that were stock launches (round 5):
//   class_sums_kernel:  part[wg][cls][c] = sum over the workgroup's rows with idx[row] == cls of g[row][c]   (c < 128, cls < nclass <= 64)
//     — the gradient of an embedding-table lookup with FEW rows (W_s: 33 tokens, node_embedding: 6 polymer types; na_model_utils.py:586-599);
//     the stock embedding backward scatters 24,000 rows into those few with atomics (119 us), the one-hot GEMM form is a library launch.
//     A wave owns a private [nclass][128] table in LDS and adds its rows in order (plain read-modify-write: deterministic).
//   wcolsum_kernel:     part[wg][c] = sum over the workgroup's rows of g[row][c] * w[row]                    — db3 of the hoisted layer 3.
// ------------------------------------------------------------------------------------------
#define CLASS_SUMS_MAX 64
static __global__ __launch_bounds__(256) void class_sums_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx, int nclass, long rows,
                                                         float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float cs_tab[];     // [4 waves][nclass][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* tw = cs_tab + (long)wave * nclass * NAMP_H;
  for (int q = lane; q < nclass * NAMP_H; q += 64) tw[q] = 0.f;
  const long nw = (long)gridDim.x * 4;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += nw) {
    const int cls = __builtin_amdgcn_readfirstlane(idx[r]);
    const float2 v = *(const float2*)(g + r * NAMP_H + 2 * lane);
    if (cls >= 0 && cls < nclass) {
      float2* d = (float2*)(tw + cls * NAMP_H + 2 * lane);
      float2 o = *d; o.x += v.x; o.y += v.y; *d = o;
    }
  }
  __syncthreads();
  for (int q = tid; q < nclass * NAMP_H; q += 256) {
    const float s_ = (cs_tab[q] + cs_tab[nclass * NAMP_H + q]) + (cs_tab[2 * nclass * NAMP_H + q] + cs_tab[3 * nclass * NAMP_H + q]);
    part[(long)blockIdx.x * nclass * NAMP_H + q] = s_;
  }
}

static __global__ __launch_bounds__(256) void wcolsum_kernel(const float* __restrict__ g, const float* __restrict__ w, long rows, float* __restrict__ part) {
  __shared__ float red[4][NAMP_H];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2 acc = make_float2(0.f, 0.f);
  const long nw = (long)gridDim.x * 4;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += nw) {
    const float2 v = *(const float2*)(g + r * NAMP_H + 2 * lane);
    const float wr = w[r];
    acc.x = fmaf(v.x, wr, acc.x); acc.y = fmaf(v.y, wr, acc.y);
  }
  red[wave][2 * lane] = acc.x; red[wave][2 * lane + 1] = acc.y;
  __syncthreads();
  if (tid < NAMP_H) part[(long)blockIdx.x * NAMP_H + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
#endif  // NAMP_TRAIN_EDGE_ONLY
