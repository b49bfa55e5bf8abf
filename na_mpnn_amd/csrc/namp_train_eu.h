// The EncLayer edge update  h_E' = LN3(h_E + dropout(W13 gelu(W12 gelu(W11b h_E + Pa + Pc) + b12) + b13))  backward
// (na_model_utils.py:236-240), mixed precision (plain bf16 products), as TWO persistent launches that own their weight gradients
// (round 5).  No row tensor is written for a later contraction: A1, A2, G3 and the parked gelu'(z1) rows of the
// round-3 launch are gone, and so are its three row contractions.
//
// The one-launch form (round 4) wanted the six-product chain, three 64-register accumulator blocks and six 32-KiB images at once: ~520
// registers, 163 spilled.  The chain is cut where its live set is smallest — at g2 = dL/dz2, one bf16 row:
//
//   launch A (edge_update_bwd_a16_kernel): recompute z1, a1, z2, a2, z3; dropout; LayerNorm3 backward in registers
//       -> dL/dx rows (the residual part of dL/dh_E, fp32, parked in the g_hE output rows), d(ln weight), d(ln bias);
//       dW13 += G3^T A2, db13 contracted on chip;  g2 = (W13^T g3) * gelu'(z2) -> G2 rows (bf16, 256 B per edge).
//       Resident images: W11b, W12, W13, W13^T (4 x 32 KiB) + two staged planes = the 160 KiB of edge_bwd_dw16_kernel.
//   launch B (edge_update_bwd_b16_kernel): recompute z1, a1, gelu'(z1);  dW12 += G2^T A1, db12;  g1 = (W12^T g2) * gelu'(z1);
//       dW11b += G1^T h_E;  dL/dh_E = dL/dx + W11b^T g1;  G1 rows (bf16) for the table-gradient gather, per-tile sums for dL/dPa.
//       = edge_bwd_dw16_kernel without its second product: resident W11b, W12^T, W11b^T (3 x 32 KiB) + two staged planes.
//
// Seven products where one launch needs six (z1 is recomputed twice); HBM per edge row: A reads 512 (h_E) + 512 (dL/dh_E'), writes 512
// (dL/dx) + 256 (G2); B reads 512 + 256 + 512, writes 512 (dL/dh_E) + 256 (G1) — 3.8 KB against ~6.9 KB measured for the round-3 launch
// before its row contractions read A1, A2, G1, G2, G3 back.
// Both kernels keep the rules of edge_bwd_dw16_kernel (namp_train_dw.h): one wave per SIMD, no vector-memory instruction of the round
// loop under a branch (rows past E compute on clamped inputs with a zero upstream gradient and store into padding rows), requests in
// the order they are consumed, LDS-only barriers.
#pragma once
#include "namp_train_dw.h"

struct EdgeUpdAArgs {
  EdgeBwdArgs b;        // hE, E_idx, Pa, Pj0 (= Pc), W1/W2/W3/W3t images (bf16), b2, b3, ln_g, g_rows, drop_*; out: G2 (bf16 rows), g_hE (dL/dx rows)
  float* dW_part;       // [grid][128][128]: dW13 = G3^T A2
  float* db_part;       // [grid][128]: db13 = sum of G3 rows
  float* dgb_part;      // [grid][2][128]: sums of g * xhat (-> d ln weight) and of g (-> d ln bias)
  long nrounds;
};

#define EUA_LDS (4 * NAMP_BIMG_BYTES + 2 * DW_ARR)
#define EUB_LDS (3 * NAMP_BIMG_BYTES + 2 * DW_ARR)

// value-only GELU of the mixed-precision mode (the polynomial of dw_gelu_split4_bf16 / gelu4_bf16mode)
__device__ __forceinline__ f4 eu_gelu4(const f4 x) {
  const f4 t = x * x;
  f4 q = (f4){NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4};
  q = q * t + NAMP_GELU4_Q3;
  q = q * t + NAMP_GELU4_Q2;
  q = q * t + NAMP_GELU4_Q1;
  q = q * t + NAMP_GELU4_Q0;
  f4 p = x * q + 0.5f;
  p = (f4){__builtin_amdgcn_fmed3f(p.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(p.y, 0.f, 1.f), __builtin_amdgcn_fmed3f(p.z, 0.f, 1.f),
           __builtin_amdgcn_fmed3f(p.w, 0.f, 1.f)};
  return x * p;
}

// Reduce-scatter of a tile's register block over the 16 rows of a DPP row: in v[t] = channels 16t + 4g + (0..3) of row m; out: the sum over the 16
// rows of channel tile t = m >> 1 (every tile ends in two lanes).  Recursive halving with the DPP mirrors as the pairings — (m, 15 - m), then (m, 7 - m)
// within 8 lanes, (m, 3 - m) within quads, (m, m ^ 1) — so that each step moves half of what is left: 32 adds + 56 selects per block, where an
// all-reduce of every value costs 128 adds, and the running sums take 4 registers instead of 32.
template <int CTRL>
__device__ __forceinline__ float eu_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ f4 eu_row_reduce_scatter(const f4 (&v)[8], const int m) {
  const bool h1 = (m & 8) != 0, h2 = (m & 4) != 0, h3 = (m & 2) != 0;
  f4 s1[4], s2[2], s3, s4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) s1[i][c] = (h1 ? v[4 + i][c] : v[i][c]) + eu_dpp<0x140>(h1 ? v[i][c] : v[4 + i][c]);        // row_mirror
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[i][c] = (h2 ? s1[2 + i][c] : s1[i][c]) + eu_dpp<0x141>(h2 ? s1[i][c] : s1[2 + i][c]);    // row_half_mirror
#pragma unroll
  for (int c = 0; c < 4; ++c) s3[c] = (h3 ? s2[1][c] : s2[0][c]) + eu_dpp<0x1B>(h3 ? s2[0][c] : s2[1][c]);                  // quad_perm [3,2,1,0]
#pragma unroll
  for (int c = 0; c < 4; ++c) s4[c] = s3[c] + eu_dpp<0xB1>(s3[c]);                                                          // quad_perm [1,0,3,2]
  return s4;
}

__device__ __forceinline__ f4 eu_unpack(const bf2 lo, const bf2 hi) { return (f4){(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]}; }

template <bool DROP>      // dropout3 active (drop_thresh != 0): a template parameter, so that the round loop has no branch around it
__global__ __launch_bounds__(64 * DW_WAVES) void edge_update_bwd_a16_kernel(const EdgeUpdAArgs aa) {
  const EdgeBwdArgs& a = aa.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* SG = smem + 4 * NAMP_BIMG_BYTES;
  char* SA = SG + DW_ARR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int wo = wave >> 1, wc = wave & 1;
  const bf8* w1 = (const bf8*)smem + lane;
  const bf8* w2 = (const bf8*)(smem + NAMP_BIMG_BYTES) + lane;
  const bf8* w3 = (const bf8*)(smem + 2 * NAMP_BIMG_BYTES) + lane;
  const bf8* w3t = (const bf8*)(smem + 3 * NAMP_BIMG_BYTES) + lane;
  copy_to_lds<4>(smem, a.W1_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + 2 * NAMP_BIMG_BYTES, a.W3_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + 3 * NAMP_BIMG_BYTES, a.W3t_img, 32, wave, DW_WAVES, lane);

  f4 dW3[4][4], db3[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) dW3[q][t] = (f4){0.f, 0.f, 0.f, 0.f};
  // LayerNorm d(weight) = sum g * xhat, d(bias) = sum g: fp32, reduced over the tile's 16 rows every round (eu_row_reduce_scatter: lane m keeps
  // channel tile m >> 1) and over the rounds in 2 x 4 registers; the four waves are added once behind the loop.  (Per-lane running sums of all 64
  // values: 17-26 spilled registers.  The round-4 form — two more staged planes and MFMAs against ones — needs LDS this launch does not have.)
  f4 kw = (f4){0.f, 0.f, 0.f, 0.f}, kb = (f4){0.f, 0.f, 0.f, 0.f};

  auto row_of = [&](const long round) {                             // clamped edge row of this lane in `round`
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;
    return e_raw < a.E ? e_raw : (a.E - 1);
  };
  auto nbr_of = [&](const long e, const int idx) {                  // global row of the neighbour (same complex)
    const int node = (int)(e / a.K);
    return node - node % a.N + idx;
  };
  // register roles.  x: h_E rows (fp32: first product's operand AND the LayerNorm's residual), behind the statistics the next round's h_E rows.
  // z1: Pa (+ Pc), first product; then b13, third product = z3, x_ln, xhat; behind dL/dx the next round's Pa rows.  A: Pc rows; b12, second product,
  // gelu'(z2) (packed away); behind the LayerNorm the next round's Pc rows.  y: a1, a2; the LayerNorm weight; g * ln_w; accumulator of the last
  // product, g2.  gr: this round's dL/dh_E' rows, dL/dx, g3.
  f4 x[8], z1[8], A[8], y[8], gr[8];
  bf2 e16[16];
  long round = blockIdx.x;
  {
    const long r0 = round < aa.nrounds ? round : 0;
    const long e0 = row_of(r0);
    const int j0 = nbr_of(e0, a.E_idx[e0]);
    const float* src = a.hE + e0 * NAMP_H + 4 * g;
    const float* pa = a.Pa + (e0 / a.K) * NAMP_H + 4 * g;
    const float* pj = a.Pj0 + (long)j0 * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { x[t] = *(const f4*)(src + 16 * t); z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
  }
  int idx_n1 = a.E_idx[row_of(round + gridDim.x < aa.nrounds ? round + gridDim.x : round)];
  __syncthreads();                                                   // the images are in place

  for (; round < aa.nrounds; round += gridDim.x) {
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;          // unclamped: rows past E store into the buffers' padding
    const bool valid = e_raw < a.E;
    const long e = valid ? e_raw : (a.E - 1);
    const long round_n = round + gridDim.x;
    const long rn = round_n < aa.nrounds ? round_n : round;        // (last rounds: their own rows again — no branch around the requests)
    const long e_n = row_of(rn);
    const int j_n = nbr_of(e_n, idx_n1);
    const long round_n2 = round_n + gridDim.x;
    idx_n1 = a.E_idx[row_of(round_n2 < aa.nrounds ? round_n2 : round)];
    // (loop-invariant vectors: their addresses are made opaque per round, or the compiler keeps all 96 registers of them across the loop)
    const float* b2p = a.b2; const float* b3p = a.b3; const float* lgp = a.ln_g;
    asm volatile("" : "+s"(b2p), "+s"(b3p), "+s"(lgp));
    // ---- z1 = W11b . h_E + (Pa + Pc)
#pragma unroll
    for (int t = 0; t < 8; ++t) z1[t] += A[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) A[t] = *(const f4*)(b2p + 16 * t + 4 * g);                       // b12 (L1)
    // this round's dL/dh_E' rows (HBM): three products to land
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = *(const f4*)(a.g_rows + e * NAMP_H + 4 * g + 16 * t);
    dw_gemm16_ahead(z1, x, w1);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = eu_gelu4(z1[t]);                                           // a1 (its derivative is launch B's business)
#pragma unroll
    for (int t = 0; t < 8; ++t) z1[t] = *(const f4*)(b3p + 16 * t + 4 * g);                      // b13 (L1)
    // ---- z2 = W12 . a1 + b12
    dw_gemm16_ahead(A, y, w2);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      y[t] = dw_gelu_split4_bf16(A[t]);                                                           // y <- a2, A <- gelu'(z2)
      e16[2 * t] = (bf2){(__bf16)A[t].x, (__bf16)A[t].y};
      e16[2 * t + 1] = (bf2){(__bf16)A[t].z, (__bf16)A[t].w};
    }
    dw_lds_barrier();                                                // the previous round's contraction has been read out
    dw_stage<false>(SA, y, wave, m, g);                              // A2 plane
    // ---- z3 = W13 . a2 + b13, dropout, x_ln = h_E + z3, LayerNorm statistics
    dw_gemm16_ahead(z1, y, w3);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = *(const f4*)(lgp + 16 * t + 4 * g);                       // LayerNorm weight (L1), ahead of the HBM prefetch below
    // the row's dropout mask as 32 bits (bit 4t + r = channel 16t + 4g + r kept): hashed once, applied here and to the gradient below
    uint32_t keep = 0xffffffffu;
    if constexpr (DROP) {
      const uint32_t key = drop_row_key(a.drop_seed, e);
      keep = 0u;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) keep |= (drop_factor(key, 16 * t + 4 * g + r, a.drop_thresh, 1.0f) != 0.f ? 1u : 0u) << (4 * t + r);
      asm volatile("" : "+v"(keep));                                  // one register across the LayerNorm, not 32 factors
    }
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) z1[t][r] *= ((keep >> (4 * t + r)) & 1u) ? a.drop_scale : 0.f;
      }
      z1[t] += x[t];
      s1 += (z1[t].x + z1[t].y) + (z1[t].z + z1[t].w);
    }
    {                                                                // the next round's h_E rows (HBM): the rest of this round to land
      const float* src = a.hE + e_n * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
    }
    const float mean = xg_sum(s1) * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { z1[t] -= mean; s2 += (z1[t].x * z1[t].x + z1[t].y * z1[t].y) + (z1[t].z * z1[t].z + z1[t].w * z1[t].w); }
    const float rstd = rsqrtf(xg_sum(s2) * (1.0f / 128.0f) + 1e-5f);
    const float vz = valid ? 1.f : 0.f;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      z1[t] *= rstd;                                                  // xhat
      gr[t] = gr[t] * vz;                                              // g = dL/dh_E' (zero past E)
      y[t] = gr[t] * y[t];                                             // g * ln_w
      m1 += (y[t].x + y[t].y) + (y[t].z + y[t].w);
      m2 += (y[t].x * z1[t].x + y[t].y * z1[t].y) + (y[t].z * z1[t].z + y[t].w * z1[t].w);
    }
    m1 = xg_sum(m1) * (1.0f / 128.0f);
    m2 = xg_sum(m2) * (1.0f / 128.0f);
    kb += eu_row_reduce_scatter(gr, m);
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = gr[t] * z1[t];                  // g * xhat (gr is rebuilt as dL/dx below)
    kw += eu_row_reduce_scatter(gr, m);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      gr[t] = (y[t] - m1 - z1[t] * m2) * rstd;                         // dL/dx of the LayerNorm = the residual part of dL/dh_E:
      *(f4*)(a.g_hE + e_raw * NAMP_H + 4 * g + 16 * t) = gr[t];         // parked in the output rows, picked up by launch B
      if constexpr (DROP) {                                            // through the dropout mask: g3 = dL/dz3
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[t][r] *= ((keep >> (4 * t + r)) & 1u) ? a.drop_scale : 0.f;
      }
    }
    {                                                                // the next round's table rows (L2): Pa -> z1, Pc -> A
      const float* pa = a.Pa + (e_n / a.K) * NAMP_H + 4 * g;
      const float* pj = a.Pj0 + (long)j_n * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
    }
    // ---- contraction: dW13 += G3^T A2, db13 += sum G3
    dw_stage<false>(SG, gr, wave, m, g);
    dw_lds_barrier();
    dw_contract<false, true>(dW3, db3, SG, SA, wo, wc, m, g);
    // ---- g2 = (W13^T g3) * gelu'(z2)  ->  G2 rows (bf16)
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = (f4){0.f, 0.f, 0.f, 0.f};
    dw_gemm16_ahead(y, gr, w3t);
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] *= eu_unpack(e16[2 * t], e16[2 * t + 1]);
    st_tile_bf16(a.G2, e_raw * NAMP_H, y, g);
  }
  // ---- the workgroup's partials
  float* o3 = aa.dW_part + (long)blockIdx.x * NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o3[(64 * wo + 16 * q + 4 * g + r) * NAMP_H + 64 * wc + 16 * t + m] = dW3[q][t][r];
  if (m == 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) aa.db_part[(long)blockIdx.x * NAMP_H + 64 * wo + 16 * (2 * wc + u) + 4 * g + r] = db3[u][r];
  }
  // LayerNorm sums of the four waves through the (now free) staged planes: lane (m even, g) holds channels 16 (m >> 1) + 4g + (0..3)
  __syncthreads();
  float* red = (float*)SG;                                          // [wave][2][128]
  if ((m & 1) == 0) {
    *(f4*)(red + (wave * 2 + 0) * NAMP_H + 16 * (m >> 1) + 4 * g) = kw;
    *(f4*)(red + (wave * 2 + 1) * NAMP_H + 16 * (m >> 1) + 4 * g) = kb;
  }
  __syncthreads();
  if (tid < 2 * NAMP_H) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < DW_WAVES; ++w) s += red[w * 2 * NAMP_H + tid];
    aa.dgb_part[(long)blockIdx.x * 2 * NAMP_H + tid] = s;
  }
}

// Launch B.  EdgeBwdDwArgs: b.hE, E_idx, Pa, Pj0 (= Pc), W1_img (W11b), W2t_img (W12^T), W1t_img (W11b^T), G2 (in: bf16 rows of launch A), g_hE_in (dL/dx rows;
// may alias g_hE), g_hE, G1 (out, bf16), g_Pa; dW_part [grid][2][128][128] (0 = dW12, 1 = dW11b), db_part [grid][128] (db12).
template <int GPA>        // 1 = per-tile sums of G1 for dL/dPa (K % 16 == 0); 2 = fp32 atomics
__global__ __launch_bounds__(64 * DW_WAVES) void edge_update_bwd_b16_kernel(const EdgeBwdDwArgs aa) {
  const EdgeBwdArgs& a = aa.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* SG = smem + 3 * NAMP_BIMG_BYTES;
  char* SA = SG + DW_ARR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int wo = wave >> 1, wc = wave & 1;
  const bf8* w1 = (const bf8*)smem + lane;
  const bf8* w2t = (const bf8*)(smem + NAMP_BIMG_BYTES) + lane;
  const bf8* w1t = (const bf8*)(smem + 2 * NAMP_BIMG_BYTES) + lane;
  copy_to_lds<4>(smem, a.W1_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + NAMP_BIMG_BYTES, a.W2t_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + 2 * NAMP_BIMG_BYTES, a.W1t_img, 32, wave, DW_WAVES, lane);

  f4 dW2[4][4], dW1[4][4], db2[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW2[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; dW1[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; }

  auto row_of = [&](const long round) {
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;
    return e_raw < a.E ? e_raw : (a.E - 1);
  };
  auto nbr_of = [&](const long e, const int idx) {
    const int node = (int)(e / a.K);
    return node - node % a.N + idx;
  };
  // register roles: as in edge_bwd_dw16_kernel.  x: h_E rows, a1, the next round's h_E rows.  z1: Pa (+ Pc), first product, gelu'(z1) (packed away);
  // behind g1 the next round's Pa.  A: Pc rows; the second product's accumulator; behind g1 the next round's Pc.  gr: g2, g1.  P: dL/dx rows, accumulates
  // the last product.  g2n: the NEXT round's G2 rows as they come from memory (bf16, 16 registers), requested a round ahead.
  f4 x[8], z1[8], gr[8], A[8], P[8];
  bf2 h16[16], d16[16];
  bf4 g2n[8];
  const __bf16* G2 = (const __bf16*)a.G2;
  long round = blockIdx.x;
  {
    const long r0 = round < aa.nrounds ? round : 0;
    const long e0 = row_of(r0);
    const int j0 = nbr_of(e0, a.E_idx[e0]);
    const float* src = a.hE + e0 * NAMP_H + 4 * g;
    const float* pa = a.Pa + (e0 / a.K) * NAMP_H + 4 * g;
    const float* pj = a.Pj0 + (long)j0 * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { x[t] = *(const f4*)(src + 16 * t); z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
#pragma unroll
    for (int t = 0; t < 8; ++t) g2n[t] = *(const bf4*)(G2 + e0 * NAMP_H + 4 * g + 16 * t);
  }
  int idx_n1 = a.E_idx[row_of(round + gridDim.x < aa.nrounds ? round + gridDim.x : round)];
  __syncthreads();

  for (; round < aa.nrounds; round += gridDim.x) {
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;
    const bool valid = e_raw < a.E;
    const long e = valid ? e_raw : (a.E - 1);
    const long round_n = round + gridDim.x;
    const long rn = round_n < aa.nrounds ? round_n : round;
    const long e_n = row_of(rn);
    const int j_n = nbr_of(e_n, idx_n1);
    const long round_n2 = round_n + gridDim.x;
    idx_n1 = a.E_idx[row_of(round_n2 < aa.nrounds ? round_n2 : round)];
    const float vz = valid ? 1.f : 0.f;
    // ---- z1 = W11b . h_E + (Pa + Pc); h_E kept as packed bf16 for the second contraction
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      z1[t] += A[t];
      h16[2 * t] = (bf2){(__bf16)x[t].x, (__bf16)x[t].y};
      h16[2 * t + 1] = (bf2){(__bf16)x[t].z, (__bf16)x[t].w};
    }
    dw_gemm16_ahead(z1, x, w1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      x[t] = dw_gelu_split4_bf16(z1[t]);                              // x <- a1, z1 <- gelu'(z1)
      d16[2 * t] = (bf2){(__bf16)z1[t].x, (__bf16)z1[t].y};
      d16[2 * t + 1] = (bf2){(__bf16)z1[t].z, (__bf16)z1[t].w};
      gr[t] = from_bf4(g2n[t]) * vz;                                  // g2 (rows past E: launch A stored zeros there, vz keeps NaN-free padding out)
    }
    // ---- contraction 1: dW12 += G2^T A1, db12 += sum G2
    dw_lds_barrier();                                                // the previous round's second contraction has been read out
    dw_stage<false>(SG, gr, wave, m, g);
    dw_stage<false>(SA, x, wave, m, g);
    {                                                                // the next round's h_E and G2 rows (HBM)
      const float* src = a.hE + e_n * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
#pragma unroll
      for (int t = 0; t < 8; ++t) g2n[t] = *(const bf4*)(G2 + e_n * NAMP_H + 4 * g + 16 * t);
    }
    dw_lds_barrier();
    dw_contract<false, true>(dW2, db2, SG, SA, wo, wc, m, g);
    // ---- g1 = (W12^T g2) * gelu'(z1)
#pragma unroll
    for (int t = 0; t < 8; ++t) A[t] = (f4){0.f, 0.f, 0.f, 0.f};
    dw_gemm16_ahead(A, gr, w2t);
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = A[t] * eu_unpack(d16[2 * t], d16[2 * t + 1]);
    {                                                                // the next round's table rows (L2): Pa -> z1, Pc -> A; then this round's dL/dx rows (HBM) -> P
      const float* pa = a.Pa + (e_n / a.K) * NAMP_H + 4 * g;
      const float* pj = a.Pj0 + (long)j_n * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
#pragma unroll
      for (int t = 0; t < 8; ++t) P[t] = *(const f4*)(a.g_hE_in + e_raw * NAMP_H + 4 * g + 16 * t);
    }
    // ---- contraction 2: dW11b += G1^T h_E
    dw_lds_barrier();                                                // contraction 1 has been read out
    dw_stage<false>(SG, gr, wave, m, g);
    dw_stage_packed(SA, h16, wave, m, g);
    dw_lds_barrier();
    st_tile_bf16(a.G1, e_raw * NAMP_H, gr, g);
    if constexpr (GPA == 1) {
      const long tile = round * DW_WAVES + wave;
      f4 keep = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f4 v = gr[t];                                                 // (zero past E: g2 was)
        v.x = dw_row_allsum(v.x); v.y = dw_row_allsum(v.y); v.z = dw_row_allsum(v.z); v.w = dw_row_allsum(v.w);
        keep = ((m & 7) == t) ? v : keep;
      }
      *(f4*)(a.g_Pa + tile * NAMP_H + 16 * (m & 7) + 4 * g) = keep;
    } else {
      float* d = a.g_Pa + (e / a.K) * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w);
      }
    }
    f4 nob[2];
    dw_contract<false, false>(dW1, nob, SG, SA, wo, wc, m, g);
    // ---- dL/dh_E = dL/dx + W11b^T g1
    dw_gemm16_ahead(P, gr, w1t);
    {
      float* d = a.g_hE + e_raw * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(d + 16 * t) = P[t];
    }
  }
  float* o2 = aa.dW_part + (long)blockIdx.x * 2 * NAMP_H * NAMP_H;
  float* o1 = o2 + NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 64 * wo + 16 * q + 4 * g + r, c = 64 * wc + 16 * t + m;
        o2[o * NAMP_H + c] = dW2[q][t][r];
        o1[o * NAMP_H + c] = dW1[q][t][r];
      }
  if (m == 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) aa.db_part[(long)blockIdx.x * NAMP_H + 64 * wo + 16 * (2 * wc + u) + 4 * g + r] = db2[u][r];
  }
}
