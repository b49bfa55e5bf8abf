// edge_bwd_dw_kernel — the message-stage backward launch that OWNS its weight gradients (round 4).
//
// edge_chain_bwd_kernel (namp_train.h) writes the row tensors A1 = gelu(z1) and G2 = dL/dz2 to HBM only so that a second launch
// (wgrad_*) can read them back and contract them over the ~10^6 edge rows: dW2 = G2^T A1, dW1b = G1^T h_E, db2 = sum G2
// (na_model_utils.py:196-283: the gradients of EncLayer.W1/W2, DecLayer.W1/W2).  Here the contraction happens inside the launch:
//
//   * PERSISTENT workgroups (one per CU, 8 waves), each walking rounds of 8 x 16 consecutive edge rows.  A wave carries its 16-row
//     tile through the recomputed chain in registers exactly as edge_chain_bwd_kernel does.
//   * When a row operand pair (G, A) of a contraction exists in registers (lane (m, g) holds channels 16t + 4g + r of row m) the
//     waves write their tiles to LDS K-MAJOR — S[channel][row] in bf16, rows contiguous — so that the MFMA operand of the
//     contraction, "8 consecutive rows of one channel", is ONE 16-byte LDS read: lane (n, g) of v_mfma_f32_16x16x32_bf16 feeds
//     A[i = n][k = 8g + j] = G[row 32ks + 8g + j][channel 16q + n], B[k][n] = A[row][channel 16t + n].  The transposition is paid by
//     the writer as 2-byte LDS stores; a 16-byte XOR swizzle of the row chunks keeps reads conflict-free without padding.
//   * Wave (wo, wc) = (wave >> 2, wave & 3) owns the dW block [64 wo .. +64) x [32 wc .. +32) of each weight — 4 x 2 accumulator
//     tiles = 32 VGPRs per weight — ACROSS ALL ITS ROUNDS; one [128 x 128] partial per workgroup and weight leaves the chip at
//     the end (the caller adds the <= 256 partials: deterministic).  db2 rides along as one more MFMA against a fragment of ones.
//   * Weights stream through a ring of two 32 KiB LDS slots by LDS-DMA one product ahead: a slot is a whole bf16 image (mixed
//     precision) or the K-half (steps s = 2h, 2h + 1; hi and mid planes) of an x3 image (split-bf16: every 128 x 128 product runs
//     as two half products with a ring point between).  That leaves 64 KiB for the staged operands: 128 rows of (G, A) in bf16, or
//     64 rows of (G_hi, G_mid, A_hi, A_mid) — the split-bf16 contraction G_mid.A_hi + G_hi.A_mid + G_hi.A_hi runs in two sub-phases
//     (waves 0-3's rows, then waves 4-7's).
//
// Not written any more: A1, G2 (2 x 590 MB per stage at cfg5 in fp32 rows).  Still written: G1 (the table-gradient gather
// dL/dPj reads it: namp_train_scatter_rows), dL/dh_E, the per-tile sums of G1 for dL/dPa.
#pragma once
#include "namp_train.h"

struct EdgeBwdDwArgs {
  EdgeBwdArgs b;        // A1, A2, G2, G3, S3, w3, g_Pj0, g_Pj1 unused
  float* dW_part;       // [gridDim.x][2][128][128]: 0 = dW2 = G2^T A1, 1 = dW1b = G1^T h_E
  float* db_part;       // [gridDim.x][128] = sum of G2 rows
  long nrounds;         // ceil(E / 128)
};

#define DW_SLOT_BYTES 32768
#define DW_STAGE_BYTES 65536
#define DW_LDS (2 * DW_SLOT_BYTES + DW_STAGE_BYTES + 512)

template <int PREC> struct DwGeom {
  static constexpr int ROWB = (PREC == 1) ? 128 : 256;     // bytes per channel row of a staged array: 64 / 128 rows of bf16
  static constexpr int ARR = 128 * ROWB;                   // one staged plane
  static constexpr int NPL = (PREC == 1) ? 2 : 1;          // planes per operand (hi, mid)
  static constexpr int NKS = (PREC == 1) ? 2 : 4;          // 32-row MFMA steps per staged set
  static constexpr int NSUB = (PREC == 1) ? 2 : 1;         // staged sets per round
};

// Write this wave's 16-row tile v (register-chain layout) into the K-major staged array at `base`: S[ch][16 wl + m] = bf16(v),
// plane 1 (x3) = bf16 of the remainder.  Row chunk (8 rows = 16 bytes) c of channel ch sits at chunk position c ^ f(ch),
// f(ch) = ch & 15 (256-byte rows) or (ch >> 1) & 7 (128-byte rows): tests/test_layout_sim.py checks the read side is conflict-free.
template <int PREC>
__device__ __forceinline__ void dw_stage(char* base, const f4 (&v)[8], const int wl, const int m, const int g) {
  constexpr int ROWB = DwGeom<PREC>::ROWB, ARR = DwGeom<PREC>::ARR;
  const int chunkv = 2 * wl + (m >> 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int chl = 4 * g + r;
    const int f = (PREC == 1) ? ((chl >> 1) & 7) : chl;
    char* p = base + chl * ROWB + ((chunkv ^ f) << 4) + ((m & 7) << 1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float val = v[t][r];
      const __bf16 hi = (__bf16)val;
      *(__bf16*)(p + t * 16 * ROWB) = hi;
      if (PREC == 1) *(__bf16*)(p + ARR + t * 16 * ROWB) = (__bf16)(val - (float)hi);
    }
  }
}

// acc[q][t] += sum over the staged rows of G[row][64 wo + 16 q + i] * A[row][32 wc + 16 t + j]   (lane (n, g) holds D[i = 4g + r][j = n])
template <int PREC, bool BIAS>
__device__ __forceinline__ void dw_contract(f4 (&acc)[4][2], f4& accb, const char* SG, const char* SA, const int wo, const int wc,
                                            const int n, const int g) {
  constexpr int ROWB = DwGeom<PREC>::ROWB, ARR = DwGeom<PREC>::ARR, NKS = DwGeom<PREC>::NKS;
  constexpr bool X3 = (PREC == 1);
  const int fsw = X3 ? (n >> 1) : n;
  const char* gb = SG + (64 * wo + n) * ROWB;
  const char* ab = SA + (32 * wc + n) * ROWB;
  bf8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int off = ((4 * ks + g) ^ fsw) << 4;
    bf8 gh[4], gm[4], ah[2], am[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gh[q] = *(const bf8*)(gb + q * 16 * ROWB + off);
      if (X3) gm[q] = *(const bf8*)(gb + ARR + q * 16 * ROWB + off);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *(const bf8*)(ab + t * 16 * ROWB + off);
      if (X3) am[t] = *(const bf8*)(ab + ARR + t * 16 * ROWB + off);
    }
    // product-major: eight independent accumulators between two MFMAs on the same one
    if (X3) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm[q], ah[t], acc[q][t], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], am[t], acc[q][t], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], ah[t], acc[q][t], 0, 0, 0);
    if (BIAS) {
      // column sums of G for the wave's bias tile q == wc: G^T . ones (every column of the result holds the sum)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q == wc) {
          if (X3) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm[q], ones, accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], ones, accb, 0, 0, 0);
        }
    }
  }
}

// K-half HALF (steps s = 2 HALF, 2 HALF + 1) of a split-bf16 128 x 128 product out of one ring slot: hi plane at byte 0, mid plane at
// byte 16384, fragment (s_local, tn) at (8 s_local + tn) KiB.  Same product order as chain_gemm_x3 (namp_device.h).
template <int HALF>
__device__ __forceinline__ void dw_half_gemm_x3(f4 (&acc)[8], const f4 (&x)[8], const char* slot, const int lane) {
  const bf8* wh = (const bf8*)slot + lane;
  const bf8* wm = (const bf8*)(slot + 16384) + lane;
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    bf8 hi, mid;
    split_x3(x[4 * HALF + 2 * sl], x[4 * HALF + 2 * sl + 1], hi, mid);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf8 w4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w4[q] = wh[(sl * 8 + 4 * h + q) * 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w4[q], mid, acc[4 * h + q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf8 wmq = wm[(sl * 8 + 4 * h + q) * 64];
        acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmq, hi, acc[4 * h + q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w4[q], hi, acc[4 * h + q], 0, 0, 0);
    }
  }
}

// MODE: BWD_ENC_MSG / BWD_DEC_MSG.  PREC: 1 split-bf16 products, 2 plain bf16 products (mixed precision; G1 rows bf16).
template <int MODE, int PREC>
__global__ __launch_bounds__(512) void edge_bwd_dw_kernel(const EdgeBwdDwArgs aa) {
  static_assert(MODE == BWD_ENC_MSG || MODE == BWD_DEC_MSG, "message stages only");
  static_assert(PREC == 1 || PREC == 2, "split-bf16 or bf16 products");
  using Ge = DwGeom<PREC>;
  constexpr bool X3 = (PREC == 1);
  constexpr bool RB = (PREC == 2);                     // bf16 G1 rows
  const EdgeBwdArgs& a = aa.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slot0 = smem;
  char* slot1 = smem + DW_SLOT_BYTES;
  char* SG = smem + 2 * DW_SLOT_BYTES;
  char* SA = SG + Ge::NPL * Ge::ARR;
  float* cstb = (float*)(smem + 2 * DW_SLOT_BYTES + DW_STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int wo = wave >> 2, wc = wave & 3;
  if (tid < NAMP_H) cstb[tid] = a.b2[tid];

  f4 dW2[4][2], dW1[4][2], db2 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 2; ++t) { dW2[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; dW1[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; }

  // ---- the weight ring.  Fill i of a round: bf16 -> image i (W1b, W2, W2^T, W1b^T); x3 -> K-half (i & 1) of image i >> 1.
  // Fill i always lands in slot i & 1 (4 / 8 fills per round: even).
  auto issue_fill = [&](const int i) {
    char* dst = (i & 1) ? slot1 : slot0;
    const int img_i = X3 ? (i >> 1) : i;
    const float* img = img_i == 0 ? a.W1_img : img_i == 1 ? a.W2_img : img_i == 2 ? a.W2t_img : a.W1t_img;
    if (X3) {
      const char* src = (const char*)img + (i & 1) * 16384;
      dma_to_lds(dst, (const float*)src, 16, wave, 8, lane);
      dma_to_lds(dst + 16384, (const float*)(src + NAMP_BIMG_BYTES), 16, wave, 8, lane);
    } else {
      dma_to_lds(dst, img, 32, wave, 8, lane);
    }
  };
  auto ring_point = [&]() { wait_dma_and_sync(); };

  // ---- per-round row bookkeeping
  struct Meta { long e; int node; int j; bool valid; float w_row; bool from1; };
  auto meta_of = [&](const long round) {
    Meta mt;
    const long e_raw = (round * 8 + wave) * 16 + m;
    mt.valid = e_raw < a.E;
    mt.e = mt.valid ? e_raw : (a.E - 1);
    mt.node = (int)(mt.e / a.K);
    const int i_loc = mt.node % a.N;
    mt.j = mt.node - i_loc + a.E_idx[mt.e];
    mt.from1 = false;
    if (MODE == BWD_DEC_MSG) {
      mt.from1 = !(a.rank[mt.j] < a.rank[mt.node]);
      mt.w_row = mt.valid ? (1.0f / 30.0f) : 0.f;
    } else {
      int ma;
      if (a.mask_attend) ma = a.mask_attend[mt.e];
      else ma = a.mask ? (a.mask[mt.node] * a.mask[mt.j]) : 1;
      mt.w_row = mt.valid ? ((float)ma * (1.0f / 30.0f)) : 0.f;
    }
    return mt;
  };

  long round = blockIdx.x;
  Meta cur = meta_of(round < aa.nrounds ? round : 0);
  f4 x[8], z1[8], pjv[8], gr[8], acc[8];
  auto load_hE = [&](const Meta& mt) {
    const float* src = a.hE + mt.e * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
  };
  load_hE(cur);
  issue_fill(0);

  // the previous round's dL/dh_E tile waits in `acc` until the next round's first ring point has passed (a store issued right
  // before a ring point would make s_waitcnt vmcnt(0) wait for its acknowledgement)
  bool have_prev = false;
  long e_prev = 0;
  bool valid_prev = false;
  auto store_prev = [&]() {
    if (have_prev && valid_prev) {
      float* d = a.g_hE + e_prev * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(d + 16 * t) = acc[t];
    }
  };

  for (; round < aa.nrounds; round += gridDim.x) {
    const Meta me = cur;
    const long round_n = round + gridDim.x;
    const bool more = round_n < aa.nrounds;
    // ================= ring point 1: fill 0 (W1b / its first half) and the h_E rows have landed
    ring_point();
    issue_fill(1);
    store_prev();
    {
      const float* pa = a.Pa + (long)me.node * NAMP_H + 4 * g;
      const float* pj = (me.from1 ? a.Pj1 : a.Pj0) + (long)me.j * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
    }
    // ---- z1 = W1b . h_E + Pa + Pj
    if constexpr (X3) {
      dw_half_gemm_x3<0>(z1, x, slot0, lane);
      ring_point();
      issue_fill(2);
      dw_half_gemm_x3<1>(z1, x, slot1, lane);
    } else {
      chain_gemm_bf16<false, false>(z1, x, (const bf8*)slot0 + lane);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) z1[t] += pjv[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = gelu_split4(z1[t]);          // x <- a1, z1 <- gelu'(z1)
    // ---- z2 = W2 . a1 + b2, g2 = w * g_node * gelu'(z2)
    ring_point();
    issue_fill(X3 ? 3 : 2);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(cstb + 16 * t + 4 * g);
    if constexpr (X3) {
      dw_half_gemm_x3<0>(acc, x, slot0, lane);
      ring_point();
      issue_fill(4);
      dw_half_gemm_x3<1>(acc, x, slot1, lane);
    } else {
      chain_gemm_bf16<false, false>(acc, x, (const bf8*)slot1 + lane);
    }
    {
      const float* src = a.g_node + (long)me.node * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        (void)gelu_split4(acc[t]);                                   // acc <- gelu'(z2)
        gr[t] = *(const f4*)(src + 16 * t) * me.w_row * acc[t];
      }
    }
    // ---- contraction 1: dW2 += G2^T A1, db2 += sum G2
    if constexpr (X3) {
      if (wave < 4) { dw_stage<PREC>(SG, gr, wave, m, g); dw_stage<PREC>(SA, x, wave, m, g); }
      ring_point();
      issue_fill(5);
      dw_contract<PREC, true>(dW2, db2, SG, SA, wo, wc, m, g);
      __syncthreads();
      if (wave >= 4) { dw_stage<PREC>(SG, gr, wave - 4, m, g); dw_stage<PREC>(SA, x, wave - 4, m, g); }
      __syncthreads();
      dw_contract<PREC, true>(dW2, db2, SG, SA, wo, wc, m, g);
    } else {
      dw_stage<PREC>(SG, gr, wave, m, g);
      dw_stage<PREC>(SA, x, wave, m, g);
      ring_point();
      issue_fill(3);
      dw_contract<PREC, true>(dW2, db2, SG, SA, wo, wc, m, g);
    }
    load_hE(me);                                                     // h_E rows again (L2): the second contraction's operand
    // ---- g1 = (W2^T g2) * gelu'(z1)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      dw_half_gemm_x3<0>(acc, gr, slot0, lane);
      ring_point();
      issue_fill(6);
      dw_half_gemm_x3<1>(acc, gr, slot1, lane);
    } else {
      chain_gemm_bf16<false, false>(acc, gr, (const bf8*)slot0 + lane);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = acc[t] * z1[t];
    // next round's indices (their loads fly under the second contraction)
    if (more) cur = meta_of(round_n);
    // ---- contraction 2: dW1b += G1^T h_E
    f4 nob = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      ring_point();                                                  // everyone is done with the staged set of contraction 1 and slot 1
      issue_fill(7);
      if (wave < 4) { dw_stage<PREC>(SG, gr, wave, m, g); dw_stage<PREC>(SA, x, wave, m, g); }
      __syncthreads();
      dw_contract<PREC, false>(dW1, nob, SG, SA, wo, wc, m, g);
      __syncthreads();
      if (wave >= 4) { dw_stage<PREC>(SG, gr, wave - 4, m, g); dw_stage<PREC>(SA, x, wave - 4, m, g); }
      __syncthreads();
    } else {
      ring_point();                                                  // everyone is done with SG / SA / slot 0; W1b^T has landed
      if (more) issue_fill(0);
      dw_stage<PREC>(SG, gr, wave, m, g);
      dw_stage<PREC>(SA, x, wave, m, g);
      __syncthreads();
    }
    // G1 rows (for the table-gradient gather) and the per-tile sums for dL/dPa go out behind a barrier, with the contraction
    // and the last product to retire under
    if (me.valid) {
#pragma unroll
      for (int t = 0; t < 8; ++t) st_row4<RB>(a.G1, me.e * NAMP_H + 4 * g + 16 * t, gr[t]);
    }
    if (a.g_Pa && a.gpa_tiles) {
      const long tile = round * 8 + wave;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f4 v = me.valid ? gr[t] : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
        }
        if (m == 0 && tile * 16 < a.E) *(f4*)(a.g_Pa + tile * NAMP_H + 16 * t + 4 * g) = v;
      }
    } else if (a.g_Pa && me.valid) {
      float* d = a.g_Pa + (long)me.node * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w);
      }
    }
    dw_contract<PREC, false>(dW1, nob, SG, SA, wo, wc, m, g);
    // ---- dL/dh_E = W1b^T g1 (+ the other consumer's rows); the next round's h_E rows are requested first
    if (more) load_hE(cur);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      acc[t] = (a.acc_hE && me.valid) ? *(const f4*)(a.g_hE_in + me.e * NAMP_H + 4 * g + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      dw_half_gemm_x3<0>(acc, gr, slot0, lane);
      ring_point();
      if (more) issue_fill(0);
      dw_half_gemm_x3<1>(acc, gr, slot1, lane);
    } else {
      chain_gemm_bf16<false, false>(acc, gr, (const bf8*)slot1 + lane);
    }
    have_prev = true; e_prev = me.e; valid_prev = me.valid;
  }
  store_prev();
  // ---- the workgroup's partial weight gradients: D[i = 4g + r][j = n] of tile (q, t) -> dW[64 wo + 16 q + 4g + r][32 wc + 16 t + n]
  float* o2 = aa.dW_part + (long)blockIdx.x * 2 * NAMP_H * NAMP_H;
  float* o1 = o2 + NAMP_H * NAMP_H;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 64 * wo + 16 * q + 4 * g + r, c = 32 * wc + 16 * t + m;
        o2[o * NAMP_H + c] = dW2[q][t][r];
        o1[o * NAMP_H + c] = dW1[q][t][r];
      }
  if (m == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) aa.db_part[(long)blockIdx.x * NAMP_H + 64 * wo + 16 * wc + 4 * g + r] = db2[r];
  }
}
