// edge_bwd_dw_kernel — the message-stage backward launch that OWNS its weight gradients (round 4).
//
// edge_chain_bwd_kernel (namp_train.h) writes the row tensors A1 = gelu(z1) and G2 = dL/dz2 to HBM only so that a second launch
// (wgrad_*) can read them back and contract them over the ~10^6 edge rows: dW2 = G2^T A1, dW1b = G1^T h_E, db2 = sum G2
// (na_model_utils.py:196-283: the gradients of EncLayer.W1/W2, DecLayer.W1/W2).  Here the contraction happens inside the launch:
//
//   * PERSISTENT workgroups (one per CU, 4 waves = one per SIMD with the whole 512-entry register file), each walking rounds of
//     4 x 16 consecutive edge rows.  A wave carries its 16-row
//     tile through the recomputed chain in registers exactly as edge_chain_bwd_kernel does.
//   * When a row operand pair (G, A) of a contraction exists in registers (lane (m, g) holds channels 16t + 4g + r of row m) the
//     waves write their tiles to LDS K-MAJOR — S[channel][row] in bf16, rows contiguous — so that the MFMA operand of the
//     contraction, "8 consecutive rows of one channel", is ONE 16-byte LDS read: lane (n, g) of v_mfma_f32_16x16x32_bf16 feeds
//     A[i = n][k = 8g + j] = G[row 32ks + 8g + j][channel 16q + n], B[k][n] = A[row][channel 16t + n].  The transposition is paid by
//     the writer as 2-byte LDS stores; a 16-byte XOR swizzle of the row chunks keeps reads conflict-free without padding.
//   * Wave (wo, wc) = (wave >> 1, wave & 1) owns the dW block [64 wo .. +64) x [64 wc .. +64) of each weight — 4 x 4 accumulator
//     tiles = 64 registers per weight — ACROSS ALL ITS ROUNDS; one [128 x 128] partial per workgroup and weight leaves the chip at
//     the end (the caller adds the <= 256 partials: deterministic).  db2 rides along as one more MFMA against a fragment of ones.
//   * Weights stream through a ring of two 32 KiB LDS slots by LDS-DMA one product ahead: a slot is a whole bf16 image (mixed
//     precision) or the K-half (steps s = 2h, 2h + 1; hi and mid planes) of an x3 image (split-bf16: every 128 x 128 product runs
//     as two half products with a ring point between).  Staged operands: the round's 64 rows of (G, A) in bf16 (2 x 16 KiB), or of
//     (G_hi, G_mid, A_hi, A_mid) for the split-bf16 contraction G_mid.A_hi + G_hi.A_mid + G_hi.A_hi (4 x 16 KiB).
//
// Not written any more: A1, G2 (2 x 590 MB per stage at cfg5 in fp32 rows).  Still written: G1 (the table-gradient gather
// dL/dPj reads it: namp_train_scatter_rows), dL/dh_E, the per-tile sums of G1 for dL/dPa.
#pragma once
#include "namp_train.h"

struct EdgeBwdDwArgs {
  EdgeBwdArgs b;        // A1, A2, G2, G3, S3, w3, g_Pj0, g_Pj1 unused
  float* dW_part;       // [gridDim.x][2][128][128]: 0 = dW2 = G2^T A1, 1 = dW1b = G1^T h_E
  float* db_part;       // [gridDim.x][128] = sum of G2 rows
  long nrounds;         // ceil(E / DW_ROWS)
};

#define DW_WAVES 4
#define DW_ROWS (16 * DW_WAVES)                  // edge rows per round and workgroup
#define DW_SLOT_BYTES 32768
#define DW_ROWB (2 * DW_ROWS)                    // bytes per channel row of a staged plane (64 rows of bf16)
#define DW_ARR (128 * DW_ROWB)                   // one staged plane: 16 KiB
#define DW_STAGE_BYTES (4 * DW_ARR)              // G_hi, G_mid, A_hi, A_mid (bf16 products use planes 0 and 2)
#define DW_LDS (2 * DW_SLOT_BYTES + DW_STAGE_BYTES + 512)
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

// workgroup barrier that waits for LDS traffic only: vector-memory requests stay in flight across it (__syncthreads() drains them too)
__device__ __forceinline__ void dw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Write this wave's 16-row tile v (register-chain layout: lane (m, g) holds channels 16t + 4g + r of row m) into the K-major staged
// plane at `base`: S[ch][16 wave + m] = bf16(v); X3: the plane behind it takes bf16 of the remainder.  Row chunk c (8 rows = 16 bytes)
// of channel ch sits at chunk position c ^ ((ch >> 1) & 7): with 128-byte channel rows a quarter-wave of fragment reads (16 channels, one
// chunk) then covers all 64 banks exactly once (tests/test_layout_sim.py).
// Ablation switches for tools/dw_time.py (never defined in the shipped build): DW_EXP_NOSTAGE / NOCONTRACT / NOGEMM / NOGELU / NOSTORE / NOLOAD
// drop the staging writes, the contraction, the chain products, the GELU evaluations, the row stores, the per-round row loads.
// Split-bf16 (X3: two planes per value), round 5: rows m and m ^ 1 of a channel are neighbours in the plane (2 bytes each) and live in neighbouring
// lanes: the even lane of a pair takes components 0, 1 of BOTH rows, the odd lane components 2, 3 (two DPP quad_perm [1,0,3,2] moves each way), so
// that a value pair goes out as one v_cvt_pk_bf16_f32 + ds_write_b32 — half the LDS write instructions: -4 % on the ring kernels (4.54 -> 4.35 ms
// per three cfg5 launches).  The one-plane bf16 form keeps single values: paired, it measured the same and cost edge_bwd_dw16_kernel its last registers.
template <bool X3>
__device__ __forceinline__ void dw_stage(char* base, const f4 (&v)[8], const int wave, const int m, const int g) {
#ifdef DW_EXP_NOSTAGE
  return;
#endif
  const int chunkv = 2 * wave + (m >> 3);
  if constexpr (!X3) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int chl = 4 * g + r;
    const int f = (chl >> 1) & 7;                       // ((16t + chl) >> 1) & 7 for every t
    char* p = base + chl * DW_ROWB + ((chunkv ^ f) << 4) + ((m & 7) << 1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float val = v[t][r];
      const __bf16 hi = (__bf16)val;
      *(__bf16*)(p + t * 16 * DW_ROWB) = hi;
    }
  }
  } else {
  const bool odd = (m & 1) != 0;
  const int c0 = 4 * g + (odd ? 2 : 0);                 // this lane's two channels (within a tile) of the row pair (m & ~1, m | 1)
  const int f = (c0 >> 1) & 7;                          // same swizzle for c0 and c0 + 1
  char* p = base + c0 * DW_ROWB + ((chunkv ^ f) << 4) + ((m & 6) << 1);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float s0 = odd ? v[t][0] : v[t][2], s1 = odd ? v[t][1] : v[t][3];          // what the partner lane packs
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xf, 0xf, true));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xf, 0xf, true));
    const float lo0 = odd ? r0 : v[t][0], hi0 = odd ? v[t][2] : r0;                  // (row m & ~1, row m | 1) of channel c0
    const float lo1 = odd ? r1 : v[t][1], hi1 = odd ? v[t][3] : r1;                  // ... of channel c0 + 1
    const bf2 a = (bf2){(__bf16)lo0, (__bf16)hi0}, b = (bf2){(__bf16)lo1, (__bf16)hi1};
    *(bf2*)(p + t * 16 * DW_ROWB) = a;
    *(bf2*)(p + DW_ROWB + t * 16 * DW_ROWB) = b;
    *(bf2*)(p + DW_ARR + t * 16 * DW_ROWB) = (bf2){(__bf16)(lo0 - (float)a[0]), (__bf16)(hi0 - (float)a[1])};
    *(bf2*)(p + DW_ARR + DW_ROWB + t * 16 * DW_ROWB) = (bf2){(__bf16)(lo1 - (float)b[0]), (__bf16)(hi1 - (float)b[1])};
  }
  }
}

// Sum over the 16 lanes of a DPP row (the 16 rows m of one channel group g): v += row_shr:8, 4, 2, 1 — the total ends in lane m = 15.  Plain
// VALU with a DPP source: the __shfl_xor butterflies of edge_chain_bwd_kernel compile to ds_bpermute_b32 (an LDS round trip each), which a
// launch with one wave per SIMD has nobody to cover (measured: 8 x 8 exposed waits per round).
template <int CTRL>
__device__ __forceinline__ float dw_dpp_add(const float v) {
  const int sh = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, sh);
}
// the same sum in EVERY lane of the row: rotations by 8, 4, 2, 1 (row_ror)
__device__ __forceinline__ float dw_row_allsum(float v) {
  v = dw_dpp_add<0x128>(v); v = dw_dpp_add<0x124>(v); v = dw_dpp_add<0x122>(v); v = dw_dpp_add<0x121>(v);
  return v;
}
__device__ __forceinline__ float dw_row_sum_to_lane15(float v) {
  v = dw_dpp_add<0x118>(v); v = dw_dpp_add<0x114>(v); v = dw_dpp_add<0x112>(v); v = dw_dpp_add<0x111>(v);
  return v;
}

// acc[q][t] += sum over the 64 staged rows of G[row][64 wo + 16 q + i] * A[row][64 wc + 16 t + j]   (lane (n, g) holds D[i = 4g + r][j = n]);
// BIAS: accb[u] += column sums of G over the rows for o-tile q = 2 wc + u (G^T . ones: every column of the result holds the sum).
template <bool X3, bool BIAS>
__device__ __forceinline__ void dw_contract(f4 (&acc)[4][4], f4 (&accb)[2], const char* SG, const char* SA, const int wo, const int wc,
                                            const int n, const int g) {
#ifdef DW_EXP_NOCONTRACT
  return;
#endif
  const int fsw = (n >> 1) & 7;
  const char* gb = SG + (64 * wo + n) * DW_ROWB;
  const char* ab = SA + (64 * wc + n) * DW_ROWB;
  bf8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
#pragma unroll 1
  for (int ks = 0; ks < DW_ROWS / 32; ++ks) {
    const int off = ((4 * ks + g) ^ fsw) << 4;
    bf8 gh[4], gm[4], ah[4], am[4];
    if constexpr (X3) {
      // split-bf16: the G fragments (hi, mid) stay for the step, the A fragments come two channel tiles at a time (48 fragment registers
      // instead of 64); product-major inside a pass: eight independent accumulators between two MFMAs on the same one
#pragma unroll
      for (int q = 0; q < 4; ++q) { gh[q] = *(const bf8*)(gb + q * 16 * DW_ROWB + off); gm[q] = *(const bf8*)(gb + DW_ARR + q * 16 * DW_ROWB + off); }
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          ah[u] = *(const bf8*)(ab + (2 * tp + u) * 16 * DW_ROWB + off);
          am[u] = *(const bf8*)(ab + DW_ARR + (2 * tp + u) * 16 * DW_ROWB + off);
        }
        __builtin_amdgcn_sched_barrier(2);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[q][2 * tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm[q], ah[u], acc[q][2 * tp + u], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[q][2 * tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], am[u], acc[q][2 * tp + u], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[q][2 * tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], ah[u], acc[q][2 * tp + u], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gh[q] = *(const bf8*)(gb + q * 16 * DW_ROWB + off);
        ah[q] = *(const bf8*)(ab + q * 16 * DW_ROWB + off);
      }
      __builtin_amdgcn_sched_barrier(2);             // all of the step's fragment reads go out before its MFMAs (one wait instead of one per read)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[q], ah[t], acc[q][t], 0, 0, 0);
    }
    if (BIAS) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf8 bh = wc ? gh[2 + u] : gh[u];
        if (X3) { const bf8 bm = wc ? gm[2 + u] : gm[u]; accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ones, accb[u], 0, 0, 0); }
        accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ones, accb[u], 0, 0, 0);
      }
    }
  }
}

// K-half HALF (steps s = 2 HALF, 2 HALF + 1) of a split-bf16 128 x 128 product out of one ring slot: hi plane at byte 0, mid plane at
// byte 16384, fragment (s_local, tn) at (8 s_local + tn) KiB.  Same product order as chain_gemm_x3 (namp_device.h).
template <int HALF>
__device__ __forceinline__ void dw_half_gemm_x3(f4 (&acc)[8], const f4 (&x)[8], const char* slot, const int lane) {
  // four groups (K-step sl, channel-tile half h) of 12 MFMAs; the hi fragments of group i + 1 are requested before the MFMAs of group i
  // issue, the mid fragments of group i right before its own (48 fragment registers; a full group ahead for both planes, 64, spills
  // beside the fill registers of edge_bwd_dw3_kernel)
  const bf8* wh = (const bf8*)slot + lane;
  const bf8* wm = (const bf8*)(slot + 16384) + lane;
  bf8 fh[2][4], fm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) fh[0][q] = wh[q * 64];
  bf8 hi, mid;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int sl = i >> 1, h = i & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) fm[q] = wm[(sl * 8 + 4 * h + q) * 64];
    if (i + 1 < 4) {
      const int sn = (i + 1) >> 1, hn = (i + 1) & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) fh[(i + 1) & 1][q] = wh[(sn * 8 + 4 * hn + q) * 64];
    }
    __builtin_amdgcn_sched_barrier(2);
    if (h == 0) split_x3(x[4 * HALF + 2 * sl], x[4 * HALF + 2 * sl + 1], hi, mid);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i & 1][q], mid, acc[4 * h + q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[q], hi, acc[4 * h + q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i & 1][q], hi, acc[4 * h + q], 0, 0, 0);
  }
}

// (Tried, round 4: the same ring filled THROUGH REGISTERS — each lane carries one 32-KiB fill as 8 x 16 bytes, requested from L2 half a
// product ahead and written with ds_write_b128 behind it, LDS-only barriers, the request schedule of edge_bwd_dw16_kernel — so that no
// wait drains the rows' requests: correct, but 1.89-1.99 ms per cfg5-sized launch against 1.62 ms for the LDS-DMA ring below and 1.72 ms
// for the round-3 launches; 50-60 spilled registers with the fill registers beside the split-bf16 fragments.  profiles/r04c.)
// MODE: BWD_ENC_MSG / BWD_DEC_MSG.  PREC: 1 split-bf16 products, 2 plain bf16 products (mixed precision; G1 rows bf16).
// One wave per SIMD (4 waves, __launch_bounds__(256): the 512-entry unified register file is one wave's): the chain's ~200 registers,
// 2 x 64 accumulator registers of weight gradients and two fragment sets in flight do not fit the 256 of a two-waves-per-SIMD launch
// (measured with the 8-wave form of this kernel: 55-190 spilled registers).  Matrix and vector time add up on this chip whichever wave
// issues them, so the second wave was only ever covering memory round trips — here that is the prefetches' job.
// ACC / GPA as in edge_bwd_dw16_kernel below; rows past E store into the buffers' padding (namp_train_edge_bwd_dw_rows).
template <int MODE, int PREC, bool ACC, int GPA>
__global__ __launch_bounds__(64 * DW_WAVES) void edge_bwd_dw_kernel(const EdgeBwdDwArgs aa) {
  static_assert(MODE == BWD_ENC_MSG || MODE == BWD_DEC_MSG, "message stages only");
  static_assert(PREC == 1 || PREC == 2, "split-bf16 or bf16 products");
  constexpr bool X3 = (PREC == 1);
  constexpr bool RB = (PREC == 2);                     // bf16 G1 rows
  const EdgeBwdArgs& a = aa.b;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slot0 = smem;
  char* slot1 = smem + DW_SLOT_BYTES;
  char* SG = smem + 2 * DW_SLOT_BYTES;
  char* SA = SG + 2 * DW_ARR;
  float* cstb = (float*)(smem + 2 * DW_SLOT_BYTES + DW_STAGE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int wo = wave >> 1, wc = wave & 1;
  if (tid < NAMP_H) cstb[tid] = a.b2[tid];

  f4 dW2[4][4], dW1[4][4], db2[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW2[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; dW1[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; }

  // ---- the weight ring.  Fill i of a round: bf16 -> image i (W1b, W2, W2^T, W1b^T); x3 -> K-half (i & 1) of image i >> 1.
  // Fill i always lands in slot i & 1 (4 / 8 fills per round: even).
  auto issue_fill = [&](const int i) {
    char* dst = (i & 1) ? slot1 : slot0;
    const int img_i = X3 ? (i >> 1) : i;
    const float* img = img_i == 0 ? a.W1_img : img_i == 1 ? a.W2_img : img_i == 2 ? a.W2t_img : a.W1t_img;
    if (X3) {
      const char* src = (const char*)img + (i & 1) * 16384;
      dma_to_lds(dst, (const float*)src, 16, wave, DW_WAVES, lane);
      dma_to_lds(dst + 16384, (const float*)(src + NAMP_BIMG_BYTES), 16, wave, DW_WAVES, lane);
    } else {
      dma_to_lds(dst, img, 32, wave, DW_WAVES, lane);
    }
  };
  auto ring_point = [&]() { wait_dma_and_sync(); };
  auto gemm = [&](f4 (&o)[8], const f4 (&in)[8], const int first_fill, const bool issue_next) {
    // one 128 x 128 product out of the ring; x3: two K-halves with a ring point between (the second half's slot was requested at the
    // ring point in front of the product; the NEXT product's first fill goes out at the middle one)
    if constexpr (X3) {
      dw_half_gemm_x3<0>(o, in, (first_fill & 1) ? slot1 : slot0, lane);
      ring_point();
      if (issue_next) issue_fill((first_fill + 2) & 7);
      dw_half_gemm_x3<1>(o, in, (first_fill & 1) ? slot0 : slot1, lane);
    } else {
      chain_gemm_bf16<false, false>(o, in, (const bf8*)((first_fill & 1) ? slot1 : slot0) + lane);
    }
  };

  // ---- per-round row bookkeeping
  struct Meta { long e; int node; int j; bool valid; float w_row; bool from1; };
  auto meta_of = [&](const long round) {
    Meta mt;
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;
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
  auto store_prev = [&]() {
    if (have_prev) {
      float* d = a.g_hE + e_prev * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(d + 16 * t) = acc[t];
    }
  };

  // Ring schedule (fill numbers within the round; bf16 | x3).  A fill is requested at the ring point after which its slot is free and
  // has one product (x3: half a product) plus the phase work behind it to land.
  //   RP1  [fill 0 landed]                 issue 1      z1 = W1b . h_E            (x3: RP, issue 2, second half)
  //   RP2  [1 | 2 landed]                  issue 2 | 3  z2 = W2 . a1 + b2         (x3: RP, issue 4, second half)
  //   RP3  [2 | 4 landed; staged G2, A1]   issue 3 | 5  dW2 += G2^T A1;  W2^T g2  (x3: RP, issue 6, second half)
  //   RP4  [3 | 6 landed; contraction 1 and the product read out]  issue 0' | 7;  stage G1, h_E;  barrier;  dW1b += G1^T h_E
  //        dL/dh_E = W1b^T g1                                                     (x3: RP, issue 0', second half)
  for (; round < aa.nrounds; round += gridDim.x) {
    const Meta me = cur;
    const long round_n = round + gridDim.x;
    const bool more = round_n < aa.nrounds;
    ring_point();                                                    // RP1
    issue_fill(1);
    store_prev();
    {
      const float* pa = a.Pa + (long)me.node * NAMP_H + 4 * g;
      const float* pj = (me.from1 ? a.Pj1 : a.Pj0) + (long)me.j * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); pjv[t] = *(const f4*)(pj + 16 * t); }
    }
    gemm(z1, x, 0, true);
#pragma unroll
    for (int t = 0; t < 8; ++t) z1[t] += pjv[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = gelu_split4(z1[t]);          // x <- a1, z1 <- gelu'(z1)
    ring_point();                                                    // RP2
    issue_fill(X3 ? 3 : 2);
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = *(const f4*)(cstb + 16 * t + 4 * g);
    gemm(acc, x, X3 ? 2 : 1, true);
    {
      const float* src = a.g_node + (long)me.node * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        (void)gelu_split4(acc[t]);                                   // acc <- gelu'(z2)
        gr[t] = *(const f4*)(src + 16 * t) * me.w_row * acc[t];      // g2 = w_ik * dL/d(K-sum) * gelu'(z2)
      }
    }
    // ---- contraction 1: dW2 += G2^T A1, db2 += sum G2
    dw_stage<X3>(SG, gr, wave, m, g);
    dw_stage<X3>(SA, x, wave, m, g);
    ring_point();                                                    // RP3
    issue_fill(X3 ? 5 : 3);
    dw_contract<X3, true>(dW2, db2, SG, SA, wo, wc, m, g);
    // ---- g1 = (W2^T g2) * gelu'(z1)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    gemm(acc, gr, X3 ? 4 : 2, true);
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = acc[t] * z1[t];
    load_hE(me);                                                     // h_E rows again (L2 / MALL): the second contraction's operand
    if (more) cur = meta_of(round_n);                                // next round's indices
    // ---- contraction 2: dW1b += G1^T h_E
    ring_point();                                                    // RP4
    if (X3) issue_fill(7);
    else if (more) issue_fill(0);
    dw_stage<X3>(SG, gr, wave, m, g);
    dw_stage<X3>(SA, x, wave, m, g);
    dw_lds_barrier();                                                // (LDS only: the row requests stay in flight)
    // G1 rows (for the table-gradient gather) and the per-tile sums for dL/dPa go out behind a barrier, with the contraction and the
    // last product to retire under
    {
      const long e_st = (round * DW_WAVES + wave) * 16 + m;         // unclamped: rows past E go to the buffers' padding
#pragma unroll
      for (int t = 0; t < 8; ++t) st_row4<RB>(a.G1, e_st * NAMP_H + 4 * g + 16 * t, gr[t]);
    }
    if constexpr (GPA == 1) {
      const long tile = round * DW_WAVES + wave;
      f4 keep = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f4 v = me.valid ? gr[t] : (f4){0.f, 0.f, 0.f, 0.f};
        v.x = dw_row_allsum(v.x); v.y = dw_row_allsum(v.y); v.z = dw_row_allsum(v.z); v.w = dw_row_allsum(v.w);
        keep = ((m & 7) == t) ? v : keep;
      }
      *(f4*)(a.g_Pa + tile * NAMP_H + 16 * (m & 7) + 4 * g) = keep;
    } else if constexpr (GPA == 2) {
      float* d = a.g_Pa + (long)me.node * NAMP_H + 4 * g;
      const float vz = me.valid ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x * vz); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y * vz);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z * vz); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w * vz);
      }
    }
    f4 nob[2];
    dw_contract<X3, false>(dW1, nob, SG, SA, wo, wc, m, g);
    // ---- dL/dh_E = W1b^T g1 (+ the other consumer's rows); the next round's h_E rows are requested first
    if (more) load_hE(cur);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      acc[t] = ACC ? *(const f4*)(a.g_hE_in + me.e * NAMP_H + 4 * g + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
    gemm(acc, gr, X3 ? 6 : 3, more);
    have_prev = true; e_prev = (round * DW_WAVES + wave) * 16 + m;
  }
  store_prev();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA in flight when the workgroup's LDS is released
  // ---- the workgroup's partial weight gradients: D[i = 4g + r][j = n] of tile (q, t) -> dW[64 wo + 16 q + 4g + r][64 wc + 16 t + n]
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


// ---- mixed precision (plain bf16 products): the WEIGHT-STATIONARY form -----------------------------------------------------------
// The four bf16 images of a message stage (W1b, W2, W2^T, W1b^T: 4 x 32 KiB) stay in LDS for the life of the persistent workgroup;
// the remaining 32 KiB hold the round's staged (G, A) planes.  No weight traffic per round, no LDS-DMA waits: the only barriers left
// are the four around the two staged contractions, and they wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier) — global loads
// (next round's rows, the tables, dL/dh_E of the other consumer) stay in flight across them.  The ring form above streams
// 128 KiB of weights per 64 rows at LDS-DMA's ~25 GB/s per CU: 5.2 us per round before any arithmetic.
#define DW16_LDS (4 * NAMP_BIMG_BYTES + 2 * DW_ARR)


// One 16-row x 128 x 128 bf16 product out of a resident image with the eight weight fragments of K-step s + 1 requested before the
// MFMAs of step s issue (two fragment sets in flight).  One wave per SIMD has nobody to cover an LDS round trip: the compiler's own
// schedule (read, wait, MFMA — nearly one for one) made this launch 13 us per 64-row round where its MFMAs need 1.3.
__device__ __forceinline__ void dw_gemm16_ahead(f4 (&acc)[8], const f4 (&x)[8], const bf8* w) {
#ifdef DW_EXP_NOGEMM
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] += x[t];
  return;
#endif
  // half-steps of four channel tiles: the four fragments of half-step i + 1 are requested before the MFMAs of half-step i issue
  // (two sets of four in flight: 32 registers — a full step ahead, 64, spills next to the launch's prefetched row streams)
  bf8 wf[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wf[0][q] = w[q * 64];
  bf8 xb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s = i >> 1, h = i & 1;
    if (i + 1 < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[(i + 1) & 1][q] = w[(((i + 1) >> 1) * 8 + 4 * ((i + 1) & 1) + q) * 64];
    }
    __builtin_amdgcn_sched_barrier(2);               // the requests stay ahead of this half-step's MFMAs; VALU (the operand pack) may cross
    if (h == 0) xb = pack_bf16<false>(x[2 * s], x[2 * s + 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i & 1][q], xb, acc[4 * h + q], 0, 0, 0);
  }
}

// dw_contract<false, BIAS> with both K-steps' fragments requested up front (16 ds_read_b128 in flight, 64 registers)
template <bool BIAS>
__device__ __forceinline__ void dw_contract16_ahead(f4 (&acc)[4][4], f4 (&accb)[2], const char* SG, const char* SA, const int wo, const int wc,
                                                    const int n, const int g) {
  const int fsw = (n >> 1) & 7;
  const char* gb = SG + (64 * wo + n) * DW_ROWB;
  const char* ab = SA + (64 * wc + n) * DW_ROWB;
  bf8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  bf8 gh[2][4], ah[2][4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int off = ((4 * ks + g) ^ fsw) << 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) { gh[ks][q] = *(const bf8*)(gb + q * 16 * DW_ROWB + off); ah[ks][q] = *(const bf8*)(ab + q * 16 * DW_ROWB + off); }
  }
  __builtin_amdgcn_sched_barrier(2);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh[ks][q], ah[ks][t], acc[q][t], 0, 0, 0);
    if (BIAS) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf8 bh = wc ? gh[ks][2 + u] : gh[ks][u];
        accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ones, accb[u], 0, 0, 0);
      }
    }
  }
}

// gelu and gelu' of the mixed-precision launch: Phi(x) = clamp01(1/2 + x Q(x^2)) with the degree-4 polynomial of the bf16 throughput mode
// (namp_device.h gelu4_bf16mode: max |error| 1.3e-3 — a third of the bf16 rounding step its result and everything it multiplies are
// rounded with), phi(x) = exp(-x^2 / 2) / sqrt(2 pi) exactly: gelu = x Phi, gelu' = Phi + x phi.  12 operations per value (packable)
// instead of the 17 of gelu_val_grad's Abramowitz-Stegun form (two transcendentals there, one here).
// in: pre-activations z; out: z <- gelu'(z), returns gelu(z)
__device__ __forceinline__ f4 dw_gelu_split4_bf16(f4& z) {
#ifdef DW_EXP_NOGELU
  { const f4 v_ = z; z = z * 0.5f; return v_; }
#endif
  const f4 x = z;
  const f4 t = x * x;
  f4 q = (f4){NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4};       // the constants of gelu4_bf16mode (namp_device.h)
  q = q * t + NAMP_GELU4_Q3;
  q = q * t + NAMP_GELU4_Q2;
  q = q * t + NAMP_GELU4_Q1;
  q = q * t + NAMP_GELU4_Q0;
  f4 p = x * q + 0.5f;
  p = (f4){__builtin_amdgcn_fmed3f(p.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(p.y, 0.f, 1.f), __builtin_amdgcn_fmed3f(p.z, 0.f, 1.f),
           __builtin_amdgcn_fmed3f(p.w, 0.f, 1.f)};
  const f4 u = t * -0.72134752044448170f;                     // -x^2 / 2 * log2(e)
  const f4 e = (f4){__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y), __builtin_amdgcn_exp2f(u.z), __builtin_amdgcn_exp2f(u.w)};
  z = (x * e) * 0.3989422804014327f + p;
  return x * p;
}

// dw_stage<false> from a tile kept as packed bf16 pairs: h[2t + (r >> 1)][r & 1] = bf16(v[t][r])
__device__ __forceinline__ void dw_stage_packed(char* base, const bf2 (&h)[16], const int wave, const int m, const int g) {
#ifdef DW_EXP_NOSTAGE
  return;
#endif
  const int chunkv = 2 * wave + (m >> 3);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int chl = 4 * g + r;
    const int f = (chl >> 1) & 7;
    char* p = base + chl * DW_ROWB + ((chunkv ^ f) << 4) + ((m & 7) << 1);
#pragma unroll
    for (int t = 0; t < 8; ++t) *(__bf16*)(p + t * 16 * DW_ROWB) = h[2 * t + (r >> 1)][r & 1];
  }
}

// ACC: the other consumer's dL/dh_E rows (g_hE_in) are added.  GPA: 1 = per-tile sums of G1 for dL/dPa (K % 16 == 0), 2 = fp32 atomics.
// No vector-memory instruction of the round loop sits under a branch: gfx9's one in-order counter makes the compiler wait for EVERYTHING
// outstanding — store acknowledgements included — at the first use of a load behind a join (measured: 6,000 of a round's 30,000 cycles at
// the top of the round with `if (valid) store` / `if (m == 15) store` in the loop).  Rows past E therefore compute on clamped inputs and
// store into PADDING: G1 / g_hE must hold 64 * ceil(E / 64) rows, g_Pa one row per 16 of those (namp_train_edge_bwd_dw_rows).
template <int MODE, bool ACC, int GPA>
__global__ __launch_bounds__(64 * DW_WAVES) void edge_bwd_dw16_kernel(const EdgeBwdDwArgs aa) {
  static_assert(MODE == BWD_ENC_MSG || MODE == BWD_DEC_MSG, "message stages only");
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
  const bf8* w2t = (const bf8*)(smem + 2 * NAMP_BIMG_BYTES) + lane;
  const bf8* w1t = (const bf8*)(smem + 3 * NAMP_BIMG_BYTES) + lane;
  copy_to_lds<4>(smem, a.W1_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + 2 * NAMP_BIMG_BYTES, a.W2t_img, 32, wave, DW_WAVES, lane);
  copy_to_lds<4>(smem + 3 * NAMP_BIMG_BYTES, a.W1t_img, 32, wave, DW_WAVES, lane);

  f4 dW2[4][4], dW1[4][4], db2[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW2[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; dW1[q][t] = (f4){0.f, 0.f, 0.f, 0.f}; }

  // Row bookkeeping in three steps, so that no step waits for a load it has just issued: (1) arithmetic (edge row, residue) + the request
  // of E_idx; (2) the neighbour's global row and the requests that depend on it (its rank / mask); (3) the row weight and table choice.
  // (row numbers are recomputed from the round where they are needed — a few integer operations — instead of being carried in registers)
  struct Meta { long e; int node; int j; bool valid; float w_row; bool from1; };
  struct Pend { int node; int i_loc; int idx; int a0, a1; };     // a0 / a1: rank[j], rank[node] or mask[node], mask[j] (or mask_attend)
  auto row_of = [&](const long round) {                             // clamped edge row of this lane in `round`
    const long e_raw = (round * DW_WAVES + wave) * 16 + m;
    return e_raw < a.E ? e_raw : (a.E - 1);
  };
  auto pend1 = [&](const long round, const int idx) {                // idx = E_idx[row_of(round)], requested TWO rounds ahead
    Pend p;
    p.node = (int)(row_of(round) / a.K);
    p.i_loc = p.node % a.N;
    p.idx = idx;
    p.a0 = p.a1 = 1;
    return p;
  };
  auto pend2 = [&](Pend& p, const long p_round) {
    const int j = p.node - p.i_loc + p.idx;
    if (MODE == BWD_DEC_MSG) { p.a0 = a.rank[j]; p.a1 = a.rank[p.node]; }
    else {
      // branch-free (no request of the round loop sits under a branch): an explicit mask_attend row is read twice — its entries are
      // 0 / 1, so a0 * a1 = a0 —, otherwise mask[i] and mask[j] (the entry point insists on one of the two)
      const int32_t* q0 = a.mask_attend ? a.mask_attend + row_of(p_round) : a.mask + p.node;
      const int32_t* q1 = a.mask_attend ? q0 : a.mask + j;
      p.a0 = *q0; p.a1 = *q1;
    }
  };
  auto pend3 = [&](const Pend& p, const long p_round) {
    Meta mt;
    const long e_raw = (p_round * DW_WAVES + wave) * 16 + m;
    mt.valid = e_raw < a.E;
    mt.e = mt.valid ? e_raw : (a.E - 1);
    mt.node = p.node; mt.j = p.node - p.i_loc + p.idx;
    if (MODE == BWD_DEC_MSG) { mt.from1 = !(p.a0 < p.a1); mt.w_row = mt.valid ? (1.0f / 30.0f) : 0.f; }
    else { mt.from1 = false; mt.w_row = mt.valid ? ((float)(p.a0 * p.a1) * (1.0f / 30.0f)) : 0.f; }
    return mt;
  };
  // Register roles.  x: h_E rows, then a1, then (from the first staging on) the NEXT round's h_E rows.  z1: Pa (+ Pj), the first
  // product, gelu'(z1); from g1 on the next round's Pa rows.  gr: dL/d(K-sum) rows, g2, g1.  A: arrives holding this round's gathered
  // Pj rows (added to z1 in the first lines), then b2, z2, the third product, and from g1 on the NEXT round's Pj rows.  P: the other
  // consumer's dL/dh_E rows, requested behind g1, accumulates the last product and is stored.
  // gfx9 has ONE in-order counter for vector-memory loads and stores: waiting for a load also waits for everything issued before it.
  // The requests of a round are therefore issued (nearly) in the order they are consumed — b2, E_idx of the next round, dL/d(K-sum) rows,
  // rank / mask of the next round, the next round's h_E rows (HBM), its table rows, the other consumer's rows (HBM); stores last — so
  // that no wait covers a younger long-latency request.
  f4 x[8], z1[8], gr[8], A[8], P[8];
  bf2 h16[16], d16[16];
  long round = blockIdx.x;
  Meta cur;
  {
    const long r0 = round < aa.nrounds ? round : 0;
    Pend p0 = pend1(r0, a.E_idx[row_of(r0)]);
    pend2(p0, r0);
    cur = pend3(p0, r0);
    const float* src = a.hE + cur.e * NAMP_H + 4 * g;
    const float* pa = a.Pa + (long)cur.node * NAMP_H + 4 * g;
    const float* pj = (cur.from1 ? a.Pj1 : a.Pj0) + (long)cur.j * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) { x[t] = *(const f4*)(src + 16 * t); z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
  }
  // E_idx of the NEXT round's rows (the one after it is requested at the top of every round: two rounds of look-ahead, so that the
  // neighbour's rank / mask can be requested a whole round before the row weight is needed)
  int idx_n1 = a.E_idx[row_of(round + gridDim.x < aa.nrounds ? round + gridDim.x : round)];
  __syncthreads();                                                   // the images are in place

#ifdef DW_EXP_STAMPS
  // phase stamps (s_memtime) of workgroup 0 into the otherwise unused S3 pointer: [round][wave][32] (tools/dw_time.py --stamps)
  int stamp_round = 0;
#define DW_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && stamp_round < 8 && a.S3) ((long long*)a.S3)[(stamp_round * DW_WAVES + wave) * 32 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#endif
  for (; round < aa.nrounds; round += gridDim.x) {
    const Meta me = cur;
    DW_STAMP(0);
    const long round_n = round + gridDim.x;
    const bool more = round_n < aa.nrounds;
    // ---- z1 = W1b . h_E + (Pa + Pj).  h_E is kept as packed bf16 (16 registers) for the second contraction instead of being read again.
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      z1[t] += A[t];
      h16[2 * t] = (bf2){(__bf16)x[t].x, (__bf16)x[t].y};
      h16[2 * t + 1] = (bf2){(__bf16)x[t].z, (__bf16)x[t].w};
    }
    // (last rounds: their own rows again — no branch around the requests)
    const long rn = more ? round_n : round;
    Pend pn = pend1(rn, idx_n1);
    pend2(pn, rn);                                                   // rank / mask of the next round's rows: consumed behind g1
    const long round_n2 = round_n + gridDim.x;
    idx_n1 = a.E_idx[row_of(round_n2 < aa.nrounds ? round_n2 : round)];
#ifndef DW_EXP_NOLOAD
#pragma unroll
    for (int t = 0; t < 8; ++t) A[t] = *(const f4*)(a.b2 + 16 * t + 4 * g);      // (L1: 512 bytes every lane re-reads)
#else
#pragma unroll
    for (int t = 0; t < 8; ++t) A[t] = x[t];
#endif
    DW_STAMP(1);
    dw_gemm16_ahead(z1, x, w1);
    DW_STAMP(2);
    // dL/d(K-sum) rows of the tile's residues (L1 / L2): requested here, consumed behind the second product
#ifndef DW_EXP_NOLOAD
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = *(const f4*)(a.g_node + (long)me.node * NAMP_H + 4 * g + 16 * t);
#else
#pragma unroll
    for (int t = 0; t < 8; ++t) gr[t] = x[t];
#endif
    // x <- a1; gelu'(z1) is kept as packed bf16 (16 registers; it multiplies a bf16-grade gradient): z1 is free for the next round's Pa rows
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      x[t] = dw_gelu_split4_bf16(z1[t]);
      d16[2 * t] = (bf2){(__bf16)z1[t].x, (__bf16)z1[t].y};
      d16[2 * t + 1] = (bf2){(__bf16)z1[t].z, (__bf16)z1[t].w};
    }
    DW_STAMP(3);
    // ---- z2 = W2 . a1 + b2;  g2 = w_ik * dL/d(K-sum) * gelu'(z2)
    dw_gemm16_ahead(A, x, w2);
    DW_STAMP(4);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      (void)dw_gelu_split4_bf16(A[t]);                                       // A <- gelu'(z2)
      gr[t] = gr[t] * me.w_row * A[t];
    }
    DW_STAMP(5);
    // ---- contraction 1: dW2 += G2^T A1, db2 += sum G2
    dw_lds_barrier();                                                // the previous round's second contraction has been read out
    DW_STAMP(6);
    dw_stage<false>(SG, gr, wave, m, g);
    dw_stage<false>(SA, x, wave, m, g);
    DW_STAMP(7);
#ifndef DW_EXP_NOLOAD
    {                                                                // the next round's h_E rows: two products and two contractions to land
      const float* src = a.hE + row_of(rn) * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
    }
#endif
    dw_lds_barrier();
    DW_STAMP(8);
    dw_contract<false, true>(dW2, db2, SG, SA, wo, wc, m, g);
    DW_STAMP(9);
    // ---- g1 = (W2^T g2) * gelu'(z1)
#pragma unroll
    for (int t = 0; t < 8; ++t) A[t] = (f4){0.f, 0.f, 0.f, 0.f};
    dw_gemm16_ahead(A, gr, w2t);
    DW_STAMP(10);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      gr[t] = A[t] * (f4){(float)d16[2 * t][0], (float)d16[2 * t][1], (float)d16[2 * t + 1][0], (float)d16[2 * t + 1][1]};
    cur = pend3(pn, rn);
#ifndef DW_EXP_NOLOAD
    {                                                                // the next round's table rows (L2): Pa -> z1, Pj -> A
      const float* pa = a.Pa + (long)cur.node * NAMP_H + 4 * g;
      const float* pj = (cur.from1 ? a.Pj1 : a.Pj0) + (long)cur.j * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) { z1[t] = *(const f4*)(pa + 16 * t); A[t] = *(const f4*)(pj + 16 * t); }
      // the other consumer's dL/dh_E rows (HBM) -> P, the last product's accumulator: two barriers, the staging, the row stores and a
      // contraction to land (requested earlier they would be 32 more live registers across three products: the launch spills)
#pragma unroll
      for (int t = 0; t < 8; ++t) P[t] = ACC ? *(const f4*)(a.g_hE_in + me.e * NAMP_H + 4 * g + 16 * t) : (f4){0.f, 0.f, 0.f, 0.f};
    }
#else
#pragma unroll
    for (int t = 0; t < 8; ++t) P[t] = gr[t];
#endif
    DW_STAMP(11);
    // ---- contraction 2: dW1b += G1^T h_E
    dw_lds_barrier();                                                // contraction 1 has been read out
    DW_STAMP(12);
    dw_stage<false>(SG, gr, wave, m, g);
    dw_stage_packed(SA, h16, wave, m, g);
    DW_STAMP(13);
    dw_lds_barrier();
    DW_STAMP(14);
#ifndef DW_EXP_NOSTORE
    const long e_st = (round * DW_WAVES + wave) * 16 + m;           // unclamped: rows past E go to the buffers' padding
    st_tile_bf16(a.G1, e_st * NAMP_H, gr, g);
    if constexpr (GPA == 1) {
      // per-tile sums of G1 over the tile's 16 rows: a rotate all-reduce inside each DPP row (every lane ends with the 32 sums of its channel
      // group), then lane m keeps channel tile m & 7 — ONE unconditional 16-byte store per lane (lanes m and m + 8 write the same bytes)
      const long tile = round * DW_WAVES + wave;
      f4 keep = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f4 v = me.valid ? gr[t] : (f4){0.f, 0.f, 0.f, 0.f};
        v.x = dw_row_allsum(v.x); v.y = dw_row_allsum(v.y); v.z = dw_row_allsum(v.z); v.w = dw_row_allsum(v.w);
        keep = ((m & 7) == t) ? v : keep;
      }
      *(f4*)(a.g_Pa + tile * NAMP_H + 16 * (m & 7) + 4 * g) = keep;
    } else if constexpr (GPA == 2) {
      float* d = a.g_Pa + (long)me.node * NAMP_H + 4 * g;
      const float vz = me.valid ? 1.f : 0.f;                         // rows past E add zeros (to the last residue's row)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        unsafeAtomicAdd(d + 16 * t + 0, gr[t].x * vz); unsafeAtomicAdd(d + 16 * t + 1, gr[t].y * vz);
        unsafeAtomicAdd(d + 16 * t + 2, gr[t].z * vz); unsafeAtomicAdd(d + 16 * t + 3, gr[t].w * vz);
      }
    }
#endif
    DW_STAMP(15);
    f4 nob[2];
    dw_contract<false, false>(dW1, nob, SG, SA, wo, wc, m, g);
    DW_STAMP(16);
    // ---- dL/dh_E = W1b^T g1 + the other consumer's rows (P)
    dw_gemm16_ahead(P, gr, w1t);
    DW_STAMP(17);
#ifndef DW_EXP_NOSTORE
    {
      float* d = a.g_hE + ((round * DW_WAVES + wave) * 16 + m) * NAMP_H + 4 * g;
#pragma unroll
      for (int t = 0; t < 8; ++t) *(f4*)(d + 16 * t) = P[t];
    }
#endif
    DW_STAMP(18);
#ifdef DW_EXP_STAMPS
    ++stamp_round;
#endif
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

