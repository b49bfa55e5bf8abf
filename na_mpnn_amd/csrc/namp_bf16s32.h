// edge_mlp_bf16s32_kernel — the edge launches of the bf16-storage path (large batches of the bf16 mode, namp.hip: encdec_bf16_storage) on
// v_mfma_f32_32x32x16_bf16: EncLayer message (model_utils.py:666-672; optionally with h_E = W_e . E + b_e, :89, in front), EncLayer edge
// update (:697-702), DecLayer message on the implicit context (:636-646).  Round 3; the 16x16x32 form it replaced: profiles/r03e.
//
// One wave = 32 rows = two consecutive 16-row tiles of the flat tile list (the halves may belong to different residues).
//   v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l&31][k = 8(l>>5) + j], B[k = 8(l>>5) + j][n = l&31], j = 0..7, and receives
//   D[i = 8(v>>2) + 4(l>>5) + (v&3)][n = l&31], v = 0..15.
// T orientation (A = weights, B = activations): lane (row r = l&31, hk = l>>5) ends with channels 32tn + 8(v>>2) + 4hk + (v&3) of ITS row —
// the operand of the next layer's K-step s' = 2tn + (v>>3) if the reduction index is enumerated k(s', hk, j) = 16s' + 8(j>>2) + 4hk + (j&3).
// So a row of 128 bf16 is stored in the order [s' 0..7][hk 0..1][j 0..7] (16-byte piece (s', hk) = what lane (r, hk) feeds into step s'), and
//   img32[s'][tn 0..3][lane][j] = bf16(W[32tn + (lane&31)][k(s', lane>>5, j)])          (32 KiB per 128 x 128 block, like the 16x16x32 image)
// serves both orientations.  Per 128 x 128 product and 32 rows: 32 MFMAs of 32 cycles — the pipe time of the 64 16x16x32 MFMAs — but HALF the
// ds_read_b128 of weight fragments, and a VALU instruction issued beside them costs 1.3-1.5 cycles instead of 1.65-2.1 (profiles/r03a).
// F orientation (A = activations, B = weights) for layer 2: lane (channel 32tn + (l&31), hk) holds rows 8(v>>2) + 4hk + (v&3): v < 8 are rows of
// the first 16-row tile, v >= 8 of the second, so each half's K-sum is 8 in-lane FMAs and ONE cross-lane step.
#pragma once
#include "namp_kernels.h"

typedef float f16v __attribute__((ext_vector_type(16)));

static __global__ void pack_image_bf16_32_kernel(const float* __restrict__ W, int ld, int col0, __bf16* __restrict__ img) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 128 * 128) return;
  const int j = e & 7, lane = (e >> 3) & 63, tn = (e >> 9) & 3, s = e >> 11;
  const int n = 32 * tn + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
  img[e] = (__bf16)W[(size_t)n * ld + col0 + k];
}

// GELU (bf16-mode polynomial) of eight accumulators -> the bf16 operand of one K-step
__device__ __forceinline__ bf8 gelu_pack8(const f16v& v, const int u) {
#ifdef B32_SCALAR
  // eight independent scalar Horner chains (no packed-fp32 forms: a v_pk_fma_f32 costs more than two v_fma_f32 beside MFMAs, profiles/r03c)
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = v[8 * u + i], t = x * x;
    float q = fmaf(NAMP_GELU4_Q4, t, NAMP_GELU4_Q3);
    q = fmaf(q, t, NAMP_GELU4_Q2); q = fmaf(q, t, NAMP_GELU4_Q1); q = fmaf(q, t, NAMP_GELU4_Q0);
    float p;
    asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(p) : "v"(x), "v"(q));
    y[i] = x * p;
  }
  bf8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (__bf16)y[i];
  return o;
#endif
  const f4 lo = (f4){v[8 * u + 0], v[8 * u + 1], v[8 * u + 2], v[8 * u + 3]};
  const f4 hi = (f4){v[8 * u + 4], v[8 * u + 5], v[8 * u + 6], v[8 * u + 7]};
  return pack_bf16<true>(lo, hi);
}

// One 32-row x 128 x 128 product on v_mfma_f32_32x32x16_bf16 with the weight fragments of K-step s + 1 requested from LDS before the MFMAs
// of step s issue (two fragment sets in flight: 32 VGPRs).  Left to itself the compiler reads two fragments, waits for them, issues two
// MFMAs, and so on: every pair of MFMAs then pays a full LDS round trip, with only two waves per SIMD to cover it.
//   FLIP: F orientation (A = activations, B = weights).  op(s) returns the activation operand of K-step s (a stored row piece, or the
//   GELU of eight accumulators of the previous layer, evaluated while the fragments are on their way).
// PIN: a scheduling barrier behind the requests that only VALU instructions may cross (`__builtin_amdgcn_sched_barrier(2)`) keeps them ahead
// of the step's MFMAs.  Measured (cfg3, same box, alternating): decoder message 0.407-0.412 -> 0.387-0.390 ms, but the edge update
// 0.59 -> 0.71 and the embedding variant slower too (they are at 256 VGPRs: the barrier makes them spill).  The edge update pins once its
// next-row prefetch is issued late (see the kernel); the embedding variant does not pin.
template <bool FLIP, bool PIN, class Op>
__device__ __forceinline__ void gemm32(f16v (&out)[4], const bf8* w, Op op) {
  bf8 wf[2][4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) wf[0][tn] = w[tn * 64];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s + 1 < 8) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) wf[(s + 1) & 1][tn] = w[((s + 1) * 4 + tn) * 64];
    }
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(2);
    const bf8 ab = op(s);
#ifdef B32_NOLDSW
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) { wf[s & 1][tn] = ab; wf[s & 1][tn][0] = (__bf16)(float)(s * 4 + tn); }
#endif
#ifdef B32_NOMFMA
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) asm volatile("" :: "v"(wf[s & 1][tn]), "v"(ab));
    continue;
#endif
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
      out[tn] = FLIP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, wf[s & 1][tn], out[tn], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s & 1][tn], ab, out[tn], 0, 0, 0);
  }
}

// Every store of the step loop is issued by every lane on every step (rows of a missing second tile go to these dump words): the number of
// stores behind a step's row / Pa requests is then a constant, for the explicit vmcnt waits below and for the compiler's own.
__device__ float g_bf16s32_dump[256];
__device__ __bf16 g_bf16s32_dump16[16 * NAMP_H];

// LDS: W1 | W2 | W3 or W_e (32 KiB each) | constants (2 KiB) | per wave: 32 row weights (128 B); edge update: + two Pa rows per wave (512 B)
#define BF16S32_LDS (3 * NAMP_BIMG_BYTES + 2048 + 8 * 128 + 8 * 512)

// EMB (first encoder message): the rows arrive as the fp32 edge features E (a.hE); h_E = W_e . E + b_e (a.eW1_img, a.eb2) is evaluated
// here, stored as bf16 rows (a.hE16_out) and fed straight into the message MLP (as edge_mlp_bf16s_kernel<MODE_ENC_MSG, true>).
template <int MODE, bool EMB = false>
__global__ __launch_bounds__(512) void edge_mlp_bf16s32_kernel(const EdgeArgs a) {
  static_assert(!EMB || MODE == MODE_ENC_MSG, "EMB: first encoder message only");
  constexpr bool EDGE = MODE == MODE_ENC_EDGE;
  constexpr bool PIN = !EDGE && !EMB;                 // see gemm32
  // the edge update pins all three products and requests the next pair's rows late (behind layer 2) instead of at the top of the step: 32
  // registers less across layers 1 and 2, which pays for the second fragment set (4 spilled registers; 0.565 -> 0.533 ms per launch).  The
  // embedding variant (fp32 rows in flight) spills 23 registers with the same treatment and keeps the compiler's schedule.
  constexpr bool EPIN = EDGE, LATE_ROW = EDGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int r = lane & 31, hk = lane >> 5, m = r & 15, half = r >> 4;
  const long ntiles = (long)a.G * a.TPN;
  const long npairs = (ntiles + 1) >> 1;
  const long stride = (long)gridDim.x * nwaves;
  long pair = (long)blockIdx.x * nwaves + wave;
  auto meta_of = [&](const long p, const int j_pre) {
    const long t = 2 * p + half;
    const bool ok = t < ntiles;
    TileMeta mt = EDGE ? tile_meta<MODE>(a, ok ? t : (ntiles - 1), m, 0) : tile_meta_pre<MODE>(a, ok ? t : (ntiles - 1), m, j_pre);
    if (!ok) { mt.valid = false; mt.w_row = 0.f; }
    return mt;
  };
  // the neighbour id a step's metadata starts from is requested a whole step before that metadata is evaluated: its dependent reads
  // (rank / mask of the neighbour) then never wait on a fresh request
  // (past the end: the last pair's ids, and the metadata below is then evaluated for that pair too — never used, but its addresses are real)
  auto pair_c = [&](const long p) { return p < npairs ? p : npairs - 1; };
  auto idx_of = [&](const long p) {
    if constexpr (EDGE) return 0;      // (the edge update has no register for it: its metadata reads the id itself)
    const long t = 2 * pair_c(p) + half;
    return a.E_idx[tile_erow<MODE>(a, t < ntiles ? t : (ntiles - 1), m)];
  };
  TileMeta cur = meta_of(pair_c(pair), idx_of(pair));
  int idx_n1 = idx_of(pair + stride);
  bf8 xn[8];
  auto row_fetch = [&](const TileMeta& mt) {
    if constexpr (EMB) {
      // fp32 E row: this lane's channels of K-step s are 16s + 4hk + (0..3) and 16s + 8 + 4hk + (0..3); rounded to bf16 on arrival
      const float* src = a.hE + mt.erow * NAMP_H + 4 * hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) xn[s] = pack_bf16<false>(*(const f4*)(src + 16 * s), *(const f4*)(src + 16 * s + 8));
    } else {
      const bf8* src = (const bf8*)(a.hE16 + mt.erow * NAMP_H) + hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) xn[s] = src[2 * s];
    }
  };
  row_fetch(cur);
  float* w_slot = (float*)(smem + 3 * NAMP_BIMG_BYTES + 2048 + wave * 128);   // 32 row weights
  // Pa, edge update only (PA_DMA): the two rows of a step by LDS-DMA a step ahead, read back as the accumulators' initial value.  The
  // message modes read Pa with plain loads next to the gathered term instead (below); the edge update has no 32 registers for that.
  constexpr bool PA_DMA = EDGE;
  char* pa_slot = smem + 3 * NAMP_BIMG_BYTES + 2048 + 8 * 128 + wave * 512;
  auto pa_fetch = [&](const TileMeta& mt) {
    const long rowA = __builtin_amdgcn_readlane(mt.node, 0), rowB = __builtin_amdgcn_readlane(mt.node, 16);     // pa_row == node
    if (lane < 32) {
      const long row = lane < 16 ? rowA : rowB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.Pa16 + row * NAMP_H + 8 * (lane & 15)),
                                       (__attribute__((address_space(3))) void*)pa_slot, 16, 0, 0);
    }
  };
  if constexpr (PA_DMA) pa_fetch(cur);
  dma_to_lds(smem, a.W1_img, 32, wave, nwaves, lane);
  dma_to_lds(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, nwaves, lane);
  if (EDGE) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.W3_img, 32, wave, nwaves, lane);
  if (EMB) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.eW1_img, 32, wave, nwaves, lane);
  float* cst = (float*)(smem + 3 * NAMP_BIMG_BYTES);          // b2 | b3 | LayerNorm-3 weight | bias   (EMB: b2 | b_e)
  if (EDGE && tid < 512) {
    const float* srcv = tid < 128 ? a.b2 : tid < 256 ? a.b3 : tid < 384 ? a.ln_g : a.ln_b;
    cst[tid] = srcv[tid & 127];
  }
  if (!EDGE && tid < 128) cst[tid] = a.b2[tid];
  if (EMB && tid >= 128 && tid < 256) cst[tid] = a.eb2[tid & 127];
  wait_dma_and_sync();
  const bf8* w1 = (const bf8*)smem + lane;
  const bf8* w2 = (const bf8*)(smem + NAMP_BIMG_BYTES) + lane;
  const bf8* w3 = (const bf8*)(smem + 2 * NAMP_BIMG_BYTES) + lane;
  // this lane's 16 channels of a 32-channel tile tn, as four f4 of a [128] vector: 32tn + 8q + 4hk + (0..3)
  auto vec16 = [&](const float* v, const int tn) {
    f16v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 t = *(const f4*)(v + 32 * tn + 8 * q + 4 * hk);
      o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
  };
  // (Tried: a phase shift of 2-8 k cycles between the two waves of a SIMD, so that one's MFMA-only layer 1 meets the other's GELU stretch:
  // cfg3 5.64-5.67 ms with and without, profiles/r03e.)
  for (; pair < npairs; pair += stride) {
    asm volatile("" ::: "memory");
    bf8 xb[8];
    const TileMeta me = cur;
    // Message modes: no LDS-DMA inside the step loop.  A global->LDS load touches both memories, and the compiler's wait-count pass answers
    // one in flight with a FULL drain of both counters at the next dependency of either kind (the first weight-fragment read of layer 1
    // drained the gathered-term and Pa requests: one exposed memory latency per step).  With plain loads only — every one issued by every
    // lane on every step, the stores included — it counts exactly: the next rows are awaited behind this step's stores without draining them.
    if constexpr (PA_DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // rows + Pa landed; the eight row stores may be in flight
    if constexpr (EMB) {
      // h_E = W_e . E + b_e for these 32 rows, rounded to bf16: stored for the later launches and used as this launch's operand
      f16v he[4];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) he[tn] = vec16(cst + 128, tn);
      gemm32<false, false>(he, w3, [&](const int s) { return xn[s]; });
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int tin = s >> 1, u = s & 1;
        xb[s] = pack_bf16<false>((f4){he[tin][8 * u], he[tin][8 * u + 1], he[tin][8 * u + 2], he[tin][8 * u + 3]},
                                 (f4){he[tin][8 * u + 4], he[tin][8 * u + 5], he[tin][8 * u + 6], he[tin][8 * u + 7]});
      }
      {
        bf8* dst = (bf8*)(me.valid ? a.hE16_out + me.erow * NAMP_H : g_bf16s32_dump16 + m * NAMP_H) + hk;
#pragma unroll
        for (int s = 0; s < 8; ++s) dst[2 * s] = xb[s];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) xb[s] = xn[s];
    }
    // gathered first-layer term of this row (bf16 fragment order) — consumed after GEMM 1
    bf8 pj[8];
    {
      const bf8* src = (const bf8*)((me.pj_from1 ? a.Pj116 : a.Pj016) + me.pj_row * NAMP_H) + hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) pj[s] = src[2 * s];
    }
    // the residue's own first-layer term Pa (same fragment order; one row per 16-row half, so the lanes of a half read the same 16 bytes)
    bf8 pa[8];
    f16v acc[4];
    if constexpr (PA_DMA) {
      const bf8* pl = (const bf8*)(pa_slot + half * 256) + hk;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const bf8 lo = pl[2 * (2 * tn)], hi = pl[2 * (2 * tn + 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[tn][j] = (float)lo[j]; acc[tn][8 + j] = (float)hi[j]; }
      }
    } else {
      const bf8* src = (const bf8*)(a.Pa16 + me.pa_row * NAMP_H) + hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) pa[s] = src[2 * s];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[tn][v] = 0.f;
    }
    if (!EDGE && hk == 0) w_slot[r] = me.w_row;
    const long np = pair + stride;
    cur = meta_of(pair_c(np), idx_n1);
    idx_n1 = idx_of(pair + 2 * stride);
    if constexpr (PA_DMA) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pa_fetch(cur);
    }
#ifdef NAMP_ABL_EARLY_ROW
    if constexpr (!LATE_ROW) row_fetch(cur);
#endif
    // ---- layer 1 (T): the stored row IS the operand
    gemm32<false, PIN || EPIN>(acc, w1, [&](const int s) { return xb[s]; });
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (PA_DMA) {
          acc[tn][j] += (float)pj[2 * tn][j];
          acc[tn][8 + j] += (float)pj[2 * tn + 1][j];
        } else {
          acc[tn][j] += (float)pa[2 * tn][j] + (float)pj[2 * tn][j];
          acc[tn][8 + j] += (float)pa[2 * tn + 1][j] + (float)pj[2 * tn + 1][j];
        }
      }
    }
#ifndef NAMP_ABL_EARLY_ROW
    if constexpr (!LATE_ROW) row_fetch(cur);                // behind the gathered term's wait (see the top of the step)
#endif
    f16v y[4];
    if constexpr (EDGE) {
      // ---- layers 2 and 3 (T), residual, LayerNorm 3, the row back as bf16
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) y[tn] = vec16(cst, tn);
      gemm32<false, EPIN>(y, w2, [&](const int s) { return gelu_pack8(acc[s >> 1], s & 1); });
      row_fetch(cur);                                       // the next pair's rows (LATE_ROW)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tn] = vec16(cst + 128, tn);
      gemm32<false, EPIN>(acc, w3, [&](const int s) { return gelu_pack8(y[s >> 1], s & 1); });
      // LayerNorm statistics over the lane's 64 channels as packed pairs in two independent chains each (one scalar chain of 64 dependent
      // adds / FMAs before), then one cross-lane step for the row's other half
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 sA = (f2){0.f, 0.f}, sB = (f2){0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[tn][j] += (float)xb[2 * tn][j]; acc[tn][8 + j] += (float)xb[2 * tn + 1][j]; }   // residual
#pragma unroll
        for (int v = 0; v < 16; v += 4) { sA += (f2){acc[tn][v], acc[tn][v + 1]}; sB += (f2){acc[tn][v + 2], acc[tn][v + 3]}; }
      }
      float sum = (sA.x + sA.y) + (sB.x + sB.y);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.0f / 128.0f);
      const f2 m2 = (f2){mean, mean};
      f2 qA = (f2){0.f, 0.f}, qB = (f2){0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int v = 0; v < 16; v += 4) {
          const f2 d0 = (f2){acc[tn][v], acc[tn][v + 1]} - m2, d1 = (f2){acc[tn][v + 2], acc[tn][v + 3]} - m2;
          acc[tn][v] = d0.x; acc[tn][v + 1] = d0.y; acc[tn][v + 2] = d1.x; acc[tn][v + 3] = d1.y;
          qA = __builtin_elementwise_fma(d0, d0, qA); qB = __builtin_elementwise_fma(d1, d1, qB);
        }
      float sq = (qA.x + qA.y) + (qB.x + qB.y);
      sq += __shfl_xor(sq, 32);
      const float rstd = rsqrtf(sq * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const f16v ga = vec16(cst + 256, tn), be = vec16(cst + 384, tn);
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const f2 o = __builtin_elementwise_fma((f2){acc[tn][v], acc[tn][v + 1]} * (f2){rstd, rstd}, (f2){ga[v], ga[v + 1]}, (f2){be[v], be[v + 1]});
          acc[tn][v] = o.x; acc[tn][v + 1] = o.y;
        }
      }
      {
        bf8* dst = (bf8*)(me.valid ? a.hE16_out + me.erow * NAMP_H : g_bf16s32_dump16 + m * NAMP_H) + hk;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int tin = s >> 1, u = s & 1;
          dst[2 * s] = pack_bf16<false>((f4){acc[tin][8 * u], acc[tin][8 * u + 1], acc[tin][8 * u + 2], acc[tin][8 * u + 3]},
                                        (f4){acc[tin][8 * u + 4], acc[tin][8 * u + 5], acc[tin][8 * u + 6], acc[tin][8 * u + 7]});
        }
      }
    } else {
      // ---- layer 2 (F): GELU of layer 1 step by step inside the MFMA loop
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const float b = cst[32 * tn + r];
#pragma unroll
        for (int v = 0; v < 16; ++v) y[tn][v] = b;
      }
      gemm32<true, PIN>(y, w2, [&](const int s) { return gelu_pack8(acc[s >> 1], s & 1); });
      // ---- K-sums of the layer-2 activations, per 16-row half
      // (packed pairs: .x = the first 16-row half (accumulator elements 0..7), .y = the second (8..15))
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 wp[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f4 wa = *(const f4*)(w_slot + 8 * q + 4 * hk), wb = *(const f4*)(w_slot + 8 * (2 + q) + 4 * hk);
        wp[4 * q] = (f2){wa.x, wb.x}; wp[4 * q + 1] = (f2){wa.y, wb.y}; wp[4 * q + 2] = (f2){wa.z, wb.z}; wp[4 * q + 3] = (f2){wa.w, wb.w};
      }
      float wsum = me.w_row;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) wsum += __shfl_xor(wsum, o);
      const int nodeA = __shfl(me.node, 0), ktA = __shfl(me.kt, 0), nodeB = __shfl(me.node, 16), ktB = __shfl(me.kt, 16);
      const bool okB = 2 * pair + 1 < ntiles;
      const int node_h = hk ? nodeB : nodeA, kt_h = hk ? ktB : ktA;
      float* dst = (hk == 0 || okB) ? a.partial + ((long)node_h * a.TPN + kt_h) * NAMP_H + r : g_bf16s32_dump + r;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f2 s01 = (f2){0.f, 0.f}, t01 = (f2){0.f, 0.f};            // two chains
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f4 g0 = gelu_prec<PREC_BF16>((f4){y[tn][4 * q], y[tn][4 * q + 1], y[tn][4 * q + 2], y[tn][4 * q + 3]});
          const f4 g1 = gelu_prec<PREC_BF16>((f4){y[tn][8 + 4 * q], y[tn][9 + 4 * q], y[tn][10 + 4 * q], y[tn][11 + 4 * q]});
          s01 = __builtin_elementwise_fma((f2){g0.x, g1.x}, wp[4 * q], s01);
          t01 = __builtin_elementwise_fma((f2){g0.y, g1.y}, wp[4 * q + 1], t01);
          s01 = __builtin_elementwise_fma((f2){g0.z, g1.z}, wp[4 * q + 2], s01);
          t01 = __builtin_elementwise_fma((f2){g0.w, g1.w}, wp[4 * q + 3], t01);
        }
        const float s0 = s01.x + t01.x, s1 = s01.y + t01.y;
        const float t = __shfl_xor(hk ? s0 : s1, 32);
        const float mine = (hk ? s1 : s0) + t;
        dst[32 * tn] = mine;
      }
      float* wdst = lane == 0 ? a.partial + (long)a.G * a.TPN * NAMP_H + (long)nodeA * a.TPN + ktA
                  : (lane == 16 && okB) ? a.partial + (long)a.G * a.TPN * NAMP_H + (long)nodeB * a.TPN + ktB : g_bf16s32_dump + 128 + lane;
      *wdst = wsum;
    }
  }
}
