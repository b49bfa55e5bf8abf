// Device-side building blocks for the NA-MPNN encoder/decoder path on gfx950.
//
// Everything here is written for one shape family: hidden width H = 128,
// 64-lane wavefronts, fp32 MFMA v_mfma_f32_16x16x4_f32.  A "tile" is 16 rows
// (16 edges of one residue, or 16 residues); one wavefront owns one tile and
// carries its activations through a whole MLP *in registers*:
//
//   D = mfma_16x16x4(a, b, C):  lane l supplies a = A[i=l&15][k=l>>4],
//                               b = B[k=l>>4][j=l&15];
//                               lane l receives D[i = 4*(l>>4)+r][j = l&15], r=0..3.
//
//   "T" orientation (A = weights, B = activations):  D[n_local][m] -> lane
//   (m = l&15, g = l>>4) ends up with output channels 16*tn + 4*g + r of ITS row m.
//   That is exactly the operand layout the next layer wants if the reduction
//   index is enumerated as  k = 16*tk + 4*g + r  — a permutation of k that the
//   weight image (pack_image_kernel) bakes in.  So a 3-layer MLP never leaves
//   the register file and needs no LDS round trip for activations.
//
//   "F" orientation (A = activations, B = weights): D[m][n_local] -> lane
//   (n_local = l&15, g) holds rows m = 4*g + r of channel 16*tn + n_local, which
//   is what the K-neighbour sum wants (4 in-lane adds + 2 cross-lane steps).
//
// Weight image of a [OUT x IN] block W (row-major, leading dimension ld):
//   img[tk][tn][lane][r] = W[16*tn + (lane&15)][16*tk + 4*(lane>>4) + r]
// one float4 per lane per (tk,tn): a wave reads 1 KiB contiguous, conflict-free
// from LDS (ds_read_b128) or fully coalesced from L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

#define NAMP_H 128
#define NAMP_TN 8                 // 128 / 16 output-channel tiles
#define NAMP_IMG_FLOATS 16384     // one 128x128 image
#define NAMP_IMG_BYTES 65536

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// GELU in its exact-erf form (torch.nn.GELU default; reference model_utils.py:600,633,678):  gelu(x) = x Phi(x)  with
//     Phi(-|x|) = erfc(|x| / sqrt 2) / 2 = 2^(-Q(|x|)),   Q(a) = 1 + a (c1 + a (c2 + ... + a c6)),
// Q a degree-6 weighted fit of -log2(erfc(a / sqrt 2) / 2) on [0, 5.6] with Q(0) = 1 exactly (beyond 5.6 Phi(-|x|) < 1.1e-8: clamped),
// so that  gelu(x) = x/2 + |x/2| (1 - 2 Phi(-|x|)).  All coefficients but two small ones are positive: Horner in fp32 is stable.
// Measured |gelu - fp64 gelu| <= 7.0e-7 over [-12, 12] — the fp32 rounding of the result at |x| ~ 4 — and 4e-7 relative near 0,
// the same as the Abramowitz-Stegun 7.1.26 form it replaces (7.2e-7; that one needs v_rcp_f32 besides v_exp_f32: 13 full-rate
// + 2 quarter-rate operations per value against 10 + 1 here, and GELU is more than half of the issue time of a split-bf16
// message launch).  The training kernels' gelu_val_grad (namp_train.h) keeps the A-S form, which shares exp(-x^2/2) with the derivative.
#define NAMP_GELU_LIM 5.6f
#define NAMP_GELU_C1 1.151147093e+00f
#define NAMP_GELU_C2 4.589156358e-01f
#define NAMP_GELU_C3 5.323827185e-02f
#define NAMP_GELU_C4 -7.977526064e-03f
#define NAMP_GELU_C5 7.398953830e-04f
#define NAMP_GELU_C6 -2.992676888e-05f
__device__ __forceinline__ float gelu_erf(float x) {
  const float a = fabsf(__builtin_amdgcn_fmed3f(x, -NAMP_GELU_LIM, NAMP_GELU_LIM));
  float q = fmaf(NAMP_GELU_C6, a, NAMP_GELU_C5);
  q = fmaf(q, a, NAMP_GELU_C4);
  q = fmaf(q, a, NAMP_GELU_C3);
  q = fmaf(q, a, NAMP_GELU_C2);
  q = fmaf(q, a, NAMP_GELU_C1);
  q = fmaf(q, a, 1.0f);
  const float e = __builtin_amdgcn_exp2f(-q);                   // Phi(-|x|)
  const float h = 0.5f * x;
  return fmaf(fabsf(h), fmaf(-2.0f, e, 1.0f), h);               // h + |h| (1 - 2 Phi(-|x|))
}

// Ablation switches for tools/kbench.py (never defined in the shipped build).
//   NAMP_ABL_NOGELU   : GELU -> identity          NAMP_ABL_LAYERS=n : stop the edge MLP after layer n
//   NAMP_ABL_NOPROLOG : edge kernel reads no per-row operands (constants instead)
//   NAMP_ABL_NOGEMM   : 128x128 tile GEMMs -> acc += x      NAMP_ABL_X1 : split-bf16 GEMMs keep only the hi.hi product
//   NAMP_ABL_NOSTORE  : fused edge update keeps its rows in registers only   NAMP_ABL_NOLN : LayerNorms -> identity
//   NAMP_ABL_NOTABLE2 : fused launches skip the second (message) table gather
//   NAMP_ABL_NOLDSW   : bf16 chain GEMMs take their weight fragment from registers (no ds_read)
//   NAMP_ABL_NODMA    : no weight staging (LDS holds garbage)  NAMP_ABL_NOTAIL : fused residue tail -> plain store
__device__ __forceinline__ f4 gelu4(f4 v) {
#ifdef NAMP_ABL_NOGELU
  return v;
#endif
  // gelu_erf() on four values, written on vectors so that the Horner steps can be selected as packed fp32 (v_pk_fma_f32);
  // the transcendental stays per element.  Same operation order per element as gelu_erf(): bit-identical results.
  const f4 a = __builtin_elementwise_abs((f4){__builtin_amdgcn_fmed3f(v.x, -NAMP_GELU_LIM, NAMP_GELU_LIM),
                                              __builtin_amdgcn_fmed3f(v.y, -NAMP_GELU_LIM, NAMP_GELU_LIM),
                                              __builtin_amdgcn_fmed3f(v.z, -NAMP_GELU_LIM, NAMP_GELU_LIM),
                                              __builtin_amdgcn_fmed3f(v.w, -NAMP_GELU_LIM, NAMP_GELU_LIM)});
  f4 q = a * NAMP_GELU_C6 + NAMP_GELU_C5;
  q = q * a + NAMP_GELU_C4;
  q = q * a + NAMP_GELU_C3;
  q = q * a + NAMP_GELU_C2;
  q = q * a + NAMP_GELU_C1;
  q = q * a + 1.0f;
  f4 e;
  e.x = __builtin_amdgcn_exp2f(-q.x); e.y = __builtin_amdgcn_exp2f(-q.y); e.z = __builtin_amdgcn_exp2f(-q.z); e.w = __builtin_amdgcn_exp2f(-q.w);
  const f4 h = v * 0.5f;
  return __builtin_elementwise_abs(h) * (e * -2.0f + 1.0f) + h;
}

// acc[tn] (+)= W . x  over TK k-tiles.  `w` points at img[tk0][0][lane]; consecutive
// tn are 64 f4 apart, consecutive tk are 64*TNW f4 apart (TNW = tn extent of the image).
//   FLIP=false: "T" orientation, FLIP=true: "F" orientation (see header comment).
//   ACT=true: x holds PRE-activations; GELU of k-tile tk is evaluated right before its MFMAs, so the VALU
//   work of tile tk+1 issues in the shadow of tile tk's 4*NTN MFMAs instead of in a separate phase.
template <int TK, int NTN, bool FLIP, bool ACT = false>
__device__ __forceinline__ void chain_gemm(f4 (&acc)[NTN], const f4 (&x)[TK], const f4* w, const int tn_stride_img) {
#pragma unroll
  for (int tk = 0; tk < TK; ++tk) {
    f4 wf[NTN];
#pragma unroll
    for (int tn = 0; tn < NTN; ++tn) wf[tn] = w[(tk * tn_stride_img + tn) * 64];
    const f4 xk = ACT ? gelu4(x[tk]) : x[tk];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) {
        if (FLIP) acc[tn] = mfma4(xk[r], wf[tn][r], acc[tn]);
        else      acc[tn] = mfma4(wf[tn][r], xk[r], acc[tn]);
      }
    }
  }
}

// ---- bf16 throughput mode (BASELINE configs[2]: "bf16 MFMA message GEMM") -------------------------------
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][n=l&15], j=0..7, and gets
// D[4(l>>4)+r][l&15] like the fp32 form.  The register chain carries over with the reduction index of step s
// enumerated as  k(s,g,j) = 32s + 16(j>>2) + 4g + (j&3): the eight values a lane feeds into step s are exactly
// its fp32 accumulators of channel tiles 2s and 2s+1, converted (round-to-nearest-even) on the fly.  Inputs and
// weights are rounded to bf16, accumulation is fp32 — about 1e-2 on log-probs, NOT the parity mode.
// Weight image: img[s][tn][lane][j] = bf16(W[16tn + (lane&15)][k(s, lane>>4, j)]), 32 KiB per 128x128 block, so
// all three layers sit in LDS at once and the edge kernel needs a single barrier.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define NAMP_BIMG_BYTES 32768

// GELU of the bf16 throughput mode: its output is rounded to bf16 (ulp 2^-8 relative) right away, so the exact-erf form
// with its two quarter-rate transcendentals per value is wasted there.  x * Phi(x) with Phi(x) - 1/2 = x * Q(x^2), Q a
// polynomial (round 4: degree 4, max |error| 1.3e-3 over the real line; rounds 1-3: degree 6, 1.9e-4), written on 4-vectors so that the
// packed-fp32 forms (v_pk_fma_f32 / v_pk_mul_f32) can be selected.
// Phi(x) - 1/2 = x Q(x^2), Q of degree 4 (round 4): Q(t) = Q0 + Q1 t + Q2 t^2 + Q3 t^3 + Q4 t^4
#define NAMP_GELU4_Q4 1.2247244342e-05f
#define NAMP_GELU4_Q3 -4.6204919395e-04f
#define NAMP_GELU4_Q2 7.1675247900e-03f
#define NAMP_GELU4_Q1 -6.1600986289e-02f
#define NAMP_GELU4_Q0 3.9660173626e-01f
__device__ __forceinline__ f4 gelu4_bf16mode(const f4 x) {
#ifdef NAMP_ABL_NOGELU
  return x;
#endif
#ifdef NAMP_ABL_GELU16_SCALAR
  f4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c = __builtin_amdgcn_fmed3f(x[i], -4.f, 4.f), t = c * c;
    float q = fmaf(2.2787273029e-08f, t, -1.5988982626e-06f);
    q = fmaf(q, t, 4.7961328822e-05f); q = fmaf(q, t, -8.1407082443e-04f); q = fmaf(q, t, 8.7726502299e-03f);
    q = fmaf(q, t, -6.4573666617e-02f); q = fmaf(q, t, 3.9788372746e-01f);
    o[i] = x[i] * fmaf(c, q, 0.5f);
  }
  return o;
#endif
  // Phi(x) = clamp01(1/2 + x Q(x^2)) with NO clamp of the argument: beyond the fit range the polynomial keeps x Q(x^2) >= 1/2 in magnitude
  // (checked on 1.2 M points up to |x| = 12; the leading coefficient is positive, so Q grows — to +inf on overflow — from there), and the
  // [0, 1] clamp is the output modifier of the last v_pk_fma_f32: one v_med3_f32 per value less than clamping x first (round 3).
  const f4 t = x * x;
#ifndef NAMP_GELU16_DEG
#define NAMP_GELU16_DEG 4
#endif
#if NAMP_GELU16_DEG == 6
  // rounds 1-3: degree 6 in x^2, max |error| 1.9e-4
  f4 q = (f4){2.2787273029e-08f, 2.2787273029e-08f, 2.2787273029e-08f, 2.2787273029e-08f};
  q = q * t + -1.5988982626e-06f;
  q = q * t + 4.7961328822e-05f;
  q = q * t + -8.1407082443e-04f;
  q = q * t + 8.7726502299e-03f;
  q = q * t + -6.4573666617e-02f;
  q = q * t + 3.9788372746e-01f;
#elif NAMP_GELU16_DEG == 2
  // degree 2 in x^2: max |error| 7.7e-3 (two bf16 ulps at |y| ~ 1) — measured only, not shipped
  f4 q = (f4){2.2650967672e-03f, 2.2650967672e-03f, 2.2650967672e-03f, 2.2650967672e-03f};
  q = q * t + -4.2712306742e-02f;
  q = q * t + 3.7477252573e-01f;
#else
  // round 4: degree 4 in x^2 (minimax over the real line, the [0, 1] clamp included; positive leading coefficient, x Q(x^2) >= 0.62 beyond
  // |x| = 4): max |error| 1.3e-3 — a third of the bf16 rounding step of the result at |y| >= 1 (2^-8 |y|) and below it for |y| > 0.33;
  // near 0 the error is x^2 (Q - Q*), i.e. vanishes.  Two packed FMAs per value pair less than the degree-6 form.  The constants are the
  // NAMP_GELU4_Q* macros above (shared with the training backward's dw_gelu_split4_bf16, namp_train_dw.h); tests/test_host_logic.py
  // parses them: test_device_gelu_bf16_mode_polynomial.
  f4 q = (f4){NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4, NAMP_GELU4_Q4};
  q = q * t + NAMP_GELU4_Q3;
  q = q * t + NAMP_GELU4_Q2;
  q = q * t + NAMP_GELU4_Q1;
  q = q * t + NAMP_GELU4_Q0;
#endif
  // (written as instructions: the compiler emits a separate `v_max_f32 ... clamp` per value instead of folding the clamp into the packed FMA)
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 half = (f2){0.5f, 0.5f};
  f2 p0, p1;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(p0) : "v"((f2){x.x, x.y}), "v"((f2){q.x, q.y}), "v"(half));
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(p1) : "v"((f2){x.z, x.w}), "v"((f2){q.z, q.w}), "v"(half));
  return x * (f4){p0.x, p0.y, p1.x, p1.y};
}

template <bool ACT>
__device__ __forceinline__ bf8 pack_bf16(const f4 lo, const f4 hi) {
  const f4 a = ACT ? gelu4_bf16mode(lo) : lo, b = ACT ? gelu4_bf16mode(hi) : hi;
  bf8 o;
  o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
  o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
  return o;
}

template <bool FLIP, bool ACT>
__device__ __forceinline__ void chain_gemm_bf16(f4 (&acc)[8], const f4 (&x)[8], const bf8* w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bf8 xb = pack_bf16<ACT>(x[2 * s], x[2 * s + 1]);
#ifdef NAMP_ABL_NOGEMM
    acc[s].x += (float)xb[0] + (float)xb[5]; acc[s + 4].y += (float)xb[2] + (float)xb[7];
    acc[s].z += (float)xb[1] + (float)xb[4]; acc[s + 4].w += (float)xb[3] + (float)xb[6];
    continue;
#endif
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) {
#ifdef NAMP_ABL_NOLDSW
      bf8 wf = xb; wf[0] = (__bf16)(float)(s * 8 + tn);
#else
      const bf8 wf = w[(s * 8 + tn) * 64];
#endif
      if (FLIP) acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, wf, acc[tn], 0, 0, 0);
      else      acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb, acc[tn], 0, 0, 0);
    }
  }
}

// ---- split-bf16 ("bf16x3") evaluation of an fp32 GEMM -------------------------------------------------------
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950, so an fp32-accurate product is cheaper as THREE bf16 products of
// split operands:  x = x_hi + x_mid + O(2^-16 x)  with x_hi = bf16(x), x_mid = bf16(x - x_hi)  (likewise W), and
//      W . x  ~=  W_hi.x_hi + W_hi.x_mid + W_mid.x_hi          (dropped terms are O(2^-16) relative, fp32 accumulation).
// 96 bf16 MFMAs (1,536 pipe cycles) replace 256 fp32 MFMAs (8,192) per 16x128x128 tile GEMM.  End to end on the N=1000
// golden the log-probs move by 3.1e-5 (bar: 1e-3) and every arg-max is unchanged — same order as the 1e-5 the exact-fp32
// build differs from the CPU reference by.  The "x3 image" of a block is its bf16 fragment image of W_hi followed by the
// one of W_mid: 2 x 32 KiB = the size of the fp32 image, so the LDS ring and its DMA schedule are unchanged.
// The register chain carries over as in the bf16 mode (k(s,g,j) = 32s + 16(j>>2) + 4g + (j&3)).
// GELU beside bf16 MFMAs: the per-element form.  VALU and MFMA time add up on this chip whatever the interleaving
// (tools/coexec_probe.hip), and packed fp32 operations beside MFMAs cost more than they save (measured -1.5 % per launch
// for the scalar form here, +2 % for it in the fp32-MFMA kernels, which keep gelu4()).
__device__ __forceinline__ f4 gelu4_scalar(const f4 v) {
#ifdef NAMP_ABL_NOGELU
  return v;
#endif
  return (f4){gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
}

// x = hi + mid (+ O(2^-16 x)) for the eight values a lane feeds into one K = 32 step (channel tiles 2s and 2s+1 of its row)
__device__ __forceinline__ void split_x3(const f4 a, const f4 b, bf8& hi, bf8& mid) {
  hi[0] = (__bf16)a.x; hi[1] = (__bf16)a.y; hi[2] = (__bf16)a.z; hi[3] = (__bf16)a.w;
  hi[4] = (__bf16)b.x; hi[5] = (__bf16)b.y; hi[6] = (__bf16)b.z; hi[7] = (__bf16)b.w;
  mid[0] = (__bf16)(a.x - (float)hi[0]); mid[1] = (__bf16)(a.y - (float)hi[1]);
  mid[2] = (__bf16)(a.z - (float)hi[2]); mid[3] = (__bf16)(a.w - (float)hi[3]);
  mid[4] = (__bf16)(b.x - (float)hi[4]); mid[5] = (__bf16)(b.y - (float)hi[5]);
  mid[6] = (__bf16)(b.z - (float)hi[6]); mid[7] = (__bf16)(b.w - (float)hi[7]);
}
// acc += W . x for one fragment pair (hi, mid images of the same tile): the three split products
__device__ __forceinline__ f4 mfma_x3(const bf8 wh, const bf8 wm, const bf8 hi, const bf8 mid, f4 acc) {
#ifdef NAMP_ABL_X1
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hi, acc, 0, 0, 0);
#endif
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hi, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hi, acc, 0, 0, 0);
}

// X1 = true: only the hi . hi product (plain bf16 operands, fp32 accumulation) out of the same x3 image — the residue-level GEMMs of the
// bf16 throughput mode (the reference's own AMP autocasts the whole model, na_run.py:216-218; the mid plane of the image is never read)
template <bool X1>
__device__ __forceinline__ f4 mfma_xs(const bf8 wh, const bf8 wm, const bf8 hi, const bf8 mid, f4 acc) {
  if constexpr (X1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hi, acc, 0, 0, 0);
  else return mfma_x3(wh, wm, hi, mid, acc);
}

// chain_gemm_x3 with the fragments requested ahead: the hi fragments of the NEXT group of four channel tiles and this group's mid fragments are
// requested before this group's twelve MFMAs issue, pinned there by a scheduling barrier that only VALU instructions may cross.  The compiler's
// own schedule of chain_gemm_x3 waits for an LDS round trip every one or two MFMAs; the throughput kernels cover that with three waves per SIMD
// (and spill with this form: see below), the sampler — one or two active waves per SIMD, a latency chain — cannot.
template <bool FLIP, bool ACT>
__device__ __forceinline__ void chain_gemm_x3_ahead(f4 (&acc)[8], const f4 (&x)[8], const bf8* w) {
  bf8 wh[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wh[0][q] = w[q * 64];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f4 a = ACT ? gelu4_scalar(x[2 * s]) : x[2 * s], b = ACT ? gelu4_scalar(x[2 * s + 1]) : x[2 * s + 1];
    bf8 hi, mid;
    split_x3(a, b, hi, mid);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = 2 * s + h;
      bf8 wm[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wm[q] = w[(NAMP_BIMG_BYTES / 16) + (s * 8 + 4 * h + q) * 64];
      if (g + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wh[(g + 1) & 1][q] = w[(((g + 1) >> 1) * 8 + 4 * ((g + 1) & 1) + q) * 64];
      }
      __builtin_amdgcn_sched_barrier(2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid, wh[g & 1][q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g & 1][q], mid, acc[4 * h + q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wm[q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[q], hi, acc[4 * h + q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wh[g & 1][q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g & 1][q], hi, acc[4 * h + q], 0, 0, 0);
      }
    }
  }
}

// (Tried, round 3: the hi fragments of the next group requested before this group's twelve MFMAs, pinned by a scheduling barrier as in
// namp_bf16s32.h's gemm32 — the 168-VGPR kernels spill: cfg3 split-bf16 13.2 -> 14.3-17.4 ms per step, cfg2 unchanged; the persistent kernel
// with 8 waves per workgroup (208-241 VGPRs, no spills) and that prefetch: 12.3 -> 12.6 ms, 13.2 without the prefetch — the third wave per SIMD
// is worth more than the fragments in flight.)
template <bool FLIP, bool ACT>
__device__ __forceinline__ void chain_gemm_x3(f4 (&acc)[8], const f4 (&x)[8], const bf8* w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f4 a = ACT ? gelu4_scalar(x[2 * s]) : x[2 * s], b = ACT ? gelu4_scalar(x[2 * s + 1]) : x[2 * s + 1];
    bf8 hi, mid;
    hi[0] = (__bf16)a.x; hi[1] = (__bf16)a.y; hi[2] = (__bf16)a.z; hi[3] = (__bf16)a.w;
    hi[4] = (__bf16)b.x; hi[5] = (__bf16)b.y; hi[6] = (__bf16)b.z; hi[7] = (__bf16)b.w;
    mid[0] = (__bf16)(a.x - (float)hi[0]); mid[1] = (__bf16)(a.y - (float)hi[1]);
    mid[2] = (__bf16)(a.z - (float)hi[2]); mid[3] = (__bf16)(a.w - (float)hi[3]);
    mid[4] = (__bf16)(b.x - (float)hi[4]); mid[5] = (__bf16)(b.y - (float)hi[5]);
    mid[6] = (__bf16)(b.z - (float)hi[6]); mid[7] = (__bf16)(b.w - (float)hi[7]);
    // product-major over groups of four channel tiles: the three products of one tile accumulate into the same registers,
    // and a dependent MFMA issued right behind its producer waits out the whole pipeline (~2x the issue time); with four
    // independent tiles between dependent issues the matrix pipe stays full
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf8 wh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wh[q] = w[(s * 8 + 4 * h + q) * 64];
#ifdef NAMP_ABL_X1
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wh[q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], hi, acc[4 * h + q], 0, 0, 0);
      }
      continue;
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid, wh[q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], mid, acc[4 * h + q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf8 wm = w[(NAMP_BIMG_BYTES / 16) + (s * 8 + 4 * h + q) * 64];
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wm, acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hi, acc[4 * h + q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (FLIP) acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, wh[q], acc[4 * h + q], 0, 0, 0);
        else      acc[4 * h + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[q], hi, acc[4 * h + q], 0, 0, 0);
      }
    }
  }
}

// The split-bf16 contraction with the x3 image streamed straight from global memory (L2 resident): the 16 fragments of step
// s+1 (hi and mid of 8 channel tiles) are requested before the MFMAs of step s issue — chain_gemm_global's schedule.
template <bool X1 = false>
__device__ __forceinline__ void chain_gemm_global_x3(f4 (&acc)[8], const f4 (&x)[8], const bf8* __restrict__ w) {
  if constexpr (X1) {
    // hi plane only: 8 fragments per step, the next step's requested before this step's MFMAs
    bf8 ch[8], nh[8];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) ch[tn] = w[tn * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) {
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) nh[tn] = w[((s + 1) * 8 + tn) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf8 hi = pack_bf16<false>(x[2 * s], x[2 * s + 1]);
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch[tn], hi, acc[tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < 4) {
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) ch[tn] = nh[tn];
      }
    }
    return;
  }
  bf8 ch[8], cm[8], nh[8], nm[8];
#pragma unroll
  for (int tn = 0; tn < 8; ++tn) { ch[tn] = w[tn * 64]; cm[tn] = w[NAMP_BIMG_BYTES / 16 + tn * 64]; }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s + 1 < 4) {
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) { nh[tn] = w[((s + 1) * 8 + tn) * 64]; nm[tn] = w[NAMP_BIMG_BYTES / 16 + ((s + 1) * 8 + tn) * 64]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    bf8 hi, mid;
    split_x3(x[2 * s], x[2 * s + 1], hi, mid);
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) acc[tn] = mfma_x3(ch[tn], cm[tn], hi, mid, acc[tn]);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < 4) {
#pragma unroll
      for (int tn = 0; tn < 8; ++tn) { ch[tn] = nh[tn]; cm[tn] = nm[tn]; }
    }
  }
}

// GELU of layer-2 pre-activations in front of the K-sum, in the form the precision's chain GEMM applies to its operand
// (PREC: 0 exact fp32, 1 bf16 throughput, 2 split-bf16 — the PREC_* codes of namp_kernels.h)
template <int PREC>
__device__ __forceinline__ f4 gelu_prec(const f4 v) {
  if constexpr (PREC == 1) return gelu4_bf16mode(v);
  else if constexpr (PREC == 2) return gelu4_scalar(v);
  else return gelu4(v);
}

// one 128 x 128 tile GEMM of the edge kernels out of a 64 KiB LDS slot: exact fp32 MFMA or the split-bf16 form
template <bool X3, bool FLIP, bool ACT>
__device__ __forceinline__ void gemm128(f4 (&acc)[8], const f4 (&x)[8], const f4* w) {
#ifdef NAMP_ABL_NOGEMM
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] += ACT ? gelu4(x[t]) : x[t];
  return;
#endif
  if constexpr (X3) chain_gemm_x3<FLIP, ACT>(acc, x, (const bf8*)w);
  else chain_gemm<8, 8, FLIP, ACT>(acc, x, w, 8);
}

// One 16-channel output tile (tn) of a 128 x 128 product out of an LDS slot (T orientation: lane (m, g) gets channels
// 16tn + 4g + r of row m): the per-residue layer 3 behind the K-sum in the fused edge kernels.  w = slot base + lane.
template <bool X3>
__device__ __forceinline__ f4 tile_gemm1(f4 o, const f4 (&x)[8], const f4* w, const int tn) {
  if constexpr (X3) {
    const bf8* wb = (const bf8*)w;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf8 hi, mid;
      split_x3(x[2 * s], x[2 * s + 1], hi, mid);
      o = mfma_x3(wb[(s * 8 + tn) * 64], wb[NAMP_BIMG_BYTES / 16 + (s * 8 + tn) * 64], hi, mid, o);
    }
    return o;
  } else {
    f4 oo[1] = {o};
    chain_gemm<8, 1, false, false>(oo, x, w + tn * 64, 8);
    return oo[0];
  }
}

// The training kernels' tile GEMM by precision code (the `x3` argument of the namp_train_* entry points): 0 exact fp32 MFMA,
// 1 split-bf16 products, 2 plain bf16 products (mixed-precision training: na_run.py trains under autocast); ACT is never
// used there (activations come from gelu_val_grad).  The bf16 image is the first 32 KiB of the 64 KiB slot.
template <int PREC, bool FLIP>
__device__ __forceinline__ void gemm128p(f4 (&acc)[8], const f4 (&x)[8], const f4* w) {
  if constexpr (PREC == 2) chain_gemm_bf16<FLIP, false>(acc, x, (const bf8*)w);
  else gemm128<PREC == 1, FLIP, false>(acc, x, w);
}

// Same contraction with the weight image streamed straight from global memory (L2 / L1 resident):
// fragments of step tk+1 are requested before the MFMAs of step tk issue, so one L2 round trip is
// always covered by 4*NTN MFMAs.  Used where each fragment is consumed once per wave or where the
// LDS copy of the image is still in flight (first layer of edge_mlp_kernel).
template <int TK, int NTN, bool FLIP>
__device__ __forceinline__ void chain_gemm_global(f4 (&acc)[NTN], const f4 (&x)[TK], const f4* __restrict__ w,
                                                  const int tn_stride_img) {
  f4 cur[NTN], nxt[NTN];
#pragma unroll
  for (int tn = 0; tn < NTN; ++tn) cur[tn] = w[tn * 64];
#pragma unroll
  for (int tk = 0; tk < TK; ++tk) {
    if (tk + 1 < TK) {
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) nxt[tn] = w[((tk + 1) * tn_stride_img + tn) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of this step's MFMAs
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) {
        if (FLIP) acc[tn] = mfma4(x[tk][r], cur[tn][r], acc[tn]);
        else      acc[tn] = mfma4(cur[tn][r], x[tk][r], acc[tn]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tk + 1 < TK) {
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) cur[tn] = nxt[tn];
    }
  }
}

// Asynchronous global -> LDS copy of `nchunks` KiB-sized chunks (one wave-instruction each,
// 64 lanes x 16 B).  The LDS image is the global image verbatim (lane-linear), which is all
// global_load_lds can do.  Completion: s_waitcnt vmcnt(0) in every issuing wave + a barrier.
__device__ __forceinline__ void dma_to_lds(char* lds_dst, const float* gsrc, int nchunks, int wave, int nwaves, int lane) {
#ifdef NAMP_ABL_NODMA
  return;
#endif
  for (int c = wave; c < nchunks; c += nwaves) {
    const char* g = (const char*)gsrc + (size_t)c * 1024 + lane * 16;
    char* d = lds_dst + c * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)d, 16, 0, 0);
  }
}

// The same copy through registers (global_load_dwordx4 -> ds_write_b128), NPER chunks per wave at a time.  LDS-DMA lands at
// ~20-25 GB/s per CU whatever is in flight (measured in the sampler, profiles/r03c; the guide's ldsdma-fill row) — fine when 250
// workgroups stream the same images and the chip-wide rate is what counts, but a one-workgroup launch (the sampler: one dependency
// level = one or two workgroups) waited 3.2 us per 64 KiB image; plain loads pull 150-230 GB/s into one CU (tools/clock_probe.hip).
// Synchronous: returns with the wave's chunks written; the caller still needs a barrier before other waves read them.
template <int NPER>
__device__ __forceinline__ void copy_to_lds(char* lds_dst, const float* gsrc, int nchunks, int wave, int nwaves, int lane) {
#ifdef NAMP_ABL_NODMA
  return;
#endif
  for (int c0 = wave * NPER; c0 < nchunks; c0 += nwaves * NPER) {
    f4 v[NPER];
#pragma unroll
    for (int u = 0; u < NPER; ++u)
      if (c0 + u < nchunks) v[u] = *(const f4*)((const char*)gsrc + (size_t)(c0 + u) * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < NPER; ++u)
      if (c0 + u < nchunks) *(f4*)(lds_dst + (c0 + u) * 1024 + lane * 16) = v[u];
  }
}

__device__ __forceinline__ void wait_dma_and_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Dropout mask of the training path (EncLayer.dropout3 on the per-edge message, na_model_utils.py:239): a counter-based
// hash of (seed, edge row, channel), so the forward and the backward launch regenerate the same mask without storing it.
// keep iff hash >= thresh (thresh = p * 2^32); kept values are scaled by 1 / (1 - p).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_row_key(uint32_t seed, long erow) {
  return mix32(seed ^ (uint32_t)erow ^ mix32((uint32_t)((unsigned long)erow >> 32) + 0x9E3779B9U));
}
__device__ __forceinline__ float drop_factor(uint32_t row_key, int channel, uint32_t thresh, float scale) {
  return mix32(row_key + (uint32_t)channel * 0x9E3779B9U) >= thresh ? scale : 0.f;
}

// Workgroup b of a grid of n (dispatched to XCD b % 8) -> position of b in the order "all of XCD 0's workgroups, then XCD 1's, ...": a bijection
// on [0, n) for any n.  Kernels that index their data with it give every XCD one contiguous range.
__device__ __forceinline__ int xcd_block_index(const int b, const int n) {
#ifdef NAMP_ABL_NOXCD
  return b;
#endif
  const int x = b & 7, slot = b >> 3, q = n >> 3, r = n & 7;
  return x * q + (x < r ? x : r) + slot;
}

// sum over the 4 lane groups g (lanes l, l^16, l^32, l^48 hold the same column)
__device__ __forceinline__ float xg_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// LayerNorm over the 128 channels of one row held as v[8] (f4 each) by the 4 lanes
// (m, g=0..3); gamma/beta are read at channel 16*tn + 4*g.  eps = 1e-5, biased variance
// (torch.nn.LayerNorm; reference model_utils.py:627-628,668-670).
__device__ __forceinline__ void layernorm_row_T(f4 (&v)[8], const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int g) {
#ifdef NAMP_ABL_NOLN
  return;
#endif
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
  const float mean = xg_sum(s) * (1.0f / 128.0f);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    v[t] -= mean;
    q += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
  }
  const float rstd = rsqrtf(xg_sum(q) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f4 ga = *(const f4*)(gamma + 16 * t + 4 * g);
    const f4 be = *(const f4*)(beta + 16 * t + 4 * g);
    v[t] = v[t] * rstd * ga + be;
  }
}
