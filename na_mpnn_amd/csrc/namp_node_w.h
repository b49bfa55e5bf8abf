// node_update_w_kernel — round 6: the residue update of the bf16 throughput mode on large batches (EncLayer / DecLayer behind the message
// K-sums: hoisted layer 3, residual, LayerNorm 1, FFN, residual, LayerNorm 2, mask, next tables — model_utils.py:619-657,659-704) as
// weight-stationary-per-step workgroups.
//
// Why: node_update_multi_kernel<2, 2> took 125 us per launch at 64,000 residues, six launches = 15 % of the cfg3 step, for 25 GFLOP and
// ~200 MB (10 us of MFMA, 40 us of HBM).  It runs 2,000 workgroups of 32 residues, one per CU at a time (135 KB of LDS), and a workgroup is a
// chain of ~10 phases, each opening with its waves' weight fragments requested from L2 (416 KB per workgroup) and closing with a barrier:
// ~16 us per workgroup whatever it computes, eight rounds of that per launch.
//
// Here a workgroup of 4 waves owns 64 residues (two workgroups per CU, out of step with each other: one's memory and GELU phases beside
// the other's products), one 16-row tile per wave END TO END (no cross-wave reduction, no activations in LDS: a
// tile's rows stay in its wave's registers through all 13 products, the result layout of one product being the operand layout of the next as
// in the edge kernels), and the 13 weight blocks of a layer (W3 | W_in, W_out x 4 hidden blocks | up to 4 table projections; 32 KiB each as
// bf16 fragment images = the hi planes of the split-bf16 images the parity mode uses) travel through a 2-slot LDS ring: block i + 2 is
// written into block i's slot behind the barrier that ends block i's product, block i + 3 requested into registers right after — one barrier
// per block.  Weight traffic out of L2 per residue: 6.5 KB (13 KB before).
// Arithmetic: that of node_update_multi_kernel<2, 2> (operands rounded to bf16, fp32 accumulation, exact-erf GELU, fp32 LayerNorms and
// residuals); the FFN's second product accumulates its four hidden blocks in one chain instead of eight per-wave partial sums.
// X3 = true: the split-bf16 (fp32-equivalent) products of the parity mode on large batches (node_update_multi_kernel<2, 1>): every matrix
// block travels as two ring entries, its hi plane (products W_hi . x_mid, W_hi . x_hi) and its mid plane (W_mid . x_hi).
#pragma once
#include "namp_kernels.h"

#define NODEW_SLOT 32768
#define NODEW_LDS (2 * NODEW_SLOT)
#define NODEW_THREADS 256
#define NODEW_ROWS 64

// acc[tn] += W[16tn .., :] . x  for one 128 x 128 block image in LDS (w = block + lane), x as the four K-step operands of the tile
__device__ __forceinline__ void gemm16(f4 (&acc)[8], const bf8 (&xb)[4], const bf8* w) {
#ifdef NW_NOGEMM
  acc[0].x += (float)xb[0][0] + (float)w[0][0];
  return;
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf8 wf[8];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) wf[tn] = w[(s * 8 + tn) * 64];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], xb[s], acc[tn], 0, 0, 0);
  }
}
// ... the hi-plane pass of a split-bf16 product: W_hi . x_mid, then W_hi . x_hi, out of one read of each fragment
__device__ __forceinline__ void gemm16_hi2(f4 (&acc)[8], const bf8 (&xh)[4], const bf8 (&xm)[4], const bf8* w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf8 wf[8];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) wf[tn] = w[(s * 8 + tn) * 64];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], xm[s], acc[tn], 0, 0, 0);
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], xh[s], acc[tn], 0, 0, 0);
  }
}
__device__ __forceinline__ void pack_rows(bf8 (&xb)[4], const f4 (&x)[8]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) xb[s] = pack_bf16<false>(x[2 * s], x[2 * s + 1]);
}
template <bool X3>
__device__ __forceinline__ void split_rows(bf8 (&xh)[4], bf8 (&xm)[4], const f4 (&x)[8]) {
  if constexpr (X3) {
#pragma unroll
    for (int s = 0; s < 4; ++s) split_x3(x[2 * s], x[2 * s + 1], xh[s], xm[s]);
  } else pack_rows(xh, x);
}

template <bool X3>
static __global__ __launch_bounds__(NODEW_THREADS, 2) void node_update_w_kernel(const NodeUpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15;
  int g = lane >> 4;
  const NodeTail& t = a.t;
  constexpr int PL = X3 ? 2 : 1;                              // ring entries per matrix block (hi plane, mid plane)
  const int nblk = PL * (9 + t.nproj);
  // ring entry i = plane i % PL of matrix block i / PL: four 8-KiB chunks `stride` bytes apart (W_in's hidden block h is 8 of its 32 channel
  // tiles in each of its four K-steps); the mid plane of an image follows its hi plane (128 KiB for W_in / W_out, 32 KiB for the square ones);
  // entries past the layer's last re-read that one (never used: the requests of the step loop stay free of control flow)
  auto block_base = [&](const int i_) -> const char* {
    const int ic = i_ < nblk ? i_ : nblk - 1;
    const int i = ic / PL, pl = ic - i * PL;
    if (i == 0) return (const char*)t.m3_img + pl * NAMP_BIMG_BYTES;
    if (i <= 8) {
      const int h = (i - 1) >> 1;
      return ((i - 1) & 1) ? (const char*)t.Wout_img + pl * 131072 + h * 32768 : (const char*)t.Win_img + pl * 131072 + h * 8192;
    }
    return (const char*)t.p[i - 9].img + pl * NAMP_BIMG_BYTES;
  };
  auto block_stride = [&](const int i_) -> long { const int i = i_ / PL; return (i >= 1 && i <= 8 && !((i - 1) & 1)) ? 32768 : 8192; };
  // acc += W_k . x for matrix block k (ring entries PL k ..): the hi pass (and the mid pass), each closed by the ring's advance
  auto product = [&](f4 (&acc)[8], const bf8 (&xh)[4], const bf8 (&xm)[4], const int k, auto&& adv, auto&& slot_) {
    if constexpr (X3) {
      gemm16_hi2(acc, xh, xm, slot_(2 * k));
      adv(2 * k);
      gemm16(acc, xh, slot_(2 * k + 1));
      adv(2 * k + 1);
    } else {
      gemm16(acc, xh, slot_(k));
      adv(k);
    }
  };
  f4 sreg[8];
  auto stage_load = [&](const int i) {
#ifdef NW_NOSTAGE
    return;
#endif
    const char* b = block_base(i) + tid * 16;
    const long st = block_stride(i);
#pragma unroll
    for (int k = 0; k < 4; ++k) { sreg[2 * k] = *(const f4*)(b + k * st); sreg[2 * k + 1] = *(const f4*)(b + k * st + 4096); }
  };
  auto stage_store = [&](const int i) {
#ifdef NW_NOSTAGE
    return;
#endif
    char* d = smem + (i & 1) * NODEW_SLOT + tid * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) { *(f4*)(d + 8192 * k) = sreg[2 * k]; *(f4*)(d + 8192 * k + 4096) = sreg[2 * k + 1]; }
  };
  auto slot = [&](const int i) { return (const bf8*)(smem + (i & 1) * NODEW_SLOT) + lane; };
  // end of block i's product: every wave is done with its slot -> block i + 2 goes in, block i + 3 is requested
  auto advance = [&](const int i) {
    __syncthreads();
    stage_store(i + 2);
    stage_load(i + 3);
  };
  const int ntile = (a.G + 15) >> 4;
  for (int tile0 = blockIdx.x * 4; tile0 < ntile; tile0 += gridDim.x * 4) {
    // (per-lane addresses are re-derived in every pass: hoisted out of this loop they would sit in ~40 registers for the whole kernel)
    asm volatile("" : "+v"(tid), "+v"(g));
    const int tile = tile0 + wave;
    const int row = 16 * tile + m;
    const bool valid = row < a.G;
    const int rr = valid ? row : (a.G - 1);
    __syncthreads();                                             // (a later pass: every wave is done with the ring)
    // ---- every request of the pass's opening in one round trip: the first two weight blocks, the K-sums of the layer-2 activations with
    // their weight sums (three row tiles per round — K = 48; a tile past the residue's last re-reads that one with weight 0), the residue's row
    f4 sreg2[8];
    stage_load(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) sreg2[k] = sreg[k];
    stage_load(1);
    f4 x[8], hv[8];
    float ws = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = (f4){0.f, 0.f, 0.f, 0.f};
    {
      const float* src = t.hV + (long)rr * NAMP_H + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) hv[c] = *(const f4*)(src + 16 * c);
    }
#ifndef NW_NOIN
    for (int p0 = 0; p0 < a.TPN; p0 += 3) {
      f4 v[3][8];
      float wv[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int p = p0 + q < a.TPN ? p0 + q : a.TPN - 1;
        const float* ps_ = a.partial + ((long)rr * a.TPN + p) * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[q][c] = *(const f4*)(ps_ + 16 * c);
        wv[q] = a.partial[(long)a.G * a.TPN * NAMP_H + (long)rr * a.TPN + p];
      }
      if (p0 == 0) {
#ifndef NW_NOSTAGE
        char* d0 = smem + tid * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) { *(f4*)(d0 + 8192 * k) = sreg2[2 * k]; *(f4*)(d0 + 8192 * k + 4096) = sreg2[2 * k + 1]; }
#endif
        stage_store(1);
        stage_load(2);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (p0 + q < a.TPN) {
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] += v[q][c];
          ws += wv[q];
        }
      }
    }
#else
    stage_store(0); stage_store(1); stage_load(2);
#endif
    bf8 xb[4], xm[X3 ? 4 : 1];
    split_rows<X3>(xb, (bf8(&)[4])xm, x);
    __syncthreads();                                             // ring entries 0, 1 in LDS
    // ---- block 0: hoisted layer 3, residual, LayerNorm 1
    {
      f4 acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = *(const f4*)(t.m3_b + 16 * c + 4 * g) * ws;
      product(acc, xb, (bf8(&)[4])xm, 0, advance, slot);
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = hv[c] + acc[c];
    }
    layernorm_row_T(x, t.ln1_g, t.ln1_b, g);
    split_rows<X3>(xb, (bf8(&)[4])xm, x);                        // x = LayerNorm-1 rows: kept for the second residual
    f4 oacc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) oacc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    // ---- blocks 1..8: the FFN, hidden block by hidden block
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      bf8 hb[4], hm[X3 ? 4 : 1];
      {
        f4 hacc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) hacc[c] = *(const f4*)(t.b_in + 128 * h + 16 * c + 4 * g);
        product(hacc, xb, (bf8(&)[4])xm, 2 * h + 1, advance, slot);
#pragma unroll
#ifndef NW_NOGELU
        for (int c = 0; c < 8; ++c) hacc[c] = gelu4(hacc[c]);
#else
        for (int c = 0; c < 8; ++c) hacc[c] *= 0.5f;
#endif
        split_rows<X3>(hb, (bf8(&)[4])hm, hacc);
      }
      product(oacc, hb, (bf8(&)[4])hm, 2 * h + 2, advance, slot);
    }
    // ---- residual, LayerNorm 2, mask; the new rows
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] += oacc[c] + *(const f4*)(t.b_out + 16 * c + 4 * g);
    layernorm_row_T(x, t.ln2_g, t.ln2_b, g);
    {
      const float mk = (t.mask && valid) ? (float)t.mask[row] : 1.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] *= mk;
      if (valid) {
        float* dst = t.hV_out + (long)row * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = x[c];
      }
    }
    split_rows<X3>(xb, (bf8(&)[4])xm, x);
    // ---- blocks 9..: the tables the next launches gather
#pragma unroll 1
    for (int pi = 0; pi < t.nproj; ++pi) {
      const ProjDesc d = t.p[pi];
      f4 acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        acc[c] = d.bias ? *(const f4*)(d.bias + 16 * c + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
        if (d.tok) acc[c] += *(const f4*)(d.tok + (long)t.S[rr] * NAMP_H + 16 * c + 4 * g);
      }
      product(acc, xb, (bf8(&)[4])xm, pi + 9, advance, slot);
      if (valid && d.out) {
        float* dst = d.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = acc[c];
      }
      __bf16* o16 = a.out16[pi];
      if (valid && o16) {
#pragma unroll
        for (int c = 0; c < 8; ++c) st_frag4(o16 + (long)row * NAMP_H, c, g, acc[c]);
      }
    }
  }
}

// node_linear_w_kernel — the residue-level table projections of large batches (node_linear_kernel: h = W_pre . x + b stored and fed on,
// then up to eight 128 x 128 projections with bias / per-token rows; model_utils.py:88,406-418,636-646) in the same form: one 16-row tile per
// wave through ALL blocks, weight blocks through the 2-slot LDS ring.  node_linear_kernel gives every (tile, projection) pair a wave of its
// own that pulls its 32-64 KiB image out of L2 and repeats the pre-stage product per projection: 62 us per launch at 64,000 residues for
// ~110 MB of rows (2 KB of weight traffic per row and projection).  X3: 1 split-bf16 products, 2 plain bf16 (hi . hi) — as node_linear_kernel<X3>.
template <int X3>
static __global__ __launch_bounds__(NODEW_THREADS, 2) void node_linear_w_kernel(const NodeLinearArgs a) {
  static_assert(X3 == 1 || X3 == 2, "split-bf16 / bf16 forms");
  constexpr bool SPLIT = X3 == 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15;
  int g = lane >> 4;
  if (a.zero && blockIdx.x == 0 && threadIdx.x < 64) a.zero[threadIdx.x] = 0u;
  constexpr int PL = SPLIT ? 2 : 1;
  const int has_pre = a.pre.img ? 1 : 0;
  const int nblk = PL * (has_pre + a.nproj);
  auto block_base = [&](const int i_) -> const char* {
    const int ic = i_ < nblk ? i_ : nblk - 1;
    const int k = ic / PL, pl = ic - k * PL;
    const float* img = (has_pre && k == 0) ? a.pre.img : a.p[k - has_pre].img;
    return (const char*)img + pl * NAMP_BIMG_BYTES;
  };
  f4 sreg[8];
  auto stage_load = [&](const int i) {
    const char* b = block_base(i) + tid * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) sreg[k] = *(const f4*)(b + k * 4096);
  };
  auto stage_store = [&](const int i) {
    char* d = smem + (i & 1) * NODEW_SLOT + tid * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) *(f4*)(d + 4096 * k) = sreg[k];
  };
  auto slot = [&](const int i) { return (const bf8*)(smem + (i & 1) * NODEW_SLOT) + lane; };
  auto advance = [&](const int i) {
    __syncthreads();
    stage_store(i + 2);
    stage_load(i + 3);
  };
  auto product = [&](f4 (&acc)[8], const bf8 (&xh)[4], const bf8 (&xm)[4], const int k) {
    if constexpr (SPLIT) {
      gemm16_hi2(acc, xh, xm, slot(2 * k));
      advance(2 * k);
      gemm16(acc, xh, slot(2 * k + 1));
      advance(2 * k + 1);
    } else {
      gemm16(acc, xh, slot(k));
      advance(k);
    }
  };
  const int ntile = (a.G_out + 15) >> 4;
  for (int tile0 = blockIdx.x * 4; tile0 < ntile; tile0 += gridDim.x * 4) {
    asm volatile("" : "+v"(tid), "+v"(g));
    const int tile = tile0 + wave;
    const int row = 16 * tile + m;
    const bool valid = row < a.G_out;
    const int rr = valid ? row : (a.G_out - 1);
    const int b = rr / a.N;
    const int src_row = (b % (a.G_src / a.N)) * a.N + (rr - b * a.N);
    __syncthreads();
    f4 sreg2[8];
    stage_load(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) sreg2[k] = sreg[k];
    stage_load(1);
    f4 x[8];
    {
      const float* src = a.X + (long)src_row * NAMP_H + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = *(const f4*)(src + 16 * c);
    }
    {
      char* d0 = smem + tid * 16;
#pragma unroll
      for (int k = 0; k < 8; ++k) *(f4*)(d0 + 4096 * k) = sreg2[k];
    }
    stage_store(1);
    stage_load(2);
    bf8 xb[4], xm[SPLIT ? 4 : 1];
    split_rows<SPLIT>(xb, (bf8(&)[4])xm, x);
    __syncthreads();
    if (has_pre) {
      f4 acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = a.pre.bias ? *(const f4*)(a.pre.bias + 16 * c + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
      product(acc, xb, (bf8(&)[4])xm, 0);
      if (valid && a.pre.out) {
        float* dst = a.pre.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = acc[c];
      }
      split_rows<SPLIT>(xb, (bf8(&)[4])xm, acc);
    }
#pragma unroll 1
    for (int pi = 0; pi < a.nproj; ++pi) {
      const ProjDesc d = a.p[pi];
      f4 acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = d.bias ? *(const f4*)(d.bias + 16 * c + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
      product(acc, xb, (bf8(&)[4])xm, has_pre + pi);
      if (d.tok) {
        const float* tk = d.tok + (long)a.S[rr] * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += *(const f4*)(tk + 16 * c);
      }
      if (valid && d.out) {
        float* dst = d.out + (long)row * NAMP_H + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f4*)(dst + 16 * c) = acc[c];
      }
      __bf16* o16 = a.out16[pi];
      if (valid && o16) {
#pragma unroll
        for (int c = 0; c < 8; ++c) st_frag4(o16 + (long)row * NAMP_H, c, g, acc[c]);
      }
    }
  }
}
