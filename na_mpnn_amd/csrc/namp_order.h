#pragma once
#include "namp_kernels.h"

// ---- decoding order on the device (round 6): decoding_order_body (namp_kernels.h) as a launch of its own — one workgroup per stream.  Replaces
// ~15 stock launches per score() / sample() call (mul, add, abs, mul, the segmented sort's launches and copies, arange, scatter).
static __global__ __launch_bounds__(512) void decoding_order_kernel(const OrderJob o) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  decoding_order_body(o, blockIdx.x, smem);
}


// ---- work lists of the level walk on the device (round 6) ----------------------------------------------------------------------------
// From level[b][v] (namp_sample_levels_dep, by visit) to what namp_decoder_sample_walk reads, for the plain branch (every visit its own
// work item): work[n][2] = (stream, visit) grouped by level, level_off[l] = number of items of a level below l (L + 2 entries),
// n_levels[0] = number of non-empty levels.  One workgroup: LDS histogram, scan, scatter through LDS cursors (the order of a level's
// items among themselves is not defined — they are independent by construction).  Replaces ~20 stock launches per design call (casts, a
// stable argsort, gathers, divisions, stack, scatter_add, cumsum, cat).
static __global__ __launch_bounds__(1024) void work_lists_kernel(const int32_t* __restrict__ level, int n, int L, int32_t* __restrict__ work,
                                                                  int32_t* __restrict__ level_off, int32_t* __restrict__ n_levels) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* cnt = (int*)smem;                    // [L + 2]
  __shared__ int part[1024];
  __shared__ int nz;
  const int tid = threadIdx.x, nb = L + 1;
  for (int i = tid; i < nb + 1; i += 1024) cnt[i] = 0;
  if (tid == 0) nz = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int l = level[i];
    atomicAdd(&cnt[l < 0 ? 0 : (l > L ? L : l)], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0 .. nb): thread t owns entries [t * per, (t + 1) * per)
  const int per = (nb + 1023) / 1024;
  int sum = 0, mine_nz = 0;
  for (int q = 0; q < per; ++q) { const int i = tid * per + q; if (i < nb) { sum += cnt[i]; mine_nz += cnt[i] > 0; } }
  part[tid] = sum;
  if (mine_nz) atomicAdd(&nz, mine_nz);
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int q = 0; q < per; ++q) {
    const int i = tid * per + q;
    if (i < nb) { const int c = cnt[i]; cnt[i] = run; level_off[i] = run; run += c; }
  }
  if (tid == 1023) level_off[nb] = part[1023];
  if (tid == 0) n_levels[0] = nz;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int l = level[i];
    const int pos = atomicAdd(&cnt[l < 0 ? 0 : (l > L ? L : l)], 1);
    work[2 * pos] = i / L; work[2 * pos + 1] = i - (i / L) * L;
  }
}

// ---- output head on the matrix pipe (round 6) ----------------------------------------------------------------------------------------
// log_softmax(W_out . h_V + b) (model_utils.py:420-421) for large batches: logits_kernel gives 33 of a wave's 64 lanes one token each and
// walks the residue's 128 channels with broadcast loads — 86 us for 64,000 residues (1.7 % of the cfg3 step) for 0.5 GFLOP and 42 MB.  Here a
// wave takes a 16-row tile: rows in the register layout of the residue kernels, W_out as an exact-fp32 fragment image [8 k-tiles][3 token
// tiles] built once per workgroup in LDS (tokens past the vocabulary: zero rows), 96 v_mfma_f32_16x16x4_f32 per tile, the soft-max over a
// row's tokens across the four lane groups, and the tile's [16][V] block — contiguous in memory — written through LDS in 16-byte pieces.
#define LOGITS_MFMA_LDS(V) (8 * 3 * 64 * 16 + 4 * 16 * (V) * 4 + 64)
static __global__ __launch_bounds__(256) void logits_mfma_kernel(const float* __restrict__ hV, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, float* __restrict__ log_probs,
                                                                 float* __restrict__ logits_out, int G, int V) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* img = (f4*)smem;                                                     // [tk 8][tn 3][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  float* slot = (float*)(smem + 8 * 3 * 64 * 16) + wave * 16 * V;          // this wave's [16][V] output block
  for (int idx = tid; idx < 8 * 3 * 64; idx += 256) {
    const int l = idx & 63, tn = (idx >> 6) % 3, tk = idx / 192;
    const int tok = 16 * tn + (l & 15);
    img[idx] = tok < V ? *(const f4*)(W + (long)tok * NAMP_H + 16 * tk + 4 * (l >> 4)) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  f4 b3[3];
#pragma unroll
  for (int tn = 0; tn < 3; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int tok = 16 * tn + 4 * g + r; b3[tn][r] = tok < V ? bias[tok] : 0.f; }
  const int ntile = (G + 15) >> 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
    const int row = 16 * tile + m;
    const int rr = row < G ? row : (G - 1);
    f4 x[8];
    const float* src = hV + (long)rr * NAMP_H + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = *(const f4*)(src + 16 * t);
    f4 acc[3] = {b3[0], b3[1], b3[2]};
    chain_gemm<8, 3, false>(acc, x, img + lane, 3);
    // soft-max over the row's V tokens: this lane holds tokens 16 tn + 4 g + r
    float mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (16 * tn + 4 * g + r < V) mx = fmaxf(mx, acc[tn][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float e = 0.f;
#pragma unroll
    for (int tn = 0; tn < 3; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (16 * tn + 4 * g + r < V) e += expf(acc[tn][r] - mx);
    e = xg_sum(e);
    const float le = logf(e);
    const int rows_here = (G - 16 * tile) < 16 ? (G - 16 * tile) : 16;
    const long nq = ((long)rows_here * V) >> 2, base = (long)16 * tile * V;     // whole 16-byte pieces of the tile's block
    for (int pass = 0; pass < (logits_out ? 2 : 1); ++pass) {
#pragma unroll
      for (int tn = 0; tn < 3; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = 16 * tn + 4 * g + r;
          if (tok < V) slot[m * V + tok] = pass ? acc[tn][r] : (acc[tn][r] - mx) - le;
        }
      float* dst = (pass ? logits_out : log_probs) + base;
      // (16 rows x V floats start at a multiple of 16 V floats = 64 V bytes: 16-byte aligned for every V)
      for (long q = lane; q < nq; q += 64) *(f4*)(dst + 4 * q) = *(const f4*)(slot + 4 * q);
      for (long q = 4 * nq + lane; q < (long)rows_here * V; q += 64) dst[q] = slot[q];
    }
  }
}
