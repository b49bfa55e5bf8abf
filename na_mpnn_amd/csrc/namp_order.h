#pragma once
#include "namp_kernels.h"

// ---- decoding order on the device (round 6) ------------------------------------------------------------------------------------------
// order[b] = argsort((mask * chain_mask + 1e-4) * |randn|) and its inverse permutation (model_utils.py:389-390; na_model_utils.py:623):
// one workgroup per stream, bitonic sort of (key, index) pairs in LDS (ascending key, ties by index; a NaN key sorts last).  The keys are
// the same fp32 operations torch runs.  Replaces ~15 stock launches per score() / sample() call (mul, add, abs, mul, the segmented sort's
// launches and copies, arange, scatter).
static __global__ __launch_bounds__(512) void decoding_order_kernel(const float* __restrict__ mask, const float* __restrict__ chain_mask,
                                                                    const float* __restrict__ randn, int B_mask, int L, int P2,
                                                                    int64_t* __restrict__ order64, int32_t* __restrict__ order32,
                                                                    int32_t* __restrict__ rank32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* key = (float*)smem;
  int* idx = (int*)(smem + (size_t)P2 * 4);
  const int b = blockIdx.x, bm = b % B_mask;
  for (int i = threadIdx.x; i < P2; i += blockDim.x) {
    float k = __builtin_inff();
    if (i < L) {
      const float cm = mask[(long)bm * L + i] * (chain_mask ? chain_mask[(long)bm * L + i] : 1.0f);
      k = (cm + 0.0001f) * fabsf(randn[(long)b * L + i]);
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
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const int v = idx[i];
    if (order64) order64[(long)b * L + i] = v;
    if (order32) order32[(long)b * L + i] = v;
    rank32[(long)b * L + v] = i;
  }
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
