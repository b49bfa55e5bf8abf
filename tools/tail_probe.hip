// Standalone probes for the residue tail of the fused cfg2 launches (4 residues per workgroup, 250 workgroups):
//   (1) how fast can 250 workgroups of 12 waves each stream the SAME 768 KiB of weights out of L2 (the tail's byte floor);
//   (2) operand / result lane mapping and issue rate of v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x4 outer product, K = 1).
//   hipcc --offload-arch=gfx950 -O3 tools/tail_probe.hip -o tools/_variants/tail_probe && tools/_variants/tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(768) void stream_k(const f4* __restrict__ w, long n_f4, float* out) {
  // every workgroup reads the whole buffer once: wave-contiguous 1 KiB pieces, DEPTH pieces in flight per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f4 s = (f4){0.f, 0.f, 0.f, 0.f};
  const long pieces = n_f4 / 64;
  for (long p = wave * DEPTH; p < pieces; p += (long)nw * DEPTH) {
    f4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = (p + d < pieces) ? w[(p + d) * 64 + lane] : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) s += v[d];
  }
  if (s.x + s.y + s.z + s.w == 12345.f) out[blockIdx.x] = s.x;
}

// the tail's own request pattern: W_in image = [tk 8][unit 32][lane 64] f4; wave w takes units w, w + 12, w + 24 (8 fragments of 1 KiB, 32 KiB
// apart, per unit), then the same over a second and third matrix (W_out, projections).  AHEAD: all of a phase's fragments requested at once.
template <bool AHEAD>
__global__ __launch_bounds__(768) void tail_pattern_k(const f4* __restrict__ w, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 s = (f4){0.f, 0.f, 0.f, 0.f};
  for (int phase = 0; phase < 3; ++phase) {
    const f4* img = w + (long)phase * 16384;                 // 256 KiB per phase
    if (AHEAD) {
      f4 v[3][8];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int u = wave + 12 * t < 32 ? wave + 12 * t : wave;
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) v[t][tk] = img[(tk * 32 + u) * 64 + lane];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) s += v[t][tk];
    } else {
      for (int u = wave; u < 32; u += 12) {
        f4 v[8];
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) v[tk] = img[(tk * 32 + u) * 64 + lane];
#pragma unroll
        for (int tk = 0; tk < 8; ++tk) s += v[tk];
      }
    }
    __syncthreads();
  }
  if (s.x + s.y + s.z + s.w == 12345.f) out[blockIdx.x] = s.x;
}

__global__ void thrash_k(const f4* __restrict__ big, long n_f4, float* out) {
  f4 s = (f4){0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (long)gridDim.x * blockDim.x) s += big[i];
  if (s.x == 12345.f) out[0] = s.x;
}

__global__ void map_k(float* out) {
  // A = 100 + lane, B = 1000 * (lane + 1): D[r] printed per lane tells which (A lane, B lane) pair lands where
  const int lane = threadIdx.x;
  f4 d = (f4){0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), (float)(1 << (lane & 3)) * (1.0f + 0.001f * (lane >> 2)), d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}

__global__ __launch_bounds__(768) void rate_k(float* out, int iters, float seed) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f4){seed, seed, seed, seed};
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.f) out[0] = s;
}

int main() {
  const long bytes = 768 << 10;
  f4* w; float* out;
  hipMalloc(&w, bytes); hipMalloc(&out, 1 << 20);
  hipMemset(w, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_stream = [&](auto kern, const char* name, int wgs, int threads) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, w, bytes / 16, out);
    hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, w, bytes / 16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("stream %-10s wgs %3d x %3d threads: %7.2f us per launch (incl. ~launch overhead), %.1f TB/s aggregate, %.1f B/clk/CU at 2.4 GHz\n", name, wgs,
           threads, us, wgs * (double)bytes / us * 1e-6, (double)bytes / (us * 2400.0));
  };
  time_stream(stream_k<1>, "depth1", 250, 768);
  time_stream(stream_k<2>, "depth2", 250, 768);
  time_stream(stream_k<4>, "depth4", 250, 768);
  time_stream(stream_k<8>, "depth8", 250, 768);
  time_stream(stream_k<8>, "depth8", 63, 768);
  time_stream(stream_k<8>, "depth8", 1, 768);
  time_stream(stream_k<4>, "depth4", 250, 256);
  {
    f4* big; const long big_bytes = 64l << 20;
    hipMalloc(&big, big_bytes); hipMemset(big, 0, big_bytes);
    auto time_pat = [&](auto kern, const char* name, bool thrash) {
      float tot = 0.f;
      const int reps = 30;
      for (int i = 0; i < reps + 3; ++i) {
        if (thrash) hipLaunchKernelGGL(thrash_k, dim3(1024), dim3(256), 0, 0, big, big_bytes / 16, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(250), dim3(768), 0, 0, w, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) tot += ms;
      }
      printf("tail pattern %-22s %s: %7.2f us per launch (one launch between events: incl. ~5-6 us of launch + event overhead)\n", name,
             thrash ? "after 64 MiB streamed through L2" : "weights L2-resident            ", tot * 1e3 / reps);
    };
    time_pat(tail_pattern_k<false>, "unit by unit", false);
    time_pat(tail_pattern_k<true>, "phase requested at once", false);
    time_pat(tail_pattern_k<false>, "unit by unit", true);
    time_pat(tail_pattern_k<true>, "phase requested at once", true);
    // the contiguous stream under the same single-launch timing, for reference
    float tot = 0.f;
    for (int i = 0; i < 33; ++i) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(stream_k<8>, dim3(250), dim3(768), 0, 0, w, bytes / 16, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 3) tot += ms;
    }
    printf("contiguous stream depth8, single-launch timing: %7.2f us\n", tot * 1e3 / 30);
    tot = 0.f;
    for (int i = 0; i < 33; ++i) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(stream_k<1>, dim3(250), dim3(768), 0, 0, w, 0L, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 3) tot += ms;
    }
    printf("empty launch, single-launch timing: %7.2f us\n", tot * 1e3 / 30);
  }
  // empty-ish launch for the overhead
  hipLaunchKernelGGL(stream_k<1>, dim3(250), dim3(768), 0, 0, w, 0L, out);
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(stream_k<1>, dim3(250), dim3(768), 0, 0, w, 0L, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  { float ms; hipEventElapsedTime(&ms, e0, e1); printf("empty launch: %.2f us\n", ms * 1e3 / 50); }

  hipLaunchKernelGGL(map_k, dim3(1), dim3(64), 0, 0, out);
  std::vector<float> h(256);
  hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  printf("mfma_f32_4x4x1: A[lane] = lane + 1, B[lane] = 2^(lane & 3) * (1 + 0.001 (lane >> 2)); D[lane][r]:\n");
  for (int l = 0; l < 64; ++l) printf("  lane %2d: %9.3f %9.3f %9.3f %9.3f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);

  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(rate_k, dim3(256), dim3(768), 0, 0, out, 2000, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_k, dim3(256), dim3(768), 0, 0, out, 20000, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  { float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 20000.0 * 8 * 3;              // MFMAs per SIMD (3 waves per SIMD)
    printf("mfma 4x4x1 rate: %.2f cycles per instruction per SIMD at 2.4 GHz (%.1f MAC/clk/SIMD)\n", ms * 1e-3 * 2.4e9 / n, 256.0 / (ms * 1e-3 * 2.4e9 / n)); }
  return 0;
}
