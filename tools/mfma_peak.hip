// Standalone probe: sustained v_mfma_f32_16x16x4_f32 rate on this chip for the occupancy shapes the
// edge kernels use (how far below the 157.3 TFLOP/s nominal peak the DVFS-limited clock sits).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_variants/mfma_peak && tools/_variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(768) void k(float* out, int iters, float seed) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){seed, seed * 2, seed * 3, seed * 4};
  float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;   // x 4 x NACC MFMAs per wave
  for (int threads : {256, 512, 768}) {
    for (int grid : {250, 256, 512}) {
      hipLaunchKernelGGL(k<8>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.0f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      double mfma = (double)grid * (threads / 64) * iters * 4 * 8;
      double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
      // cycles per MFMA per SIMD assuming waves are spread evenly over 256 CUs x 4 SIMDs
      double waves_per_simd = (double)grid * (threads / 64) / 1024.0;
      double us = ms * 1e3;
      printf("threads=%4d grid=%4d  %8.1f us  %7.1f TFLOP/s  (%.2f waves/SIMD avg; implied clock if 32 cyc/MFMA: %.2f GHz)\n",
             threads, grid, us, tf, waves_per_simd, waves_per_simd * iters * 32 * 32 / us / 1e3);
    }
  }
  return 0;
}
