// Standalone probe: do VALU work (exp-heavy, like GELU / RBF generation) and bf16 MFMAs of DIFFERENT waves on one SIMD
// overlap?  Each wave alternates [VALU block][MFMA block] per step, 12 waves per CU (3 per SIMD), one workgroup per CU.
//   MODE 0: MFMA blocks only   1: VALU blocks only   2: both, step by step   3: both, VALU of step s+1 interleaved by hand
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_probe.hip -o tools/_variants/coexec_probe && tools/_variants/coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float valu_block(float x) {       // 8 values x (2 transcendentals + ~10 ops), like gelu on 8 values
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = x + 0.1f * i;
    const float e = __builtin_amdgcn_exp2f(-(v * v));
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(v), 0.27f, 1.0f));
    float q = fmaf(1.06f, t, -1.45f); q = fmaf(q, t, 1.42f); q = fmaf(q, t, -0.28f); q = fmaf(q, t, 0.25f);
    s += fmaf(-(q * t), e, 1.0f) * v;
  }
  return s;
}

template <int MODE, int NSLEEP>
__global__ __launch_bounds__(768) void k(float* out, int iters, float seed) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f4){seed, seed * 2, seed * 3, seed * 4};
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + threadIdx.x * 1e-3f); b[j] = (__bf16)(seed - j); }
  float x = seed + threadIdx.x * 1e-3f;
  if (NSLEEP) { const int w = threadIdx.x >> 8; for (int q = 0; q < w; ++q) __builtin_amdgcn_s_sleep(NSLEEP); }
  if (MODE == 3) {
    // software pipeline inside the wave: the VALU block of step s+1 is issued BETWEEN the MFMAs of step s
    for (int it = 0; it < iters; ++it) {
      const float xn = valu_block(x) * 1e-3f + seed;            // independent of this step's MFMAs
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);      // 5 VALU / TRANS
      }
      x = xn; b[0] = (__bf16)x;
    }
  } else
  for (int it = 0; it < iters; ++it) {
    if (MODE != 0) { x = valu_block(x) * 1e-3f + seed; if (MODE != 1) b[0] = (__bf16)x; }
    if (MODE != 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = x;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NSLEEP>
void run(const char* name, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, NSLEEP>), dim3(256), dim3(768), 0, 0, d, iters, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, NSLEEP>), dim3(256), dim3(768), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-44s %8.1f us  = %6.1f ns per step per SIMD (3 waves)\n", name, ms * 1e3, ms * 1e6 / iters);
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  run<0, 0>("MFMA only (24 x 16x16x32 bf16 per step)", d);
  run<1, 0>("VALU only (8 gelu-like values per step)", d);
  run<2, 0>("both, alternating per step", d);
  run<3, 0>("both, VALU of step s+1 between the MFMAs of step s", d);
  return 0;
}
