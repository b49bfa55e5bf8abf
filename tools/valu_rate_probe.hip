// VALU issue-rate probe (round 3): cycles per wave-instruction of the candidates for a cheaper bf16-mode GELU, alone and beside
// v_mfma_f32_16x16x32_bf16 of the other waves (3 waves per SIMD, the edge kernels' occupancy; 256 workgroups).
//   fma32   : v_fma_f32 (64 values per instruction)        pkfma32 : v_pk_fma_f32 (128)
//   pkfma16 : v_pk_fma_f16 (128 values per instruction)    cvtpk   : v_cvt_pkrtz_f16_f32    dot2 : v_dot2_f32_f16
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define NCH 16        // independent chains per wave
#define NOPS 8        // dependent instructions per chain and step

template <int KIND, bool MFMA>
__global__ __launch_bounds__(768) void k(float* out, int iters, float seed) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f4){seed, seed * 2, seed * 3, seed * 4};
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + threadIdx.x * 1e-3f); b[j] = (__bf16)(seed - j); }
  float x[NCH]; f2 p[NCH]; h2 h[NCH];
  for (int c = 0; c < NCH; ++c) { x[c] = seed + c + threadIdx.x * 1e-3f; p[c] = (f2){x[c], x[c] * 0.5f}; h[c] = (h2){(_Float16)x[c], (_Float16)(x[c] * 0.5f)}; }
  const float cf = 0.999f, df = 1e-3f;
  const f2 cp = (f2){0.999f, 0.998f}, dp = (f2){1e-3f, 2e-3f};
  const h2 ch = (h2){(_Float16)0.999f, (_Float16)0.998f}, dh = (h2){(_Float16)1e-3f, (_Float16)2e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (KIND == 0) x[c] = __builtin_fmaf(x[c], cf, df);
        if (KIND == 1) p[c] = __builtin_elementwise_fma(p[c], cp, dp);
        if (KIND == 2) h[c] = __builtin_elementwise_fma(h[c], ch, dh);
        if (KIND == 3) { h[c] = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[c], x[(c + 1) % NCH])); x[c] += (float)h[c][0]; }     // cvt + add (the add keeps the chain alive)
        if (KIND == 4) x[c] = __builtin_amdgcn_fdot2(h[c], ch, x[c], false);
        if (KIND == 5) x[c] = __builtin_amdgcn_fmed3f(x[c], -4.f, cf);
      }
    if (MFMA) {
      b[0] = (__bf16)x[0];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) s += x[c] + p[c].x + p[c].y + (float)h[c][0] + (float)h[c][1];
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678f) out[0] = s;
}

template <int KIND, bool MFMA>
float run(float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, MFMA>), dim3(256), dim3(768), 0, 0, d, iters, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<KIND, MFMA>), dim3(256), dim3(768), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e6f / iters;        // ns per step
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  const char* names[6] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f16", "v_cvt_pkrtz_f16_f32 + v_cvt_f32_f16 + v_add_f32 (per instruction)", "v_dot2_f32_f16", "v_med3_f32"};
  float alone[6], with[6];
  alone[0] = run<0, false>(d); with[0] = run<0, true>(d);
  alone[1] = run<1, false>(d); with[1] = run<1, true>(d);
  alone[2] = run<2, false>(d); with[2] = run<2, true>(d);
  alone[3] = run<3, false>(d); with[3] = run<3, true>(d);
  alone[4] = run<4, false>(d); with[4] = run<4, true>(d);
  alone[5] = run<5, false>(d); with[5] = run<5, true>(d);
  // MFMA alone: KIND 6 = no VALU
  const float m = run<6, true>(d);
  printf("3 waves per SIMD, %d instructions per wave and step; 24 x v_mfma_f32_16x16x32_bf16 per wave and step alone: %.1f ns\n", NCH * NOPS, m);
  printf("| instruction | alone: ns per step | ns per wave-instruction (x3 waves) | beside the MFMAs: ns per step | extra ns per wave-instruction |\n|---|---:|---:|---:|---:|\n");
  for (int i = 0; i < 6; ++i) {
    const int n = (i == 3 ? 3 : 1) * NCH * NOPS * 3;
    printf("| %s | %.1f | %.3f | %.1f | %.3f |\n", names[i], alone[i], alone[i] / n, with[i], (with[i] - m) / n);
  }
  return 0;
}
