// Second MFMA / VALU co-execution probe (round 2): ROLE-SPECIALISED waves.
//
// tools/coexec_probe.hip let every wave alternate [VALU block][MFMA block] and found the two times ADD.  The
// microarchitecture guide reports the opposite for a matrix-only wave beside a VALU-only wave.  This probe separates the
// roles: 12 waves per CU (3 per SIMD, one workgroup per CU like the edge kernels).  Waves w, w+4, w+8 of a workgroup land
// on the same SIMD (checked below through HW_REG_HW_ID), so role = wave / 4 puts one wave of each role on every SIMD.
//
//   A  every wave: 24 MFMA (16x16x32 bf16) per step                                  -> T_m
//   B  every wave: VALU block per step (8 values: erf-GELU with exp2 + rcp)          -> T_v (transcendental)
//   C  every wave: VALU block per step (8 values: polynomial GELU, FMA only)         -> T_p
//   D  every wave: B then A per step (the round-1 "both" arm)
//   E  specialised: role 0 = 72 MFMA per step, roles 1, 2 = 12 values each per step  (same work per SIMD as D)
//   F  as E with the polynomial block (same work per SIMD as C + A)
//   G  as E, MFMA wave at s_setprio 3
//   H  every wave: C then A per step (polynomial "both")
//   I  every wave: per step 24 MFMA with the FMA-only block hand-interleaved 1 MFMA : 4 VALU
//
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_probe2.hip -o /tmp/coexec_probe2 && /tmp/coexec_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float gelu_trans(float x) {          // the shipped erf-GELU: 11 full-rate + 2 quarter-rate ops
  const float v = x * 0.84932180028801904f;
  const float e = __builtin_amdgcn_exp2f(-(v * v));
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(v), 0.27273943f, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f); q = fmaf(q, t, -0.284496736f); q = fmaf(q, t, 0.254829592f);
  const float y = fmaf(-(q * t), e, 1.0f);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), y, h);
}

__device__ __forceinline__ float gelu_poly(float x) {           // FMA-only stand-in: clamp + degree-13 odd polynomial (15 ops)
  const float c = fminf(fmaxf(x, -4.5f), 4.5f);
  const float t = c * c;
  float q = 1.1e-9f;
  q = fmaf(q, t, -1.3e-8f); q = fmaf(q, t, 1.0e-7f); q = fmaf(q, t, -3.7e-6f); q = fmaf(q, t, 7.5e-5f);
  q = fmaf(q, t, -1.0e-3f); q = fmaf(q, t, 9.5e-3f); q = fmaf(q, t, -6.5e-2f); q = fmaf(q, t, 3.9e-1f);
  return x * fmaf(c, q, 0.5f);
}

template <bool POLY, int NV>
__device__ __forceinline__ float valu_block(float x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float v = x + 0.1f * i;
    s += POLY ? gelu_poly(v) : gelu_trans(v);
  }
  return s;
}

template <int NM>
__device__ __forceinline__ void mfma_block(f4 (&acc)[8], const bf8 a, const bf8 b) {
#pragma unroll
  for (int r = 0; r < NM / 8; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
}

enum { ARM_A, ARM_B, ARM_C, ARM_D, ARM_E, ARM_F, ARM_G, ARM_H, ARM_I };

template <int ARM>
__global__ __launch_bounds__(768) void k(float* out, int* simd_of_wave, int iters, float seed) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f4){seed, seed * 2, seed * 3, seed * 4};
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + threadIdx.x * 1e-3f); b[j] = (__bf16)(seed - j); }
  float x = seed + threadIdx.x * 1e-3f;
  const int wave = threadIdx.x >> 6;
  const int role = wave >> 2;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && simd_of_wave) {
    // HW_REG_HW_ID (id 4): SIMD_ID = bits [5:4]
    simd_of_wave[wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
  if (ARM == ARM_G && role == 0) __builtin_amdgcn_s_setprio(3);
  for (int it = 0; it < iters; ++it) {
    if (ARM == ARM_A) mfma_block<24>(acc, a, b);
    if (ARM == ARM_B) x = valu_block<false, 8>(x) * 1e-3f + seed;
    if (ARM == ARM_C) x = valu_block<true, 8>(x) * 1e-3f + seed;
    if (ARM == ARM_D) { x = valu_block<false, 8>(x) * 1e-3f + seed; b[0] = (__bf16)x; mfma_block<24>(acc, a, b); }
    if (ARM == ARM_H) { x = valu_block<true, 8>(x) * 1e-3f + seed; b[0] = (__bf16)x; mfma_block<24>(acc, a, b); }
    if (ARM == ARM_E || ARM == ARM_G) {
      if (role == 0) mfma_block<72>(acc, a, b);
      else x = valu_block<false, 12>(x) * 1e-3f + seed;
    }
    if (ARM == ARM_F) {
      if (role == 0) mfma_block<72>(acc, a, b);
      else x = valu_block<true, 12>(x) * 1e-3f + seed;
    }
    if (ARM == ARM_I) {
      const float xn = valu_block<true, 8>(x) * 1e-3f + seed;
      mfma_block<24>(acc, a, b);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // 5 VALU
      }
      x = xn;
    }
  }
  float s = x;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678f) out[0] = s;
}

template <int ARM>
void run(const char* name, float* d, int* simd) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<ARM>), dim3(256), dim3(768), 0, 0, d, simd, iters, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<ARM>), dim3(256), dim3(768), 0, 0, d, nullptr, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("| %-78s | %8.1f | %7.1f |\n", name, ms * 1e3, ms * 1e6 / iters);
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  int* simd; (void)hipMalloc(&simd, 64);
  (void)hipMemset(simd, 0xff, 64);
  printf("| arm (256 workgroups x 12 waves, 2000 steps) | kernel us | ns per step and SIMD |\n|---|---:|---:|\n");
  run<ARM_A>("A  every wave: 24 MFMA 16x16x32 bf16", d, simd);
  run<ARM_B>("B  every wave: 8 erf-GELU values (exp2 + rcp)", d, simd);
  run<ARM_C>("C  every wave: 8 polynomial-GELU values (FMA only)", d, simd);
  run<ARM_D>("D  every wave: B then A", d, simd);
  run<ARM_H>("H  every wave: C then A", d, simd);
  run<ARM_E>("E  specialised: 1 wave 72 MFMA || 2 waves 12 erf-GELU values each", d, simd);
  run<ARM_G>("G  as E, MFMA wave at s_setprio 3", d, simd);
  run<ARM_F>("F  specialised: 1 wave 72 MFMA || 2 waves 12 polynomial values each", d, simd);
  run<ARM_I>("I  every wave: 24 MFMA with the polynomial block interleaved 1 : 5", d, simd);
  int h[16];
  (void)hipMemcpy(h, simd, 48, hipMemcpyDeviceToHost);
  printf("\nSIMD_ID of waves 0..11 of workgroup 0 (HW_REG_HW_ID[5:4]):");
  for (int i = 0; i < 12; ++i) printf(" %d", (h[i] >> 4) & 3);
  printf("\n");
  return 0;
}
