// Third MFMA / VALU co-execution probe (round 3): the same question as coexec_probe2.hip for the OTHER matrix shapes —
// v_mfma_f32_32x32x16_bf16 (32 cycles per issue: the microarchitecture guide reports up to 5 single-issue fillers hidden
// per gap, one wave per SIMD) and v_mfma_f32_16x16x4_f32 (the exact-fp32 headline's instruction, 32 cycles) — and for
// 1 / 2 / 3 waves per SIMD.  Every arm runs the same work per wave and step:
//     M = matrix block: 24 x 16x16x32 bf16  |  12 x 32x32x16 bf16 (same FLOPs)  |  12 x 16x16x4 f32 (same pipe cycles as 24 x 16x16x32)
//     V = VALU block  : 8 values of an FMA-only degree-8 polynomial GELU stand-in (12 VALU each) or of the shipped erf form
// arms: M alone, V alone, V then M, V interleaved between the MFMAs by sched_group_barrier.
// Output: ns per step and SIMD from HIP events + shader cycles per step from s_memtime (wave 0 of workgroup 0).
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_probe3.hip -o /tmp/coexec_probe3 && /tmp/coexec_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float gelu_poly(float x) {           // FMA-only: clamp + 8 Horner steps + 2 (12 VALU)
  const float c = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);
  const float t = c * c;
  float q = 2.2787273029e-08f;
  q = fmaf(q, t, -1.5988982626e-06f); q = fmaf(q, t, 4.7961328822e-05f); q = fmaf(q, t, -8.1407082443e-04f);
  q = fmaf(q, t, 8.7726502299e-03f); q = fmaf(q, t, -6.4573666617e-02f); q = fmaf(q, t, 3.9788372746e-01f);
  q = fmaf(q, t, 1.0e-3f); q = fmaf(q, t, 0.25f);
  return x * fmaf(c, q, 0.5f);
}
__device__ __forceinline__ float gelu_erf(float x) {            // the shipped exact-erf form: 10 VALU + 1 v_exp_f32
  const float a = fabsf(__builtin_amdgcn_fmed3f(x, -5.6f, 5.6f));
  float q = fmaf(-2.992676888e-05f, a, 7.398953830e-04f);
  q = fmaf(q, a, -7.977526064e-03f); q = fmaf(q, a, 5.323827185e-02f); q = fmaf(q, a, 4.589156358e-01f);
  q = fmaf(q, a, 1.151147093e+00f); q = fmaf(q, a, 1.0f);
  const float e = __builtin_amdgcn_exp2f(-q);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), fmaf(-2.0f, e, 1.0f), h);
}

template <int VB>
__device__ __forceinline__ float valu_block(float x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = x + 0.1f * i;
    s += VB == 1 ? gelu_poly(v) : gelu_erf(v);
  }
  return s;
}

struct Acc { f4 a[8]; f16v b[4]; };

template <int MF>
__device__ __forceinline__ void mfma_block(Acc& c, const bf8 a, const bf8 b, const float fa, const float fb) {
  if constexpr (MF == 1) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) c.a[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c.a[i], 0, 0, 0);
  } else if constexpr (MF == 2) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) c.b[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c.b[i], 0, 0, 0);
  } else if constexpr (MF == 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) c.a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c.a[i], 0, 0, 0);
  }
}

// MF: 0 none, 1 16x16x32 bf16 (24), 2 32x32x16 bf16 (12), 3 16x16x4 f32 (12).  VB: 0 none, 1 polynomial, 2 erf.  ILV: interleave
template <int MF, int VB, bool ILV>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int iters, float seed) {
  Acc c;
  for (int i = 0; i < 8; ++i) c.a[i] = (f4){seed, seed * 2, seed * 3, seed * 4};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c.b[i][j] = seed * j;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j + threadIdx.x * 1e-3f); b[j] = (__bf16)(seed - j); }
  float x = seed + threadIdx.x * 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VB != 0 && !ILV) { x = valu_block<VB>(x) * 1e-3f + seed; if (MF) b[0] = (__bf16)x; }
    if constexpr (MF != 0 && !ILV) mfma_block<MF>(c, a, b, x, seed);
    if constexpr (ILV) {
      const float xn = valu_block<VB>(x) * 1e-3f + seed;
      mfma_block<MF>(c, a, b, seed, seed);
      constexpr int NM = MF == 1 ? 24 : 12, PER = MF == 1 ? 5 : 10;
#pragma unroll
      for (int q = 0; q < NM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);      // PER VALU
      }
      x = xn;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (cyc && blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  float s = x;
  for (int i = 0; i < 8; ++i) s += c.a[i].x + c.a[i].y + c.a[i].z + c.a[i].w;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c.b[i][j];
  if (s == 12345.678f) out[0] = s;
}

template <int MF, int VB, bool ILV>
void run(const char* name, float* d, long long* cyc) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int waves = 4; waves <= 12; waves += 4) {
    hipLaunchKernelGGL((k<MF, VB, ILV>), dim3(256), dim3(64 * waves), 0, 0, d, cyc, iters, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MF, VB, ILV>), dim3(256), dim3(64 * waves), 0, 0, d, cyc, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("| %-66s | %d | %8.1f | %7.1f | %7.0f |\n", name, waves / 4, ms * 1e3, ms * 1e6 / iters, (double)h / iters);
  }
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  long long* cyc; (void)hipMalloc(&cyc, 8);
  printf("| arm (256 workgroups, 2000 steps; per wave and step) | waves / SIMD | kernel us | ns per step | s_memtime ticks per step |\n|---|---:|---:|---:|---:|\n");
  run<0, 1, false>("V   8 polynomial values (FMA only, 12 VALU each)", d, cyc);
  run<0, 2, false>("Ve  8 erf-GELU values (10 VALU + v_exp_f32 each)", d, cyc);
  run<1, 0, false>("M16 24 x v_mfma_f32_16x16x32_bf16", d, cyc);
  run<2, 0, false>("M32 12 x v_mfma_f32_32x32x16_bf16", d, cyc);
  run<3, 0, false>("Mf  12 x v_mfma_f32_16x16x4_f32", d, cyc);
  run<1, 1, false>("V then M16", d, cyc);
  run<2, 1, false>("V then M32", d, cyc);
  run<3, 1, false>("V then Mf", d, cyc);
  run<1, 1, true>("V interleaved into M16 (1 : 5)", d, cyc);
  run<2, 1, true>("V interleaved into M32 (1 : 10)", d, cyc);
  run<3, 1, true>("V interleaved into Mf (1 : 10)", d, cyc);
  run<2, 2, false>("Ve then M32", d, cyc);
  run<2, 2, true>("Ve interleaved into M32 (1 : 10)", d, cyc);
  run<3, 2, true>("Ve interleaved into Mf (1 : 10)", d, cyc);
  return 0;
}
