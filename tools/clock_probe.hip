// Effective shader clock of a launch that occupies ONE CU vs the whole chip (round 3: why does a one-workgroup sampler step take ~90 us?).
// Each workgroup runs a dependent-FMA loop for a fixed number of iterations and reads s_memtime (shader clock) and the constant
// 100 MHz wall clock before / after; also a dependent global-load chain (pointer chase over an L2-resident 1 MiB buffer) for the
// load-to-use latency in ns.   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float* out, long long* t, int iters) {
  float x = threadIdx.x * 1e-3f;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 1e-3f);
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
  if (x == 123.f) out[0] = x;
}
__global__ void chase(const int* next, int* out, long long* t, int steps) {
  int p = threadIdx.x;
  const long long w0 = wall_clock64();
  for (int i = 0; i < steps; ++i) p = next[p];
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = w1 - w0;
  if (p == -1) out[0] = p;
}
// one workgroup of `blockDim/64` waves streams `bytes` of an L2-resident buffer, U loads of 1 KiB (16 B per lane) in flight per wave
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void stream(const f4* __restrict__ buf, float* out, long long* t, long n_f4, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
  const long long w0 = wall_clock64();
  for (int r = 0; r < reps; ++r)
    for (long base = (long)wave * U * 64; base + U * 64 <= n_f4; base += (long)nw * U * 64) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = buf[base + u * 64 + lane];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = w1 - w0;
  if (acc.x == 123.f) out[0] = acc.x;
}
template <int U>
void run_stream(const f4* buf, float* d, long long* t, long n_f4, int waves, int grid) {
  long long ht;
  hipLaunchKernelGGL(stream<U>, dim3(grid), dim3(waves * 64), 0, 0, buf, d, t, n_f4, 20);
  hipLaunchKernelGGL(stream<U>, dim3(grid), dim3(waves * 64), 0, 0, buf, d, t, n_f4, 20);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
  printf("| %d workgroup(s) x %2d waves, %2d KiB in flight per wave | %.1f |\n", grid, waves, U, 20.0 * n_f4 * 16 / (ht * 10.0));
}

int main() {
  float* d; long long* t; (void)hipMalloc(&d, 4); (void)hipMalloc(&t, 16);
  const int n = 1 << 18;                                  // 1 MiB of ints: L2 resident, 64 lanes chase 64 independent chains
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 40503 + 12345) % n);
  int* nx; (void)hipMalloc(&nx, n * 4); (void)hipMemcpy(nx, h.data(), n * 4, hipMemcpyHostToDevice);
  int* o; (void)hipMalloc(&o, 4);
  long long ht[2];
  printf("| launch | shader cycles | wall ticks (100 MHz) | effective shader clock (GHz) |\n|---|---:|---:|---:|\n");
  for (int grid : {1, 8, 256}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin, dim3(grid), dim3(512), 0, 0, d, t, 2000000);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
      printf("| spin, %3d workgroups x 512 threads, run %d | %lld | %lld | %.3f |\n", grid, rep, ht[0], ht[1], (double)ht[0] / (ht[1] * 10.0));
    }
  }
  printf("\n| launch | ns per dependent load (64-lane pointer chase, 1 MiB buffer) |\n|---|---:|\n");
  for (int grid : {1, 256}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(chase, dim3(grid), dim3(64), 0, 0, nx, o, t, 20000);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(ht, t, 8, hipMemcpyDeviceToHost);
      printf("| chase, %3d workgroups, run %d | %.1f |\n", grid, rep, ht[0] * 10.0 / 20000);
    }
  }
  f4* wbuf; const long n_f4 = 2621440 / 16;                 // 2.5 MiB: the sampler's decoder weights per step
  (void)hipMalloc(&wbuf, n_f4 * 16); (void)hipMemset(wbuf, 0, n_f4 * 16);
  printf("\n| one CU streaming a 2.5 MiB L2-resident buffer (20 passes) | GB/s per workgroup |\n|---|---:|\n");
  for (int waves : {4, 8, 16}) {
    run_stream<4>(wbuf, d, t, n_f4, waves, 1);
    run_stream<8>(wbuf, d, t, n_f4, waves, 1);
    run_stream<16>(wbuf, d, t, n_f4, waves, 1);
    run_stream<32>(wbuf, d, t, n_f4, waves, 1);
  }
  run_stream<16>(wbuf, d, t, n_f4, 8, 8);
  run_stream<16>(wbuf, d, t, n_f4, 8, 64);
  return 0;
}
