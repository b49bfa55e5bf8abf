// Round 6: what the address pattern of a row tile costs on the vector-memory path.  Every wave streams 8-KiB tiles (32 rows x 256 B) with eight
// 16-byte-per-lane instructions per tile, as the bf16-storage edge kernels do, under three lane -> address maps:
//   P0  fragment order B per ROW   (shipped): lane (r = l & 31, hk = l >> 5), instruction s: row r, piece 2s + hk   -> 64 isolated 16-byte pieces
//   P1  fully contiguous: instruction s, lane l: byte s * 1024 + 16 l
//   P2  fragment order per 16-ROW TILE: lane (m = l & 15, half = (l >> 4) & 1, hk = l >> 5): tile half, piece (2s + hk), row m -> 4 runs of 256 B
// modes: r = loads only, w = stores only, rw = load, add, store (the edge update's traffic).  256 workgroups x 512 threads, persistent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int P, int MODE>
__global__ __launch_bounds__(512) void probe(const f4* __restrict__ in, f4* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long stride = (long)gridDim.x * 8;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long t = (long)blockIdx.x * 8 + wave; t < ntiles; t += stride) {
    f4 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      long off;                      // in 16-byte units within the 8-KiB tile
      if (P == 0) off = (lane & 31) * 16 + 2 * s + (lane >> 5);
      else if (P == 1) off = s * 64 + lane;
      else off = ((lane >> 4) & 1) * 256 + (2 * s + (lane >> 5)) * 16 + (lane & 15);
      if (MODE != 1) v[s] = in[t * 512 + off]; else v[s] = acc + (float)s;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      long off;
      if (P == 0) off = (lane & 31) * 16 + 2 * s + (lane >> 5);
      else if (P == 1) off = s * 64 + lane;
      else off = ((lane >> 4) & 1) * 256 + (2 * s + (lane >> 5)) * 16 + (lane & 15);
      if (MODE == 0) acc += v[s]; else out[t * 512 + off] = v[s] + 1.0f;
    }
  }
  if (MODE == 0 && acc.x == 12345.f) out[0] = acc;
}
template <int P, int MODE> float run(const f4* in, f4* out, long nt) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<P, MODE>), dim3(256), dim3(512), 0, 0, in, out, nt);
  hipEventRecord(a, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<P, MODE>), dim3(256), dim3(512), 0, 0, in, out, nt);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e3f;
}
int main() {
  const long nt = 96000;                       // 96,000 x 8 KiB = 786 MB: the h_E rows of the cfg3 batch in bf16
  f4 *in, *out; hipMalloc(&in, nt * 8192); hipMalloc(&out, nt * 8192); hipMemset(in, 0, nt * 8192);
  const char* mn[3] = {"read ", "write", "r + w"};
  float t[3][3];
  t[0][0] = run<0, 0>(in, out, nt); t[1][0] = run<1, 0>(in, out, nt); t[2][0] = run<2, 0>(in, out, nt);
  t[0][1] = run<0, 1>(in, out, nt); t[1][1] = run<1, 1>(in, out, nt); t[2][1] = run<2, 1>(in, out, nt);
  t[0][2] = run<0, 2>(in, out, nt); t[1][2] = run<1, 2>(in, out, nt); t[2][2] = run<2, 2>(in, out, nt);
  for (int m = 0; m < 3; ++m)
    printf("%s  P0 per-row pieces %7.1f us (%5.2f TB/s)   P1 contiguous %7.1f us (%5.2f TB/s)   P2 16-row tiles %7.1f us (%5.2f TB/s)\n", mn[m],
           t[0][m], (m == 2 ? 2 : 1) * nt * 8192.0 / t[0][m] * 1e-6, t[1][m], (m == 2 ? 2 : 1) * nt * 8192.0 / t[1][m] * 1e-6,
           t[2][m], (m == 2 ? 2 : 1) * nt * 8192.0 / t[2][m] * 1e-6);
  return 0;
}
