// edge_mlp_bf16p_kernel — round 6: the bf16-storage edge launches (namp_bf16s32.h: same rows, tables, images and results) re-sequenced so
// that a wave never sits in front of the matrix pipe with vector work pending.
//
// What round 5's counters said about edge_mlp_bf16s32_kernel (profiles/r05h_pmc_issue_counters_cfg3.txt): 96 MFMAs of 32 cycles and ~1,290
// VALU instructions per 32-row tile, a wave 21 % of its life at an s_waitcnt, 44 % waiting to issue, 34 % issuing — 10,100 cycles per tile
// and SIMD where the matrix pipe needs 3,072 and the vector pipe ~5,000.  Its stream is K-step major: eight steps of {GELU of eight values
// (~45 dependent packed operations), then four MFMAs back to back}: the wave stands in front of a busy pipe for three of every four MFMAs
// with its own GELU work behind them, the first layer is 32 MFMAs with no vector work at all, and the two waves of a SIMD run the same
// program, so their matrix bursts and their vector stretches coincide as often as they interleave (MI355X_MICROARCH "Two waves per
// SIMD": an in-order wave cannot slip an MFMA into idle pipe time unless the idle slot is at that point of ITS program).
//
// Here a tile is a chain of BLOCKS, one accumulator tile (32 rows x 32 channels, eight K-steps) each, accumulator-major:
//     layer 1: tn 0 1 2 3   layer 2: tn 0 1 2 3   (edge update) layer 3: tn 0 1 2 3
// and every MFMA slot carries a slice of the PREVIOUS block's epilogue (bias / gathered term, GELU, conversion to the next layer's operand —
// or the K-sum, or the residual and LayerNorm sums) and of the NEXT block's initial value: one MFMA, then ~14 vector instructions, for the
// whole tile.  It works because a block's results are complete after its eight MFMAs (not at the end of the layer), and the next layer's
// first block consumes its K-steps in the order the previous layer's blocks finish — the last block's operands (K-steps 6, 7) are made while
// K-steps 0..5 issue.  Weight fragments come from LDS three slots ahead through a four-entry ring.  Scheduling barriers pin the order (the
// compiler's own schedule hoists the packed chains and groups the MFMAs again).  Accumulation order per accumulator and every rounding point
// are those of edge_mlp_bf16s32_kernel: the two produce identical bits (tests/test_gpu_parity.py::test_bf16p_equals_bf16s32).
#pragma once
#include <type_traits>
#include "namp_bf16s32.h"

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#ifndef P32_NOFENCE
#define P32_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define P32_SB()
#endif
#ifndef P32_DIST
#define P32_DIST 3                  // weight fragments are requested this many MFMA slots ahead
#endif

__device__ __forceinline__ float bf16_lo(const unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(const unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// four consecutive bf16 of a bf8 (half h) as fp32
__device__ __forceinline__ f4 bf8_quad(const bf8& p, const int h) {
  return (f4){(float)p[4 * h], (float)p[4 * h + 1], (float)p[4 * h + 2], (float)p[4 * h + 3]};
}
__device__ __forceinline__ void set_quad(bf8& p, const int h, const f4 v) {
  p[4 * h] = (__bf16)v.x; p[4 * h + 1] = (__bf16)v.y; p[4 * h + 2] = (__bf16)v.z; p[4 * h + 3] = (__bf16)v.w;
}

// -DP32_STAMPS: per-wave time line (s_memtime deltas between fixed points of a tile, summed over the wave's tiles in scalar registers, written
// once at the end by the waves of workgroup 0; read back with namp_debug_p32_stamps).  Points: 0 top of the tile .. 1 rows awaited .. 2 requests
// for this tile issued, next metadata .. 3/4/5/6 behind each quarter of the MFMA slots .. 7 last epilogue .. 8 LayerNorm / K-sum tail .. 9 stores issued
#ifdef P32_STAMPS
__device__ unsigned long long g_p32_stamps[4][8][12];
#define P32_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          st_sum[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define P32_STAMP(i)
#endif

// LDS: W1 | W2 | W3 or W_e (32 KiB each) | constants (2 KiB) | per wave: 32 row weights (128 B)
#define BF16P_LDS (3 * NAMP_BIMG_BYTES + 2048 + 8 * 128)

// LN2P (edge update only): LayerNorm 3 with the two-pass variance of edge_mlp_bf16s32_kernel (bit-identical results: the equality test's
// instantiation).  The product's instantiation accumulates sum and sum of squares inside the layer-3 slots (var = E[x^2] - mean^2: rows of
// O(1) magnitude, 128 channels — 1e-6 relative in fp32) and normalises with two fused multiply-adds per value.
template <int MODE, bool EMB = false, bool LN2P = false>
__global__ __launch_bounds__(512) void edge_mlp_bf16p_kernel(const EdgeArgs a) {
  static_assert(!EMB || MODE == MODE_ENC_MSG, "EMB: first encoder message only");
  static_assert(!LN2P || MODE == MODE_ENC_EDGE, "LN2P: edge update only");
  constexpr bool EDGE = MODE == MODE_ENC_EDGE;
  constexpr int NL = EDGE ? 3 : EMB ? 3 : 2;               // layers of the chain (EMB: the embedding product in front)
  constexpr int NB = 4 * NL, NSLOT = 8 * NB;
  constexpr int L1 = EMB ? 1 : 0, L2 = L1 + 1, L3 = EDGE ? 2 : -1, LE = EMB ? 0 : -1;       // layer index of each product
  constexpr int D = P32_DIST, RING = D + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int r = lane & 31, hk = lane >> 5, m = r & 15, half = r >> 4;
  const long ntiles = (long)a.G * a.TPN;
  const long npairs = (ntiles + 1) >> 1;
  const long stride = (long)gridDim.x * nwaves;
  long pair = (long)blockIdx.x * nwaves + wave;
  // Metadata of a pair of tiles in two stages, so that no request is ever waited for where it was issued (round 5's kernels evaluated a
  // tile's metadata in one piece at the top of the tile before: the edge update's neighbour id, the messages' mask / rank words — a dependent
  // load whose wait also drained the gathered rows requested just before it; 2,800 / 1,300 of a tile's 17,500 / 12,000 cycles, profiles/r06a):
  //   meta_issue (in a free MFMA slot of the first block): everything that follows from the tile number and the neighbour id — the id itself was
  //     requested a tile earlier —, and the requests for the mask / rank words;
  //   meta_finish (in a free slot of the last block): row weight and table choice from those words.
  struct MetaRaw { int r0, r1; };
  auto pair_c = [&](const long p) { return p < npairs ? p : npairs - 1; };
  auto idx_of = [&](const long p) {
    const long t = 2 * pair_c(p) + half;
    return a.E_idx[tile_erow<MODE>(a, t < ntiles ? t : (ntiles - 1), m)];
  };
  auto meta_issue = [&](const long p, const int j_loc, TileMeta& t, bool& ok) {
    const long tl = 2 * p + half;
    ok = tl < ntiles;
    const long tile = ok ? tl : (ntiles - 1);
    t.node = (int)((unsigned)tile / (unsigned)a.TPN);
    t.kt = (int)tile - t.node * a.TPN;
    const int b_dec = t.node / a.N;
    const int i_loc = t.node - b_dec * a.N;
    const int node_enc = (MODE == MODE_DEC_MSG) ? ((b_dec % (a.G_enc / a.N)) * a.N + i_loc) : t.node;
    const int k = 16 * t.kt + m;
    t.valid = k < a.K;
    t.erow = (long)node_enc * a.K + (t.valid ? k : 0);
    t.w_row = 0.f;
    t.pa_row = t.node;
    MetaRaw raw = {1, 1};
    typedef const __attribute__((address_space(1))) int32_t* gptr;
    if constexpr (MODE == MODE_DEC_MSG) {
      const int j_dec = b_dec * a.N + j_loc;
      raw.r0 = *(gptr)(a.rank + j_dec); raw.r1 = *(gptr)(a.rank + t.node);
      t.pj_row = j_dec;                                                   // (backward table; meta_finish switches to the forward one)
      t.pj_from1 = false;
    } else {
      const int j = t.node - i_loc + j_loc;
      t.pj_from1 = false;
      t.pj_row = j;
      if constexpr (MODE == MODE_ENC_MSG) {
        const gptr one = (gptr)&g_meta_one;
        const gptr p0 = a.mask_attend ? (gptr)(a.mask_attend + t.erow) : a.mask ? (gptr)(a.mask + t.node) : one;
        const gptr p1 = (!a.mask_attend && a.mask) ? (gptr)(a.mask + j) : one;
        raw.r0 = *p0; raw.r1 = *p1;
      }
    }
    return raw;
  };
  auto meta_finish = [&](TileMeta& t, const MetaRaw raw, const bool ok, const int j_loc) {
    if constexpr (MODE == MODE_DEC_MSG) {
      const bool bwd = raw.r0 < raw.r1;
      const int b_dec = t.node / a.N;
      const int i_loc = t.node - b_dec * a.N;
      const int node_enc = (b_dec % (a.G_enc / a.N)) * a.N + i_loc;
      t.pj_from1 = !bwd;
      t.pj_row = bwd ? t.pj_row : (long)(node_enc - i_loc + j_loc);
      t.w_row = t.valid ? (1.0f / 30.0f) : 0.f;
    } else if constexpr (MODE == MODE_ENC_MSG) {
      t.w_row = t.valid ? ((float)(raw.r0 * raw.r1) * (1.0f / 30.0f)) : 0.f;
    }
    if (!ok) { t.valid = false; t.w_row = 0.f; }
  };
  TileMeta cur;
  int idx_n1;
  {
    bool ok0;
    const int j0 = idx_of(pair);
    const MetaRaw raw0 = meta_issue(pair_c(pair), j0, cur, ok0);
    meta_finish(cur, raw0, ok0, j0);
    idx_n1 = idx_of(pair + stride);
  }
  bf8 xn[8];
  // EMB: the next pair's fp32 E rows are requested in two halves while the message product runs (the gathered rows and the embedded rows are
  // dead by then) and rounded to bf16 two blocks later: 32 registers of raw rows in flight at a time
  f4 xraw[8];
  auto raw_fetch = [&](const TileMeta& mt, const int h) {
    const float* src = a.hE + mt.erow * NAMP_H + 4 * hk + 64 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) { xraw[2 * s] = *(const f4*)(src + 16 * s); xraw[2 * s + 1] = *(const f4*)(src + 16 * s + 8); }
  };
  auto raw_pack = [&](const int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xn[4 * h + s] = pack_bf16<false>(xraw[2 * s], xraw[2 * s + 1]);
  };
  auto row_fetch = [&](const TileMeta& mt) {
    if constexpr (EMB) {
      raw_fetch(mt, 0); raw_pack(0); raw_fetch(mt, 1); raw_pack(1);             // (prologue only)
    } else {
#ifdef P32_NOMEM
      const bf8* src = (const bf8*)(a.hE16 + (mt.erow & 1023) * NAMP_H) + hk;
#else
      const bf8* src = (const bf8*)(a.hE16 + mt.erow * NAMP_H) + hk;
#endif
#pragma unroll
      for (int s = 0; s < 8; ++s) xn[s] = src[2 * s];
    }
  };
  row_fetch(cur);
  float* w_slot = (float*)(smem + 3 * NAMP_BIMG_BYTES + 2048 + wave * 128);
  // Pa (the residue's own first-layer term, one row per 16-row tile): plain loads, the lanes of a tile reading the same 16 bytes.  (Round 3's
  // edge update fetched it by LDS-DMA for want of registers: one such request in flight makes every later wait of the compiler a full drain of
  // the memory counter — the gathered rows were awaited with vmcnt(0) behind it.)  The edge update still STARTS its first-layer accumulators
  // from Pa and adds the gathered term behind the product, the messages add both behind it: round 3's rounding order, bit for bit.
  constexpr bool PA_INIT = EDGE;
  int par = 0;
  dma_to_lds(smem, a.W1_img, 32, wave, nwaves, lane);
  dma_to_lds(smem + NAMP_BIMG_BYTES, a.W2_img, 32, wave, nwaves, lane);
  if (EDGE) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.W3_img, 32, wave, nwaves, lane);
  if (EMB) dma_to_lds(smem + 2 * NAMP_BIMG_BYTES, a.eW1_img, 32, wave, nwaves, lane);
  float* cst = (float*)(smem + 3 * NAMP_BIMG_BYTES);          // b2 | b3 | LayerNorm-3 weight | bias   (EMB: b2 | b_e)
  if (EDGE && tid < 512) {
    const float* srcv = tid < 128 ? a.b2 : tid < 256 ? a.b3 : tid < 384 ? a.ln_g : a.ln_b;
    cst[tid] = srcv[tid & 127];
  }
  if (!EDGE && tid < 128) cst[tid] = a.b2[tid];
  if (EMB && tid >= 128 && tid < 256) cst[tid] = a.eb2[tid & 127];
  wait_dma_and_sync();
  const bf8* wimg = (const bf8*)smem + lane;                 // image of layer slot i: wimg + i * (NAMP_BIMG_BYTES / 16)
  auto vec16 = [&](const float* v, const int tn) {
    f16v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 t = *(const f4*)(v + 32 * tn + 8 * q + 4 * hk);
      o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
  };
  typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef P32_STAMPS
  unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (; pair < npairs; pair += stride) {
    asm volatile("" ::: "memory");
    P32_STAMP(0);
    const TileMeta me = cur;
    bf8 xb[8], a1[8], a2[8];
    if constexpr (!EMB) {
#pragma unroll
      for (int s = 0; s < 8; ++s) xb[s] = xn[s];
    }
#ifdef P32_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_sum[10] += 1;
#endif
    P32_STAMP(1);
    bf8 pj[8], pa[8];
    auto gather_fetch = [&]() {
      const bf8* src = (const bf8*)((me.pj_from1 ? a.Pj116 : a.Pj016) + me.pj_row * NAMP_H) + hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) pj[s] = src[2 * s];
      const bf8* srca = (const bf8*)(a.Pa16 + me.pa_row * NAMP_H) + hk;
#pragma unroll
      for (int s = 0; s < 8; ++s) pa[s] = srca[2 * s];
    };
    if constexpr (!EMB) gather_fetch();               // (EMB: behind the embedding product — its registers are the fp32 rows' until then)
    f16v A[NB];
    // initial value of block b: its LDS reads are issued in front of an MFMA of the block before (init_issue), what has to be computed from
    // them behind a later one (init_finish) — never a wait for LDS right behind the request
    float ini_b = 0.f;
    auto init_issue = [&](auto Bc) {
      constexpr int b = decltype(Bc)::value, li = b / 4, tn = b % 4;
      if constexpr (li == L1) {
        // edge update: from Pa (init_finish); message modes: zero (the first MFMA takes a zero C operand)
      } else if constexpr (li == LE) {
        A[b] = vec16(cst + 128, tn);
      } else if constexpr (li == L2) {
        if constexpr (EDGE) A[b] = vec16(cst, tn);
        else ini_b = cst[32 * tn + r];
      } else {
        A[b] = vec16(cst + 128, tn);
      }
    };
    auto init_finish = [&](auto Bc) {
      constexpr int b = decltype(Bc)::value, li = b / 4, tn = b % 4;
      if constexpr (li == L1 && PA_INIT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { A[b][j] = (float)pa[2 * tn][j]; A[b][8 + j] = (float)pa[2 * tn + 1][j]; }
      } else if constexpr (li == L2 && !EDGE) {
#pragma unroll
        for (int v = 0; v < 16; ++v) A[b][v] = ini_b;
      }
    };
    init_issue(std::integral_constant<int, 0>{});
    init_finish(std::integral_constant<int, 0>{});
    if (!EDGE && hk == 0) w_slot[r] = me.w_row;
    TileMeta nxt;
    MetaRaw nraw = {1, 1};
    bool nok = true;
    int idx_cur = idx_n1;
    // K-sum state of the message modes (F orientation: lane = channel 32tn + r, rows 8(v>>2) + 4hk + (v&3))
    f2 wp[8];
    float ksx = 0.f, ktx = 0.f, ksy = 0.f, kty = 0.f;
    f2 sA = (f2){0.f, 0.f}, sB = (f2){0.f, 0.f};             // edge update: LayerNorm sums
    f2 sQA = (f2){0.f, 0.f}, sQB = (f2){0.f, 0.f};           // ... and sums of squares (one-pass form)
    float* kdst = nullptr;
    const bool okB = 2 * pair + 1 < ntiles;
    int nodeA = 0, ktA = 0, nodeB = 0, ktB = 0;
    // one quad (accumulator elements 4q .. 4q+3) of the epilogue of block b
    auto epilogue = [&](auto Bc, auto Qc) {
      constexpr int b = decltype(Bc)::value, q = decltype(Qc)::value, li = b / 4, tn = b % 4, u = q >> 1, h = q & 1;
      f4 z = (f4){A[b][4 * q], A[b][4 * q + 1], A[b][4 * q + 2], A[b][4 * q + 3]};
      if constexpr (li == LE) {
        set_quad(xb[2 * tn + u], h, z);
        if constexpr (h == 1) {
          bf8* dst = (bf8*)(me.valid ? a.hE16_out + me.erow * NAMP_H : g_bf16s32_dump16 + m * NAMP_H) + hk;
          dst[2 * (2 * tn + u)] = xb[2 * tn + u];
        }
      } else if constexpr (li == L1) {
        if constexpr (PA_INIT) z += bf8_quad(pj[2 * tn + u], h);
        else z += bf8_quad(pa[2 * tn + u], h) + bf8_quad(pj[2 * tn + u], h);
        set_quad(a1[2 * tn + u], h, gelu4_bf16mode(z));
      } else if constexpr (li == L2 && EDGE) {
        set_quad(a2[2 * tn + u], h, gelu4_bf16mode(z));
      } else if constexpr (li == L2) {
        // message modes: weighted K-sum of the block's channel over its 16 + 16 rows; quads 0, 1 = first 16-row tile, 2, 3 = second
        const f4 g = gelu4_bf16mode(z);
        if constexpr (q == 0) { ksx = 0.f; ktx = 0.f; ksy = 0.f; kty = 0.f; }
        if constexpr (u == 0) {
          ksx = fmaf(g.x, wp[4 * h].x, ksx); ktx = fmaf(g.y, wp[4 * h + 1].x, ktx);
          ksx = fmaf(g.z, wp[4 * h + 2].x, ksx); ktx = fmaf(g.w, wp[4 * h + 3].x, ktx);
        } else {
          ksy = fmaf(g.x, wp[4 * h].y, ksy); kty = fmaf(g.y, wp[4 * h + 1].y, kty);
          ksy = fmaf(g.z, wp[4 * h + 2].y, ksy); kty = fmaf(g.w, wp[4 * h + 3].y, kty);
        }
        if constexpr (q == 3) {
          const float s0 = ksx + ktx, s1 = ksy + kty;
          const float t = __shfl_xor(hk ? s0 : s1, 32);
          kdst[32 * tn] = (hk ? s1 : s0) + t;
        }
      } else {
        // edge update, layer 3: residual and the LayerNorm sums
        z += bf8_quad(xb[2 * tn + u], h);
        A[b][4 * q] = z.x; A[b][4 * q + 1] = z.y; A[b][4 * q + 2] = z.z; A[b][4 * q + 3] = z.w;
        sA += (f2){z.x, z.y}; sB += (f2){z.z, z.w};
        if constexpr (!LN2P) {
          sQA = __builtin_elementwise_fma((f2){z.x, z.y}, (f2){z.x, z.y}, sQA);
          sQB = __builtin_elementwise_fma((f2){z.z, z.w}, (f2){z.z, z.w}, sQB);
        }
      }
    };
    // ---- the slots
    bf8 wf[RING];
    auto wfrag = [&](auto Gc) {
      constexpr int g = decltype(Gc)::value, b = g / 8, s = g % 8, li = b / 4, tn = b % 4;
      constexpr int slot = li == LE ? 2 : li == L1 ? 0 : li == L2 ? 1 : 2;
#ifdef P32_NOLDSW
      wf[g % RING] = xn[g % 8];
#else
      wf[g % RING] = wimg[slot * (NAMP_BIMG_BYTES / 16) + (s * 4 + tn) * 64];
#endif
    };
    P32_STAMP(2);
    static_for<0, D>([&](auto Gc) { wfrag(Gc); });
    static_for<0, NSLOT>([&](auto Gc) {
      constexpr int g = decltype(Gc)::value, b = g / 8, s = g % 8, li = b / 4, tn = b % 4;
      constexpr bool FLIP = !EDGE && li == L2;
      if constexpr (g > 0 && g % (NSLOT / 4) == 0) P32_STAMP(2 + g / (NSLOT / 4));
      if constexpr (g + D < NSLOT) wfrag(std::integral_constant<int, g + D>{});
      if constexpr (s == 5 && b + 1 < NB) init_issue(std::integral_constant<int, b + 1>{});
      P32_SB();
      {
        bf8 ab;
        if constexpr (li == LE) ab = xn[s];
        else if constexpr (li == L1) ab = xb[s];
        else if constexpr (li == L2) ab = a1[s];
        else ab = a2[s];
        constexpr bool ZERO = (li == L1 && !PA_INIT && s == 0);
        f16v c = A[b];
        if constexpr (ZERO) {
#pragma unroll
          for (int v = 0; v < 16; ++v) c[v] = 0.f;
        }
#ifdef P32_NOMFMA
        asm volatile("" :: "v"(wf[g % RING]), "v"(ab));
        A[b] = c;
#else
        A[b] = FLIP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, wf[g % RING], c, 0, 0, 0)
                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g % RING], ab, c, 0, 0, 0);
#endif
      }
      P32_SB();
      if constexpr (b >= 1 && s >= 1 && s <= 4) epilogue(std::integral_constant<int, b - 1>{}, std::integral_constant<int, s - 1>{});
      // the next pair's rows: behind the last use of this pair's rows as an operand (message modes) / behind layer 2 (edge update: their
      // 32 registers are free once a1 is dead)
      if constexpr (s == 7 && b + 1 < NB) init_finish(std::integral_constant<int, b + 1>{});
      if constexpr (s == 6 && b == 0) {
        // the next pair: tile arithmetic, mask / rank requests, its Pa rows (edge update); the neighbour ids of the pair after it
        idx_cur = idx_n1;
        nraw = meta_issue(pair_c(pair + stride), idx_cur, nxt, nok);
        idx_n1 = idx_of(pair + 2 * stride);
      }
      if constexpr (s == 6 && b == NB - 1) meta_finish(nxt, nraw, nok, idx_cur);
      if constexpr (!EMB && s == 6 && b == (EDGE ? 8 : 4)) row_fetch(nxt);
      if constexpr (EMB && s == 6 && b == 3) gather_fetch();
      if constexpr (EMB && s == 6 && b == 7) raw_fetch(nxt, 0);
      if constexpr (EMB && s == 6 && b == 9) { raw_pack(0); raw_fetch(nxt, 1); }
      if constexpr (!EDGE && b == 4 * L2 && s == 0) {
        // message modes: row weights of the two 16-row tiles as (first, second) pairs; where the K-sums go
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const f4 wa = *(const f4*)(w_slot + 8 * q2 + 4 * hk), wb = *(const f4*)(w_slot + 8 * (2 + q2) + 4 * hk);
          wp[4 * q2] = (f2){wa.x, wb.x}; wp[4 * q2 + 1] = (f2){wa.y, wb.y}; wp[4 * q2 + 2] = (f2){wa.z, wb.z}; wp[4 * q2 + 3] = (f2){wa.w, wb.w};
        }
        nodeA = __shfl(me.node, 0); ktA = __shfl(me.kt, 0); nodeB = __shfl(me.node, 16); ktB = __shfl(me.kt, 16);
        const int node_h = hk ? nodeB : nodeA, kt_h = hk ? ktB : ktA;
        kdst = (hk == 0 || okB) ? a.partial + ((long)node_h * a.TPN + kt_h) * NAMP_H + r : g_bf16s32_dump + r;
      }
      P32_SB();
    });
    // the last block's epilogue has no MFMAs left to ride
    P32_STAMP(6);
    static_for<0, 4>([&](auto Qc) { epilogue(std::integral_constant<int, NB - 1>{}, Qc); });
    if constexpr (EMB) raw_pack(1);
    P32_STAMP(7);
    if constexpr (EDGE) {
      f16v* acc = &A[4 * L3];
      float sum = (sA.x + sA.y) + (sB.x + sB.y);
      if constexpr (LN2P) {
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 128.0f);
        const f2 m2 = (f2){mean, mean};
        f2 qA = (f2){0.f, 0.f}, qB = (f2){0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
          for (int v = 0; v < 16; v += 4) {
            const f2 d0 = (f2){acc[tn][v], acc[tn][v + 1]} - m2, d1 = (f2){acc[tn][v + 2], acc[tn][v + 3]} - m2;
            acc[tn][v] = d0.x; acc[tn][v + 1] = d0.y; acc[tn][v + 2] = d1.x; acc[tn][v + 3] = d1.y;
            qA = __builtin_elementwise_fma(d0, d0, qA); qB = __builtin_elementwise_fma(d1, d1, qB);
          }
        float sq = (qA.x + qA.y) + (qB.x + qB.y);
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          const f16v ga = vec16(cst + 256, tn), be = vec16(cst + 384, tn);
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const f2 o = __builtin_elementwise_fma((f2){acc[tn][v], acc[tn][v + 1]} * (f2){rstd, rstd}, (f2){ga[v], ga[v + 1]}, (f2){be[v], be[v + 1]});
            acc[tn][v] = o.x; acc[tn][v + 1] = o.y;
          }
        }
      } else {
        float sq = (sQA.x + sQA.y) + (sQB.x + sQB.y);
        sum += __shfl_xor(sum, 32);
        sq += __shfl_xor(sq, 32);
        const float mean = sum * (1.0f / 128.0f);
        const float var = fmaxf(fmaf(-mean, mean, sq * (1.0f / 128.0f)), 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
        const f2 r2 = (f2){rstd, rstd}, n2 = (f2){-mean * rstd, -mean * rstd};
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          const f16v ga = vec16(cst + 256, tn), be = vec16(cst + 384, tn);
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const f2 t_ = __builtin_elementwise_fma((f2){acc[tn][v], acc[tn][v + 1]}, r2, n2);
            const f2 o = __builtin_elementwise_fma(t_, (f2){ga[v], ga[v + 1]}, (f2){be[v], be[v + 1]});
            acc[tn][v] = o.x; acc[tn][v + 1] = o.y;
          }
        }
      }
      P32_STAMP(8);
#ifdef P32_NOMEM
      bf8* dst = (bf8*)(g_bf16s32_dump16 + m * NAMP_H) + hk;
#else
      bf8* dst = (bf8*)(me.valid ? a.hE16_out + me.erow * NAMP_H : g_bf16s32_dump16 + m * NAMP_H) + hk;
#endif
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int tin = s >> 1, u = s & 1;
        dst[2 * s] = pack_bf16<false>((f4){acc[tin][8 * u], acc[tin][8 * u + 1], acc[tin][8 * u + 2], acc[tin][8 * u + 3]},
                                      (f4){acc[tin][8 * u + 4], acc[tin][8 * u + 5], acc[tin][8 * u + 6], acc[tin][8 * u + 7]});
      }
    } else {
      float wsum = me.w_row;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) wsum += __shfl_xor(wsum, o);
      float* wdst = lane == 0 ? a.partial + (long)a.G * a.TPN * NAMP_H + (long)nodeA * a.TPN + ktA
                  : (lane == 16 && okB) ? a.partial + (long)a.G * a.TPN * NAMP_H + (long)nodeB * a.TPN + ktB : g_bf16s32_dump + 128 + lane;
      *wdst = wsum;
    }
    P32_STAMP(9);
    cur = nxt;
    par ^= 1;
  }
#ifdef P32_STAMPS
  if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) g_p32_stamps[MODE + (EMB ? 3 : 0)][wave][i] = st_sum[i];
  }
#endif
}
#ifdef P32_STAMPS
extern "C" int namp_debug_p32_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p32_stamps), sizeof(g_p32_stamps), 0, hipMemcpyDeviceToHost);
}
#endif
