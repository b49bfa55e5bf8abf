/*
 * namp.h — C ABI of libnamp_hip.so: the MI355X (gfx950) implementation of the NA-MPNN
 * message-passing encoder/decoder hot path.
 *
 * The reference (baker-laboratory/NA-MPNN) has no FFI of its own: the path lives in Python
 * (inference/model_utils.py, na_model_utils.py) and runs on stock ATen kernels.  This header
 * is therefore the boundary a maintainer would bind with ctypes; every
 * entry point cites the reference function it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / HIP types.  `stream` is a hipStream_t passed as void*.
 *   - every pointer is a DEVICE pointer owned by the caller, 16-byte aligned, dense row-major.
 *   - hidden width is fixed at H = 128 (the only width the reference instantiates).
 *   - index tensors are int32 (the wrapper converts the reference's int64 once per forward).
 *   - calls are asynchronous on `stream`, re-entrant across streams, hold no device memory.
 *   - return 0 on success, negative NAMP_E* on error; text via namp_last_error() (thread local).
 *   - a "weight image" is the 64 KiB MFMA-fragment permutation of a [128 x 128] block of an
 *     nn.Linear weight ([out,in] row-major) produced by namp_pack_image().
 */
#ifndef NAMP_H_
#define NAMP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAMP_ABI_VERSION 4   /* 2: message phases write K-sums + weight sums; namp_node_update takes the message MLP's W3 / b3; 3: namp_train_edge_bwd takes g_hE_in; 4: *_simg images in the weight structs; symmetry groups in the level decoders */
#define NAMP_HIDDEN 128
#define NAMP_MAX_LAYERS 8
#define NAMP_MAX_K 192

#define NAMP_OK 0
#define NAMP_EINVAL (-1)   /* bad argument (null / misaligned pointer, unsupported size) */
#define NAMP_ELAUNCH (-2)  /* HIP launch or runtime failure */
#define NAMP_EWORKSPACE (-3)

int namp_abi_version(void);
const char* namp_last_error(void);

/* ---- weights ------------------------------------------------------------------------- */

/* Permute the block W[0:out_f, col0:col0+in_f] of an nn.Linear weight (leading dimension ld)
 * into fragment-image order.  out_f, in_f multiples of 16.  img: out_f*in_f floats. */
int namp_pack_image(const float* W, int ld, int col0, int out_f, int in_f, float* img, void* stream);

/* Precision of the per-edge message / edge-update GEMMs.  flags = 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32), the
 * arithmetic BASELINE configs[1] names; the Python surface's default is NAMP_FLAG_X3 below (fp32-equivalent to 2^-16).
 * NAMP_FLAG_BF16: inputs and weights rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16) — BASELINE
 * configs[2]'s throughput mode; ~1e-2 on log-probs, not parity-grade.  Residue-level math stays fp32. */
#define NAMP_FLAG_BF16 1
int namp_pack_image_bf16(const float* W, int ld, int col0, void* img, void* stream);   /* [128x128] block -> 32 KiB */
/* ... for v_mfma_f32_32x32x16_bf16: img[s' 0..7][tn 0..3][lane][j] = bf16(W[32tn + (lane&31)][16s' + 8(j>>2) + 4(lane>>5) + (j&3)]) */
int namp_pack_image_bf16_32(const float* W, int ld, int col0, void* img, void* stream);
/* The message launch of the bf16-STORAGE path on its own (namp_encdec_fwd takes that path for large batches of the bf16 mode: h_E and the
 * gathered first-layer tables are kept as rows of 128 bf16 in "fragment order B", channel c at element 16(c>>4) + 8((c>>2)&1) + 4((c>>3)&1) + (c&3)
 * — the operand order of v_mfma_f32_32x32x16_bf16; images from namp_pack_image_bf16_32).  mode 0 = EncLayer message (model_utils.py:666-672),
 * 1 = DecLayer message on the implicit context (model_utils.py:636-646).  Output as namp_enc_message / namp_dec_message:
 * partial [G][ceil(K/16)][128] K-sums of the layer-2 activations + [G][ceil(K/16)] weight sums. */
int namp_bf16s_message(int mode, const void* hE16, const int32_t* E_idx, const int32_t* mask, const int32_t* rank,
                       const void* Pa16, const void* Pj016, const void* Pj116, const void* W1_img, const void* W2_img, const float* b2,
                       float* partial, int B_dec, int B_enc, int N, int K, void* stream);
/* NAMP_FLAG_X3: the per-edge GEMMs as THREE bf16 products of split operands (x = x_hi + x_mid, W = W_hi + W_mid;
 * W.x ~= W_hi.x_hi + W_hi.x_mid + W_mid.x_hi, fp32 accumulate): fp32-equivalent to ~2^-16 per product at 3/16 of the fp32
 * MFMA cost — the default PARITY mode of the Python surface (log-probs move by 3e-5 against exact fp32 on the N=1000
 * golden, bar 1e-3, arg-max unchanged).  The x3 image = bf16 fragment image of W_hi followed by that of W_mid (64 KiB). */
#define NAMP_FLAG_X3 2
int namp_pack_image_x3(const float* W, int ld, int col0, void* img, void* stream);     /* [128x128] block -> 64 KiB */
/* x3 image of a general block [out_f x in_f] (out_f % 16 == 0, in_f % 32 == 0): the residue-level FFN weights
 * (PositionWiseFeedForward W_in [512 x 128], W_out [128 x 512], model_utils.py:595-604); out_f*in_f floats of output. */
int namp_pack_image_x3_general(const float* W, int ld, int col0, int out_f, int in_f, void* img, void* stream);
/* Many weight images in ONE launch (round 5; the training step packs ~116 per step).  `table_dev` is a DEVICE array of ndesc descriptors sorted by
 * first_block: kind 1 = x3 image of a [128 x 128] block (namp_pack_image_x3), 2 = bf16 image of a [128 x 128] block (namp_pack_image_bf16), 3 = x3
 * image of a general [out_f x in_f] block (namp_pack_image_x3_general); transposed != 0 packs the image of the block's TRANSPOSE from the same
 * storage (element (n, k) = W[k * ld + n]; out_f / in_f then describe the transposed block).  Descriptor i owns workgroups [first_block_i,
 * first_block_i + ceil(out_f * in_f / 256)); nblocks = their total. */
typedef struct NampPack { const float* W; void* img; int ld, out_f, in_f, kind, transposed, first_block; } NampPack;
int namp_pack_images(const NampPack* table_dev, int ndesc, int nblocks, void* stream);
/* The featuriser's edge_embedding.weight [128 x 5200] (model_utils.py:484) for the split-bf16 form of its GEMM: positional
 * k-tile as an fp32 fragment tile, then one 48 KiB hi|mid block per group of 6 atom pairs; 128*5200 floats in all, the
 * size of namp_pack_image's output for the same matrix. */
int namp_pack_feat_x3(const float* W, int ld, float* img, void* stream);

/* EncLayer parameters (inference/model_utils.py:659-679).  W1/W11 are split by input block:
 * a = h_V_i columns [0,128), b = h_E_ik [128,256), c = h_V_j [256,384). */
typedef struct NampEncLayerW {
  const float *W1a_img, *W1b_img, *W1c_img, *b1;
  const float *W2_img, *b2, *W3_img, *b3;
  const float *W11a_img, *W11b_img, *W11c_img, *b11;
  const float *W12_img, *b12, *W13_img, *b13;
  const float *Win_img, *b_in, *Wout_img, *b_out;   /* dense.W_in [512x128], dense.W_out [128x512] */
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  /* bf16 throughput mode: 32 KiB bf16 images of the six per-edge blocks (namp_pack_image_bf16) */
  const float *W1b_bimg, *W2_bimg, *W3_bimg, *W11b_bimg, *W12_bimg, *W13_bimg;
  const float *W1b_ximg, *W2_ximg, *W3_ximg, *W11b_ximg, *W12_ximg, *W13_ximg;   /* x3 images (namp_pack_image_x3) */
  /* optional: x3 images of the residue-level blocks (namp_pack_image_x3_general for W_in / W_out).  When all that a launch
   * needs are present, large batches (>= 64 residues per CU) run the residue update as split-bf16 products too. */
  const float *Win_ximg, *Wout_ximg, *W1a_ximg, *W1c_ximg, *W11a_ximg, *W11c_ximg;
  /* optional: bf16 images in the 32x32x16 operand order (namp_pack_image_bf16_32) of the blocks the bf16-STORAGE launches multiply
   * (large batches of the bf16 mode, namp_encdec_fwd): without them those batches take the fp32-storage launches. */
  const float *W1b_simg, *W2_simg, *W11b_simg, *W12_simg, *W13_simg;
  int64_t flags;                                     /* NAMP_FLAG_BF16 / NAMP_FLAG_X3: precision of the per-edge GEMMs */
} NampEncLayerW;

/* DecLayer parameters (inference/model_utils.py:619-634).  W1 [128x512] split by input block:
 * a = h_V_i [0,128), e = h_E_ik [128,256), s = h_S_j [256,384), v = h_V_j [384,512).
 * tok = W_s.weight . W1s^T  ([vocab x 128]), the per-token image of the sequence embedding. */
typedef struct NampDecLayerW {
  const float *W1a_img, *W1e_img, *W1s_img, *W1v_img, *b1, *tok;
  const float *W2_img, *b2, *W3_img, *b3;
  const float *Win_img, *b_in, *Wout_img, *b_out;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *W1e_bimg, *W2_bimg, *W3_bimg;        /* bf16 images (throughput mode) */
  const float *W1e_ximg, *W2_ximg, *W3_ximg;        /* x3 images */
  const float *Win_ximg, *Wout_ximg, *W1a_ximg, *W1v_ximg;   /* optional residue-level x3 images, as in NampEncLayerW */
  const float *W1e_simg, *W2_simg;                            /* optional 32x32x16-order bf16 images, as in NampEncLayerW */
  int64_t flags;
} NampDecLayerW;

/* ProteinFeaturesNA parameters (inference/model_utils.py:470-486): edge_embedding.weight [128 x 5200] as a
 * 325-k-tile fragment image, embeddings.linear [16 x 66] + bias in plain layout, norm_edges. */
typedef struct NampFeatW {
  const float *Wedge_img, *pos_w, *pos_b, *ln_g, *ln_b;
  const float *Wedge_ximg;         /* optional (namp_pack_feat_x3): when set, the 5184 RBF columns of the embedding GEMM run as
                                      split-bf16 products (and a requested h_E uses We_ximg); Wedge_img may then be NULL */
} NampFeatW;

typedef struct NampModelW {
  int32_t n_enc, n_dec, vocab;
  int32_t reserved;                /* 0; 2 = namp_featurize evaluates its GEMM as plain bf16 products on the x3 image's hi half
                                      (mixed-precision training, feat.Wedge_ximg set) */
  const float *Wv_img, *Wv_b;      /* W_v (model_utils.py:35,88) */
  const float *We_img, *We_b;      /* W_e (model_utils.py:48,89) */
  const float *Wout_w, *Wout_b;    /* W_out [vocab x 128] plain layout (model_utils.py:65) */
  NampEncLayerW enc[NAMP_MAX_LAYERS];
  NampDecLayerW dec[NAMP_MAX_LAYERS];
  NampFeatW feat;
  const float* We_ximg;            /* x3 image of W_e (the embedding fused in front of EncLayer 0, namp_encdec_fwd) */
  const float* Wv_ximg;            /* optional x3 image of W_v (with enc[0].W1a_ximg / W1c_ximg: namp_encdec_fwd's first launch as split-bf16 products) */
  const float* We_bimg;            /* optional bf16 image of W_e: the edge embedding of the bf16 throughput mode on bf16 MFMA */
  const float* We_simg;            /* optional 32x32x16-order bf16 image of W_e: the embedding inside the first bf16-storage launch */
} NampModelW;

/* ---- a1/a3: neighbour gather ----------------------------------------------------------- */

/* gather_nodes (model_utils.py:713-721): out[b,i,k,:] = nodes[b, idx[b,i,k], :]   (C floats) */
int namp_gather_nodes_f32(const float* nodes, const int32_t* idx, float* out,
                          int B, int N, int K, int C, void* stream);
/* The general form: T tables of R rows, M lookups per table: out[t,m,:] = tables[t, idx[t,m], :].  gather_nodes is
 * (T=B, R=N, M=N*K), gather_edges (T=B*N, R=N, M=K), gather_nodes_t (model_utils.py:723-727) (T=B, R=N, M=K). */
int namp_gather_rows_f32(const float* tables, const int32_t* idx, float* out, long T, int R, int M, int C, void* stream);
/* gather_edges (model_utils.py:707-711): out[b,i,k,:] = edges[b,i,idx[b,i,k],:]   (edges [B,N,N,C]) */
int namp_gather_edges_f32(const float* edges, const int32_t* idx, float* out, int B, int N, int K, int C, void* stream);
/* cat_neighbors_nodes (model_utils.py:729-732): out = [h_neighbors (C1) | h_nodes[idx] (C2)] */
int namp_cat_neighbors_nodes_f32(const float* h_nodes, const float* h_neighbors, const int32_t* idx,
                                 float* out, int B, int N, int K, int C1, int C2, void* stream);

/* ---- building blocks (also what bench.py times per kernel) ----------------------------- */

/* out_p = X . W_p^T + bias_p (+ tok_p[S]) for nproj <= 8 blocks; X has G_src = B_src*N rows and
 * is broadcast over output batches (row n reads batch (n / N) % B_src). */
typedef struct NampProj { const float* img; const float* bias; const float* tok; float* out; } NampProj;
/* pre (nullable, needs B_out == B_src): h = pre->img . X + pre->bias is computed first, stored to
 * pre->out, and the projections apply to h — W_v followed by the first layer's tables in one launch. */
int namp_node_linear(const float* X, const int32_t* S, int B_out, int B_src, int N,
                     const NampProj* proj, int nproj, const NampProj* pre, void* stream);

/* h_E = W_e . E + b_e (model_utils.py:89) */
int namp_edge_embed(const float* We_img, const float* We_b, const float* E, float* h_E,
                    int B, int N, int K, void* stream);

/* Message phase of EncLayer (model_utils.py:684-690).  The last layer of the message MLP is linear and is followed by a
 * weighted sum over the K neighbours, so it is applied AFTER that sum, once per residue, by the residue kernel:
 *     sum_k w_k (W3 a_k + b3) = W3 (sum_k w_k a_k) + b3 sum_k w_k,   a_k = gelu(W2.gelu(W1.[h_V_i|h_E_ik|h_V_j])), w_k = mask_i*mask_j/30.
 * partial holds, for T = ceil(K/16) 16-neighbour tiles per residue,  [B*N][T][128] floats  sum_{k in tile} w_k a_k  followed by
 * [B*N][T] floats  sum_{k in tile} w_k  — (B*N*T*129 floats in all); namp_node_update(..., partial, W3_img, b3, ...) finishes it.
 * Pa = W1a.h_V + b1 and Pc = W1c.h_V come from namp_node_linear.
 * mask_attend may be NULL (then mask_i*mask_j, as every reference call site passes). */
int namp_enc_message(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* mask,
                     const int32_t* mask_attend, const float* Pa, const float* Pc, float* partial,
                     int B, int N, int K, void* stream);
/* Edge update of EncLayer (model_utils.py:699-703): h_E' = LN3(h_E + MLP'([h_V'_i|h_E|h_V'_j])).
 * h_E_out may alias h_E. */
int namp_enc_edge_update(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx,
                         const float* Pa, const float* Pc, float* h_E_out, int B, int N, int K, void* stream);
/* Residue tail shared by EncLayer / DecLayer (model_utils.py:690-697, 646-656):
 * h_V' = mask * LN2(x + FFN(x)),  x = LN1(h_V + dh),
 *   dh = m3 . sum_t partial[n][t] + m3_b * sum_t wsum[n][t]   when m3_img (fp32 image of the message MLP's W3, + bias m3_b) is given
 *        (partial as written by namp_enc_message / namp_dec_message: K-sums followed by their weight sums),
 *   dh = sum_t partial[n][t]                                   when m3_img is NULL (partial [G][T][128] holds whole messages),
 * fused with nproj (0..8) projections of h_V' (as namp_node_linear) that the next edge kernels gather. */
int namp_node_update(const float* ln1_g, const float* ln1_b, const float* Win_img, const float* b_in,
                     const float* Wout_img, const float* b_out, const float* ln2_g, const float* ln2_b,
                     const float* h_V, const float* partial, const float* m3_img, const float* m3_b, const int32_t* mask,
                     float* h_V_out, const NampProj* proj, int nproj, const int32_t* S, int G, int K, void* stream);
/* Message phase of DecLayer on the implicit context h_ESV (model_utils.py:416-418, 640-646):
 * first layer = W1e.h_E_ik + Pa[i] + (rank[j]<rank[i] ? Pbw[j] : Pfw[j]).  Decoder batch b uses
 * encoder batch b % B_enc (the reference's .repeat(B_decoder, ...), model_utils.py:399-404). */
int namp_dec_message(const NampDecLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* rank,
                     const float* Pa, const float* Pbw, const float* Pfw, float* partial,
                     int B_dec, int B_enc, int N, int K, void* stream);
/* Fused forms: message phase + residue tail + projections in ONE launch (the workgroup that summed a
 * residue's messages also updates it).  Same results as namp_enc_message + namp_node_update resp.
 * namp_dec_message + namp_node_update; preferred while B*N <= namp_fused_tail_max_residues().
 * The projection outputs must not alias Pa / Pc / Pbw (other workgroups are still gathering those). */
int namp_enc_message_update(const NampEncLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* mask,
                            const int32_t* mask_attend, const float* Pa, const float* Pc, const float* h_V,
                            float* h_V_out, const NampProj* proj, int nproj, int B, int N, int K, void* stream);
/* EncLayer l-1's edge update (tables ePa / ePc = W11a.h_V + b11, W11c.h_V; h_E updated IN PLACE) fused in front of
 * EncLayer l's message phase + residue tail: the updated edge rows go from LayerNorm3 straight into the message MLP
 * in registers, so h_E is read once and one launch carries model_utils.py:699-703 of layer l-1 and :684-697 of layer l.
 * fp32 only.  Same results as namp_enc_edge_update(w_prev) followed by namp_enc_message_update(w). */
int namp_enc_edge_message_update(const NampEncLayerW* w_prev, const float* ePa, const float* ePc, float* h_E,
                                 const NampEncLayerW* w, const int32_t* E_idx, const int32_t* mask,
                                 const int32_t* mask_attend, const float* Pa, const float* Pc, const float* h_V,
                                 float* h_V_out, const NampProj* proj, int nproj, int B, int N, int K, void* stream);
int namp_dec_message_update(const NampDecLayerW* w, const float* h_E, const int32_t* E_idx, const int32_t* rank,
                            const float* Pa, const float* Pbw, const float* Pfw, const float* h_V, const int32_t* mask,
                            float* h_V_out, const NampProj* proj, int nproj, const int32_t* S,
                            const float* head_w, const float* head_b, float* log_probs, float* logits, int vocab,
                            int B_dec, int B_enc, int N, int K, void* stream);
/* head_w (nullable): when given, the launch also writes log_softmax(head_w . h_V' + head_b) — the W_out
 * head of model_utils.py:420-421 — for the residues it updates (logits optional). */
int namp_fused_tail_max_residues(void);
/* log_softmax(W_out . h_V + b) (model_utils.py:420-421); logits may be NULL. */
int namp_logits_log_softmax(const float* Wout_w, const float* Wout_b, const float* h_V,
                            float* log_probs, float* logits, int G, int vocab, void* stream);

/* ---- a4 / a7 / a8 / a10: layer- and model-level operators -------------------------------- */

size_t namp_workspace_bytes(int B_enc, int B_dec, int N, int K);

/* EncLayer.forward (model_utils.py:681-704) with dropout inactive.  h_V_out / h_E_out may alias
 * the inputs.  mask_attend NULL -> mask_i*mask_j. */
int namp_enc_layer_fwd(const NampEncLayerW* w, const float* h_V, const float* h_E, const int32_t* E_idx,
                       const int32_t* mask, const int32_t* mask_attend, float* h_V_out, float* h_E_out,
                       void* ws, size_t ws_bytes, int B, int N, int K, void* stream);

/* DecLayer.forward (model_utils.py:636-657) as an operator on a MATERIALISED context, dropout inactive:
 * h_ESV [B,N,K,384] = the reference's h_E argument ([h_E | h_S_j | h_V_j] in score() / sample(), model_utils.py:407-418);
 * mask_attend optional float [B,N,K]; mask_V optional [B,N].  Exact fp32 MFMA.  The model itself never calls this
 * (its decoder gathers the context implicitly: namp_decoder_fwd); it is the drop-in for callers that hold h_ESV.
 * ws: namp_workspace_bytes(B, B, N, K). */
int namp_dec_layer_fwd(const NampDecLayerW* w, const float* h_V, const float* h_ESV, const int32_t* mask_V,
                       const float* mask_attend, float* h_V_out, void* ws, size_t ws_bytes, int B, int N, int K,
                       void* stream);

/* namp_encdec_fwd can run a small batch (at most one workgroup per CU; K in 33..48; 3 + 3 layers; fp32-class precision) as
 * ONE persistent launch after the node_linear launch: h_E stays in registers from the edge embedding to the last DecLayer,
 * stages are separated by in-kernel grid barriers.  Results are bit-identical to the launch chain.  OFF by default — on
 * MI355X the barriers cost more than the launch boundaries they replace —; namp_set_persistent(1) or the
 * environment variable NAMP_PERSISTENT=1 selects it; returns the previous setting.  Two persistent launches are never in
 * flight on different streams (the second caller gets the chain).  Co-residency is NOT guarded against OTHER kernels holding
 * CUs on other streams: a workgroup that cannot become resident makes a bounded grid-barrier spin give up; the launch then
 * overwrites the log-probabilities of the affected workgroups' residues with NaN (the call itself has already returned
 * NAMP_OK), so a timeout cannot pass for a valid result.  Keep the device to this stream while the mode is on.
 * namp_persistent_status: synchronous read-back of the barrier state of the last persistent launch that used `ws`
 * (0 = every grid barrier completed; otherwise the code of the barrier that gave up — the outputs are then invalid). */
int namp_set_persistent(int on);
/* The bf16-storage edge launches of namp_encdec_fwd (large batches of the bf16 throughput mode) exist in two instruction sequencings
 * with bit-identical results: edge_mlp_bf16s32_kernel (round 3) and edge_mlp_bf16p_kernel (round 6; its embedding variant, bit 2, is the slower one and off by default).  mask: bit 0 the two
 * message launches, bit 1 the edge update, bit 2 the first encoder message with the fused edge embedding; bit 3 selects the round-6
 * residue update of that path and of the split-bf16 large-batch path (node_update_w_kernel<false / true>: same rounding points, another
 * summation order in the FFN's second product / of the three split products);
 * bit 4 (with bit 1) runs the edge update's LayerNorm with the round-3 kernel's two-pass variance instead of the one-pass sums (the
 * bit-equality test's form).  Bit 5 (independent of the precision mode): the edge-feature launch of namp_featurize in parts (see
 * namp_featurize_split_bytes); bits 6-7: one or two parts more than two (measured slower; A/B only).  Returns the previous mask.  Environment: NAMP_BF16P (default 43).  For A/B timing and the equality tests. */
int namp_set_bf16p(int mask);
int namp_persistent_status(const void* ws, size_t ws_bytes, int B, int N, int K, int32_t* code);

/* ---- a11: graph construction + edge features ------------------------------------------------
 * ProteinFeaturesNA.forward, eval mode (model_utils.py:528-593) without the node one-hot (a 6-row table lookup
 * the caller does): virtual atoms, kNN on CA + ref atom with the reference's masking (E_idx int32 [B,L,K],
 * K = min(top_k, L), ascending distance, ties by index), and E = LayerNorm(edge_embedding([positional | RBF])).
 * X [B,L,16,3] f32, X_m / masks / R_idx / chain_labels int32 [B,L(,16)]; atom order of run.py:15-19; ref_atom =
 * index of na_ref_atom (15 = C1').  E and/or h_E = W_e.E + b_e [B,L,K,128] are written (pass NULL to skip one). */
size_t namp_featurize_workspace_bytes(int B, int L);
/* Optional extra: with namp_featurize_workspace_bytes(B, L) + namp_featurize_split_bytes(B, L, top_k) bytes of workspace a batch of at most one
 * round of the chip (one complex of up to ~1,000 residues) runs its edge-feature launch in parts: the residue blocks that hold nucleotides
 * (39-54 atom-pair chunks against ~10 of a protein block: the launch lasts as long as its longest workgroup there) are walked by two
 * workgroups each, every second chunk, longest blocks first, and a finishing launch adds the partial rows, normalises and embeds.  Results
 * equal to ~1e-5 after the LayerNorm (another order of fp32 sums).  0 when the split does not apply.  When both E and h_E are requested the
 * two-part form needs no extra (partial rows go to the output buffers). */
size_t namp_featurize_split_bytes(int B, int L, int top_k);
int namp_featurize(const NampModelW* w, const float* X, const int32_t* X_m, const int32_t* mask, const int32_t* R_idx,
                   const int32_t* chain_labels, const int32_t* protein_mask, const int32_t* dna_mask,
                   const int32_t* rna_mask, int top_k, int ref_atom, int32_t* E_idx, float* E, float* h_E,
                   void* ws, size_t ws_bytes, int B, int L, void* stream);
/* namp_featurize and namp_decoding_order (below) in the same launches — what score() / forward() / sample() from coordinates call
 * (model_utils.py:389 next to :528-593): the B_order = B * batch_size sorts of (order_mask * order_chain_mask + 1e-4) * |randn| depend on
 * none of the features, so they run as the first workgroups of the edge-feature launch instead of a launch (or a side stream) of their
 * own.  order_mask / order_chain_mask [B, L] float (chain mask NULL = ones), randn [B_order, L]; outputs as namp_decoding_order. */
int namp_featurize_ordered(const NampModelW* w, const float* X, const int32_t* X_m, const int32_t* mask, const int32_t* R_idx,
                           const int32_t* chain_labels, const int32_t* protein_mask, const int32_t* dna_mask,
                           const int32_t* rna_mask, int top_k, int ref_atom, int32_t* E_idx, float* E, float* h_E,
                           void* ws, size_t ws_bytes, int B, int L, const float* order_mask, const float* order_chain_mask,
                           const float* randn, int64_t* order64, int32_t* order32, int32_t* rank32, int B_order, void* stream);

/* ProteinMPNN.encode after featurisation (model_utils.py:88-94): (V, E, E_idx, mask) -> h_V, h_E.
 * E may be NULL when h_E already holds W_e.E + b_e (written by namp_featurize). */
int namp_encoder_fwd(const NampModelW* w, const float* V, const float* E, const int32_t* E_idx,
                     const int32_t* mask, float* h_V, float* h_E,
                     void* ws, size_t ws_bytes, int B, int N, int K, void* stream);

/* Parallel decoder of score() / training forward() (model_utils.py:406-421 ==
 * na_model_utils.py:610-642): encoder outputs + S + decoding ranks -> log_probs [B_dec,N,vocab].
 * rank[b][i] = position of residue i in decoding order of decoder batch b; rank all-zero gives
 * unconditional_probs (model_utils.py:343-361: nothing is "backward").
 * S / mask / rank: [B_dec, N] int32.  logits may be NULL.  h_V_dec (optional, [B_dec*N,128])
 * receives the last decoder layer's h_V. */
int namp_decoder_fwd(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                     const int32_t* S, const int32_t* mask, const int32_t* rank,
                     float* log_probs, float* logits, float* h_V_dec,
                     void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream);

/* ProteinMPNN.score's device path (model_utils.py:88-94 + 406-421) in one call: (V, E, E_idx, mask, S, rank) -> h_V, h_E
 * (encoder outputs, as namp_encoder_fwd) and log_probs [B,N,vocab] (as namp_decoder_fwd with B_dec == B_enc).  While the
 * batch takes the fused residue tail (fp32, B*N <= namp_fused_tail_max_residues()) the encoder/decoder boundary is
 * fused as well (decoder tables projected by the last EncLayer's launch, last edge update in front of DecLayer 0's
 * message phase); otherwise it is namp_encoder_fwd + namp_decoder_fwd.  ws_bytes >= 2 * namp_workspace_bytes(B,B,N,K).
 * In the bf16 throughput mode on a batch beyond the fused-tail regime (with E given) h_E and the gathered tables are also
 * STORED in bf16 (fragment order) between launches — those launches are bound by HBM bytes, not MFMA — and the h_E buffer
 * serves as that store: it does not hold fp32 h_E afterwards (use namp_encoder_fwd when h_E itself is wanted). */
int namp_encdec_fwd(const NampModelW* w, const float* V, const float* E, const int32_t* E_idx, const int32_t* mask,
                    const int32_t* S, const int32_t* rank, float* h_V, float* h_E, float* log_probs, float* logits,
                    void* ws, size_t ws_bytes, int B, int N, int K, void* stream);

/* ---- a9: autoregressive sampler ------------------------------------------------------------
 * ProteinMPNN.sample (model_utils.py:101-327: plain branch :126-218, symmetry-tied branch :219-326), after encode(): B_dec independent sample
 * streams over B_enc encoded complexes (stream b uses complex b % B_enc), one persistent launch.
 *   mask        [B_enc,N]  the residue mask: a masked residue decodes from an all-zero context (mask_bw / mask_fw, :135-137)
 *   mask_dec    [B_dec,N]  OUTPUT mask of the residue update per stream (the caller may reproduce the reference's use of stream 0's
 *                          mask at every step, model_utils.py:186 — see na_mpnn_amd/model.py)
 *   chain_mask  [B_enc,N]  mask*chain_mask: 1 = design, 0 = keep S_true       S_true [B_enc,N]
 *   bias        [B_enc,N,vocab] added to the logits before the temperature softmax (model_utils.py:196)
 *   order, rank [B_dec,N]  decoding order and its inverse                       uniform [B_dec,N] in [0,1),
 *                          consumed one per step by an inverse-CDF draw (replaces torch.multinomial, :209)
 *   S_forced    [B_dec,N]  optional: use this token instead of the draw (teacher forcing)
 *   group_first / group_last [B_dec,N] (both or neither): symmetry-tied sampling (model_utils.py:219-326) — consecutive
 *                          visits of `order` form groups; the members' logits are summed with sym_weights [B_enc,N]
 *                          (NULL = 1) and ONE token is drawn per group.  group_first[v] = first visit of v's group,
 *                          group_last[v] = 1 on the visit that closes it.  NULL: every visit is its own group.
 *   pair_bias   [B_enc,N,vocab,N,vocab] optional (model_utils.py:116,170-172): adds sum_j pair_bias[i,:,j,S_j]
 *   special_tokens  bit t set = token t is never drawn (UNK, DX, RX, MAS, PAD: model_utils.py:199-203)
 * Outputs: S_out int32 [B_dec,N]; probs_out / logp_out [B_dec,N,vocab] = chain_mask * (sampling
 * distribution / log_softmax(logits)) as model_utils.py:211-212. */
/* Work lists of namp_decoder_sample_walk for the plain branch (every visit its own work item), built on the device from the levels of
 * namp_sample_levels_dep: work [B_dec * N][2] int32 = (stream, visit) grouped by level (the order inside a level is not defined: its items
 * are independent), level_off [N + 2] int32 = items of a level below l, n_levels[0] = number of non-empty levels.  N <= 16000. */
int namp_sample_work_lists(const int32_t* level, int32_t* work, int32_t* level_off, int32_t* n_levels, int B_dec, int N, void* stream);

/* Decoding order on the device: order[b] = argsort((mask[b % B_mask] * chain_mask[b % B_mask] + 1e-4) * |randn[b]|) (ascending; ties by
 * index; NaN keys last) and rank = its inverse permutation (model_utils.py:389-390, na_model_utils.py:623 — the reference's
 * torch.argsort + one-hot einsum), one launch.  mask / chain_mask [B_mask, L] float (chain_mask NULL = ones), randn [B, L] float; outputs
 * order64 [B, L] int64 and / or order32 [B, L] int32 (either may be NULL), rank32 [B, L] int32.  L <= 8192. */
int namp_decoding_order(const float* mask, const float* chain_mask, const float* randn, int64_t* order64, int32_t* order32, int32_t* rank32,
                        int B, int B_mask, int L, void* stream);

size_t namp_sample_workspace_bytes(int B_enc, int B_dec, int N, int K);
/* The same for a model of n_dec decoder layers (1 .. NAMP_MAX_LAYERS; 0 on bad arguments).  namp_sample_workspace_bytes sizes for
 * NAMP_MAX_LAYERS layers — n_dec h_E-sized first-layer tables are what a call carves, so a three-layer model needs well under half of it.
 * The level walk keeps its grid-barrier words in the last 4 KiB of the n_dec-sized region; a larger workspace is accepted. */
size_t namp_sample_workspace_bytes_n(int B_enc, int B_dec, int N, int K, int n_dec);
int namp_decoder_sample(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                        const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                        const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                        const int32_t* group_first, const int32_t* group_last, const float* sym_weights,
                        const float* pair_bias,
                        float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                        void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream);

/* ---- a12: training (backward) -------------------------------------------------------------
 * Gradient side of EncLayer / DecLayer / the edge featuriser (na_model_utils.py:196-283, 349-517, 589-646) for the
 * reference's training step (na_run.py:218-238).  The residue-level ops (LayerNorms, FFN, the hoisted first-layer
 * tables, logits, loss) are [B*N,128]-sized and stay with the caller's autograd; these entry points carry the
 * per-edge work.  Activations are recomputed in the backward pass, the reference's torch.utils.checkpoint policy
 * (na_model_utils.py:606,637).  mode: 0 = EncLayer message, 1 = DecLayer message, 2 = EncLayer edge update.
 * All images are fp32 fragment images (namp_pack_image) — or, with x3 != 0, x3 images (namp_pack_image_x3: the GEMMs then
 * run as split-bf16 products like the forward path's default mode); "t" images are those of the transposed blocks.
 *
 * namp_train_edge_fwd: the forward of one per-edge MLP from raw images.  mode 0/1: out = the K-sums of the layer-2 activations
 *   and their weight sums, [B*N][ceil(K/16)][128] + [B*N][ceil(K/16)] floats (as namp_enc_message / namp_dec_message;
 *   B_dec == B_enc; W3_img / b3 are not read: the caller applies layer 3 per residue, dh = W3 . sum_t S[t] + b3 sum_t w[t]);
 *   mode 2: out = per edge [B*N*K][128]
 *   either the bare message W13.gelu(W12.gelu(W11.[..])) + b13 (ln_g NULL, drop_p 0) or the finished edge update (below).
 * namp_train_edge_bwd: given g_out = dL/d(sum_k w_k a2_k) [B*N][128] (modes 0/1: the gradient of the K-sum the forward wrote,
 *   i.e. W3^T dL/d(dh); the 1/30 scale and mask_attend are applied inside; W3t_img, A2, G3, S3, w3 are not used and may be NULL)
 *   or dL/d(message) per edge [B*N*K][128] (mode 2), recompute the chain and write per edge row the activations
 *   A1 = gelu(z1), A2 = gelu(z2) (mode 2), the gradients G1 = dL/dz1, G2 = dL/dz2 (mode 2: G3 == g_out) and g_hE = dL/dh_E.
 *   Then dW2 = G2^T A1, dW1b = G1^T h_E (mode 2 also dW3 = G3^T A2) by namp_train_wgrad, db2 = sum G2;
 *   dL/dPa[i] = sum_k G1[i,k], dL/dPj[j] += G1[i,k] — the last two are accumulated by the
 *   launch itself with fp32 atomics into g_Pa / g_Pj0 (/ g_Pj1: rows that gathered Pfw) when those ZEROED [B*N][128]
 *   buffers are given (each optional; NULL: the caller reduces G1 — namp_train_scatter_rows does dL/dPj without atomics).
 *   The `x3` argument of the namp_train_* entry points is a precision code: 0 exact fp32 MFMA, 1 split-bf16 products, 2 plain
 *   bf16 products (mixed-precision training).  For namp_train_edge_bwd, adding 4 makes the launch ADD its dL/dh_E to the rows
 *   of g_hE_in (another consumer's gradient of the same h_E, [B*N*K][128]; read only) and write the sum to g_hE — g_hE_in may
 *   equal g_hE (in place) when the caller owns that buffer exclusively; without the flag g_hE_in is ignored; adding 8 (both backward
 *   entry points, K % 16 == 0) makes g_Pa a [B*N*K/16][128] buffer of per-tile sums written with plain stores — the caller adds
 *   a residue's K/16 tiles — instead of a zeroed [B*N][128] buffer accumulated with fp32 atomics (deterministic).
 * namp_train_wgrad: dW_part[c] = sum over row chunk c of G[row]^T (gelu_A ? gelu(A[row]) : A[row]), db_part[c] = sum G[row];
 *   c < namp_train_wgrad_chunks(rows); the caller adds the chunks.  dW_part [chunks][128][128], db_part [chunks][128] or NULL.
 * namp_train_feat_wgrad: gradient of features.edge_embedding.weight [128 x 5200] with the RBF features regenerated
 *   on the fly: X18 [B*L][18][3] (16 atoms + Cb + N_na), M18 [B*L][18] 0/1 floats, E_pos [B*L*K][16] the positional
 *   features, g_pre [B*L*K][128] = dL/d(pre-LayerNorm edge embedding); dW_part [namp_train_feat_wgrad_chunks][128][5200];
 *   x3 != 0: the contraction over edges as split-bf16 products (as namp_train_wgrad).
 * namp_featurize with w->feat.ln_g == NULL writes the pre-LayerNorm rows to E (the training forward). */
/* namp_edge_embed / namp_node_linear with the product evaluation of the training step's precision code (0 exact fp32 MFMA with fp32
 * images, 1 split-bf16 with x3 images, 2 plain bf16: bf16 image for the edge embedding, the hi plane of x3 images for the residue
 * products) — W_e (na_model_utils.py:598) and the hoisted first-layer tables of EncLayer / DecLayer (:218-236, 610-636) and their
 * data gradients inside the training step. */
int namp_edge_embed_prec(const float* We_img, const float* We_b, const float* E, float* h_E, int prec, int B, int N, int K,
                         void* stream);
/* out[n] = sum_{q < n} W_q . X_q[n] over [G][128] rows: the data gradient of several residue-level linear maps of one input in one launch (images as
 * namp_node_linear_prec takes them: fp32 fragment images for precision code 0, x3 images for 1 / 2). */
int namp_node_linear_sum(const float* const* X, const float* const* img, int n, float* out, int G, int prec, void* stream);
int namp_node_linear_prec(const float* X, int G, const NampProj* proj, int nproj, int prec, void* stream);
int namp_train_edge_fwd(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                        const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                        const float* W2_img, const float* W3_img, const float* b2, const float* b3,
                        const float* ln_g, const float* ln_b, float drop_p, uint32_t drop_seed, float* out,
                        int x3, int B, int N, int K, void* stream);
/* The whole EncLayer edge update with its tail (na_model_utils.py:236-240), forward = namp_train_edge_fwd(mode 2, ln_g, ln_b,
 * drop_p, drop_seed): out = LayerNorm3(h_E + dropout(message)); the dropout mask is a counter-based hash of (drop_seed, edge
 * row, channel), regenerated — not stored — by the backward launch.  Backward: g_out = dL/d(out) per edge row; recomputes
 * the message (6 GEMMs per row: 3 recompute + 3 data gradients), differentiates LayerNorm3 and the mask in registers and
 * writes A1, A2, G1, G2, G3 (for namp_train_wgrad), g_hE (chain + residual path), the table gradients (atomics, as
 * namp_train_edge_bwd) and per-workgroup partial sums dgb_part [namp_train_edge_update_bwd_groups][2][128] of
 * d(ln weight) = sum g*xhat and d(ln bias) = sum g.  A caller that walks the batch in slices of complexes (all pointers advanced to the slice, B =
 * its complexes: the row tensors then hold one slice at a time) passes the slice's first edge row as drop_row0, so that the mask is the forward's. */
int namp_train_edge_update_bwd_groups(int B, int N, int K);
int namp_train_edge_update_bwd(const float* h_E, const int32_t* E_idx, const float* Pa, const float* Pc, const float* W1_img,
                               const float* W2_img, const float* W3_img, const float* W3t_img, const float* W2t_img,
                               const float* W1t_img, const float* b2, const float* b3, const float* ln_g, float drop_p,
                               uint32_t drop_seed, long drop_row0, const float* g_out, float* A1, float* A2, float* G1, float* G2, float* G3,
                               float* g_hE, float* g_Pa, float* g_Pc, float* dgb_part, int x3, int B, int N, int K, void* stream);
/* The same backward as TWO persistent launches that own their weight gradients (round 5; mixed precision only: x3 & 3 == 2, bit 3 as above) — no
 * A1 / A2 / G3 rows and no row contractions behind it.  Launch A: recompute, LayerNorm3 + dropout backward (dL/dx rows parked in g_hE, d ln sums),
 * dW3 / db3 on chip, G2 rows (bf16 workspace).  Launch B: dW2 / db2 and dW1b on chip, dL/dh_E = dL/dx + W1b^T g1 (in place in g_hE), G1 rows (bf16,
 * for namp_train_scatter_rows_bf16), g_Pa.  Row buffers G2, G1 and g_hE hold namp_train_edge_bwd_dw_rows() rows, g_Pa one row per 16 of those
 * (bit 3) or [G][128] zeroed.  With groups = namp_train_edge_bwd_dw_groups(): dW_part = [groups][128][128] (dW3) followed by [groups][2][128][128]
 * (0 = dW2, 1 = dW1b); db_part = [groups][128] (db3) followed by [groups][128] (db2); dgb_part [groups][2][128] (sum g * xhat, sum g).
 * Replaces namp_train_edge_update_bwd + its three row contractions (na_model_utils.py:236-240). */
int namp_train_edge_update_bwd_dw(const float* h_E, const int32_t* E_idx, const float* Pa, const float* Pc, const float* W1_img,
                                  const float* W2_img, const float* W3_img, const float* W3t_img, const float* W2t_img,
                                  const float* W1t_img, const float* b2, const float* b3, const float* ln_g, float drop_p,
                                  uint32_t drop_seed, const float* g_out, float* G2, float* G1, float* g_hE, float* g_Pa,
                                  float* dW_part, float* db_part, float* dgb_part, int x3, int B, int N, int K, void* stream);
int namp_train_edge_bwd(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                        const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                        const float* W2_img, const float* W3t_img, const float* W2t_img, const float* W1t_img,
                        const float* b2, const float* g_out, float* A1, float* A2, float* G1, float* G2, float* G3,
                        float* g_hE, const float* g_hE_in, float* g_Pa, float* g_Pj0, float* g_Pj1, float* S3, float* w3, int x3, int B, int N,
                        int K, void* stream);
/* Message-stage backward that owns its weight gradients (round 4; modes 0 / 1, precision codes 1 / 2 with the same flag bits 4 and 8
 * as namp_train_edge_bwd): ONE persistent launch recomputes the chain, walks it backwards AND contracts the row operands over the
 * edges on chip — the A1 / G2 row tensors and the namp_train_wgrad launches of na_model_utils.py:196-283's W1 / W2 gradients
 * disappear.  Outputs: G1 rows (bf16 in precision code 2; for namp_train_scatter_rows), g_hE, g_Pa as namp_train_edge_bwd, and per
 * workgroup c < namp_train_edge_bwd_dw_groups(B, N, K): dW_part[c][0] = partial of dW2 = G2^T A1, dW_part[c][1] = partial of
 * dW1b = G1^T h_E ([128][128] each), db_part[c] = partial of db2 = sum G2 ([128]); the caller adds the partials. */
int namp_train_edge_bwd_dw_groups(int B, int N, int K);
/* Rows G1 and g_hE must have room for (g_Pa: one row per 16 of them when the per-tile flag is set): the launches store whole 64-row rounds,
 * rows past B*N*K into this padding (no vector-memory instruction of their round loop sits under a branch). */
long namp_train_edge_bwd_dw_rows(int B, int N, int K);
int namp_train_edge_bwd_dw(int mode, const float* h_E, const int32_t* E_idx, const int32_t* mask, const int32_t* mask_attend,
                           const int32_t* rank, const float* Pa, const float* Pj0, const float* Pj1, const float* W1_img,
                           const float* W2_img, const float* W2t_img, const float* W1t_img, const float* b2, const float* g_out,
                           float* G1, float* g_hE, const float* g_hE_in, float* g_Pa, float* dW_part, float* db_part, int x3, int B,
                           int N, int K, void* stream);
/* dL/dPj = transpose of the neighbour gather, as a gather over the reverse adjacency: rev_edge [B*N*K] = edge ids sorted by
 * the table row they gathered (global row b*N + E_idx), rev_off [B*N+1] their offsets per row; out0[j] = sum of G1[e] over the
 * edges of row j (sel[e] != 0 when sel is given; the others go to out1: DecLayer's Pbw / Pfw).  Deterministic. */
int namp_train_scatter_rows(const float* G1, const int32_t* rev_edge, const int32_t* rev_off, const uint8_t* sel,
                            float* out0, float* out1, int G, void* stream);
/* Mixed-precision mode (precision code 2): the backward entry points write A1, A2, G1, G2, G3 as bf16 [E][128] row tensors
 * (plain channel order; the float* parameters then point at bf16 storage), namp_train_wgrad takes them with bit 4 (G is
 * bf16; required) / bit 5 (A is bf16) added to its precision argument, and the table-gradient gather reads G1 here. */
/* n <= 8 row contractions over the SAME rows in one launch (fp32 row tensors; precision code 1 or 2): G[q], A[q], dW_part[q],
 * db_part[q] (may be NULL) are host arrays of device pointers, each pair as in namp_train_wgrad.  Bit 6 of x3: ADD to the partials an earlier launch
 * over other rows left there (every workgroup adds to its own chunk slot; the earlier launch must have at least as many chunks).  chunks: the
 * number of row chunks (= partials per contraction), 0 = namp_train_wgrad_chunks(rows); a caller that walks the rows in slices picks n * chunks
 * close to the 512 workgroups the chip holds at a time.  Rows past the last chunk's end are not read: chunks whose range is empty store zeros. */
int namp_train_wgrad_multi(const float* const* G, const float* const* A, int n, int x3, long rows, int chunks, float* const* dW_part,
                           float* const* db_part, void* stream);
int namp_train_scatter_rows_bf16(const void* G1, const int32_t* rev_edge, const int32_t* rev_off, const uint8_t* sel,
                                 float* out0, float* out1, int G, void* stream);
/* Residue tail of EncLayer / DecLayer in training (na_model_utils.py:236-247, 268-283), one launch each way:
 *     x1 = LayerNorm1(h_V + dropout1(dh));  z = W_in x1 + b_in;  y = x1 + dropout2(W_out gelu(z) + b_out);  out = mask * LayerNorm2(y)
 * (dh = the K-sum of the messages / 30; both dropouts are counter-based hashes of (seed, row, channel), regenerated in
 * backward; GEMMs as split-bf16 products).  Win_ximg / Wout_ximg: namp_pack_image_x3_general of W_in [512 x 128] / W_out
 * [128 x 512]; WoutT_ximg / WinT_ximg: the same of their transposes.  Forward keeps x1 [G][128], z [4][G][128] (hidden units in
 * four 128-wide blocks, block-major) and y [G][128] for the backward launch, which writes dL/dh_V, dL/d(dh), the row tensors
 * g_f [G][128], g_z and h = gelu(z) [4][G][128] — dW_out[:, 128q:] = g_f^T h_q, dW_in[128q:, :] = g_z_q^T x1, db_out = sum g_f,
 * db_in_q = sum g_z_q through namp_train_wgrad — and part [namp_train_tail_groups(G)][4][128]: per-workgroup sums of d ln2 weight,
 * d ln2 bias, d ln1 weight, d ln1 bias. */
int namp_train_tail_groups(int G);
int namp_train_tail_fwd(const float* h_V, const float* dh, const int32_t* mask, const float* ln1_g, const float* ln1_b,
                        const float* Win_ximg, const float* b_in, const float* Wout_ximg, const float* b_out, const float* ln2_g,
                        const float* ln2_b, float drop_p, uint32_t seed1, uint32_t seed2, float* out, float* x1, float* z, float* y,
                        int G, void* stream);
int namp_train_tail_bwd(const float* h_V, const float* dh, const int32_t* mask, const float* ln1_g, const float* ln2_g,
                        const float* WoutT_ximg, const float* WinT_ximg, float drop_p, uint32_t seed1, uint32_t seed2,
                        const float* x1, const float* z, const float* y, const float* g_out, float* g_hV, float* g_dh, float* g_f,
                        float* g_z, float* h, float* part, int G, void* stream);
/* LayerNorm over the 128 channels of [rows][128] (features.norm_edges on the edge embedding, na_model_utils.py:509) and its
 * backward: gx = dL/dx; dgb_part [namp_train_ln_rows_groups(rows)][2][128] = per-workgroup partial sums of d(weight), d(bias). */
int namp_train_ln_rows_groups(long rows);
int namp_train_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* out, long rows, void* stream);
int namp_train_ln_rows_bwd(const float* x, const float* g, const float* gamma, float* gx, float* dgb_part, long rows, void* stream);
int namp_train_wgrad_chunks(long rows);
int namp_train_wgrad(const float* G, const float* A, int gelu_A, int x3, long rows, float* dW_part, float* db_part, void* stream);
int namp_train_feat_wgrad_chunks(long edges);
long namp_train_feat_wgrad_ws_ints(long edges);       /* int32 elements of tile_ws (atom-presence words per 64-edge tile) */
/* (round 5) M18 == NULL: X18 is the PACKED atom array [B*L][18][4] = (x, y, z, mask) — one 16-byte request per gathered atom; split-bf16 / bf16 only. */
/* g16 (optional, with the packed atoms and precision code 1 / 2): g_pre once more as bf16 tiles [ceil(E / 64)][128 channels][64 edges] — split-bf16: the
 * remainders in a second array namp_train_g16_elems(E) elements behind — as namp_train_embed_ln_bwd writes them: the launch then stages its tiles by
 * 16-byte copies instead of converting and transposing the fp32 rows in every column group (half of its time at cfg5). */
long namp_train_g16_elems(long rows);
int namp_train_feat_wgrad(const float* X18, const float* M18, const int32_t* E_idx, const float* E_pos, const float* g_pre, const void* g16,
                          float* dW_part, int32_t* tile_ws, int x3, int B, int L, int K, void* stream);

/* ---- training: loss and optimiser step (round 3) ----------------------------------------------
 * namp_train_loss_smoothed: the per-residue label-smoothed loss of na_model_utils.py:111-146 in fp64 (backward = 0: loss[G]) and its
 *   gradient with respect to log_probs (backward = 1: g_log_probs[G][V] = -target * g_loss[G]); masks float32 [G], restype tables
 *   float32 [V] (0 / 1), eps_scale3 = HOST array {weight/21, weight/5, weight/5} as float32, ppm_mask int32 [G] + aligned_ppm fp64
 *   [G][V] optional (the specificity model's targets, :134).  The reduction sum(loss * mask) / tokens stays with the caller.
 * namp_train_adam_step: gradient clip (max_norm > 0: torch.nn.utils.clip_grad_norm_, scaled gradients written back) + one
 *   torch.optim.Adam step over ALL parameter tensors in one launch (three with clipping).  Plan arrays (device): blk_tensor /
 *   blk_off [nblocks] = the tensor and first element of each block of namp_train_adam_chunk() elements, numel [ntensors], ptrs
 *   [4][ntensors] = param / grad / exp_avg / exp_avg_sq addresses (fp32).  step_size = lr / (1 - beta1^t), bias_correction2_sqrt =
 *   sqrt(1 - beta2^t) (host doubles rounded to float, like torch).  ws: nblocks + 2 floats (ws[0] = gradient norm, ws[1] = clip
 *   coefficient afterwards); may be NULL without clipping. */
int namp_train_loss_smoothed(int backward, const int32_t* S, const float* log_probs, const float* protein_mask, const float* dna_mask,
                             const float* rna_mask, const float* protein_restypes, const float* dna_restypes, const float* rna_restypes,
                             const float* eps_scale3, double weight, const int32_t* ppm_mask, const double* aligned_ppm,
                             double* loss, const double* g_loss, float* g_log_probs, long G, int V, void* stream);
int namp_train_adam_chunk(void);
int namp_train_adam_step(const int32_t* blk_tensor, const long long* blk_off, const long long* numel, const unsigned long long* ptrs,
                         int ntensors, int nblocks, float max_norm, double beta1, double beta2, float step_size, float bias_correction2_sqrt,
                         float eps, float* ws, void* stream);

/* Positional edge features of the training step (PositionalEncodings, na_model_utils.py:537-541) as two launches (round 5):
 *   namp_train_pos_features: d_out[e] = chain_i == chain_j ? clip(R_i - R_j + 32, 0, 64) : 65 and E_pos[e][16] = pos_w[:, d] + pos_b for every edge
 *     (pos_w = embeddings.linear.weight [16][66], j = E_idx[e] within the complex) — the input of namp_train_feat_wgrad's positional block.
 *   namp_train_pos_grad: with gp[e][k] = sum_c g[e][c] * Wedge[c][k] (k < 16; Wedge row stride ld) per-workgroup partials part[groups][67][16]:
 *     rows 0..65 = sum of gp over the edges of class d (-> d pos_w^T), row 66 = sum over all edges (-> d pos_b).  groups =
 *     namp_train_pos_grad_groups(edges); add them with namp_reduce_sum.  Deterministic. */
int namp_train_pos_features(const int32_t* R_idx, const int32_t* chain, const int32_t* E_idx, const float* pos_w, const float* pos_b,
                            int32_t* d_out, float* E_pos, int B, int L, int K, void* stream);
int namp_train_pos_grad_groups(long edges);
int namp_train_pos_grad(const float* g, const float* Wedge, int ld, const int32_t* d, float* part, long edges, void* stream);

/* Reverse adjacency for namp_train_scatter_rows (round 5; four small launches instead of a stock 64-bit radix sort + bincount + cumsum): offsets
 * [B*N + 1] and edges [B*N*K] such that edges[offsets[j] .. offsets[j+1]) are the edge rows e = (b, i, k) with b*N + E_idx[b,i,k] == j in ascending
 * order (= a stable sort of the edges by target row).  ws: 2*B*N + B*N*K int32 of scratch. */
int namp_train_reverse_adjacency(const int32_t* E_idx, int32_t* offsets, int32_t* edges, int32_t* ws, int B, int N, int K, void* stream);

/* norm_edges + W_e of the training copy (na_model_utils.py:509,598) without the normalised rows in memory (round 5):
 *   namp_edge_embed_ln:       h_E = W_e . LayerNorm(Y) + b_e over the [B*N*K][128] PRE-LayerNorm rows Y (one launch; precision code as namp_edge_embed_prec);
 *   namp_train_embed_ln_bwd:  g_pre = dL/dY from g = dL/dh_E: the W_e^T product and the LayerNorm backward in one pass (Wt_img = image of W_e^T at the
 *                             precision code, 1 or 2); leaves (mean, rstd) per row in ln_stats [rows][2] and per-workgroup sums
 *                             dgb_part [namp_train_embed_ln_bwd_groups(rows)][2][128] of d(ln weight) = sum g_E xhat and d(ln bias) = sum g_E;
 *                             g16 (optional; the tail tile zeroed by the caller when rows % 64 != 0): g_pre also as the bf16 tiles of
 *                             namp_train_feat_wgrad (1 or 2 arrays of namp_train_g16_elems(rows) bf16 by the precision code);
 *   namp_train_wgrad_ln:      dW_e = sum_rows g^T LayerNorm(Y) (+ db_e = sum g) as namp_train_wgrad, LayerNorm(Y) re-derived from Y and ln_stats. */
int namp_edge_embed_ln(const float* We_img, const float* We_b, const float* ln_g, const float* ln_b, const float* Y, float* h_E, int prec,
                       int B, int N, int K, void* stream);
int namp_train_embed_ln_bwd_groups(long rows);
int namp_train_embed_ln_bwd(const float* g, const float* Y, const float* Wt_img, const float* ln_g, float* g_pre, float* ln_stats,
                            float* dgb_part, void* g16, int x3, long rows, void* stream);
int namp_train_wgrad_ln(const float* G, const float* Y, const float* ln_stats, const float* ln_g, const float* ln_b, int x3, long rows,
                        float* dW_part, float* db_part, void* stream);

/* Two residue-level reductions of the training step (round 5; per-workgroup partials, groups = namp_train_rows_groups(rows), add with namp_reduce_sum):
 *   namp_train_class_sums: part[groups][nclass][128] = per-class sums of the rows g [rows][128] by idx [rows] (nclass <= 64; rows with idx outside
 *     [0, nclass) are skipped) — the gradient of a few-row embedding lookup: W_s (na_model_utils.py:626) and node_embedding (:586).
 *   namp_train_wcolsum:    part[groups][128] = sum_rows g[row][:] * w[row] — db3 of a message stage with its third layer behind the K-sum. */
int namp_train_rows_groups(long rows);
int namp_train_class_sums(const float* g, const int32_t* idx, int nclass, long rows, float* part, void* stream);
int namp_train_wcolsum(const float* g, const float* w, long rows, float* part, void* stream);

/* Sums over partials, up to 16 segments in ONE launch (round 5): dst[a * Mb + b] = sum_{i < n} src[a * sa + i * sn + b] for a < A, b < Mb (element
 * strides; fp32).  Covers the reductions behind the training launches — [n][M] partials of weight gradients (A = 1, sa = 0, sn = M), per-tile rows
 * [G][T][128] -> [G][128] (A = G, sa = T * 128, sn = 128) — which were ~100 stock reduction launches per cfg5 step.  Deterministic (fixed order). */
typedef struct NampReduce { const float* src; float* dst; long long A, Mb, sa, sn; int n; int reserved; } NampReduce;
int namp_reduce_sum(const NampReduce* seg, int nseg, void* stream);

/* Level-parallel form of the sampler.  The step for residue i depends only on the neighbours visited before it, so visits can be
 * grouped into dependency levels and every level decoded in one launch over all streams: ~64 launches instead of 1000 sequential
 * steps at N = 1000, K = 48.  Same arithmetic per residue and the same uniform per visit as namp_decoder_sample, hence identical draws.
 *   namp_sample_levels: level[b][t] (int32, indexed by VISIT t of stream b) = 1 + max level of the earlier neighbours.
 *   namp_decoder_sample_levels: work = int32 pairs (stream, visit) sorted by level [nwork][2] (device);
 *     level_counts = HOST array of n_levels counts (the caller reads the histogram back; this library never synchronises).
 *   namp_decoder_sample_walk (round 3): the same levels in ONE persistent launch — no kernel boundary per level (whose cold L2 made
 *     the level's workgroups re-fetch the decoder weights at ~45 GB/s) and no host read-back: level_off = DEVICE int32 array,
 *     level_off[l] = index of the first pair of level l in `work`, every entry behind the last level = nwork (at least N + 2
 *     entries).  namp_decoder_sample_walk_grid workgroups (<= min(128, half the device's CUs), one per CU: the device must not be shared with other
 *     streams meanwhile) walk the levels with a grid barrier in between; a barrier that gives up (bounded spin) makes the launch
 *     overwrite log_probs with NaN.  K <= 128 (returns 0 workgroups otherwise: use the per-level launches).  Identical draws.
 * Symmetry-tied sampling (model_utils.py:219-326; group_first / group_last / sym_weights as in namp_decoder_sample): a work item is a
 *   GROUP — work = (stream, the group's first visit), work_n[item] = its number of (consecutive) visits, nwork = number of groups over
 *   all streams; the item's workgroup slot runs the members one after the other (a member sees the decoder states of the members
 *   before it, undrawn tokens as in the sequential walk) and draws once.  namp_sample_levels_dep with the group arrays gives every
 *   visit of a group the group's level: 1 + the highest level among its members' dependencies in earlier groups.  Without symmetry
 *   groups: group_first = group_last = sym_weights = work_n = NULL and nwork = B_dec * N.
 *   Deferred group draw (namp_decoder_sample_walk only; close / close_off / zbuf all set or all NULL): a group may then be SPLIT over
 *   several work items of the same level — consecutive runs of its visits, e.g. single members when no member is a graph neighbour of
 *   another (a member that reads an earlier member's decoder state must share the earlier member's run) — which different workgroups
 *   decode in parallel; each member's logits go to zbuf (float [B_dec][N][vocab], scratch), and after the level's grid barrier one wave
 *   per group adds them in visit order (the walk's own fma chain: identical bits) and draws.  close = int32 pairs (stream, LAST visit)
 *   of every group sorted by level, close_off[l] = first pair of level l (like level_off; entries behind the last level = number of groups). */
int namp_sample_levels(const int32_t* E_idx, const int32_t* order, const int32_t* rank, int32_t* level, int B_dec, int B_enc,
                       int N, int K, void* stream);
/* ... with extra dependencies per residue, dep_idx int32 [B_enc][N][D] (-1 = none): `pair_bias` (optional argument of the two level
 * decoders below, [B_enc][N][vocab][N][vocab] as in namp_decoder_sample) makes the step of residue i read the token of every j whose
 * block pair_bias[i, :, j, :] is not all zero; with those j listed here the levels respect that, and the level decoders — which
 * treat every residue later in the decoding order as undecoded (PAD), like the sequential walk sees it — give identical draws. */
int namp_sample_levels_dep(const int32_t* E_idx, const int32_t* order, const int32_t* rank, const int32_t* dep_idx, int D,
                           const int32_t* group_first, const int32_t* group_last, int32_t* level,
                           int B_dec, int B_enc, int N, int K, void* stream);
int namp_decoder_sample_levels(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                               const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                               const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                               const int32_t* group_first, const int32_t* group_last, const float* sym_weights, const float* pair_bias,
                               const int32_t* work, const int32_t* work_n, const int32_t* level_counts, int n_levels,
                               float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                               void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream);

int namp_decoder_sample_walk_grid(int B_dec, int N, int K);
int namp_decoder_sample_walk(const NampModelW* w, const float* h_V_enc, const float* h_E, const int32_t* E_idx,
                             const int32_t* mask, const int32_t* mask_dec, const int32_t* chain_mask, const int32_t* S_true, const float* bias,
                             const int32_t* order, const int32_t* rank, const float* uniform, const int32_t* S_forced,
                             const int32_t* group_first, const int32_t* group_last, const float* sym_weights, const float* pair_bias,
                             const int32_t* work, const int32_t* work_n, int nwork, const int32_t* level_off,
                             const int32_t* close, const int32_t* close_off, float* zbuf,
                             float temperature, uint64_t special_tokens, int32_t* S_out, float* probs_out, float* logp_out,
                             void* ws, size_t ws_bytes, int B_dec, int B_enc, int N, int K, void* stream);

/* ---- measurement hook (bench.py) ------------------------------------------------------------
 * When enabled (thread-local), every kernel launch made through this ABI is bracketed by HIP
 * events on the launch stream; namp_profile_collect() waits for them and returns the summed
 * milliseconds and launch counts per kernel kind, then clears the records.  Off by default. */
#define NAMP_KIND_GATHER 0
#define NAMP_KIND_NODE_LINEAR 1
#define NAMP_KIND_EDGE_EMBED 2
#define NAMP_KIND_ENC_MESSAGE 3
#define NAMP_KIND_ENC_EDGE 4
#define NAMP_KIND_NODE_UPDATE 5
#define NAMP_KIND_DEC_MESSAGE 6
#define NAMP_KIND_LOGITS 7
#define NAMP_KIND_FEATURES 8
#define NAMP_KIND_ENC_EDGE_MESSAGE 9   /* namp_enc_edge_message_update: edge update of layer l-1 + message of layer l */
#define NAMP_KIND_ENC_EDGE_DEC_MESSAGE 10   /* namp_encdec_fwd: last edge update + DecLayer 0 message */
#define NAMP_KIND_ENCDEC_PERSISTENT 11   /* namp_encdec_fwd: the whole encoder + decoder pass as one persistent launch */
#define NAMP_NUM_KINDS 12
int namp_profile_enable(int on);
int namp_profile_collect(float* ms_per_kind, int32_t* launches_per_kind, int nkinds);

#ifdef __cplusplus
}
#endif
#endif /* NAMP_H_ */
