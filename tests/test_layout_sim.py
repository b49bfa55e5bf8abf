"""Lane-level numpy simulation of the register-resident MLP chain (csrc/namp_device.h).

Validates, on the CPU, the host-side weight-image permutation (na_mpnn_amd.pack.image_index,
which restates pack_image_kernel) together with the k-permutation argument that lets the
accumulators of one v_mfma_f32_16x16x4_f32 layer feed the next layer's operand registers.
The MFMA operand/result lane maps are the ones documented for gfx950:
  a: A[i=l&15][k=l>>4]   b: B[k=l>>4][j=l&15]   d[r]: D[i=4*(l>>4)+r][j=l&15]
"""
import numpy as np

from na_mpnn_amd.pack import image_index, plan


def mfma16x16x4(a, b, c):
    """a,b: [64] per-lane scalars; c: [64,4] -> d [64,4]"""
    lanes = np.arange(64)
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[lanes & 15, lanes >> 4] = a
    B[lanes >> 4, lanes & 15] = b
    D = A @ B
    d = np.empty((64, 4))
    for r in range(4):
        d[:, r] = D[4 * (lanes >> 4) + r, lanes & 15]
    return d + c


def image(W):
    out_f, in_f = W.shape
    n, k = image_index(out_f, in_f)
    return W[n, k].reshape(in_f // 16, out_f // 16, 64, 4)     # [tk][tn][lane][r]


def chain(img, x, acc, flip):
    """x: [TK][64][4] activation registers, acc: [NTN][64][4]."""
    TK, NTN = img.shape[0], img.shape[1]
    for tk in range(TK):
        for r in range(4):
            for tn in range(NTN):
                wf = img[tk, tn, :, r]
                acc[tn] = mfma16x16x4(x[tk][:, r], wf, acc[tn]) if flip else mfma16x16x4(wf, x[tk][:, r], acc[tn])
    return acc


def rows_to_T(X):
    """[16 rows, C] -> T-layout registers [C/16][64][4]: lane (m,g) holds channels 16t+4g+r."""
    lanes = np.arange(64)
    m, g = lanes & 15, lanes >> 4
    C = X.shape[1]
    out = np.empty((C // 16, 64, 4))
    for t in range(C // 16):
        for r in range(4):
            out[t][:, r] = X[m, 16 * t + 4 * g + r]
    return out


def T_to_rows(regs):
    lanes = np.arange(64)
    m, g = lanes & 15, lanes >> 4
    C = 16 * regs.shape[0]
    X = np.empty((16, C))
    for t in range(regs.shape[0]):
        for r in range(4):
            X[m, 16 * t + 4 * g + r] = regs[t][:, r]
    return X


def F_to_rows(regs):
    """F layout: lane (n_local, g) holds rows 4g+r of channel 16t + n_local."""
    lanes = np.arange(64)
    nl, g = lanes & 15, lanes >> 4
    X = np.empty((16, 16 * regs.shape[0]))
    for t in range(regs.shape[0]):
        for r in range(4):
            X[4 * g + r, 16 * t + nl] = regs[t][:, r]
    return X


def test_three_layer_chain_matches_matmul():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((16, 128))
    W1, W2, W3 = (rng.standard_normal((128, 128)) / 11 for _ in range(3))   # asymmetric on purpose
    b1, b2, b3 = (rng.standard_normal(128) for _ in range(3))
    act = np.tanh
    ref = act(act(X @ W1.T + b1) @ W2.T + b2) @ W3.T + b3

    x = rows_to_T(X)
    acc = rows_to_T(np.broadcast_to(b1, (16, 128)))
    h1 = act(chain(image(W1), x, acc, flip=False))
    np.testing.assert_allclose(T_to_rows(h1), act(X @ W1.T + b1), rtol=1e-12, atol=1e-12)
    acc = rows_to_T(np.broadcast_to(b2, (16, 128)))
    h2 = act(chain(image(W2), h1, acc, flip=False))
    # layer 3 in the T orientation (edge update) ...
    accT = rows_to_T(np.broadcast_to(b3, (16, 128)))
    np.testing.assert_allclose(T_to_rows(chain(image(W3), h2, accT, flip=False)), ref, rtol=1e-11, atol=1e-11)
    # ... and in the F orientation (message sum): bias enters as b3[16t + n_local] on every r
    lanes = np.arange(64)
    accF = np.stack([np.repeat(b3[16 * t + (lanes & 15)][:, None], 4, 1) for t in range(8)])
    np.testing.assert_allclose(F_to_rows(chain(image(W3), h2, accF, flip=True)), ref, rtol=1e-11, atol=1e-11)


def test_ffn_slices_match_matmul():
    """node_ffn_kernel: wave w owns hidden units [64w,64w+64): tn tiles 4w..4w+3 of the W_in image,
    tk tiles 4w..4w+3 of the W_out image; the 8 partial outputs add up to the full FFN."""
    rng = np.random.default_rng(1)
    X = rng.standard_normal((16, 128))
    Win, Wout = rng.standard_normal((512, 128)) / 11, rng.standard_normal((128, 512)) / 22
    b_in = rng.standard_normal(512)
    ref = np.tanh(X @ Win.T + b_in) @ Wout.T
    img_in, img_out = image(Win), image(Wout)          # [8][32][64][4], [32][8][64][4]
    x = rows_to_T(X)
    total = np.zeros((16, 128))
    for w in range(8):
        acc = rows_to_T(np.broadcast_to(b_in, (16, 512)))[4 * w:4 * w + 4]
        hid = np.tanh(chain(img_in[:, 4 * w:4 * w + 4], x, acc.copy(), flip=False))
        out = chain(img_out[4 * w:4 * w + 4], hid, np.zeros((8, 64, 4)), flip=False)
        total += T_to_rows(out)
    np.testing.assert_allclose(total, ref, rtol=1e-11, atol=1e-11)


def test_plan_offsets_are_aligned_and_disjoint():
    items, total = plan(3, 3, 33)
    end = 0
    for name, (off, n, _) in items.items():
        assert off % 64 == 0 and off >= end, name       # 256-byte aligned, no overlap
        end = off + n
    assert end <= total
    for l in range(3):
        for f in ("W1a_img", "W1b_img", "W1c_img", "W11a_img", "Win_img", "Wout_img", "ln3_b"):
            assert f"enc{l}.{f}" in items
        for f in ("W1a_img", "W1e_img", "W1s_img", "W1v_img", "tok", "ln2_b"):
            assert f"dec{l}.{f}" in items
