#!/usr/bin/env python
"""Per-kernel timing at the cfg2 shape for one build of libnamp_hip.so (NAMP_LIB_PATH selects it).

    NAMP_LIB_PATH=/path/to/variant.so python tools/kbench.py [--reps 200] [--B 1]

Used with ablation builds (see NAMP_ABL_* in csrc/namp_device.h) to attribute kernel time.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip, spec, synth          # noqa: E402
from na_mpnn_amd.pack import PackedWeights        # noqa: E402


def timeit(fn, reps):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--N", type=int, default=1000)
    ap.add_argument("--K", type=int, default=48)
    ap.add_argument("--precision", default="x3", choices=["x3", "fp32", "bf16"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = hip.lib()
    B, N, K = args.B, args.N, args.K
    w = synth.make_weights(0)
    P = PackedWeights({k: torch.from_numpy(v).to(dev) for k, v in w.items()}, 3, 3, spec.VOCAB, dev)
    P.set_precision(args.precision)
    G = B * N
    tpn = (K + 15) // 16
    hE = torch.randn(G, K, 128, device=dev)
    hE2 = torch.empty_like(hE)
    hV = torch.randn(G, 128, device=dev)
    hV2 = torch.empty_like(hV)
    Pa, Pc, Pf = (torch.randn(G, 128, device=dev) for _ in range(3))
    T = [torch.empty(G, 128, device=dev) for _ in range(4)]
    partial = torch.randn(G * tpn * 129 + 3, device=dev)       # K-sums + weight sums (include/namp.h)
    idx = torch.randint(0, N, (B, N, K), device=dev, dtype=torch.int32)
    mask = torch.ones(G, dtype=torch.int32, device=dev)
    rank = torch.randperm(N, device=dev).to(torch.int32).repeat(B)
    S = torch.randint(0, 25, (G,), device=dev, dtype=torch.int32)
    s = hip.current_stream()
    e0, d0 = P.enc_layer(0), P.dec_layer(0)
    res = {}
    res["enc_message"] = timeit(lambda: L.namp_enc_message(e0, hE.data_ptr(), idx.data_ptr(), mask.data_ptr(), None,
                                                           Pa.data_ptr(), Pc.data_ptr(), partial.data_ptr(), B, N, K, s), args.reps)
    res["enc_edge_update"] = timeit(lambda: L.namp_enc_edge_update(e0, hE.data_ptr(), idx.data_ptr(), Pa.data_ptr(),
                                                                   Pc.data_ptr(), hE2.data_ptr(), B, N, K, s), args.reps)
    res["dec_message"] = timeit(lambda: L.namp_dec_message(d0, hE.data_ptr(), idx.data_ptr(), rank.data_ptr(), Pa.data_ptr(),
                                                           Pc.data_ptr(), Pf.data_ptr(), partial.data_ptr(), B, B, N, K, s), args.reps)
    res["edge_embed"] = timeit(lambda: L.namp_edge_embed(P.addr("We_img"), P.addr("We_b"), hE.data_ptr(), hE2.data_ptr(),
                                                         B, N, K, s), args.reps)
    a = lambda n: P.addr("enc0." + n)
    for npj in (0, 2, 4):
        proj = (hip.NampProj * 4)(*[hip.NampProj(a(nm), None, None, T[i].data_ptr())
                                    for i, nm in enumerate(["W11a_img", "W11c_img", "W1a_img", "W1c_img"])])
        res[f"node_update_p{npj}"] = timeit(lambda: L.namp_node_update(
            a("ln1_g"), a("ln1_b"), a("Win_img"), a("b_in"), a("Wout_img"), a("b_out"), a("ln2_g"), a("ln2_b"),
            hV.data_ptr(), partial.data_ptr(), a("W3_img"), a("b3"), mask.data_ptr(), hV2.data_ptr(), proj, npj, None, G, K, s), args.reps)
    for npj in (0, 2, 4):
        proj = (hip.NampProj * 4)(*[hip.NampProj(a(nm), None, None, T[i].data_ptr())
                                    for i, nm in enumerate(["W11a_img", "W11c_img", "W1a_img", "W1c_img"])])
        res[f"enc_msg_upd_p{npj}"] = timeit(lambda: L.namp_enc_message_update(
            e0, hE.data_ptr(), idx.data_ptr(), mask.data_ptr(), None, Pa.data_ptr(), Pc.data_ptr(), hV.data_ptr(),
            hV2.data_ptr(), proj, npj, B, N, K, s), args.reps)
    e1 = P.enc_layer(1)
    proj4 = (hip.NampProj * 4)(*[hip.NampProj(a(nm), None, None, T[i].data_ptr())
                                 for i, nm in enumerate(["W11a_img", "W11c_img", "W1a_img", "W1c_img"])])
    res["enc_edge_msg_upd_p4"] = timeit(lambda: L.namp_enc_edge_message_update(
        e0, Pa.data_ptr(), Pc.data_ptr(), hE.data_ptr(), e1, idx.data_ptr(), mask.data_ptr(), None, Pa.data_ptr(), Pc.data_ptr(),
        hV.data_ptr(), hV2.data_ptr(), proj4, 4, B, N, K, s), args.reps)
    proj = (hip.NampProj * 2)(hip.NampProj(a("W1a_img"), a("b1"), None, T[0].data_ptr()),
                              hip.NampProj(a("W1c_img"), None, None, T[1].data_ptr()))
    res["node_linear_p2"] = timeit(lambda: L.namp_node_linear(hV.data_ptr(), None, B, B, N, proj, 2, None, s), args.reps)
    logp = torch.empty(G, 33, device=dev)
    res["logits"] = timeit(lambda: L.namp_logits_log_softmax(P.addr("Wout_w"), P.addr("Wout_b"), hV.data_ptr(),
                                                             logp.data_ptr(), None, G, 33, s), args.reps)
    tag = os.path.basename(os.environ.get("NAMP_LIB_PATH", "default"))
    print(tag, args.precision, " ".join(f"{k}={v:.1f}us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
