"""A/B of the bf16-storage edge launches inside the cfg3 step (B=64 x N=1000, K=48, bf16 mode): per-launch HIP-event times by launch kind for
each value of the namp_set_bf16p mask (0 = round-3 kernels, 7 = round-6 sequencing), alternating, in ONE process on one box.
    python tools/cfg3_ab.py [--masks 0,11] [--reps 3] [--steps 20]        (NAMP_LIB_PATH selects a variant build)"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from na_mpnn_amd import hip                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,7")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    L = hip.lib()
    dev = torch.device("cuda:0")
    args = argparse.Namespace(no_pmc=True, no_cpu_baseline=True, verbose=False, detail_out=None, min_seconds=1.0)
    tag = os.path.basename(os.environ.get("NAMP_LIB_PATH", "default"))
    for rep in range(a.reps):
        for m in [int(x) for x in a.masks.split(",")]:
            L.namp_set_bf16p(m)
            o, r = bench.encdec_bench(args, dev, 0, 1, None, "cfg3", "bf16", a.steps, 5, profile_steps=a.steps)
            pk = o["per_kernel"]
            print(tag, "mask", m, "ms/step %.4f" % o["ms_per_step"], " ".join(f"{k}={v['avg_ms'] * 1e3:.1f}us" for k, v in pk.items()), flush=True)


if __name__ == "__main__":
    main()
