"""Phase attribution of the sampler step from the NAMP_ABL_STAMPS build (tools/build_variants.sh stamps:-DNAMP_ABL_STAMPS):
    NAMP_LIB_PATH=tools/_variants/stamps.so python tools/sample_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from na_mpnn_amd import hip, spec, synth
from na_mpnn_amd.model import ProteinMPNN
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
torch.manual_seed(0)
w = synth.make_weights(0)
L = C.CDLL(os.environ["NAMP_LIB_PATH"])
names = ["between steps (barrier / launch)", "index chain + row gather", "layer argument block read", "product 2 (W2 . gelu(z1))", "product 3 + K-sum",
         "tail: projections (rest of the tail phase)", "head + draw", "tail: next images copied + K-sums read", "tail: LayerNorm 1", "tail: W_in + GELU",
         "tail: W_out partials", "tail: mask, h_V' out (rest of LN2 phase)", "(12)", "tail: partial sums read", "tail: LayerNorm 2", "(15)"]
for n, k, bs, walk in ((97, 32, 1, True), (97, 32, 1, False), (400, 32, 30, True)):
    m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=k, atom_dict=spec.atom_dict(), restype_to_int=spec.restype_to_int(),
                    polytype_to_int=spec.polytype_to_int())
    m.load_state_dict({k_: torch.from_numpy(v) for k_, v in w.items()}); m = m.to(dev).eval()
    m.sample_level_walk = walk
    cx = synth.make_complex(seed=3, n=n)
    fd = {k_: torch.from_numpy(np.ascontiguousarray(v))[None].to(dev) for k_, v in cx.items()}
    fd.update({"batch_size": bs, "temperature": 0.1, "bias": torch.zeros(1, n, 33, device=dev), "symmetry_residues": [[]],
               "symmetry_weights": [[]], "randn": torch.randn(bs, n, device=dev)})
    out = m.sample(fd); torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    L.namp_debug_stamps(buf, 1)
    reps = 5
    for _ in range(reps):
        out = m.sample(fd)
    torch.cuda.synchronize()
    L.namp_debug_stamps(buf, 1)
    lv = int(out["levels"])
    tot = sum(buf[:16])
    print(f"N={n} K={k} batch_size={bs} {'persistent walk' if walk else 'launch per level'}: {lv} levels; workgroup 0, us per level (3 layers summed):")
    for i, nm in enumerate(names):
        print(f"   {nm:40s} {buf[i] * 0.01 / reps / lv:8.2f}")
    print(f"   {'total':40s} {tot * 0.01 / reps / lv:8.2f}")
