R=$GRAFT_REPO_ROOT
cd $R
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import gc, time, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from na_mpnn_amd import spec, synth, train
from na_mpnn_amd.model import ProteinMPNN
import train_time as tt
dev = torch.device("cuda:0")
rti = spec.restype_to_int()
m = ProteinMPNN(num_letters=33, vocab=33, k_neighbors=48, dropout=0.1, atom_dict=spec.atom_dict(), restype_to_int=rti, polytype_to_int=spec.polytype_to_int(), augment_eps=0.1)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0).items()}); m.to(dev).train(); m.message_precision = "bf16"
fd = tt.make_batch(16, 1500, dev)
opt = train.get_std_opt(m.parameters(), 128, 0)
rm, rn = train.polymer_restype_tables(rti, 33, dev)
no_loss = torch.tensor([rti[t] for t in ("UNK", "DX", "RX", "MAS", "PAD")], device=dev)
step = lambda: train.train_step(m, opt, fd, rm, rn, no_loss, loss_tokens=6000.0, gradient_norm=1.0)
for _ in range(3): step()
for label in ("gc on", "gc off"):
    if label == "gc off":
        gc.collect(); gc.disable()
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    print(label, "host ms max %.1f median %.1f | total max %.1f median %.1f" % (max(a for a, _ in ts), sorted(a for a, _ in ts)[20], max(b for _, b in ts), sorted(b for _, b in ts)[20]), "outliers:", [round(b) for _, b in ts if b > 40])
    print("   gc counts", gc.get_count(), "mem reserved GiB", torch.cuda.memory_reserved() / 2**30)
PY
