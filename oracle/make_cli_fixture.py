#!/usr/bin/env python
"""TEST INFRASTRUCTURE — expected output FILES of the design CLI, made with the CPU oracle.

The reference's inference/run.py cannot run in the build container (prody is absent), so its file formats are pinned this
way, with ORACLE-side code only (oracle/pdb_ref.py: parse_PDB, featurize, run.py's token tables / residue codes / sequence
strings; oracle/cpu_ref.py: the model) — the product's reader and CLI are what the fixture tests, so they take no part in it: a small mixed protein / DNA / RNA complex with a ligand and waters is written as PDB and as mmCIF
(tests/golden/cli/input.{pdb,cif}); the oracle's `sample()` (proven bit-identical to the reference's, make_goldens.py G5)
draws sequences under a fixed torch seed; and the output files are written with the reference's own format strings
(run.py:445-455 native header, :501-511 per-sample header, :426-443 specificity keys, np.format_float_positional with 4
digits).  The draws (decoding-order noise + sampled tokens) are stored next to them so that the product CLI can be
teacher-forced to the same sequences on the GPU (`--forced_draws_npz`) and compared with the expected files BYTE for byte.

    python oracle/make_cli_fixture.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from na_mpnn_amd import pdbio, synth   # noqa: E402   (only to WRITE the synthetic input files: pdbio.write_pdb / write_mmcif)
from oracle import cpu_ref, pdb_ref    # noqa: E402   (everything the expected files are computed with is oracle-side)

OUT = os.path.join(ROOT, "tests", "golden", "cli")
SEED, T, BS, K = 7, 1.0, 2, 32


def build_input():
    cx = synth.make_complex(seed=77, n=48, n_chains=3)
    int_to_res = dict(enumerate(pdb_ref.RESTYPES))
    letters = ["ABC"[c] for c in cx["chain_labels"]]
    names = [int_to_res[int(s)] for s in cx["S"]]
    pdb = os.path.join(OUT, "input.pdb")
    pdbio.write_pdb(pdb, cx["X"], cx["X_m"], names, letters, cx["R_idx"])
    lines = open(pdb).read().splitlines()[:-1]
    n = len(lines)
    lines += ["HETATM%5d MG    MG A 301    %8.3f%8.3f%8.3f  1.00 20.00          MG" % (n + 1, 1.0, 2.0, 3.0),
              "HETATM%5d  O   HOH A 401    %8.3f%8.3f%8.3f  1.00 30.00           O" % (n + 2, 4.0, 5.0, 6.0), "END"]
    open(pdb, "w").write("\n".join(lines) + "\n")
    pdbio.write_mmcif(os.path.join(OUT, "input.cif"), cx["X"], cx["X_m"], names, letters, cx["R_idx"], name="input")
    return pdb


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    pdb = build_input()
    # the oracle's restatement of parse_PDB / featurize / run.py's bookkeeping — NOT the product's reader
    P = pdb_ref.parse_PDB(pdb, na_shared_tokens=True)
    L = len(P["S"])
    fixed = [f"A{P['R_idx'][0]}", f"A{P['R_idx'][1]}"]
    encoded = pdb_ref.encoded_residues(P)
    chain_mask = np.array([int(e not in fixed) for e in encoded], np.int32)                       # run.py:259 (all chains designed)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt)[None]
    fd = {"X": t(P["X"], torch.float32), "X_m": t(P["X_m"], torch.int32), "mask": t(P["mask"], torch.int32),
          "S": t(P["S"], torch.int64), "R_idx": t(pdb_ref.featurize_R_idx(P["R_idx"]), torch.int64),
          "chain_labels": t(P["chain_labels"], torch.int32), "chain_mask": t(chain_mask, torch.int32),
          "R_polymer_type": t(P["R_polymer_type"], torch.int64)}
    for k_ in ("protein_mask", "dna_mask", "rna_mask", "rna_mask_for_token_conversion"):
        fd[k_] = t(P[k_], torch.int32)
    rti, alphabet, int_to_str, dna_to_rna = pdb_ref.token_tables(True)
    omit = torch.tensor([float(c in "X" + "bdhuy") for c in alphabet])                            # run.py:135-139
    fd.update({"batch_size": BS, "temperature": T, "bias": (-1e8 * omit[None, None, :]).repeat(1, L, 1),
               "symmetry_residues": [[]], "symmetry_weights": [[]]})
    torch.manual_seed(SEED)
    fd["randn"] = torch.randn(BS, L)
    w = {k: torch.from_numpy(v) for k, v in synth.make_weights(0).items()}
    out = cpu_ref.sample(w, fd, K)
    S, lp, sp = out["S"], out["log_probs"], out["sampling_probs"]
    # reproducible by teacher forcing
    again = cpu_ref.sample(w, fd, K, S_forced=S)
    assert torch.equal(again["S"], S) and torch.equal(again["log_probs"], lp)
    cmask = (fd["mask"] * fd["chain_mask"]).float()
    lpr = -(torch.nn.functional.one_hot(S, 33) * lp).sum(-1)
    loss = (lpr * cmask).sum(-1) / (cmask.sum(-1) + 1e-8)
    rec = ((fd["S"][:1] == S) * cmask).sum(-1) / cmask.sum(-1)
    # keep the 4-digit prints away from a rounding boundary, so that 1e-5 of device arithmetic cannot flip a digit
    for v in list(np.exp(-loss.numpy())) + list(rec.numpy()):
        frac = (float(v) * 1e4) % 1.0
        assert abs(frac - 0.5) > 0.02, f"value {v} too close to a 4-digit rounding boundary: change SEED"
    name, ckpt = "input", "random_init_seed_0"
    entries = ['>{}, T={}, seed={}, num_res={}, batch_size={}, number_of_batches={}, model_path={}\n{}'.format(
        name, T, SEED, (fd["mask"] * fd["chain_mask"]).sum().numpy(), BS, 1, ckpt,
        pdb_ref.sequence_string(P["S"], P, int_to_str, dna_to_rna))]
    for ix in range(BS):
        conf = np.format_float_positional(np.exp(-loss[ix].numpy()), unique=False, precision=4)
        srec = np.format_float_positional(rec[ix].numpy(), unique=False, precision=4)
        entries.append('>{}, id={}, T={}, seed={}, overall_confidence={} seq_rec={}\n{}'.format(
            name, ix + 1, T, SEED, conf, srec, pdb_ref.sequence_string(S[ix].numpy(), P, int_to_str, dna_to_rna)))
    open(os.path.join(OUT, "expected.fa"), "w").write("\n".join(entries))
    np.savez(os.path.join(OUT, "expected_specificity.npz"),
             predicted_ppm=np.mean(sp.numpy().astype(np.float64), axis=0), true_sequence=P["S"].astype(np.int64),
             chain_labels=P["chain_labels"], mask=P["mask"], protein_mask=P["protein_mask"], dna_mask=P["dna_mask"],
             rna_mask=P["rna_mask"], encoded_residues=encoded, encoded_residues_dict=dict(zip(encoded, range(L))), restype_to_int=rti)
    np.savez(os.path.join(OUT, "forced_draws.npz"), randn=fd["randn"].numpy(), S_forced=S.numpy().astype(np.int32),
             log_probs=lp.numpy(), per_residue_loss=lpr.numpy())
    print(open(os.path.join(OUT, "expected.fa")).read())
    print("fixture written to", OUT)


if __name__ == "__main__":
    main()
