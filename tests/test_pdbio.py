"""f3: the prody-free PDB reader (na_mpnn_amd/pdbio.py) — write/read round trips of synthetic complexes and,
in the build container only, the facts SURVEY App. B records for the reference's two example files."""
import os

import numpy as np
import pytest

from na_mpnn_amd import pdbio, spec, synth

REF_EXAMPLES = "/root/reference/inference/examples"
INT_TO_RES = {v: k for k, v in spec.restype_to_int().items()}


def _write(tmp_path, cx, chain_letters="ABCDEFGH"):
    names = [INT_TO_RES[int(s)] for s in cx["S"]]
    letters = [chain_letters[c] for c in cx["chain_labels"]]
    path = os.path.join(tmp_path, "x.pdb")
    pdbio.write_pdb(path, cx["X"], cx["X_m"], names, letters, cx["R_idx"])
    return path, letters


def test_round_trip_mixed_complex(tmp_path):
    cx = synth.make_complex(seed=21, n=60, n_chains=3)
    path, letters = _write(str(tmp_path), cx)
    P = pdbio.parse_pdb(path, na_shared_tokens=False)
    assert P["chain_letters"] == letters
    assert np.array_equal(P["X_m"], cx["X_m"]) and np.abs(P["X"] - cx["X"]).max() < 6e-4     # %8.3f
    for k in ("mask", "protein_mask", "dna_mask", "rna_mask", "R_idx", "chain_labels", "S"):
        assert np.array_equal(P[k], cx[k]), k
    assert np.array_equal(P["R_polymer_type"], cx["R_polymer_type"])
    # shared DNA/RNA tokens (run.py:112-117): RNA residues get the DNA token ids
    Ps = pdbio.parse_pdb(path, na_shared_tokens=True)
    rna = cx["rna_mask"].astype(bool)
    assert np.array_equal(Ps["S"][rna], cx["S"][rna] - 5) and np.array_equal(Ps["S"][~rna], cx["S"][~rna])
    assert np.array_equal(Ps["rna_mask_for_token_conversion"], cx["rna_mask"])


def test_incomplete_backbones_are_masked_and_filters(tmp_path):
    cx = synth.make_complex(seed=22, n=40, n_chains=2)
    cx["X_m"][3, 2] = 0            # protein residue without C
    dna = np.where(cx["dna_mask"] == 1)[0]
    cx["X_m"][dna[0], 6] = 0       # 5'-terminal nucleotide without P
    path, letters = _write(str(tmp_path), cx)
    P = pdbio.parse_pdb(path)
    assert P["mask"][3] == 0 and P["mask"][dna[0]] == 0 and P["mask"].sum() == 38
    assert P["R_polymer_type"][3] == spec.polytype_to_int()["UNK"]
    assert P["S"][3] == cx["S"][3]           # a known residue name keeps its token (data_utils.py:333-345)
    only = pdbio.parse_pdb(path, chains=[letters[0]])
    assert set(only["chain_letters"]) == {letters[0]}
    na = pdbio.parse_pdb(path, parse_na_only=True)
    assert na["protein_mask"].sum() == 0 and len(na["S"]) == int((cx["dna_mask"] + cx["rna_mask"]).sum())
    lines = open(path).read().splitlines()
    lines.insert(0, lines[0][:54] + "  0.00" + lines[0][60:])      # occupancy 0 atom is dropped
    lines.insert(0, "HETATM    1  O   HOH A 999      0.000   0.000   0.000  1.00  0.00           O")
    open(path, "w").write("\n".join(lines) + "\n")
    assert len(pdbio.parse_pdb(path)["S"]) == 40


def test_renumber_insertion_codes():
    assert pdbio.renumber(np.array([5, 6, 6, 6, 7, 10])).tolist() == [5, 6, 7, 8, 9, 12]


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference example files only exist in the build container")
def test_reference_example_files():
    p = pdbio.parse_pdb(os.path.join(REF_EXAMPLES, "4oqu.pdb"))
    assert len(p["S"]) == 97 and set(p["chain_letters"]) == {"A"}            # 97-nt RNA, one chain
    assert p["rna_mask"].sum() == 97 and p["mask"].sum() == 97
    q = pdbio.parse_pdb(os.path.join(REF_EXAMPLES, "1am9.pdb"))
    assert len(q["S"]) == 389 and len(set(q["chain_letters"])) == 8          # 313 aa + 76 nt, 8 chains
    assert q["protein_mask"].sum() == 313 and q["dna_mask"].sum() == 72 and q["mask"].sum() == 385
