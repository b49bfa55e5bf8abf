#!/usr/bin/env python
"""TEST INFRASTRUCTURE — expected per-residue arrays of the input parser, made with the oracle's restatement of
`parse_PDB` (oracle/pdb_ref.py), never with the product's reader:

* tests/golden/pdb/edge_cases.pdb + edge_cases_expected.npz — a hand-built file that exercises the selection rules of
  data_utils.py:232-345: a ligand chain that appears first (chain numbering counts every chain of the file), alternate
  locations, zero occupancy, a HETATM amino acid of ProDy's non-standard table (MSE), an amino acid without CA, residue names
  outside ProDy's tables (UNK, PSU), insertion codes, a 5'-terminal nucleotide without phosphate, waters, a second MODEL;
* tests/golden/pdb/{4oqu,1am9}_expected.npz — the same arrays for the reference's two example inputs
  (/root/reference/inference/examples: present in the build container only; the files themselves are not copied).

    python oracle/make_pdb_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pdb_ref    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pdb")
KEYS = ("X", "X_m", "mask", "R_idx", "chain_labels", "protein_mask", "dna_mask", "rna_mask", "rna_mask_for_token_conversion",
        "R_polymer_type", "S")
PROT = ["N", "CA", "C", "O"]
DNA = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "C1'"]
RNA = DNA[:-1] + ["O2'", "C1'"]


def edge_case_lines():
    rng = np.random.default_rng(11)
    out, serial = [], [1]

    def atom(rec, name, resname, chain, resnum, icode=" ", alt=" ", occ=1.0, b=10.0, el=None):
        x, y, z = rng.uniform(-20, 20, 3)
        nm = name if len(name) == 4 else " " + name
        out.append("%-6s%5d %-4s%1s%3s %1s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" % (
            rec, serial[0], nm, alt, resname, chain, resnum, icode, x, y, z, occ, b, el or name[0]))
        serial[0] += 1

    def residue(resname, chain, resnum, names, rec="ATOM", icode=" ", **kw):
        for n in names:
            atom(rec, n, resname, chain, resnum, icode, **kw)

    out.append("MODEL        1")
    atom("HETATM", "ZN", "ZN", "L", 900, el="ZN", b=33.0)                 # a ligand chain FIRST: shifts every chain index by one
    residue("ALA", "A", 1, PROT + ["CB"])
    for n in PROT:                                                          # GLY 2: alternate locations on CA (B is dropped)
        if n == "CA":
            atom("ATOM", n, "GLY", "A", 2, alt="A", occ=0.6)
            atom("ATOM", n, "GLY", "A", 2, alt="B", occ=0.4)
        else:
            atom("ATOM", n, "GLY", "A", 2)
    residue("MSE", "A", 3, PROT + ["SE"], rec="HETATM")                     # ProDy non-standard amino acid -> `protein`; token UNK
    residue("SER", "A", 4, ["N", "C", "O", "OG"])                           # no CA: not `protein` for ProDy -> "other atoms"
    for n in PROT:                                                          # LYS 5: O with occupancy 0 -> dropped -> incomplete backbone
        atom("ATOM", n, "LYS", "A", 5, occ=0.0 if n == "O" else 1.0)
    residue("UNK", "A", 6, PROT)                                            # not in ProDy's amino-acid tables -> "other atoms"
    residue("THR", "A", 7, PROT)
    residue("THR", "A", 7, PROT, icode="A")                                 # insertion code: same number, own residue
    residue("VAL", "A", 8, PROT)
    out.append("TER")
    residue("DT", "B", 1, DNA[3:])                                          # 5'-terminal nucleotide without phosphate: masked
    residue("DA", "B", 2, DNA)
    residue("DG", "B", 3, DNA)
    residue("DC", "B", 4, DNA[:-1])                                         # no C1': no reference atom -> not a residue at all
    out.append("TER")
    residue("A", "C", 10, RNA)
    residue("PSU", "C", 11, RNA, rec="HETATM")                              # modified nucleotide outside ProDy's table -> "other atoms"
    residue("U", "C", 12, RNA)
    residue("G", "C", 13, [a for a in RNA if a != "O2'"])                   # RNA name, DNA-complete backbone -> dna_mask, token by name
    out.append("TER")
    atom("HETATM", "O", "HOH", "A", 501, el="O")
    atom("HETATM", "O", "HOH", "C", 502, el="O")
    atom("HETATM", "MG", "MG", "C", 601, el="MG", b=44.0)
    out.append("ENDMDL")
    out.append("MODEL        2")
    residue("ALA", "A", 1, PROT)                                            # second model: ignored
    out.append("ENDMDL")
    out.append("END")
    return out


def expected(path, **kw):
    P = pdb_ref.parse_PDB(path, **kw)
    d = {k: P[k] for k in KEYS}
    d["chain_letters"] = np.array(P["chain_letters"])
    d["icodes"] = np.array([str(i) for i in P["icodes"]])
    d["na_chain_letters"] = np.array(list(P["na_chain_letters"]))
    d["encoded_residues"] = np.array(pdb_ref.encoded_residues(P))
    d["R_idx_renumbered"] = pdb_ref.featurize_R_idx(P["R_idx"])
    oa, bb = P["other_atoms"], P["backbone"]
    d["other_atom_names"] = np.array([] if oa is None else [f"{r}:{c}:{n}:{a}" for r, c, n, a in zip(oa.resname, oa.chid, oa.resnum, oa.name)])
    d["n_backbone_atoms"] = np.int64(0 if bb is None else len(bb))
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    pdb = os.path.join(OUT, "edge_cases.pdb")
    open(pdb, "w").write("\n".join(edge_case_lines()) + "\n")
    for tag, kw in (("shared", dict(na_shared_tokens=True)), ("legacy", dict(na_shared_tokens=False)),
                    ("missing", dict(na_shared_tokens=True, load_residues_with_missing_atoms=1)),
                    ("naonly", dict(na_shared_tokens=True, parse_na_only=True)), ("chainsBC", dict(na_shared_tokens=True, chains=["B", "C"]))):
        d = expected(pdb, **kw)
        np.savez_compressed(os.path.join(OUT, f"edge_cases_expected_{tag}.npz"), **d)
        print(tag, "L =", len(d["S"]), "S =", d["S"].tolist(), "chain_labels =", d["chain_labels"].tolist(), "mask =", d["mask"].tolist())
    ex = "/root/reference/inference/examples"
    if os.path.isdir(ex):
        for name in ("4oqu", "1am9"):
            d = expected(os.path.join(ex, name + ".pdb"), na_shared_tokens=True)
            d.pop("X")                                  # coordinates stay with the reference's file; a digest pins them
            d["X_checksum"] = np.float64(pdb_ref.parse_PDB(os.path.join(ex, name + ".pdb"), na_shared_tokens=True)["X"].astype(np.float64).sum())
            np.savez_compressed(os.path.join(OUT, f"{name}_expected.npz"), **d)
            print(name, "L =", len(d["S"]), "protein/dna/rna/masked =", int(d["protein_mask"].sum()), int(d["dna_mask"].sum()),
                  int(d["rna_mask"].sum()), int((d["mask"] == 0).sum()), "chains =", sorted(set(d["chain_letters"].tolist())))


if __name__ == "__main__":
    main()
