"""Minimal PDB ATOM-record reader/writer producing the NA-MPNN ``feature_dict``.

Restates the semantics of the reference's prody-based ``parse_PDB`` + ``featurize``
(/root/reference/inference/data_utils.py:84-439) without prody (absent from this image):

* atoms with occupancy > 0, altloc blank or 'A', first MODEL only;
* residues are keyed chain / resnum / insertion-code and ordered by the first appearance of their
  reference atom (CA for amino acids, C1' for nucleotides; data_utils.py:258-276);
* the 16 backbone atoms in the order N, CA, C, O, OP1, OP2, P, O5', C5', C4', O4', C3', O3', C2', O2', C1';
* polymer masks from backbone completeness (protein: N CA C O; RNA: the 12 NA atoms; DNA: the 11 without
  O2', minus RNA; data_utils.py:316-321), ``mask`` = their sum, ``R_polymer_type`` PP/DNA/RNA/UNK;
* tokens from residue names with polymer-dependent unknowns (UNK / DX / RX; data_utils.py:333-345),
  optionally sharing DNA tokens for RNA (``na_shared_tokens``, run.py:112-117);
* ``featurize``: residue numbers bumped by the count of repeated numbers (insertion codes; data_utils.py:409-416).

Parity note: prody cannot be imported here, so this reader is checked by write->read round trips and against
the facts SURVEY App. B records for the reference's two example files (when /root/reference is present).
"""
from __future__ import annotations

import numpy as np

from . import spec

PROTEIN_NAMES = set(spec.RESTYPES[:20]) | {"UNK", "ASX", "GLX", "CSO", "HIP", "HSD", "HSE", "HSP", "MSE", "SEC", "SEP",
                                           "TPO", "PTR", "XLE", "XAA", "PYL"}
NUCLEIC_NAMES = {"A", "C", "G", "U", "T", "I", "N", "DA", "DC", "DG", "DT", "DU", "DI", "DN", "DX", "RX",
                 "ADE", "CYT", "GUN", "GUA", "THY", "URA", "PSU", "5MC", "OMC", "OMG", "OMU", "1MA", "2MG", "7MG", "M2G", "H2U", "5MU"}
PROTEIN_BB = ["N", "CA", "C", "O"]
DNA_BB = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "C1'"]
RNA_BB = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"]


def _atom_records(path, chains=None):
    """Yield (name, resname, chain, resnum, icode, xyz) of ATOM/HETATM records of the first model."""
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec == "ENDMDL":
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            if line[16] not in (" ", "A"):
                continue
            try:
                occ = float(line[54:60]) if line[54:60].strip() else 1.0
                xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
                resnum = int(line[22:26])
            except ValueError:
                continue
            if occ <= 0:
                continue
            chain = line[21]
            if chains and chain not in chains:
                continue
            yield line[12:16].strip(), line[17:20].strip(), chain, resnum, line[26].strip(), xyz


def parse_pdb(path, chains=None, parse_na_only=False, na_shared_tokens=True, load_residues_with_missing_atoms=False):
    """-> dict of numpy arrays (no batch dimension) + 'chain_letters', 'icodes', 'name'."""
    atom_index = {a: i for i, a in enumerate(spec.ATOM_TYPES)}
    rti = spec.restype_to_int(na_shared_tokens)
    residues, order = {}, []        # key -> {"resname", "kind", "atoms": {name: xyz}}
    chain_order = []
    for name, resname, chain, resnum, icode, xyz in _atom_records(path, set(chains) if chains else None):
        kind = "protein" if resname in PROTEIN_NAMES else ("nucleic" if resname in NUCLEIC_NAMES else None)
        if kind is None or (parse_na_only and kind != "nucleic"):
            continue
        key = (chain, resnum, icode)
        r = residues.get(key)
        if r is None:
            r = residues[key] = {"resname": resname, "kind": kind, "atoms": {}, "ref": False}
        r["atoms"][name] = xyz
        if name == ("CA" if kind == "protein" else "C1'") and not r["ref"]:
            r["ref"] = True
            order.append(key)                       # residue order = order of reference atoms
            if chain not in chain_order:
                chain_order.append(chain)
    n = len(order)
    X = np.zeros((n, spec.N_ATOMS, 3), np.float32)
    X_m = np.zeros((n, spec.N_ATOMS), np.int32)
    for i, key in enumerate(order):
        for a, xyz in residues[key]["atoms"].items():
            j = atom_index.get(a)
            if j is not None:
                X[i, j] = xyz
                X_m[i, j] = 1
    resnames = [residues[k]["resname"] for k in order]
    prod = lambda names: np.prod(X_m[:, [atom_index[a] for a in names]], axis=-1).astype(np.int32)
    if load_residues_with_missing_atoms:
        protein_mask = np.array([r in spec.RESTYPES[:21] for r in resnames], np.int32)
        dna_mask = np.array([r in ("DA", "DC", "DG", "DT", "DX") for r in resnames], np.int32)
        rna_mask = np.array([r in ("A", "C", "G", "U", "RX") for r in resnames], np.int32)
    else:
        protein_mask = prod(PROTEIN_BB)
        rna_mask = prod(RNA_BB)
        dna_mask = prod(DNA_BB) - rna_mask           # RNA also carries every DNA backbone atom
    mask = protein_mask + dna_mask + rna_mask
    pti = spec.polytype_to_int()
    R_polymer_type = (protein_mask * pti["PP"] + dna_mask * pti["DNA"] + rna_mask * pti["RNA"] +
                      (1 - protein_mask - dna_mask - rna_mask) * pti["UNK"]).astype(np.int64)
    S = np.empty(n, np.int32)
    for i, rn in enumerate(resnames):
        unk = "DX" if dna_mask[i] == 1 else ("RX" if rna_mask[i] == 1 else "UNK")
        S[i] = rti.get(rn, rti[unk])
    chain_letters = [k[0] for k in order]
    return {
        "X": X, "X_m": X_m, "mask": mask.astype(np.int32), "S": S,
        "R_idx": np.array([k[1] for k in order], np.int32),
        "chain_labels": np.array([chain_order.index(c) for c in chain_letters], np.int32),
        "protein_mask": protein_mask, "dna_mask": dna_mask, "rna_mask": rna_mask,
        "rna_mask_for_token_conversion": X_m[:, atom_index["O2'"]].copy(),
        "R_polymer_type": R_polymer_type,
        "chain_letters": chain_letters, "icodes": [k[2] for k in order],
        "na_chain_letters": [c for i, c in enumerate(chain_letters) if dna_mask[i] or rna_mask[i]],
    }


def renumber(R_idx):
    """data_utils.featurize: consecutive equal residue numbers (insertion codes) are pushed apart."""
    out, count, prev = [], 0, -100000
    for r in R_idx.tolist():
        if r == prev:
            count += 1
        out.append(r + count)
        prev = r
    return np.array(out, dtype=np.int64)


def to_feature_dict(parsed, chain_mask, device):
    """Batch-of-one torch feature_dict with the reference's dtypes (data_utils.py:361-439)."""
    import torch
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=device)[None]
    fd = {"X": t(parsed["X"], torch.float32), "X_m": t(parsed["X_m"], torch.int32), "mask": t(parsed["mask"], torch.int32),
          "S": t(parsed["S"], torch.int32), "R_idx": t(renumber(parsed["R_idx"]), torch.int64),
          "R_idx_original": t(parsed["R_idx"], torch.int32), "chain_labels": t(parsed["chain_labels"], torch.int32),
          "chain_mask": t(chain_mask, torch.int32), "R_polymer_type": t(parsed["R_polymer_type"], torch.int64)}
    for k in ("protein_mask", "dna_mask", "rna_mask", "rna_mask_for_token_conversion"):
        fd[k] = t(parsed[k], torch.int32)
    return fd


def write_pdb(path, X, X_m, resnames, chain_letters, R_idx, icodes=None):
    """Write the 16 backbone atoms as ATOM records (used by tests and for round trips)."""
    lines, serial = [], 1
    for i in range(X.shape[0]):
        for a, name in enumerate(spec.ATOM_TYPES):
            if not X_m[i, a]:
                continue
            nm = name if len(name) == 4 else " " + name
            ic = (icodes[i] if icodes else "") or " "
            lines.append("ATOM  %5d %-4s %3s %1s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" % (
                serial, nm, resnames[i], chain_letters[i], R_idx[i], ic, X[i, a, 0], X[i, a, 1], X[i, a, 2], 1.0, 0.0,
                name[0]))
            serial += 1
    lines.append("END")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
