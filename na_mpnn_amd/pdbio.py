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

# Residue-name classes of the reference's prody selections ("protein", "nucleic", "water": prody/atomic/flags.py as
# documented for ProDy 2.x — standard + non-standard amino acids; nucleobases, nucleotides, nucleosides).  prody is not
# installable here, so these tables are a restatement from its documentation, not a pinned import.
PROTEIN_NAMES = set(spec.RESTYPES[:20]) | {"ASX", "GLX", "CSO", "HIP", "HSD", "HSE", "HSP", "MSE", "SEC", "SEP", "TPO", "PTR",
                                           "XLE", "XAA"}      # (UNK is in neither of prody's amino-acid tables: "other atoms")
NUCLEIC_NAMES = {"GUN", "ADE", "CYT", "THY", "URA",                                    # nucleobase
                 "DA", "DC", "DG", "DT", "DU", "A", "C", "G", "T", "U",                # nucleotide
                 "AMP", "ADP", "ATP", "CDP", "CTP", "GMP", "GDP", "GTP", "TMP", "TTP", "UMP", "UDP", "UTP",   # nucleoside
                 "DX", "RX"}                                                           # the reference's own unknown-NA names
WATER_NAMES = {"HOH", "DOD", "WAT", "TIP3", "H2O", "OH2", "TIP", "TIP2", "TIP4", "SOL"}
PROTEIN_BB = ["N", "CA", "C", "O"]
DNA_BB = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "C1'"]
RNA_BB = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"]
LEGACY_ATOM_NAMES = {"O1P": "OP1", "O2P": "OP2"}       # pre-remediation PDB names; '*' -> "'" is applied to every name


class Atom:
    """One coordinate record of either file format."""
    __slots__ = ("het", "serial", "name", "altloc", "resname", "chain", "resnum", "icode", "xyz", "occ", "bfac", "element")

    def __init__(self, het, serial, name, altloc, resname, chain, resnum, icode, xyz, occ, bfac, element):
        self.het, self.serial, self.name, self.altloc, self.resname, self.chain = het, serial, name, altloc, resname, chain
        self.resnum, self.icode, self.xyz, self.occ, self.bfac, self.element = resnum, icode, xyz, occ, bfac, element


def _pdb_atoms(path):
    """ATOM / HETATM records of the first MODEL of a PDB file (prody parsePDB: model 1, HETATM included)."""
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec == "ENDMDL":
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            try:
                occ = float(line[54:60]) if line[54:60].strip() else 1.0
                bfac = float(line[60:66]) if line[60:66].strip() else 0.0
                xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
                resnum = int(line[22:26])
                serial = int(line[6:11]) if line[6:11].strip().isdigit() else 0
            except ValueError:
                continue
            yield Atom(rec == "HETATM", serial, line[12:16].strip(), line[16].strip(), line[17:20].strip(), line[21],
                       resnum, line[26].strip(), xyz, occ, bfac, line[76:78].strip() if len(line) >= 78 else "")


def _split_cif_row(line):
    """Tokens of one mmCIF data line: whitespace separated, with '...' / "..." quoting."""
    out, i, n = [], 0, len(line)
    while i < n:
        c = line[i]
        if c.isspace():
            i += 1
        elif c in "'\"":
            j = i + 1
            while j < n and not (line[j] == c and (j + 1 == n or line[j + 1].isspace())):
                j += 1
            out.append(line[i + 1:j]); i = j + 1
        else:
            j = i
            while j < n and not line[j].isspace():
                j += 1
            out.append(line[i:j]); i = j
    return out


def _mmcif_atoms(path):
    """The `_atom_site` loop of an mmCIF file, first model, with the author (auth_*) identifiers PDB files carry —
    chain = auth_asym_id, residue number = auth_seq_id, insertion code = pdbx_PDB_ins_code."""
    cols, rows, in_loop, in_site = [], [], False, False
    with open(path) as fh:
        for raw in fh:
            line = raw.rstrip("\n")
            st = line.strip()
            if st == "loop_":
                if in_site and rows:
                    break
                in_loop, in_site, cols = True, False, []
                continue
            if in_loop and st.startswith("_"):
                if st.startswith("_atom_site."):
                    in_site = True
                    cols.append(st.split()[0][len("_atom_site."):])
                elif in_site:
                    break
                continue
            if in_site:
                if not st or st.startswith("#") or st.startswith("data_"):
                    if rows:
                        break
                    continue
                rows.append(_split_cif_row(line))
            elif st and not st.startswith("_"):
                in_loop = False
    if not cols:
        raise ValueError(f"{path}: no _atom_site loop found")
    ix = {c: i for i, c in enumerate(cols)}
    need = ("Cartn_x", "Cartn_y", "Cartn_z")
    if any(c not in ix for c in need):
        raise ValueError(f"{path}: _atom_site lacks coordinates")
    get = lambda row, key, alt=None, default="": (row[ix[key]] if key in ix else (row[ix[alt]] if alt and alt in ix else default))
    nul = lambda v: "" if v in (".", "?") else v
    first_model = None
    for row in rows:
        if len(row) < len(cols):
            continue
        model = get(row, "pdbx_PDB_model_num", default="1")
        if first_model is None:
            first_model = model
        if model != first_model:
            break
        try:
            xyz = (float(row[ix["Cartn_x"]]), float(row[ix["Cartn_y"]]), float(row[ix["Cartn_z"]]))
            resnum = int(get(row, "auth_seq_id", "label_seq_id", "0"))
            occ = float(nul(get(row, "occupancy", default="1.0")) or 1.0)
            bfac = float(nul(get(row, "B_iso_or_equiv", default="0.0")) or 0.0)
            serial = int(get(row, "id", default="0"))
        except ValueError:
            continue
        yield Atom(get(row, "group_PDB", default="ATOM") == "HETATM", serial, get(row, "auth_atom_id", "label_atom_id"),
                   nul(get(row, "label_alt_id")), get(row, "auth_comp_id", "label_comp_id"), get(row, "auth_asym_id", "label_asym_id"),
                   resnum, nul(get(row, "pdbx_PDB_ins_code")), xyz, occ, bfac, nul(get(row, "type_symbol")))


def read_atoms(path, chains=None, normalize_legacy_names=False, chain_order=None, ca_residues=None):
    """Coordinate records the reference's parse_PDB keeps before any polymer logic (data_utils.py:232-238): first model,
    altloc blank or 'A' (prody's default), occupancy > 0, optionally only the given chains.  `.cif` / `.mmcif` files go
    through the mmCIF reader.  normalize_legacy_names maps pre-remediation nucleic atom names (O1P, O2P, C1*, ...) to the
    current ones; the reference (prody) does not, so it is off by default.  chain_order (a list) receives the chain ids in
    order of first appearance over EVERY parsed record — before the occupancy / chain filters, hetero atoms and waters
    included: that is the numbering prody's getChindices() reports (data_utils.py:303).  ca_residues (a set) receives the
    (chain, number, insertion code) keys of the residues that have an atom named CA, likewise over every parsed record."""
    low = str(path).lower()
    src = _mmcif_atoms(path) if low.endswith((".cif", ".mmcif")) else _pdb_atoms(path)
    chains = set(chains) if chains else None
    records = [a for a in src if a.altloc in ("", "A")]
    for a in records:                                   # properties of the WHOLE parsed structure, like prody's flags / hierarchy
        if chain_order is not None and a.chain not in chain_order:
            chain_order.append(a.chain)
        if ca_residues is not None and a.name == "CA":
            ca_residues.add((a.chain, a.resnum, a.icode))
    for a in records:
        if a.occ <= 0 or (chains and a.chain not in chains):
            continue
        if normalize_legacy_names:
            a.name = LEGACY_ATOM_NAMES.get(a.name.replace("*", "'"), a.name.replace("*", "'"))
        yield a


def parse_pdb(path, chains=None, parse_na_only=False, na_shared_tokens=True, load_residues_with_missing_atoms=False,
              normalize_legacy_names=False):
    """parse_PDB of the reference (data_utils.py:84-405) for PDB and mmCIF files -> dict of numpy arrays (no batch
    dimension) + 'chain_letters', 'icodes', 'na_chain_letters', and the atom records the backbone writer needs
    ('backbone_atoms': protein N/CA/C/O and the 12 nucleic backbone atoms; 'other_atoms': neither polymer nor water)."""
    atom_index = {a: i for i, a in enumerate(spec.ATOM_TYPES)}
    rti = spec.restype_to_int(na_shared_tokens)
    residues, order = {}, []        # key -> {"resname", "kind", "atoms": {name: xyz}}
    chain_order = []
    backbone_atoms, other_atoms = [], []
    bb_names = {"protein": set(PROTEIN_BB), "nucleic": set(RNA_BB)}
    ca_residues = set()
    for at in read_atoms(path, chains, normalize_legacy_names, chain_order, ca_residues):
        name, resname, chain = at.name, at.resname, at.chain
        # prody's `protein` flag: a qualifying residue name AND an atom named CA in the residue (manual, Atom Flags)
        kind = "protein" if (resname in PROTEIN_NAMES and (chain, at.resnum, at.icode) in ca_residues) else \
            ("nucleic" if resname in NUCLEIC_NAMES else None)
        if parse_na_only and kind != "nucleic":
            continue
        if kind is None:
            if resname not in WATER_NAMES:
                other_atoms.append(at)
            continue
        if name in bb_names[kind]:
            backbone_atoms.append(at)
        key = (chain, at.resnum, at.icode)
        r = residues.get(key)
        if r is None:
            r = residues[key] = {"resname": resname, "kind": kind, "atoms": {}, "ref": False}
        r["atoms"][name] = at.xyz
        if name == ("CA" if kind == "protein" else "C1'") and not r["ref"]:
            r["ref"] = True
            order.append(key)                       # residue order = order of reference atoms
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
        "backbone_atoms": backbone_atoms, "other_atoms": other_atoms,
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


def _pdb_line(het, serial, name, resname, chain, resnum, icode, xyz, occ, bfac, element):
    nm = name if len(name) >= 4 else " " + name          # PDB columns 13-16: short names start in column 14
    return "%-6s%5d %-4s %3s %1s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" % (
        "HETATM" if het else "ATOM", serial % 100000, nm, resname[-3:], (chain or " ")[0], resnum, icode or " ",
        xyz[0], xyz[1], xyz[2], occ, bfac, element or name[0])


def write_backbone_pdb(path, parsed, resnames3, bfactors):
    """The design output of run.py:475-491: the parsed BACKBONE atoms (protein N, CA, C, O; the 12 nucleic backbone atoms)
    with every residue renamed to its designed 3-letter name and the B-factor column carrying the per-residue confidence,
    followed by the non-polymer, non-water atoms of the input ('backbone + other_atoms').  resnames3 / bfactors are per
    residue, in the order of parsed['chain_letters'] / ['R_idx'].  The reference writes this with prody's writePDB; this
    writer emits standard fixed-column ATOM / HETATM records (prody absent: byte-for-byte parity with its writer unpinned)."""
    by_res = {}
    for i, (c, r) in enumerate(zip(parsed["chain_letters"], parsed["R_idx"].tolist())):
        by_res.setdefault((c, int(r)), i)             # the reference selects "chain c and resnum r": insertion codes share it
    # PDB has ONE column for the chain id: mmCIF auth_asym_id values longer than that would collide once truncated
    long_ids = sorted({at.chain for at in list(parsed["backbone_atoms"]) + list(parsed["other_atoms"]) if at.chain and len(at.chain) > 1})
    if long_ids:
        raise ValueError(f"write_backbone_pdb: chain ids {long_ids} do not fit the PDB format's one-character chain column "
                         "(mmCIF input with multi-character auth_asym_id): rename the chains before asking for --output_pdbs")
    lines, serial = [], 1
    for at in parsed["backbone_atoms"]:
        i = by_res.get((at.chain, at.resnum))
        rn = resnames3[i] if i is not None else at.resname
        bf = float(bfactors[i]) if i is not None else at.bfac
        lines.append(_pdb_line(at.het, serial, at.name, rn, at.chain, at.resnum, at.icode, at.xyz, at.occ, bf, at.element))
        serial += 1
    for at in parsed["other_atoms"]:
        # run.py:336-338: other_atoms.setBetas(other_bfactors * 0.0) — the non-polymer atoms are written with B-factor 0.00
        lines.append(_pdb_line(at.het, serial, at.name, at.resname, at.chain, at.resnum, at.icode, at.xyz, at.occ, 0.0, at.element))
        serial += 1
    lines.append("END")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def write_mmcif(path, X, X_m, resnames, chain_letters, R_idx, icodes=None, name="x"):
    """Minimal mmCIF with one `_atom_site` loop (tests and round trips; the column set wwPDB files carry)."""
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id",
            "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "auth_seq_id",
            "auth_comp_id", "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    out = [f"data_{name}", "#", "loop_"] + ["_atom_site." + c for c in cols]
    serial = 1
    for i in range(X.shape[0]):
        for a, an in enumerate(spec.ATOM_TYPES):
            if not X_m[i, a]:
                continue
            q = f'"{an}"' if "'" in an else an
            ic = (icodes[i] if icodes else "") or "?"
            out.append(" ".join(["ATOM", str(serial), an[0], q, ".", resnames[i], chain_letters[i], "1", str(i + 1), ic,
                                 "%.3f" % X[i, a, 0], "%.3f" % X[i, a, 1], "%.3f" % X[i, a, 2], "1.00", "0.00", str(int(R_idx[i])),
                                 resnames[i], chain_letters[i], q, "1"]))
            serial += 1
    out.append("#")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
