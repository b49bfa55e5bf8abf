"""CPU restatement of the reference's input parsing (`parse_PDB` + `featurize`, /root/reference/inference/data_utils.py:84-439)
for model_type "na_mpnn" — TEST INFRASTRUCTURE ONLY: imported by tests/, oracle/make_cli_fixture.py and
oracle/make_pdb_fixture.py, never by the product (`na_mpnn_amd/pdbio.py` is written independently of this file and vice versa).

The reference parses with ProDy (README.md:15 pins ProDy v2.6.1), which is not installed in this image and cannot be
(no network).  `parse_PDB` below follows the reference statement by statement; the handful of ProDy calls it makes are
served by `_Atoms`, a restatement of their PUBLISHED behaviour (ProDy manual: parsePDB, Atom Flags, Atom Selections, HierView):

* parsePDB: ATOM / HETATM records of the first model; alternate locations blank or 'A' (`altloc='A'` default); fixed columns;
* select('occupancy > 0'), 'chain X or chain Y', 'name N or name CA ...', boolean and / or / not over flags;
* flag `protein`: residue name in ProDy's standard + non-standard amino-acid tables AND the residue has an atom named CA
  (manual, Atom Flags: "Residue must also have an atom named CA in addition to having a qualifying residue name");
* flag `nucleic`: nucleobase / nucleotide / nucleoside residue names; flag `water`: the water residue names;
* a selection that matches nothing is `None`;
* getChindices: index of the atom's chain in the hierarchical view of the WHOLE parsed structure — chains numbered by first
  appearance of (segment name, chain id) over all atoms, hetero atoms and waters included — not of the selection.

Parity: **unpinned against ProDy itself** (it cannot run here); pinned against the product by construction of two independent
implementations that must agree on the synthetic CLI fixture and on the reference's example files
(tests/test_pdbio.py::test_reference_examples_match_the_oracle_arrays, fixtures written by oracle/make_pdb_fixture.py).
"""
from __future__ import annotations

import numpy as np

# ProDy 2.6.1 atomic/flags.py tables (restated from the manual)
STDAA = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP",
         "TYR", "VAL"}
NONSTDAA = {"ASX", "GLX", "CSO", "HIP", "HSD", "HSE", "HSP", "MSE", "SEC", "SEP", "TPO", "PTR", "XLE", "XAA"}
NUCLEOBASE = {"GUN", "ADE", "CYT", "THY", "URA"}
NUCLEOTIDE = {"DA", "DC", "DG", "DT", "DU", "A", "C", "G", "T", "U"}
NUCLEOSIDE = {"AMP", "ADP", "ATP", "CDP", "CTP", "GMP", "GDP", "GTP", "TMP", "TTP", "UMP", "UDP", "UTP"}
WATER = {"HOH", "DOD", "WAT", "TIP3", "H2O", "OH2", "TIP", "TIP2", "TIP4", "SOL"}


class _Atoms:
    """The slice of ProDy's AtomGroup / Selection API that parse_PDB uses, over parallel numpy arrays."""
    FIELDS = ("name", "resname", "chid", "resnum", "icode", "coords", "occ", "element", "het", "serial", "beta", "chindex", "protein",
              "nucleic", "water")

    def __init__(self, **f):
        for k in self.FIELDS:
            setattr(self, k, f[k])

    def __len__(self):
        return len(self.name)

    def pick(self, keep):
        keep = np.asarray(keep, bool)
        if not keep.any():
            return None                                     # ProDy: a selection matching nothing is None
        return _Atoms(**{k: getattr(self, k)[keep] for k in self.FIELDS})

    # the getters the reference calls
    def getCoords(self): return self.coords
    def getResnums(self): return self.resnum
    def getChids(self): return self.chid
    def getIcodes(self): return self.icode
    def getResnames(self): return self.resname
    def getChindices(self): return self.chindex
    def getElements(self): return self.element


def parsePDB(path):
    """ProDy parsePDB(path) with its defaults, as far as parse_PDB depends on it."""
    rows = []
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break                                       # first model only
            if rec not in ("ATOM  ", "HETATM"):
                continue
            alt = line[16]
            if alt not in (" ", "A"):
                continue
            try:
                xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
                resnum = int(line[22:26])
            except ValueError:
                continue
            def num(s):
                try:
                    return float(s)
                except ValueError:
                    return 0.0                              # ProDy warns and leaves the zero the array was created with
            rows.append((line[12:16].strip(), line[17:21].strip(), line[21].strip(), resnum, line[26].strip(), xyz, num(line[54:60]),
                         line[76:78].strip() if len(line) >= 78 else "", rec == "HETATM", line[6:11].strip(), num(line[60:66])))
    n = len(rows)
    name = np.array([r[0] for r in rows], dtype=object)
    resname = np.array([r[1] for r in rows], dtype=object)
    chid = np.array([r[2] for r in rows], dtype=object)
    resnum = np.array([r[3] for r in rows], dtype=np.int64)
    icode = np.array([r[4] for r in rows], dtype=object)
    # hierarchical view: chains by first appearance of the chain id (no segment names in PDB files read this way)
    order, chindex = {}, np.zeros(n, np.int64)
    for i, c in enumerate(chid):
        chindex[i] = order.setdefault(c, len(order))
    # residues = runs of atoms with the same (chain, resnum, icode); `protein` needs a qualifying name AND a CA atom in the residue
    has_ca = {}
    for i in range(n):
        key = (chid[i], int(resnum[i]), icode[i])
        has_ca[key] = has_ca.get(key, False) or name[i] == "CA"
    aa = STDAA | NONSTDAA
    protein = np.array([resname[i] in aa and has_ca[(chid[i], int(resnum[i]), icode[i])] for i in range(n)], bool)
    nucleic = np.array([r in (NUCLEOBASE | NUCLEOTIDE | NUCLEOSIDE) for r in resname], bool)
    water = np.array([r in WATER for r in resname], bool)
    return _Atoms(name=name, resname=resname, chid=chid, resnum=resnum, icode=icode, coords=np.array([r[5] for r in rows], np.float64).reshape(n, 3),
                  occ=np.array([r[6] for r in rows], np.float64), element=np.array([r[7] for r in rows], dtype=object),
                  het=np.array([r[8] for r in rows], bool), serial=np.array([r[9] for r in rows], dtype=object),
                  beta=np.array([r[10] for r in rows], np.float64), chindex=chindex, protein=protein, nucleic=nucleic, water=water)


RESTYPES = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR",
            "VAL", "UNK", "DA", "DC", "DG", "DT", "DX", "A", "C", "G", "U", "RX", "MAS", "PAD"]                 # data_utils.py:189-223
ATOM_TYPES = ["N", "CA", "C", "O", "OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"]   # :156-158
POLYTYPES = ["PP", "DNA", "RNA", "UNK", "MAS", "PAD"]                                                         # :145-152


def get_aligned_coordinates(macromolecule_atoms, reference_atom_dict, atom_name):
    """data_utils.py:56-82."""
    sel = macromolecule_atoms.pick(macromolecule_atoms.name == atom_name)
    xyz = np.zeros([len(reference_atom_dict), 3], np.float32)
    m = np.zeros([len(reference_atom_dict)], np.int32)
    if sel is not None:
        for i in range(len(sel)):
            code = sel.chid[i] + "_" + str(sel.resnum[i]) + "_" + sel.icode[i]
            if code in reference_atom_dict:
                xyz[reference_atom_dict[code], :] = sel.coords[i]
                m[reference_atom_dict[code]] = 1
    return xyz, m


def parse_PDB(input_path, chains=(), parse_na_only=False, na_shared_tokens=False, load_residues_with_missing_atoms=0):
    """data_utils.py:84-405 for model_type == "na_mpnn", parse_all_atoms=False -> dict of numpy arrays and lists with the
    reference's keys (X, X_m, mask, R_idx, chain_labels, chain_letters, na_chain_letters, protein/dna/rna masks,
    rna_mask_for_token_conversion, R_polymer_type, S, Y, Y_t, Y_m, chain_list) + 'icodes', 'backbone', 'other_atoms'."""
    polytype_to_int = dict(zip(POLYTYPES, range(len(POLYTYPES))))
    atom_order = dict(zip(ATOM_TYPES, range(len(ATOM_TYPES))))
    protein_bb = ["N", "CA", "C", "O"]                                                                      # :165-168
    dna_bb = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "C1'"]
    rna_bb = ["OP1", "OP2", "P", "O5'", "C5'", "C4'", "O4'", "C3'", "O3'", "C2'", "O2'", "C1'"]
    restype_to_int = dict(zip(RESTYPES, range(len(RESTYPES))))
    if na_shared_tokens:                                                                                    # :225-230
        for r, d in (("A", "DA"), ("C", "DC"), ("G", "DG"), ("U", "DT"), ("RX", "DX")):
            restype_to_int[r] = restype_to_int[d]
    atoms = parsePDB(input_path)
    atoms = atoms.pick(atoms.occ > 0)                                                                       # :238
    if chains:
        atoms = atoms.pick(np.isin(atoms.chid, list(chains)))                                               # :239-243
    if parse_na_only:
        atoms = atoms.pick(atoms.nucleic)                                                                   # :245-246
    macro = atoms.pick(atoms.protein | atoms.nucleic)                                                       # :248
    backbone = atoms.pick((atoms.protein & np.isin(atoms.name, protein_bb)) | (atoms.nucleic & np.isin(atoms.name, rna_bb)))   # :250-258
    other_atoms = atoms.pick(~atoms.protein & ~atoms.nucleic & ~atoms.water)                                # :259
    ref = macro.pick((macro.protein & (macro.name == "CA")) | (macro.nucleic & (macro.name == "C1'")))      # :262-269
    na_ref = macro.pick(macro.nucleic & (macro.name == "C1'"))                                              # :274
    reference_atom_dict = {}
    for i in range(len(ref)):
        reference_atom_dict[ref.chid[i] + "_" + str(ref.resnum[i]) + "_" + ref.icode[i]] = i               # :276-279
    L = len(reference_atom_dict)
    xyz, xyz_m = np.zeros([L, 16, 3], np.float32), np.zeros([L, 16], np.int32)
    for a in ATOM_TYPES:                                                                                    # :283-286
        xyz[:, atom_order[a]], xyz_m[:, atom_order[a]] = get_aligned_coordinates(macro, reference_atom_dict, a)
    chain_labels = np.array(ref.getChindices(), dtype=np.int32)                                             # :303
    R_idx = np.array(ref.resnum, dtype=np.int32)
    S = ref.getResnames()
    idx = lambda names: [atom_order[a] for a in names]
    if load_residues_with_missing_atoms:                                                                    # :307-316
        protein_mask = np.array([r in RESTYPES[:21] for r in S], np.int32)
        dna_mask = np.array([r in RESTYPES[21:26] and not (r in RESTYPES[:21]) for r in S], np.int32)
        rna_mask = np.array([r in RESTYPES[26:31] for r in S], np.int32)
    else:                                                                                                   # :317-322
        protein_mask = np.prod(xyz_m[:, idx(protein_bb)], axis=-1)
        rna_mask = np.prod(xyz_m[:, idx(rna_bb)], axis=-1)
        dna_mask = np.prod(xyz_m[:, idx(dna_bb)], axis=-1) - rna_mask
    rna_tok = xyz_m[:, atom_order["O2'"]]                                                                   # :324
    mask = protein_mask + dna_mask + rna_mask
    R_polymer_type = protein_mask * polytype_to_int["PP"] + dna_mask * polytype_to_int["DNA"] + rna_mask * polytype_to_int["RNA"] + \
        (1 - protein_mask - dna_mask - rna_mask) * polytype_to_int["UNK"]                                   # :328-331
    S_int = []
    for i, AA in enumerate(list(S)):                                                                        # :333-345
        unk = "UNK" if protein_mask[i] == 1 else "DX" if dna_mask[i] == 1 else "RX" if rna_mask[i] == 1 else "UNK"
        S_int.append(restype_to_int.get(AA, restype_to_int[unk]))
    chain_letters = list(ref.chid)
    na_chain_ids = [] if na_ref is None else [c for i, c in enumerate(chain_letters) if dna_mask[i] or rna_mask[i]]   # :381-388
    return {"X": xyz, "X_m": xyz_m, "mask": mask.astype(np.int32), "R_idx": R_idx, "chain_labels": chain_labels, "chain_letters": chain_letters,
            "na_chain_letters": na_chain_ids, "protein_mask": protein_mask.astype(np.int32), "dna_mask": dna_mask.astype(np.int32),
            "rna_mask": rna_mask.astype(np.int32), "rna_mask_for_token_conversion": rna_tok.astype(np.int32),
            "R_polymer_type": R_polymer_type.astype(np.int64), "S": np.array(S_int, np.int32), "icodes": list(ref.icode),
            "chain_list": sorted(set(chain_letters)), "backbone": backbone, "other_atoms": other_atoms}


def featurize_R_idx(R_idx):
    """data_utils.featurize (:407-417): equal consecutive residue numbers (insertion codes) are pushed apart."""
    out, count, prev = [], 0, -100000
    for r in list(R_idx):
        if prev == r:
            count += 1
        out.append(int(r) + count)
        prev = r
    return np.array(out, np.int64)


# ------------------------------------------------------------------------------------------------------------
# inference/run.py, output side (test infrastructure: the expected files of tests/golden/cli are written with these)
# ------------------------------------------------------------------------------------------------------------
RESTYPE_3_TO_1 = {"ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H", "ILE": "I",
                  "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W", "TYR": "Y", "VAL": "V",
                  "UNK": "X", "DA": "a", "DC": "c", "DG": "g", "DT": "t", "DX": "x", "A": "b", "C": "d", "G": "h", "U": "u", "RX": "y",
                  "MAS": "-", "PAD": "+"}                                                                  # run.py:68-102


def token_tables(na_shared_tokens):
    """run.py:104-128 -> restype_to_int, alphabet, restype_INTtoSTR, dna_char_to_rna_char."""
    restype_to_int = dict(zip(RESTYPES, range(len(RESTYPES))))
    int_to_restype = dict(zip(range(len(RESTYPES)), RESTYPES))
    alphabet = [RESTYPE_3_TO_1[int_to_restype[i]] for i in range(len(int_to_restype))]
    dna_char_to_rna_char = {}
    if na_shared_tokens:
        for r, d in (("A", "DA"), ("C", "DC"), ("G", "DG"), ("U", "DT"), ("RX", "DX")):
            restype_to_int[r] = restype_to_int[d]
            dna_char_to_rna_char[RESTYPE_3_TO_1[d]] = RESTYPE_3_TO_1[r]
    str_to_int = {RESTYPE_3_TO_1[k]: v for k, v in restype_to_int.items()}
    int_to_str = {}
    for k, v in str_to_int.items():
        if v not in int_to_str:
            int_to_str[v] = k
    return restype_to_int, alphabet, int_to_str, dna_char_to_rna_char


def encoded_residues(parsed):
    """run.py:250-256: chain letter + residue number + insertion code per residue."""
    return [str(c) + str(r) + ic for c, r, ic in zip(parsed["chain_letters"], list(parsed["R_idx"]), parsed["icodes"])]


def sequence_string(tokens, parsed, int_to_str, dna_char_to_rna_char):
    """run.py:393-405 / :482-499: one character per residue (RNA residues — rna_mask_for_token_conversion — take the RNA letter of a
    shared DNA token), chains in the order of the SORTED chain list (data_utils.py:396-401 mask_c), joined by '/'."""
    chars = []
    for i, AA in enumerate(list(tokens)):
        ch = int_to_str[int(AA)]
        chars.append(dna_char_to_rna_char.get(ch, ch) if parsed["rna_mask_for_token_conversion"][i] == 1 else ch)
    seq_np = np.array(chars)
    out = []
    for chain in parsed["chain_list"]:
        m = np.array([chain == item for item in parsed["chain_letters"]], dtype=bool)
        out += list(seq_np[m]) + ["/"]
    return "".join(out)[:-1]
