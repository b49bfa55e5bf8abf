"""Command-line front end with the reference's flags and output files.

Mirrors the contract of /root/reference/inference/run.py — the flags of :519-555, the mode defaults of
:559-583, ``seqs/<name>.fa`` with the header lines of :445-455 / :501-511 and ``specificity/<name>.npz`` with the
keys of :426-443, ``backbones/<name>_<id>.pdb`` (:475-491) — on top of ``na_mpnn_amd.model.ProteinMPNN`` and
``na_mpnn_amd.pdbio`` (no prody; PDB and mmCIF input).

    python -m na_mpnn_amd.cli --mode design --pdb_path in.pdb --out_folder out/ [--checkpoint_na_mpnn ckpt.pt]

``--random_init_seed N`` replaces the checkpoint by seeded synthetic weights (the trained checkpoints are not
part of the reference tree).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys

import numpy as np
import torch

from . import pdbio, spec, synth
from .model import ProteinMPNN


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    a = p.add_argument
    a("--model_type", type=str, default="na_mpnn")
    a("--checkpoint_na_mpnn", type=str, default=None)
    a("--random_init_seed", type=int, default=None, help="use seeded synthetic weights instead of a checkpoint")
    a("--out_folder", type=str, required=True)
    a("--file_ending", type=str, default="")
    a("--pdb_path", type=str, default="")
    a("--fixed_pos_by_pdb", type=str, default="")
    a("--zero_indexed", type=int, default=0)
    a("--seed", type=int, default=0)
    a("--batch_size", type=int, default=None)
    a("--number_of_batches", type=int, default=1)
    a("--temperature", type=float, default=None)
    a("--save_stats", type=int, default=0)
    a("--chains_to_design", type=str, default=None)
    a("--omit_AA", type=str, default="X")
    a("--fixed_residues", type=str, default="")
    a("--redesigned_residues", type=str, default="")
    a("--parse_these_chains_only", type=str, default="")
    a("--bias_AA", type=str, default="")
    a("--pair_bias_AA", type=str, default="", help="pair bias for sequence neighbours, e.g. 'KK:-10.0,KE:-10.0'")
    a("--symmetry_residues", type=str, default="", help="tied residues, e.g. 'A12,A13,A14|C2,C3'")
    a("--symmetry_weights", type=str, default="", help="weights matching --symmetry_residues, e.g. '1.0,1.0,1.0|-1.0,2.0'")
    a("--na_shared_tokens", type=int, default=1)
    a("--parse_na_only", type=int, default=0)
    a("--design_na_only", type=int, default=0)
    a("--k_neighbors", type=int, default=None)
    a("--catch_failed_inferences", type=int, default=0)
    a("--output_pdbs", type=int, default=1, help="1 - write backbones/<name>_<id>.pdb with the designed residue names")
    a("--output_sequences", type=int, default=1)
    a("--output_specificity", type=int, default=0)
    a("--load_residues_with_missing_atoms", type=int, default=0)
    a("--mode", type=str, default=None)
    a("--device", type=str, default="cuda:0")
    a("--forced_draws_npz", type=str, default="", help="testing: npz with 'randn' [batches*batch_size, L] (decoding-order noise) and "
      "'S_forced' [batches*batch_size, L] (the tokens every draw is forced to) — makes a run reproducible across devices")
    return p


def apply_mode_defaults(args):
    """run.py:559-583."""
    def need(v):
        if v is None:
            print("Choose mode from: design, specificity")
            sys.exit()
        return v
    m = {"design": ("./models/design_model/s_19137.pt", 1, 0.1), "specificity": ("./models/specificity_model/s_70114.pt", 30, 0.6)}.get(args.mode)
    if args.checkpoint_na_mpnn is None and args.random_init_seed is None:
        args.checkpoint_na_mpnn = need(m and m[0])
    if args.batch_size is None:
        args.batch_size = need(m and m[1])
    if args.temperature is None:
        args.temperature = need(m and m[2])
    return args


def seq_string(tokens, rna_flag, int_to_str, dna_to_rna, chain_letters):
    chars = [dna_to_rna.get(int_to_str[int(t)], int_to_str[int(t)]) if rna_flag[i] == 1 else int_to_str[int(t)]
             for i, t in enumerate(tokens)]
    out, seen = [], []
    for c in chain_letters:
        if c not in seen:
            seen.append(c)
    for c in sorted(seen):
        out.append("".join(ch for ch, cl in zip(chars, chain_letters) if cl == c))
    return "/".join(out)


def make_pair_bias(chain_labels, R_idx, pair_bias_AA):
    """[1,L,33,L,33] bias between sequence neighbours on one chain (semantics of data_utils.py:7-16):
    out[0,i,a,i+1,b] = pair_bias_AA[a,b] and out[0,i+1,a,i,b] = pair_bias_AA[b,a] where R_idx[i+1]-R_idx[i]==1."""
    L = R_idx.shape[0]
    adj = ((R_idx[1:] - R_idx[:-1]) == 1) & (chain_labels[1:] == chain_labels[:-1])
    out = torch.zeros(1, L, pair_bias_AA.shape[0], L, pair_bias_AA.shape[1], dtype=torch.float32, device=pair_bias_AA.device)
    i = torch.nonzero(adj)[:, 0]
    out[0, i, :, i + 1, :] = pair_bias_AA
    out[0, i + 1, :, i, :] = pair_bias_AA.t()
    return out


def main(argv=None):
    args = apply_mode_defaults(build_parser().parse_args(argv))
    if args.model_type != "na_mpnn":
        print("Choose --model_type flag from currently available models")
        sys.exit()
    seed = args.seed if args.seed else int(np.random.randint(0, high=99999, size=1, dtype=int)[0])
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    device = torch.device(args.device)
    shared = bool(args.na_shared_tokens)
    rti = spec.restype_to_int(shared)
    alphabet = [spec.RESTYPE_3TO1[r] for r in spec.RESTYPES]
    str_to_int = {spec.RESTYPE_3TO1[k]: v for k, v in rti.items()}
    int_to_str = {}
    for k, v in str_to_int.items():
        int_to_str.setdefault(v, k)
    dna_to_rna = {spec.RESTYPE_3TO1[d]: spec.RESTYPE_3TO1[r] for d, r in
                  (("DA", "A"), ("DC", "C"), ("DG", "G"), ("DT", "U"), ("DX", "RX"))} if shared else {}

    k_neighbors = args.k_neighbors if args.k_neighbors is not None else 32          # run.py:176-182
    model = ProteinMPNN(node_features=128, edge_features=128, hidden_dim=128, num_encoder_layers=3, num_decoder_layers=3,
                        k_neighbors=k_neighbors, model_type=args.model_type, vocab=33, num_letters=33,
                        atom_dict=spec.atom_dict(), restype_to_int=rti, polytype_to_int=spec.polytype_to_int())
    if args.random_init_seed is not None:
        ckpt_name = f"random_init_seed_{args.random_init_seed}"
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(args.random_init_seed).items()})
    else:
        ckpt_name = args.checkpoint_na_mpnn
        model.load_state_dict(torch.load(ckpt_name, map_location="cpu", weights_only=False)["model_state_dict"])
    model.to(device).eval()

    bias_AA = torch.zeros(33, device=device)
    if args.bias_AA:
        for item in args.bias_AA.split(","):
            aa, val = item.split(":")
            bias_AA[str_to_int[aa]] = float(val)
    pair_bias_AA = None
    if args.pair_bias_AA:                                                           # run.py:215-223
        pair_bias_AA = torch.zeros(33, 33, device=device)
        for item in args.pair_bias_AA.split(","):
            pair, val = item.split(":")
            pair_bias_AA[str_to_int[pair[0]], str_to_int[pair[1]]] = float(val)
    omit_list = args.omit_AA + ("bdhuy" if shared else "")
    omit_AA = torch.tensor([float(c in omit_list) for c in alphabet], device=device)

    base = args.out_folder if args.out_folder.endswith("/") else args.out_folder + "/"
    os.makedirs(base + "seqs", exist_ok=True)
    if args.output_pdbs:
        os.makedirs(base + "backbones", exist_ok=True)
    if args.output_specificity:
        os.makedirs(base + "specificity", exist_ok=True)
    if args.fixed_pos_by_pdb:
        with open(args.fixed_pos_by_pdb) as fh:
            fixed_pos_by_pdb = json.load(fh)
            fixed_pos_by_pdb = {k: (v.split() if isinstance(v, str) else v) for k, v in fixed_pos_by_pdb.items()}
    else:
        fixed_pos_by_pdb = {args.pdb_path: args.fixed_residues.split()}

    for pdb, fixed_residues in fixed_pos_by_pdb.items():
        name = os.path.basename(pdb)
        if name[-4:] in (".pdb", ".cif"):
            name = name[:-4]
        try:
            run_one(args, model, pdb, name, fixed_residues, device, seed, ckpt_name, bias_AA, omit_AA, int_to_str, dna_to_rna,
                    rti, base, pair_bias_AA)
        except Exception as e:                     # run.py:585-617
            if not args.catch_failed_inferences:
                raise
            os.makedirs(base + "failed_inferences", exist_ok=True)
            with open(base + "failed_inferences/" + name + ".txt", "w") as fh:
                fh.write(repr(e))


def run_one(args, model, pdb, name, fixed_residues, device, seed, ckpt_name, bias_AA, omit_AA, int_to_str, dna_to_rna, rti, base,
            pair_bias_AA=None):
    P = pdbio.parse_pdb(pdb, chains=list(args.parse_these_chains_only) or None, parse_na_only=bool(args.parse_na_only),
                        na_shared_tokens=bool(args.na_shared_tokens),
                        load_residues_with_missing_atoms=bool(args.load_residues_with_missing_atoms))
    L = len(P["S"])
    encoded = [f"{c}{r}{ic}" for c, r, ic in zip(P["chain_letters"], P["R_idx"].tolist(), P["icodes"])]
    encoded_dict = dict(zip(encoded, range(L)))
    fixed_positions = np.array([int(e not in fixed_residues) for e in encoded], np.int32)
    if args.redesigned_residues:
        red = args.redesigned_residues.split()
        redesigned = np.array([int(e not in red) for e in encoded], np.int32)
    else:
        redesigned = np.zeros(L, np.int32)
    chains = args.chains_to_design.split(",") if isinstance(args.chains_to_design, str) else P["chain_letters"]
    if args.design_na_only:
        chains = [c for c in chains if c in P["na_chain_letters"]]
    chain_mask = np.array([c in chains for c in P["chain_letters"]], np.int32) * fixed_positions * (1 - redesigned)

    if args.symmetry_residues:                                                      # run.py:313-332
        sym_res = [[encoded_dict[t] for t in grp.split(",")] for grp in args.symmetry_residues.split("|")]
        sym_w = ([[float(v) for v in grp.split(",")] for grp in args.symmetry_weights.split("|")] if args.symmetry_weights
                 else [[1.0] * len(grp) for grp in sym_res])
    else:
        sym_res, sym_w = [[]], [[]]

    with torch.no_grad():
        fd = pdbio.to_feature_dict(P, chain_mask, device)
        fd.update({"batch_size": args.batch_size, "temperature": args.temperature,
                   "bias": (-1e8 * omit_AA[None, None, :] + bias_AA).repeat(1, L, 1),
                   "symmetry_residues": sym_res, "symmetry_weights": sym_w})
        if pair_bias_AA is not None:
            fd["pair_bias"] = make_pair_bias(fd["chain_labels"][0], fd["R_idx"][0], pair_bias_AA)
        S_l, lp_l, sp_l, loss_l, lpr_l = [], [], [], [], []
        cmask = (fd["mask"] * fd["chain_mask"]).float()
        forced = np.load(args.forced_draws_npz) if args.forced_draws_npz else None
        model.sample_check_walk = True      # verify the persistent level walk's barriers per design call; falls back to per-level launches
        for ib in range(args.number_of_batches):
            fd["randn"] = torch.randn(args.batch_size, L, device=device)
            if forced is not None:
                rows = slice(ib * args.batch_size, (ib + 1) * args.batch_size)
                fd["randn"] = torch.from_numpy(forced["randn"][rows]).to(device)
                fd["S_forced"] = torch.from_numpy(forced["S_forced"][rows]).to(device)
            out = model.sample(fd)
            onehot = torch.nn.functional.one_hot(out["S"], 33)
            lpr = -(onehot * out["log_probs"]).sum(-1)                                   # get_score, data_utils.py:36-52
            lpr_l.append(lpr)
            loss_l.append((lpr * cmask).sum(-1) / (cmask.sum(-1) + 1e-8))
            S_l.append(out["S"]); lp_l.append(out["log_probs"]); sp_l.append(out["sampling_probs"])
        S_stack, sp_stack, loss_stack, lpr_stack = torch.cat(S_l), torch.cat(sp_l), torch.cat(loss_l), torch.cat(lpr_l)
        rec = ((fd["S"][:1] == S_stack) * cmask).sum(-1) / cmask.sum(-1)               # get_seq_rec, data_utils.py:18-30

    rna_flag = P["rna_mask_for_token_conversion"]
    entries = ['>{}, T={}, seed={}, num_res={}, batch_size={}, number_of_batches={}, model_path={}\n{}'.format(
        name, args.temperature, seed, (fd["mask"] * fd["chain_mask"]).sum().cpu().numpy(), args.batch_size, args.number_of_batches, ckpt_name,
        seq_string(P["S"], rna_flag, int_to_str, dna_to_rna, P["chain_letters"]))]
    for ix in range(S_stack.shape[0]):
        conf = np.format_float_positional(np.exp(-loss_stack[ix].cpu().numpy()), unique=False, precision=4)
        srec = np.format_float_positional(rec[ix].cpu().numpy(), unique=False, precision=4)
        entries.append('>{}, id={}, T={}, seed={}, overall_confidence={} seq_rec={}\n{}'.format(
            name, ix if args.zero_indexed else ix + 1, args.temperature, seed, conf, srec,
            seq_string(S_stack[ix].cpu().numpy(), rna_flag, int_to_str, dna_to_rna, P["chain_letters"])))
        if args.output_pdbs:                                                          # run.py:475-491
            one_to_three = {v: k for k, v in spec.RESTYPE_3TO1.items()}
            chars = [dna_to_rna.get(int_to_str[int(t)], int_to_str[int(t)]) if rna_flag[i] == 1 else int_to_str[int(t)]
                     for i, t in enumerate(S_stack[ix].cpu().numpy())]
            lp_res = lpr_stack[ix].cpu().numpy()
            pdbio.write_backbone_pdb(base + "backbones/" + name + "_" + str(ix if args.zero_indexed else ix + 1) + ".pdb" + args.file_ending,
                                     P, [one_to_three[c] for c in chars], np.exp(-lp_res) * (lp_res > 0.01).astype(np.float32))
    if args.output_sequences:
        with open(base + "seqs/" + name + ".fa" + args.file_ending, "w") as fh:
            fh.write("\n".join(entries))
    if args.output_specificity:
        np.savez(os.path.join(base, "specificity", name + ".npz"),
                 predicted_ppm=np.mean(sp_stack.cpu().numpy().astype(np.float64), axis=0),
                 true_sequence=P["S"].astype(np.int64), chain_labels=P["chain_labels"], mask=P["mask"],
                 protein_mask=P["protein_mask"], dna_mask=P["dna_mask"], rna_mask=P["rna_mask"],
                 encoded_residues=encoded, encoded_residues_dict=encoded_dict, restype_to_int=rti)
    if args.save_stats:
        os.makedirs(base + "stats", exist_ok=True)
        torch.save({"generated_sequences": S_stack.cpu(), "sampling_probs": sp_stack.cpu(), "log_probs": torch.cat(lp_l).cpu(),
                    "native_sequence": fd["S"][0].cpu(), "mask": fd["mask"][0].cpu(), "chain_mask": fd["chain_mask"][0].cpu(),
                    "seed": seed, "temperature": args.temperature}, base + "stats/" + name + ".pt")


if __name__ == "__main__":
    main()
