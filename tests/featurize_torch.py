"""Test helper: ProteinFeaturesNA.forward (inference/model_utils.py:528-593) as stock PyTorch-ROCm tensor ops, chunked over
residues.  An on-device cross-check of the HIP featuriser; NOT part of the product path (na_mpnn_amd never imports tests/)."""
import torch
from torch import nn

from na_mpnn_amd import spec


@torch.no_grad()
def featurize_torch(self, fd, chunk=128):
    """`self` is a na_mpnn_amd.model.ProteinMPNN on the device; returns V, E, E_idx (int64) like the reference."""
    X, mask = fd["X"], fd["mask"]
    ad = self.atom_dict
    X = self._noised_X(fd)
    B, L = X.shape[:2]
    K = int(min(self.k_neighbors, L))
    Ca = X[:, :, ad["CA"]]
    Cb = self._virtual(X[:, :, ad["N"]], Ca, X[:, :, ad["C"]], -0.58273431, 0.56802827, -0.54067466)
    C1p = X[:, :, ad["C1'"]]
    Nna = self._virtual(X[:, :, ad["O4'"]], C1p, X[:, :, ad["C2'"]], -0.56967352, 0.51055973, -0.53122153)
    X18 = torch.cat((X, Cb[:, :, None], Nna[:, :, None]), -2)
    dna_m, rna_m = self._na_masks(fd)
    M18 = torch.cat((fd["X_m"], fd["protein_mask"][:, :, None], (rna_m + dna_m)[:, :, None]), -1).float()
    P = Ca + X[:, :, ad[self.na_ref_atom]]
    mf = mask.float()
    m2 = mf[:, None, :] * mf[:, :, None]
    D = m2 * torch.sqrt(((P[:, None] - P[:, :, None]) ** 2).sum(-1) + 1e-6)
    D = D + (1. - m2) * D.max(-1, keepdim=True)[0]
    E_idx = torch.topk(D, K, dim=-1, largest=False)[1]
    del D, m2
    fp = self.features
    mu = torch.linspace(2., 22., spec.NUM_RBF, device=X.device)
    R_idx, chain = fd["R_idx"].long(), fd["chain_labels"].long()
    bidx = torch.arange(B, device=X.device)[:, None, None]
    Wpos, bpos = fp.embeddings.linear.weight, fp.embeddings.linear.bias
    E = torch.empty(B, L, K, self.edge_features, device=X.device)
    for i0 in range(0, L, chunk):
        i1 = min(L, i0 + chunk)
        j = E_idx[:, i0:i1]                                        # [B,c,K]
        Xj, Mj = X18[bidx, j], M18[bidx, j]                         # [B,c,K,18,3], [B,c,K,18]
        Dab = torch.sqrt(((X18[:, i0:i1, None, :, None, :] - Xj[:, :, :, None, :, :]) ** 2).sum(-1) + 1e-6)
        rbf = torch.exp(-(((Dab[..., None] - mu) / 1.25) ** 2))
        rbf = rbf * M18[:, i0:i1, None, :, None, None] * Mj[:, :, :, None, :, None]
        off = R_idx[:, i0:i1, None] - R_idx[bidx, j]
        same = (chain[:, i0:i1, None] == chain[bidx, j]).long()
        d = torch.clip(off + spec.MAX_REL, 0, 2 * spec.MAX_REL) * same + (1 - same) * (2 * spec.MAX_REL + 1)
        pos = Wpos.t()[d] + bpos                                   # one-hot @ W^T == column select
        feat = torch.cat((pos, rbf.reshape(B, i1 - i0, K, -1)), -1)
        E[:, i0:i1] = nn.functional.layer_norm(feat @ self.edge_weight18().t(), (self.edge_features,),
                                               fp.norm_edges.weight, fp.norm_edges.bias, 1e-5)
    return self._node_features(fd), E, E_idx

