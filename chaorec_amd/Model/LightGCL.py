"""LightGCL with the reference's surface (Model/LightGCL.py:16-247) -- LightGCN over the [U, I] normalised interaction matrix
and its transpose, contrasted with the same propagation through a rank-q SVD of that matrix -- `torch.spmm` family (SURVEY
8(f).1): the 2 L products of a forward are `chaorec_amd.sparse.mm` over the rectangular CSR and its transpose (:96-108), the
ranking is `ranking.gene_ranklist` over the layer-summed tables of the last training forward (:224-247).  The low-rank
channel is q-column GEMMs ([q, I] x [I, D], [U, q] x [q, D]: :131-139), the contrast two [B, n] score matrices: dense torch.

The truncated SVD (:40-46: `torch.svd_lowrank`, a randomised range finder with two power iterations) is run once at
construction with the SAME algorithm on the HIP SpMM -- Gaussian test matrix, QR after every product, the small [q, I] factor
decomposed exactly --; its draws come from torch's generator on the host, so the factors differ from a reference run's by the
randomness both have (the golden test loads the reference run's factors, and checks this construction against the exact
top-q triplets).

Same constructor, parameters in the reference's creation order; the adjacency counts a repeated interaction twice
(`csr_matrix` sums duplicates, :60-63), dropout is fixed at 0 there (:30) and the branch is kept."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse


def svd_lowrank(adj, r, niter=2):
    """torch.svd_lowrank(A, q, niter) (torch/_lowrank.py: Halko et al. 5.1 on the TALL orientation of A) for a graph.CSR A
    [m, n] through `sparse.mm`, with the Gaussian test matrix r [min(m, n), q] handed in.  -> (U [m, q], s [q], V [n, q])"""
    tall = adj if adj.n_rows >= adj.n_cols else adj.t()
    flat = tall.t()
    q = r.shape[1]
    mm = lambda a, x: sparse.mm(a, F.pad(x, (0, (-q) % 4)).contiguous())[:, :q]          # (the SpMM's rows are float4s: zero columns ride along)
    with torch.no_grad():
        Q = torch.linalg.qr(mm(tall, r.to(adj.val.device))).Q
        for _ in range(niter):
            Q = torch.linalg.qr(mm(flat, Q)).Q
            Q = torch.linalg.qr(mm(tall, Q)).Q
        Ub, s, Vh = torch.linalg.svd(mm(flat, Q).T, full_matrices=False)          # B = Q^T A = (A^T Q)^T  [q, n]
        u, v = Q @ Ub, Vh.T
    return (u, s, v) if adj.n_rows >= adj.n_cols else (v, s, u)


class LightGCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, aggr_mode,
                 ssl_alpha, ssl_temp, device):
        super(LightGCL, self).__init__()
        self.num_user, self.num_item, self.user_item_dict, self.dim_E = num_user, num_item, user_item_dict, dim_E
        self.edge_index, self.n_layers, self.aggr_mode, self.device = edge_index, n_layers, aggr_mode, device
        self.q, self.dropout, self.temp, self.lambda_1, self.lambda_2 = 5, 0.0, ssl_temp, ssl_alpha, reg_weight
        self.act = nn.LeakyReLU(0.5)
        e = np.asarray(edge_index)
        self._user, self._item = e[:, 0], e[:, 1] - num_user
        self.adj_norm = self.create_adjust_matrix().to(device)
        r = torch.randn(min(num_user, num_item), self.q, dtype=torch.float32)       # (the reference's draw, in its place in the sequence)
        self._set_svd(*(svd_lowrank(self.adj_norm, r) if device.type == "cuda" else (None, None, None)))
        self.E_u_0 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_user, dim_E)))
        self.E_i_0 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_item, dim_E)))
        L = n_layers
        self.E_u_list, self.E_i_list = [None] * (L + 1), [None] * (L + 1)
        self.Z_u_list, self.Z_i_list = [None] * (L + 1), [None] * (L + 1)
        self.G_u_list, self.G_i_list = [None] * (L + 1), [None] * (L + 1)
        self.E_u_list[0] = self.G_u_list[0] = self.E_u_0
        self.E_i_list[0] = self.G_i_list[0] = self.E_i_0
        self.E_u = self.E_i = self.restore_user_e = self.restore_item_e = None
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def _set_svd(self, svd_u, s, svd_v):
        if svd_u is None:
            self.u_mul_s = self.v_mul_s = self.ut = self.vt = None
            return
        self.u_mul_s, self.v_mul_s = svd_u @ torch.diag(s), svd_v @ torch.diag(s)
        self.ut, self.vt = svd_u.T, svd_v.T

    def create_adjust_matrix(self):
        """:58-72: count / sqrt(row sum * column sum), the sums counting a repeated interaction as often as it is listed."""
        U, I = self.num_user, self.num_item
        key, cnt = torch.unique(torch.from_numpy(self._user.astype(np.int64)) * I + torch.from_numpy(self._item.astype(np.int64)),
                                return_counts=True)
        u, i, w = torch.div(key, I, rounding_mode="floor"), key % I, cnt.to(torch.float32)
        rowD = torch.zeros(U, dtype=torch.float32).index_add_(0, u, w)
        colD = torch.zeros(I, dtype=torch.float32).index_add_(0, i, w)
        # (numpy float32 scalars there, :69-70: product, power 0.5 and quotient round in fp32; its scalar powf and this
        #  correctly-rounded square root may differ in the last bit)
        val = w / torch.sqrt(rowD[u] * colD[i])
        return graph.coo_to_csr_coalesced(u, i, val, U, I)

    def sparse_dropout(self, matrix, dropout):
        if dropout == 0.0:
            return matrix
        return graph.CSR(matrix.rowptr, matrix.col, F.dropout(matrix.val, p=dropout), matrix.n_rows, matrix.n_cols)

    def forward(self):
        for layer in range(1, self.n_layers + 1):
            self.Z_u_list[layer] = sparse.mm(self.sparse_dropout(self.adj_norm, self.dropout), self.E_i_list[layer - 1])
            self.Z_i_list[layer] = sparse.mm(self.sparse_dropout(self.adj_norm, self.dropout).t(), self.E_u_list[layer - 1])
            self.E_u_list[layer], self.E_i_list[layer] = self.Z_u_list[layer], self.Z_i_list[layer]
        self.E_u, self.E_i = sum(self.E_u_list), sum(self.E_i_list)
        return self.E_u, self.E_i

    def bpr_loss(self, E_u_norm, E_i_norm, user, pos_item, neg_item):
        u_e, pi_e, ni_e = E_u_norm[user], E_i_norm[pos_item], E_i_norm[neg_item]
        loss1 = -(torch.mul(u_e, pi_e).sum(dim=1) - torch.mul(u_e, ni_e).sum(dim=1)).sigmoid().log().mean()
        loss_reg = 0
        for param in self.parameters():
            loss_reg += param.norm(2).square()
        return loss1 + loss_reg * self.lambda_2

    def ssl_loss(self, E_u_norm, E_i_norm, user, pos_item):
        for layer in range(1, self.n_layers + 1):
            self.G_u_list[layer] = self.u_mul_s @ (self.vt @ self.E_i_list[layer - 1])
            self.G_i_list[layer] = self.v_mul_s @ (self.ut @ self.E_u_list[layer - 1])
        G_u_norm, G_i_norm = sum(self.G_u_list), sum(self.G_i_list)
        neg_score = torch.log(torch.exp(G_u_norm[user] @ E_u_norm.T / self.temp).sum(1) + 1e-8).mean()
        neg_score += torch.log(torch.exp(G_i_norm[pos_item] @ E_i_norm.T / self.temp).sum(1) + 1e-8).mean()
        pos_score = (torch.clamp((G_u_norm[user] * E_u_norm[user]).sum(1) / self.temp, -5.0, 5.0)).mean() + \
                    (torch.clamp((G_i_norm[pos_item] * E_i_norm[pos_item]).sum(1) / self.temp, -5.0, 5.0)).mean()
        return self.lambda_1 * (-pos_score + neg_score)

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        E_u_norm, E_i_norm = self.forward()
        self.restore_user_e, self.restore_item_e = E_u_norm, E_i_norm
        return self.bpr_loss(E_u_norm, E_i_norm, users, pos_items, neg_items) + self.ssl_loss(E_u_norm, E_i_norm, users, pos_items)

    def gene_ranklist(self, topk=50, to_cpu=True):
        res = torch.cat((self.restore_user_e.detach(), self.restore_item_e.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
