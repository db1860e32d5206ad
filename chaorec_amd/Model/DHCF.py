"""DHCF with the reference's surface (Model/DHCF.py:15-198) -- dual-channel hypergraph collaborative filtering: every layer
applies to the user table the operator  M_u = Dv Hu De^2 Hu^T Dv + I  with the hyperedge incidence  Hu = [H | H (H^T H)]
(direct interactions and their two-hop closure), and the mirror operator built from H^T to the item table, then one shared
Linear.

The reference MATERIALISES Hu in every forward (:37-40: two sparse-sparse products, [U, 2 I] with the counts of all 3-step
walks: dense for any real graph) and multiplies the chain with `torch.linalg.multi_dot`.  Here nothing is materialised:

    Hu De^2 Hu^T y  =  H (De1^2 (H^T y))  +  H H^T H (De2^2 (H^T H H^T y))

is eight `chaorec_amd.sparse.mm` launches over the interaction CSR and its transpose (the hot-path SpMM with autograd), and
the degree vectors Dv, De -- row and column sums of Hu, constants of the graph -- are computed ONCE at construction by the
same chain applied to ones (fp64 on the host: the counts are integers, the reference's fp32 sparse sums are exact up to
2^24).  The shared Linear runs on the MFMA GEMM (`ops.linear`), BPR is the fused kernel, the ranking is
`ranking.gene_ranklist` over the tables of the last training forward (:170-198).

Quirks kept: the layers live in a plain Python LIST (:117-118), so their weights are not parameters of the model -- the
optimizer never sees them, `named_parameters()` is the two embedding tables --, and the bias is `torch.Tensor(n)`
(:24: uninitialised memory in the reference; zeros here, tests copy the reference run's values in)."""
import numpy as np
import torch
from torch import nn

from .. import graph, ops, ranking, sparse


class DJconv(nn.Module):
    """:15-69.  The operator pair of one layer + the shared Linear."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        nn.init.xavier_uniform_(self.weight)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, ops_u, ops_i, U, I):
        M_u, M_i = ops_u(U) + U, ops_i(I) + I
        wt = self.weight.t().contiguous()
        return ops.linear(M_u, wt, self.bias), ops.linear(M_i, wt, self.bias)


class _TwoHop:
    """y -> Dv Hu De^2 Hu^T Dv y for Hu = [H | H H^T H], H the CSR `h` ([n, m]) with transpose `ht`."""

    def __init__(self, h, ht, device):
        self.h, self.ht = h, ht
        n, m = h.n_rows, h.n_cols
        # degrees on the host in fp64 (exact integer counts): row sums of Hu = H 1 + H H^T H 1, column sums = [H^T 1 | H^T H H^T 1]
        hd = torch.sparse_csr_tensor(h.rowptr.cpu(), h.col.cpu().long(), h.val.cpu().double(), (n, m))
        htd = torch.sparse_csr_tensor(ht.rowptr.cpu(), ht.col.cpu().long(), ht.val.cpu().double(), (m, n))
        mv = lambda a, x: (a @ x.unsqueeze(1)).squeeze(1)
        one_n, one_m = torch.ones(n, dtype=torch.float64), torch.ones(m, dtype=torch.float64)
        row_sum = mv(hd, one_m) + mv(hd, mv(htd, mv(hd, one_m)))
        col1, col2 = mv(htd, one_n), mv(htd, mv(hd, mv(htd, one_n)))
        f32 = lambda x: x.to(torch.float32)
        # :26-27: (sum + 1e-7)^-1/2 in fp32; De enters squared
        self.dv = torch.pow(f32(row_sum) + 1e-7, -0.5).unsqueeze(1).to(device)
        self.de1_sq = (torch.pow(f32(col1) + 1e-7, -0.5) ** 2).unsqueeze(1).to(device)
        self.de2_sq = (torch.pow(f32(col2) + 1e-7, -0.5) ** 2).unsqueeze(1).to(device)

    def __call__(self, x):
        y = self.dv * x
        t = sparse.mm(self.ht, y)                                                   # H^T y                     [m, D]
        direct = sparse.mm(self.h, self.de1_sq * t)
        w = sparse.mm(self.ht, sparse.mm(self.h, t))                                # H^T H H^T y               [m, D]
        two_hop = sparse.mm(self.h, sparse.mm(self.ht, sparse.mm(self.h, self.de2_sq * w)))
        return self.dv * (direct + two_hop)


class DHCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, dropout,
                 device):
        super(DHCF, self).__init__()
        self.device, self.num_user, self.num_item = device, num_user, num_item
        self.user_item_dict, self.reg_weight, self.dim_embedding = user_item_dict, reg_weight, dim_E
        self.use_sparse = True
        self.user_e = self.item_e = None
        e = torch.as_tensor(np.asarray(edge_index)).long()
        u, i = e[:, 0], e[:, 1] - num_user
        # :96-106: H as given (an uncoalesced COO tensor: a repeated interaction counts twice in every product)
        ones = torch.ones(len(u))
        self.interaction_matrix = graph.coo_to_csr_coalesced(u, i, ones, num_user, num_item).to(device)
        self.interaction_matrix_t = graph.coo_to_csr_coalesced(i, u, ones, num_item, num_user).to(device)
        self.interaction_matrix._t, self.interaction_matrix_t._t = self.interaction_matrix_t, self.interaction_matrix
        # (:110-117 in the reference's order of draws, on the HOST generator like every other model here -- the reference moves
        #  the tables to its device before the xavier draw --, then moved)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.user_embedding, self.item_embedding = self.user_embedding.to(device), self.item_embedding.to(device)
        self.layers = [DJconv(dim_E, dim_E).to(device) for _ in range(n_layers)]          # (a list: not registered, :117)
        self.dropout = [nn.Dropout(dropout).to(device) for _ in range(n_layers)]
        self._ops_u = _TwoHop(self.interaction_matrix, self.interaction_matrix_t, device)
        self._ops_i = _TwoHop(self.interaction_matrix_t, self.interaction_matrix, device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def forward(self):
        """:120-136: every layer's output concatenated behind the ego rows."""
        U, I = self.user_embedding.weight, self.item_embedding.weight
        U_out, I_out = U, I
        for idx, layer in enumerate(self.layers):
            U, I = self.dropout[idx](U), self.dropout[idx](I)
            U, I = layer(self._ops_u, self._ops_i, U, I)
            U_out, I_out = torch.concat((U_out, U), dim=1), torch.concat((I_out, I), dim=1)
        self.user_e, self.item_e = U_out, I_out
        return U_out, I_out

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        U_out, I_out = self.forward()
        return self.bpr_loss(users, pos_items, neg_items, U_out, I_out) + self.regularization_loss(users, pos_items, neg_items, U_out, I_out)

    def bpr_loss(self, users, pos_items, neg_items, U_out, I_out):
        """:151-164."""
        return ops.bpr_loss(U_out.contiguous(), I_out.contiguous(), users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items, U_out, I_out):
        """:166-175."""
        return self.reg_weight * (ops.mean_all(U_out[users] ** 2) + ops.mean_all(I_out[pos_items] ** 2) + ops.mean_all(I_out[neg_items] ** 2))

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:177-198: the tables of the last training forward, history at 1e-6."""
        result = torch.cat([self.user_e.detach(), self.item_e.detach()], 0)
        return ranking.gene_ranklist(result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
