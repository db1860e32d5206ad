"""XSimGCL with the reference's surface (Model/XSimGCL.py:34-202) -- one more member of the `torch.sparse.mm` family through
the adapter alone: the propagate is `chaorec_amd.sparse.mm`, the ranking the shared `ranking.gene_ranklist`;
every other line of arithmetic is the reference's own torch expression.  ONE perturbed forward per step (SimGCL needs
three): the contrastive views are the final mean and the output of layer `layer_cl` of the same pass.

Same constructor, parameters (`user_embedding`, `item_embedding`, created in the reference's order), `forward(perturbed)`,
`bpr_loss`, `regularization_loss`, `cal_cl_loss`, `loss`, `gene_ranklist` -- which, unlike most of the family, runs a FRESH
unperturbed forward on the current weights (:175-178) instead of ranking the last training forward.  Differences: the
adjacency is built vectorised (graph.binary_sym_norm_csr: the scipy path's values, :64-105); the noise is drawn on the
embeddings' device (the reference hard-codes `.cuda()`, :113) through `noise_fn` (default torch.rand_like)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse
from .SimGCL import InfoNCE


class XSimGCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, ssl_temp, ssl_reg,
                 device):
        super(XSimGCL, self).__init__()
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers = dim_E, reg_weight, n_layers
        self.ssl_temp, self.ssl_reg, self.device = ssl_temp, ssl_reg, device
        self.eps = 0.2                                   # perturbation radius (:48)
        self.layer_cl = 1                                # the layer whose output is the second view (:49)
        self.noise_fn = torch.rand_like
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        e = torch.as_tensor(edge_index).long()
        self.sparse_norm_adj = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.user_emb = self.item_emb = None

    def forward(self, perturbed=False):
        """:107-127: x_{k+1} = A x_k [+ sign(x) * normalize(noise) * eps]; mean over layers 1..L; the view of layer layer_cl."""
        ego = torch.cat([self.user_embedding.weight, self.item_embedding.weight], 0)
        layers, view = [], ego
        for k in range(self.n_layers):
            ego = sparse.mm(self.sparse_norm_adj, ego)
            if perturbed:
                ego = ego + torch.sign(ego) * F.normalize(self.noise_fn(ego), dim=-1) * self.eps
            layers.append(ego)
            if k == self.layer_cl - 1:
                view = ego
        out = torch.mean(torch.stack(layers, dim=1), dim=1)
        u, i = torch.split(out, [self.num_user, self.num_item])
        if perturbed:
            return (u, i) + tuple(torch.split(view, [self.num_user, self.num_item]))
        return u, i

    def bpr_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        u, p, n = user_emb[users], item_emb[pos_items], item_emb[neg_items]
        return -torch.mean(torch.log(torch.sigmoid((u * p).sum(1) - (u * n).sum(1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        return self.reg_weight * (torch.mean(user_emb[users] ** 2) + torch.mean(item_emb[pos_items] ** 2))

    def cal_cl_loss(self, users, pos_items, user_view1, user_view2, item_view1, item_view2):
        return InfoNCE(user_view1[users], user_view2[users], self.ssl_temp) + \
            InfoNCE(item_view1[pos_items], item_view2[pos_items], self.ssl_temp)

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        ru, ri, cu, ci = self.forward(True)
        return (self.bpr_loss(users, pos_items, neg_items, ru, ri) + self.regularization_loss(users, pos_items, neg_items, ru, ri)
                + self.ssl_reg * self.cal_cl_loss(users, pos_items, ru, cu, ri, ci))

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:172-202 (mask value 1e-6; a fresh unperturbed forward on the current weights)."""
        with torch.no_grad():
            self.user_emb, self.item_emb = self.forward()
        res = torch.cat((self.user_emb, self.item_emb), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
