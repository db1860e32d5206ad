"""SimGCL with the reference's surface (Model/SimGCL.py:17-200) -- a member of the `torch.sparse.mm` family that took NO
per-model kernel work: the propagate is `chaorec_amd.sparse.mm` (the CSR SpMM kernel behind
torch.sparse.mm's signature), the ranking is the shared `ranking.gene_ranklist`; every other line of arithmetic is the
reference's own torch expression.  Three full-graph forwards per step (one clean, two perturbed).

Same constructor, parameters (`user_embedding`, `item_embedding`, created in the reference's order: same seed, same
weights), `forward(perturbed)`, `bpr_loss`, `regularization_loss`, `cal_cl_loss`, `loss`, `gene_ranklist`.
Differences: the adjacency is built vectorised (graph.binary_sym_norm_csr: same fp64 arithmetic, same fp32 values as the
scipy path, :64-112); the noise is drawn on the embeddings' device (the reference hard-codes `.cuda()`, :121) through
`noise_fn` (default torch.rand_like) so that a test can feed the numbers the reference drew."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse


def InfoNCE(view1, view2, temperature, b_cos=True):
    """Model/SimGCL.py:16-32: mean over the batch of -log softmax_row(view1 view2^T / t)[i, i]."""
    if b_cos:
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
    logits = (view1 @ view2.T) / temperature
    return -torch.diag(F.log_softmax(logits, dim=1)).mean()


class SimGCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, ssl_temp, ssl_reg,
                 device):
        super(SimGCL, self).__init__()
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers = dim_E, reg_weight, n_layers
        self.ssl_temp, self.ssl_reg, self.device = ssl_temp, ssl_reg, device
        self.eps = 0.1                                   # perturbation radius (:49)
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
        """:114-131: x_{k+1} = A x_k [+ sign(x) * normalize(noise) * eps]; mean over layers 1..L (the ego layer is NOT in it)."""
        ego = torch.cat([self.user_embedding.weight, self.item_embedding.weight], 0)
        layers = []
        for _ in range(self.n_layers):
            ego = sparse.mm(self.sparse_norm_adj, ego)
            if perturbed:
                ego = ego + torch.sign(ego) * F.normalize(self.noise_fn(ego), dim=-1) * self.eps
            layers.append(ego)
        out = torch.mean(torch.stack(layers, dim=1), dim=1)
        return torch.split(out, [self.num_user, self.num_item])

    def bpr_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        u, p, n = user_emb[users], item_emb[pos_items], item_emb[neg_items]
        return -torch.mean(torch.log(torch.sigmoid((u * p).sum(1) - (u * n).sum(1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        return self.reg_weight * (torch.mean(user_emb[users] ** 2) + torch.mean(item_emb[pos_items] ** 2))

    def cal_cl_loss(self, users, pos_items):
        u1, i1 = self.forward(perturbed=True)
        u2, i2 = self.forward(perturbed=True)
        return InfoNCE(u1[users], u2[users], self.ssl_temp) + InfoNCE(i1[pos_items], i2[pos_items], self.ssl_temp)

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        self.user_emb, self.item_emb = self.forward()
        return (self.bpr_loss(users, pos_items, neg_items, self.user_emb, self.item_emb)
                + self.regularization_loss(users, pos_items, neg_items, self.user_emb, self.item_emb)
                + self.ssl_reg * self.cal_cl_loss(users, pos_items))

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:169-199 (mask value 1e-6, the embeddings of the last training forward)."""
        res = torch.cat((self.user_emb.detach(), self.item_emb.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
