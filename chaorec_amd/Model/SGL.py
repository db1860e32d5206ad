"""SGL with the reference's surface (Model/SGL.py:23-265) -- LightGCN plus an InfoNCE between two randomly sub-sampled views
of the graph, re-drawn every step -- `torch.sparse.mm` family (the file calls it through `import torch.sparse as torch_sp`,
:13,114-116).  The reference rebuilds a scipy Laplacian per view and step (:61-104: np.random.choice, csr products, a
host-to-device copy); here the STRUCTURE is built once -- one symmetric CSR over the distinct interactions -- and a view is
a VALUE array over it: the surviving copies of every pair, normalised by the surviving degrees (zero sums -> 1e-10, :97-99),
through the dynamic-values HIP SpMM (`sparse.DroppedAdj`), MMGCL's treatment.  The main view is `ops.layer_mean_propagate`,
the ranking `ranking.gene_ranklist` over the tables of the last training forward (:241-265).

Same constructor and parameters.  `ssl_aug_type` is fixed at 'ed' there (:40); 'nd' and 'rw' are kept.  The draws come from
the device generator; `edge_keep_fn(n_listed, ratio)` / `node_keep_fn(U, I, ratio)` replay stored draws in the golden test."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


def _subset(n, k, dev):
    """k of range(n), uniformly without replacement (np.random.choice(n, k, replace=False), :69-70,88): the first k of a
    random order, taken as the argsort of device uniforms."""
    return torch.rand(n, device=dev).argsort()[:k]


class SGL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, aggr_mode, ssl_temp,
                 ssl_reg, device):
        super(SGL, self).__init__()
        self.item_emb_final = self.user_emb_final = None
        self.num_user, self.num_item, self.user_item_dict, self.dim_E = num_user, num_item, user_item_dict, dim_E
        self.reg_weight, self.n_layers, self.aggr_mode, self.device = reg_weight, n_layers, aggr_mode, device
        self.ssl_aug_type, self.ssl_temp, self.ssl_reg, self.ssl_ratio = 'ed', ssl_temp, ssl_reg, 0.1
        self.edge_index = edge_index
        self._pairs = sparse.PairStructure(edge_index, num_user, num_item, device)      # distinct interactions + one symmetric [N, N] structure
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self.n_listed, self._lower, both = self._pairs.n_listed, self._pairs.lower, self._pairs.csr
        both.val.copy_(self._values(self._ew))
        self.norm_adj = both
        self._structure = self._pairs.structure
        self.user_embeddings = nn.Embedding(num_user, dim_E)
        self.item_embeddings = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embeddings.weight)
        nn.init.xavier_uniform_(self.item_embeddings.weight)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.edge_keep_fn = self.node_keep_fn = None

    # ---- graphs (:61-104) -----------------------------------------------------------------------------------------------------
    def _values(self, w):
        U, N = self.num_user, self.num_user + self.num_item
        deg = torch.zeros(N, dtype=torch.float32, device=w.device).index_add_(0, self._eu, w).index_add_(0, U + self._ei, w)
        d = torch.pow(torch.where(deg == 0, torch.full_like(deg, 1e-10), deg), -0.5)
        val = (d[self._eu] * w) * d[U + self._ei]
        return torch.cat([val, val[self._lower]])

    def create_adj_mat(self, is_subgraph=False, aug_type='ed'):
        if not (is_subgraph and self.ssl_ratio > 0):
            return self.norm_adj
        dev, ratio = self._eu.device, self.ssl_ratio
        if aug_type == 'nd':                                                        # :69-86 (the kept pairs with their multiplicities)
            if self.node_keep_fn is not None:
                ku, ki = (k.to(dev) for k in self.node_keep_fn(self.num_user, self.num_item, ratio))
            else:
                ku = torch.ones(self.num_user, dtype=torch.bool, device=dev)
                ki = torch.ones(self.num_item, dtype=torch.bool, device=dev)
                ku[_subset(self.num_user, int(self.num_user * ratio), dev)] = False
                ki[_subset(self.num_item, int(self.num_item * ratio), dev)] = False
            w = torch.where(ku[self._eu] & ki[self._ei], self._ew, torch.zeros_like(self._ew))
        else:                                                                       # :87-93: listed copies kept, copies of a pair add up
            if self.edge_keep_fn is not None:
                keep = self.edge_keep_fn(self.n_listed, ratio).to(dev)
            else:
                keep = torch.zeros(self.n_listed, dtype=torch.bool, device=dev)
                keep[_subset(self.n_listed, int(self.n_listed * (1 - ratio)), dev)] = True
            w = self._pairs.kept_copies(keep)
        val = self._values(w)
        return sparse.DroppedAdj(self._structure, val, val)                         # (symmetric: its own transpose)

    # ---- :106-140 -------------------------------------------------------------------------------------------------------------
    def gcn(self, norm_adj):
        ego = torch.cat([self.user_embeddings.weight, self.item_embeddings.weight], dim=0)
        if isinstance(norm_adj, graph.CSR):
            mean = ops.layer_mean_propagate(ego, norm_adj, self.n_layers)
        else:
            x, total = ego, ego
            for k in range(self.n_layers):
                x = sparse.mm(norm_adj[k] if isinstance(norm_adj, list) else norm_adj, x)
                total = total + x
            mean = total / (self.n_layers + 1)
        return torch.split(mean, [self.num_user, self.num_item], dim=0)

    def forward(self):
        if self.ssl_aug_type in ['nd', 'ed']:
            sub_graph1 = self.create_adj_mat(is_subgraph=True, aug_type=self.ssl_aug_type)
            sub_graph2 = self.create_adj_mat(is_subgraph=True, aug_type=self.ssl_aug_type)
        else:
            sub_graph1, sub_graph2 = [], []
            for _ in range(self.n_layers):
                sub_graph1.append(self.create_adj_mat(is_subgraph=True, aug_type=self.ssl_aug_type))
                sub_graph2.append(self.create_adj_mat(is_subgraph=True, aug_type=self.ssl_aug_type))
        user_emb, item_emb = self.gcn(self.norm_adj)
        user_s1, item_s1 = self.gcn(sub_graph1)
        user_s2, item_s2 = self.gcn(sub_graph2)
        return user_emb, item_emb, user_s1, item_s1, user_s2, item_s2

    # ---- :142-239 -------------------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        u, p, n = user_emb[users], item_emb[pos_items], item_emb[neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def ssl_loss(self, users, items, user_s1, item_s1, user_s2, item_s2):
        total = 0
        for ids, s1, s2 in ((users, user_s1, user_s2), (items, item_s1, item_s2)):
            e1, e2 = F.normalize(s1, dim=1), F.normalize(s2, dim=1)
            b1, b2 = F.embedding(ids, e1), F.embedding(ids, e2)
            pos = torch.sum(b1 * b2, dim=-1)
            tot = torch.matmul(b1, torch.transpose(e2, 0, 1))
            total = total + torch.logsumexp((tot - pos[:, None]) / self.ssl_temp, dim=1)
        return torch.sum(total)

    def regularization_loss(self, users, pos_items, neg_items):
        return self.reg_weight * (torch.mean(self.user_embeddings.weight[users] ** 2) + torch.mean(self.item_embeddings.weight[pos_items] ** 2)
                                  + torch.mean(self.item_embeddings.weight[neg_items] ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        user_emb, item_emb, user_s1, item_s1, user_s2, item_s2 = self.forward()
        self.user_emb_final, self.item_emb_final = user_emb, item_emb
        return (self.bpr_loss(users, pos_items, neg_items, user_emb, item_emb) + self.regularization_loss(users, pos_items, neg_items)
                + self.ssl_reg * self.ssl_loss(users, pos_items, user_s1, item_s1, user_s2, item_s2))

    def gene_ranklist(self, topk=50, to_cpu=True):
        res = torch.cat((self.user_emb_final.detach(), self.item_emb_final.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
