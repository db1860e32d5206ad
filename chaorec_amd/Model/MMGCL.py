"""MMGCL with the reference's surface (Model/MMGCL.py:18-424) -- three LightGCN-style encoders (id / projected visual /
projected textual item rows over one user-item graph), read out by two Linears, plus a contrastive loss between two
augmented views per step: one on an EDGE-dropped graph, one with a NODE-dropped graph for one randomly chosen modality.

The reference rebuilds a scipy Laplacian and a torch COO tensor for both augmentations in every step (:119-145,193-212:
random.sample over the edge list on the host, two sparse products with diagonal matrices, a host-to-device copy).  Here the
graph's STRUCTURE never changes: the augmented graphs are value arrays over the one symmetric CSR -- keep mask -> kept
degrees -> d^-1/2 keep d^-1/2 per entry, a handful of elementwise device launches -- multiplied by the hot-path SpMM in its
dynamic-values mode (`sparse.DroppedAdj` through `sparse.mm`, forward and backward).  The unperturbed encoders are the fused
layer-mean propagate (`ops.layer_mean_propagate`), the feature projections and read-outs run on the MFMA GEMM
(`ops.linear`), BPR is the fused kernel, the ranking is `ranking.gene_ranklist` over the tables of the last training
forward (:399-424).

Same constructor, parameters in the reference's creation order.  Randomness: the kept edges, the dropped users / items and
the masked modality are drawn on the device / by numpy as the reference does by `random.sample` / `np.random.choice`;
`edge_keep_fn(n_edges, rate) -> bool [n_edges]`, `node_keep_fn(n_users, n_items, rate) -> (bool [U], bool [I])` and
`modality_fn() -> 0 | 1` replace the draws (the golden test feeds the reference run's)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


class MMGCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 reg_weight, n_layers, ssl_alpha, ssl_temp, dropout, device):
        super(MMGCL, self).__init__()
        self.result_item = self.result_user = self.t_dense_emb = self.v_dense_emb = None
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers = dim_E, reg_weight, n_layers
        self.ssl_alpha, self.ssl_temp, self.device = ssl_alpha, ssl_temp, device
        self.ssl_task = "ED+MM+CN"
        self.dropout_rate = dropout
        self.dropout = nn.Dropout(p=dropout)
        self.p_vat = [0.5, 0.5]

        self.user_embeddings = nn.Embedding(num_user, dim_E)
        self.item_embeddings = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embeddings.weight)
        nn.init.xavier_uniform_(self.item_embeddings.weight)
        self.register_buffer("v_feat", F.normalize(v_feat, dim=1), persistent=False)
        self.v_dense = nn.Linear(v_feat.shape[1], dim_E)
        nn.init.xavier_uniform_(self.v_dense.weight)
        self.register_buffer("t_feat", F.normalize(t_feat, dim=1), persistent=False)
        self.t_dense = nn.Linear(t_feat.shape[1], dim_E)
        nn.init.xavier_uniform_(self.t_dense.weight)
        self.read_user = nn.Linear(3 * dim_E, dim_E)
        self.read_item = nn.Linear(3 * dim_E, dim_E)
        nn.init.xavier_uniform_(self.read_user.weight)
        nn.init.xavier_uniform_(self.read_item.weight)

        # the distinct (user, item) pairs in row-major order (= scipy's csr.nonzero(), what :121,139 index), how often each is
        # listed, and one symmetric [N, N] structure over them
        self._pairs = sparse.PairStructure(edge_index, num_user, num_item, device)
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self._lower, both = self._pairs.lower, self._pairs.csr
        both.val.copy_(self._values(self._ew))                                      # :77-109 on the multiplicities: norm_adj
        self.norm_adj = both
        self._structure = self._pairs.structure
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.edge_keep_fn = self.node_keep_fn = self.modality_fn = None

    # ---- graphs -------------------------------------------------------------------------------------------------------------
    def _values(self, w):
        """:77-109 for the pair weights w [n_edges] (0 = dropped): row sums of [[0, R], [R^T, 0]], zero sums -> 1e-10,
        d^-1/2 w d^-1/2 per entry of the symmetric CSR (fp32 like scipy's float32 matrices)."""
        U, N = self.num_user, self.num_user + self.num_item
        deg = torch.zeros(N, dtype=torch.float32, device=w.device).index_add_(0, self._eu, w).index_add_(0, U + self._ei, w)
        d = torch.pow(torch.where(deg == 0, torch.full_like(deg, 1e-10), deg), -0.5)
        val = (d[self._eu] * w) * d[U + self._ei]
        return torch.cat([val, val[self._lower]])

    def _perturbed(self, keep):
        """binary graph of the kept pairs, normalised by ITS degrees (:119-145: both augmentations reset the weights to 1)"""
        val = self._values(keep.to(torch.float32))
        return sparse.DroppedAdj(self._structure, val, val)                        # (symmetric: its own transpose)

    def random_graph_augment(self, aug_type):
        """:193-205.  1: node dropout (:119-134), 0: edge dropout (:136-145)."""
        dev, rate = self._eu.device, self.dropout_rate
        if aug_type == 0:
            if self.edge_keep_fn is not None:
                keep = self.edge_keep_fn(self.n_edges, rate).to(dev)
            else:
                keep = torch.zeros(self.n_edges, dtype=torch.bool, device=dev)
                keep[torch.randperm(self.n_edges, device=dev)[:int(self.n_edges * (1 - rate))]] = True
        else:
            if self.node_keep_fn is not None:
                ku, ki = (k.to(dev) for k in self.node_keep_fn(self.num_user, self.num_item, rate))
            else:
                ku = torch.ones(self.num_user, dtype=torch.bool, device=dev)
                ki = torch.ones(self.num_item, dtype=torch.bool, device=dev)
                ku[torch.randperm(self.num_user, device=dev)[:int(self.num_user * rate)]] = False
                ki[torch.randperm(self.num_item, device=dev)[:int(self.num_item * rate)]] = False
            keep = ku[self._eu] & ki[self._ei]
        return self._perturbed(keep)

    graph_reconstruction = random_graph_augment

    # ---- :147-191 -----------------------------------------------------------------------------------------------------------
    def sgl_encoder(self, user_emb, item_emb, perturbed_adj=None):
        ego = torch.cat([user_emb, item_emb], 0)
        if perturbed_adj is None:
            mean = ops.layer_mean_propagate(ego, self.norm_adj, self.n_layers)
        else:
            x, total = ego, ego
            for k in range(self.n_layers):
                x = sparse.mm(perturbed_adj[k] if isinstance(perturbed_adj, list) else perturbed_adj, x)
                total = total + x
            mean = total / (self.n_layers + 1)
        return torch.split(mean, [self.num_user, self.num_item])

    def forward(self):
        users_emb, items_emb = self.user_embeddings.weight, self.item_embeddings.weight
        self.v_dense_emb = ops.linear(self.v_feat, self.v_dense.weight, self.v_dense.bias)
        self.t_dense_emb = ops.linear(self.t_feat, self.t_dense.weight, self.t_dense.bias)
        i_emb_u, i_emb_i = self.sgl_encoder(users_emb, items_emb)
        v_emb_u, v_emb_i = self.sgl_encoder(users_emb, self.v_dense_emb)
        t_emb_u, t_emb_i = self.sgl_encoder(users_emb, self.t_dense_emb)
        user = ops.linear(torch.cat([i_emb_u, v_emb_u, t_emb_u], dim=1), self.read_user.weight, self.read_user.bias)
        item = ops.linear(torch.cat([i_emb_i, v_emb_i, t_emb_i], dim=1), self.read_item.weight, self.read_item.bias)
        return user, item

    # ---- :214-287: the two augmented views, batch rows only after the propagation -----------------------------------------------
    def _read_views(self, views, user, pos_item, neg_item, neg_from):
        """views: (u, i) tables of the id / visual / textual encoders.  neg_from: which encoders' rows the 'negative' read-out
        takes from neg_item (:240 the textual one only, :282 the visual and the textual one)."""
        (iu, ii), (vu, vi), (tu, ti) = views
        users_sub = self.read_user(torch.cat([iu[user], vu[user], tu[user]], dim=1))
        items_sub = self.read_item(torch.cat([ii[pos_item], vi[pos_item], ti[pos_item]], dim=1))
        neg_items_sub = self.read_item(torch.cat([ii[pos_item], vi[neg_item] if "v" in neg_from else vi[pos_item], ti[neg_item]], dim=1))
        return F.normalize(users_sub, dim=1), F.normalize(items_sub, dim=1), F.normalize(neg_items_sub, dim=1)

    def modality_edge_dropout_emb(self, user, pos_item, neg_item):
        users_emb, items_emb = self.user_embeddings.weight, self.item_embeddings.weight
        adj = self.graph_reconstruction(aug_type=0)
        views = [self.sgl_encoder(users_emb, x, adj) for x in (items_emb, self.v_dense_emb, self.t_dense_emb)]
        return self._read_views(views, user, pos_item, neg_item, "t")

    def modality_masking_emb(self, user, pos_item, neg_item):
        users_emb, items_emb = self.user_embeddings.weight, self.item_embeddings.weight
        adj = self.graph_reconstruction(aug_type=1)
        modality = int(self.modality_fn()) if self.modality_fn is not None else int(np.random.choice(2, p=self.p_vat))   # 0: image
        views = [self.sgl_encoder(users_emb, items_emb),
                 self.sgl_encoder(users_emb, self.v_dense_emb, adj if modality == 0 else None),
                 self.sgl_encoder(users_emb, self.t_dense_emb, adj if modality == 1 else None)]
        return self._read_views(views, user, pos_item, neg_item, "vt")

    def cal_multiview_MM_ED_CN(self, users, pos_items, neg_items):
        """:289-344, ssl_task "ED+MM+CN": user view 1 against item view 1 and against item view 2 (the third, negated term
        is computed there and never added)."""
        users_sub_1, items_sub_1, _ = self.modality_edge_dropout_emb(users, pos_items, neg_items)
        _, items_sub_2, _ = self.modality_masking_emb(users, pos_items, neg_items)
        labels = torch.arange(users_sub_1.shape[0], device=users_sub_1.device)
        return F.cross_entropy(torch.mm(users_sub_1, items_sub_1.T) / self.ssl_temp, labels) + \
            F.cross_entropy(torch.mm(users_sub_1, items_sub_2.T) / self.ssl_temp, labels)

    # ---- :357-397 -----------------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items, u_g, i_g):
        return ops.bpr_loss(u_g.contiguous(), i_g.contiguous(), users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items, u_g, i_g):
        return self.reg_weight * (ops.mean_all(u_g[users] ** 2) + ops.mean_all(i_g[pos_items] ** 2) + ops.mean_all(i_g[neg_items] ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        user, item = self.forward()
        self.result_user, self.result_item = user, item
        return self.bpr_loss(users, pos_items, neg_items, user, item) + \
            self.ssl_alpha * self.cal_multiview_MM_ED_CN(users, pos_items, neg_items)          # (:394: no regularisation term)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:399-424: the tables of the last training forward, history at 1e-6."""
        result = torch.cat([self.result_user.detach(), self.result_item.detach()], 0)
        return ranking.gene_ranklist(result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
