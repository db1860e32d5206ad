"""LATTICE with the reference's surface (Model/LATTICE.py:48-244) -- MICRO's predecessor: ONE item-item graph learned from
both projected modalities (softmax-weighted sum of their cosine-kNN graphs, normalised, mixed with the raw-feature graphs),
the item ids propagated over it, LightGCN over the user-item graph, BPR.

The reference keeps the item graph DENSE -- [I, I] floats with at most 4 topk non-zeros per row, `torch.mm` per layer
(:91-105: 925 MB at sports, out of reach at config 5).  Here it is what it is: a sparse operand over the union of the four
kNN patterns whose VALUES carry gradient into the projections and the modality weights (`sparse.LearnedAdj`, rebuilt on the
first batch of every epoch: train_and_evaluate.py:96-103), multiplied by the HIP SpMM; the other steps use it detached.  The
user-item encoder is `ops.layer_mean_propagate` over `graph.lightgcn_csr` (LATTICEGCNConv is BasicGCN.GCNConv, :16-31), the
projections `ops.linear`, the ranking `ranking.gene_ranklist` over the table of the last forward."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse
from .MICRO import knn_sym_entries


class LATTICE(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, feat_embed_dim,
                 reg_weight, n_layers, mm_layers, ii_topk, aggr_mode, lambda_coeff, device):
        super(LATTICE, self).__init__()
        self.result = self.item_adj = None
        self.num_user, self.num_item, self.dim_E = num_user, num_item, dim_E
        self.weight_size = [dim_E, 64, 64]
        self.topk, self.device, self.feat_embed_dim = ii_topk, device, feat_embed_dim
        self.lambda_coeff, self.mm_layers, self.n_layers = lambda_coeff, mm_layers, n_layers
        self.user_item_dict, self.reg_weight = user_item_dict, reg_weight
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        self.graph = graph.lightgcn_csr(edge_index, num_user + num_item).to(device)
        with torch.no_grad():
            self._image_original = tuple(t.to(device) for t in knn_sym_entries(v_feat.to(device), ii_topk))
            self._text_original = tuple(t.to(device) for t in knn_sym_entries(t_feat.to(device), ii_topk))
        self.image_trs = nn.Linear(v_feat.shape[1], feat_embed_dim)
        self.text_trs = nn.Linear(t_feat.shape[1], feat_embed_dim)
        self.modal_weight = nn.Parameter(torch.Tensor([0.5, 0.5]))
        self.softmax = nn.Softmax(dim=0)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def _learned_graph(self, image_feats, text_feats):
        """:91-101: (1 - lambda) D^-1/2 (w0 kNN(image) + w1 kNN(text)) D^-1/2 + lambda (w0 image_original + w1 text_original),
        D = the row sums of the weighted sum; the union of the four entry lists, row-major."""
        n, lam = self.num_item, self.lambda_coeff
        w = self.softmax(self.modal_weight)
        (idx_v, val_v), (idx_t, val_t) = (knn_sym_entries(f, self.topk, normalise=False) for f in (image_feats, text_feats))
        rowsum = w[0] * val_v.view(n, -1).sum(dim=1) + w[1] * val_t.view(n, -1).sum(dim=1)
        d = torch.pow(rowsum, -0.5)
        d = d.masked_fill(torch.isinf(d), 0.)
        (idx_vo, val_vo), (idx_to, val_to) = self._image_original, self._text_original
        parts = ((idx_v, (1 - lam) * d[idx_v[0]] * (w[0] * val_v) * d[idx_v[1]]), (idx_t, (1 - lam) * d[idx_t[0]] * (w[1] * val_t) * d[idx_t[1]]),
                 (idx_vo, lam * w[0] * val_vo), (idx_to, lam * w[1] * val_to))
        key = torch.cat([i[0] * n + i[1] for i, _ in parts])
        uniq, inverse = torch.unique(key, return_inverse=True)
        val = torch.zeros(uniq.numel(), dtype=torch.float32, device=key.device).index_add(0, inverse, torch.cat([v for _, v in parts]))
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(torch.bincount(torch.div(uniq, n, rounding_mode="floor"), minlength=n), 0, out=rowptr[1:])
        return sparse.LearnedAdj(rowptr, uniq % n, val, n, n)

    def forward(self, build_item_graph=False):
        """:87-118."""
        if build_item_graph:
            image_feats = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
            text_feats = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
            self.item_adj = self._learned_graph(image_feats, text_feats)
        else:
            if self.item_adj is None:
                raise AttributeError("LATTICE.forward(build_item_graph=False) before any build: there is no item graph to detach")
            if isinstance(self.item_adj, sparse.LearnedAdj):
                self.item_adj = self.item_adj.detach()
        h = self.item_embedding.weight
        for _ in range(self.mm_layers):
            h = sparse.mm(self.item_adj, h)
        ego = torch.cat((self.user_embedding.weight, self.item_embedding.weight), dim=0)
        u_g, i_g = torch.split(ops.layer_mean_propagate(ego, self.graph, self.n_layers), [self.num_user, self.num_item], dim=0)
        self.result = torch.cat((u_g, i_g + F.normalize(h, p=2, dim=1)), dim=0)
        return self.result

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        u, p, n = embeddings[users], embeddings[self.num_user + pos_items], embeddings[self.num_user + neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items, embeddings):
        u, p, n = embeddings[users], embeddings[self.num_user + pos_items], embeddings[self.num_user + neg_items]
        return self.reg_weight * (torch.mean(u ** 2) + torch.mean(p ** 2) + torch.mean(n ** 2))

    def loss(self, users, pos_items, neg_items, build_item_graph):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        embeddings = self.forward(build_item_graph)
        return self.bpr_loss(users, pos_items, neg_items, embeddings) + self.regularization_loss(users, pos_items, neg_items, embeddings)

    def gene_ranklist(self, topk=50, to_cpu=True):
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
