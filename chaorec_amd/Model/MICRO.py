"""MICRO with the reference's surface (Model/MICRO.py:93-260) -- item-item graphs LEARNED from the projected modality
features (cosine kNN, symmetric normalisation, mixed with the graph of the raw features), the item ids propagated over
them, an attention between the two views, LightGCN over the user-item graph and a contrast of each view with the fusion.

The item graphs are rebuilt on the first batch of every epoch (`build_item_graph=True`, train_and_evaluate.py:96-103) and
on that step their VALUES carry gradient into the projections: `sparse.LearnedAdj` -- the HIP SpMM over the union structure
of this build, d value[k] = <gy[row_k], x[col_k]> -- built on the device without the [I, I] similarity matrix (rows in
chunks) and without the reference's Python loop over I x topk index pairs (:27-31).  Every other step multiplies with the
detached graph, an ordinary constant CSR.  The user-item encoder is `ops.layer_mean_propagate` over GCNConv's graph
(`graph.lightgcn_csr`: no self loops, degrees over the bidirectional list), the Linears are `ops.linear`, the ranking
`ranking.gene_ranklist` over the table of the last forward.

Kept quirk: the n_ii_layer loop re-applies the graph to the ORIGINAL ids every time (:195-198): one product whatever
mm_layers says."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


def knn_sym_entries(feats, topk, chunk=4096, normalise=True):
    """:16-51 with is_sparse=True, norm_type='sym': cosine similarity of the rows, each row's `topk` largest (itself among
    them), weight d^-1/2[row] w d^-1/2[col] with d = the row's kept weights summed (1 / 0 -> 0) -- or the kept cosines
    themselves (normalise=False: LATTICE mixes two such graphs before it normalises).  Differentiable in `feats`.
    -> (idx [2, n topk], val [n topk])"""
    x = feats.div(torch.norm(feats, p=2, dim=-1, keepdim=True))
    n = x.shape[0]
    vals, inds = [], []
    for s in range(0, n, chunk):
        v, i = torch.topk(x[s:s + chunk] @ x.T, topk, dim=-1)
        vals.append(v)
        inds.append(i)
    knn_val, knn_ind = torch.cat(vals), torch.cat(inds)
    row = torch.arange(n, device=x.device).repeat_interleave(topk)
    col, w = knn_ind.flatten(), knn_val.flatten()
    if not normalise:
        return torch.stack([row, col]), w
    dis = knn_val.sum(dim=1).pow(-0.5)
    dis = dis.masked_fill(dis == float('inf'), 0)
    return torch.stack([row, col]), dis[row] * w * dis[col]


class MICRO(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, n_layer, reg_weight,
                 ii_topk, mm_layers, ssl_temp, lambda_coeff, ssl_alpha, aggr_mode, device):
        super().__init__()
        self.text_adj = self.image_adj = self.text_item_embeds = self.image_item_embeds = self.h = self.result = None
        self.num_user, self.num_item, self.dim_E = num_user, num_item, dim_E
        self.n_ui_layers, self.topk, self.sparse, self.norm_type = n_layer, ii_topk, True, 'sym'
        self.tau, self.lambda_coeff, self.n_ii_layer = ssl_temp, lambda_coeff, mm_layers
        self.device, self.user_item_dict, self.reg_weight, self.beta = device, user_item_dict, reg_weight, ssl_alpha
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
        self.image_original_adj = graph.coo_to_csr_coalesced(*self._image_original[0], self._image_original[1], num_item, num_item)
        self.text_original_adj = graph.coo_to_csr_coalesced(*self._text_original[0], self._text_original[1], num_item, num_item)
        self.image_trs = nn.Linear(v_feat.shape[1], dim_E)
        self.text_trs = nn.Linear(t_feat.shape[1], dim_E)
        self.softmax = nn.Softmax(dim=-1)
        self.query = nn.Sequential(nn.Linear(dim_E, dim_E), nn.Tanh(), nn.Linear(dim_E, 1, bias=False))
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    # ---- :176-187 -------------------------------------------------------------------------------------------------------------
    def _learned_graph(self, feats, original):
        """(1 - lambda) kNN(feats) + lambda original: the union of the two entry lists (an entry of both sums its two terms),
        row-major, the learned half differentiable."""
        n = self.num_item
        idx, w = knn_sym_entries(feats, self.topk)
        idx_o, val_o = original
        key = torch.cat((idx[0] * n + idx[1], idx_o[0] * n + idx_o[1]))
        uniq, inverse = torch.unique(key, return_inverse=True)
        val = torch.zeros(uniq.numel(), dtype=torch.float32, device=key.device).index_add(
            0, inverse, torch.cat(((1 - self.lambda_coeff) * w, self.lambda_coeff * val_o)))
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(torch.bincount(torch.div(uniq, n, rounding_mode="floor"), minlength=n), 0, out=rowptr[1:])
        return sparse.LearnedAdj(rowptr, uniq % n, val, n, n)

    def sim(self, z1, z2):
        return torch.mm(F.normalize(z1), F.normalize(z2).t())

    def batched_contrastive_loss(self, z1, z2, batch_size=1024):
        """:147-167."""
        num_nodes = z1.size(0)
        f = lambda x: torch.exp(x / self.tau)
        losses = []
        for i in range((num_nodes - 1) // batch_size + 1):
            lo, hi = i * batch_size, (i + 1) * batch_size
            refl_sim, between_sim = f(self.sim(z1[lo:hi], z1)), f(self.sim(z1[lo:hi], z2))
            losses.append(-torch.log(between_sim[:, lo:hi].diag() / (refl_sim.sum(1) + between_sim.sum(1) - refl_sim[:, lo:hi].diag())))
        return torch.cat(losses).mean()

    def forward(self, build_item_graph=False):
        """:169-225."""
        if build_item_graph:
            image_feats = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
            text_feats = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
            self.image_adj = self._learned_graph(image_feats, self._image_original)
            self.text_adj = self._learned_graph(text_feats, self._text_original)
        else:
            # (the projections feed nothing on these steps: the reference computes and drops them, :170-171)
            if self.image_adj is None or self.text_adj is None:
                raise AttributeError("MICRO.forward(build_item_graph=False) before any build: there is no item graph to detach")
            if isinstance(self.image_adj, sparse.LearnedAdj):
                self.image_adj, self.text_adj = self.image_adj.detach(), self.text_adj.detach()
        ids = self.item_embedding.weight
        for _ in range(self.n_ii_layer):
            self.image_item_embeds = sparse.mm(self.image_adj, ids)
        for _ in range(self.n_ii_layer):
            self.text_item_embeds = sparse.mm(self.text_adj, ids)
        q = lambda x: ops.linear(torch.tanh(ops.linear(x, self.query[0].weight, self.query[0].bias)), self.query[2].weight)
        weight = self.softmax(torch.cat([q(self.image_item_embeds), q(self.text_item_embeds)], dim=-1))
        self.h = weight[:, 0].unsqueeze(dim=1) * self.image_item_embeds + weight[:, 1].unsqueeze(dim=1) * self.text_item_embeds
        ego = torch.cat((self.user_embedding.weight, ids), dim=0)
        u_g, i_g = torch.split(ops.layer_mean_propagate(ego, self.graph, self.n_ui_layers), [self.num_user, self.num_item], dim=0)
        self.result = torch.cat((u_g, i_g + F.normalize(self.h, p=2, dim=1)), dim=0)
        return self.result

    # ---- :227-260 -------------------------------------------------------------------------------------------------------------
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
        contrastive_loss = self.beta * (self.batched_contrastive_loss(self.image_item_embeds, self.h)
                                        + self.batched_contrastive_loss(self.text_item_embeds, self.h))
        return (self.bpr_loss(users, pos_items, neg_items, embeddings)
                + self.regularization_loss(users, pos_items, neg_items, embeddings) + contrastive_loss)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """the table of the last forward, history at 1e-6"""
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
