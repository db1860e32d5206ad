"""SMORE with the reference's surface (Model/SMORE.py:108-455) -- spectrum-filtered modality features gating the item ids,
propagated over two item-item kNN graphs, their max-pooled fusion and the user-item graph, then attention / preference gates
and an InfoNCE between the side and the content view -- through the hot-path adapters alone: all 3 + 3 + L sparse products
of a forward (:306-343) are `chaorec_amd.sparse.mm` on the HIP SpMM, every Linear ([I, 4096] feature projections as well as
the D x D gates) is `ops.linear` on the MFMA GEMM, the ranking is `ranking.gene_ranklist` over the tables of the last forward
(:425-455).  The spectrum filter is three real FFTs over the D columns (:272-292: hipFFT through torch.fft), softmax /
sigmoid gates and InfoNCE are dense torch work on [N, D] and [B, B].

Same constructor, parameters in the reference's creation order.  The graphs are built once, on the device: the weighted
D^-1/2 A D^-1/2 of :241-262 (repeated interactions add up there: lil assignment of a coo matrix), its user-item block R,
the two cosine-kNN graphs with their symmetric normalisation over the kept weights (:24-106; rows in chunks, not one
[I, I] matrix) and their union with the larger weight (:215-239)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


def knn_sym_graph(feats, topk, chunk=4096):
    """:24-28,61-76,93-106 with is_sparse=True, norm_type='sym': cosine similarity of the rows, each row's `topk` largest
    (itself among them), weight d^-1/2[row] w d^-1/2[col] with d = the row's kept weights summed.  -> (idx [2, I topk], val)"""
    x = feats.detach().float()
    x = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    n = x.shape[0]
    vals, inds = [], []
    for s in range(0, n, chunk):
        v, i = torch.topk(x[s:s + chunk] @ x.T, topk, dim=-1)
        vals.append(v)
        inds.append(i)
    knn_val, knn_ind = torch.cat(vals), torch.cat(inds)
    row = torch.arange(n, device=x.device).repeat_interleave(topk)
    col, w = knn_ind.flatten(), knn_val.flatten()
    deg = torch.zeros(n, device=x.device).scatter_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    return torch.stack([row, col]), dis[row] * w * dis[col]


class SMORE(nn.Module):
    def __init__(self, n_users, n_items, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 reg_weight, n_ui_layers, ii_topk, dropout_rate, dataset, device):
        super(SMORE, self).__init__()
        self.n_users, self.n_items, self.user_item_dict = n_users, n_items, user_item_dict
        self.batch_size, self.sparse, self.cl_loss = 1024, True, 0.01
        self.n_ui_layers, self.embedding_dim, self.n_layers = n_ui_layers, dim_E, 1
        self.reg_weight = float(reg_weight)
        self.image_knn_k = self.text_knn_k = ii_topk
        self.dropout_rate = dropout_rate
        self.dropout = nn.Dropout(p=dropout_rate)
        self.device, self.result = device, None
        D = dim_E

        self.user_embedding = nn.Embedding(n_users, D)
        self.item_id_embedding = nn.Embedding(n_items, D)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        self.norm_adj, self.R = self.get_adj_mat(edge_index)

        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        idx, val = knn_sym_graph(v_feat.to(device), ii_topk)
        self.image_original_adj = graph.coo_to_csr_coalesced(idx[0], idx[1], val, n_items, n_items).to(device)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        idx_t, val_t = knn_sym_graph(t_feat.to(device), ii_topk)
        self.text_original_adj = graph.coo_to_csr_coalesced(idx_t[0], idx_t[1], val_t, n_items, n_items).to(device)
        self.fusion_adj = self.max_pool_fusion(idx, val, idx_t, val_t)

        self.image_trs = nn.Linear(v_feat.shape[1], D)
        self.text_trs = nn.Linear(t_feat.shape[1], D)
        self.softmax = nn.Softmax(dim=-1)
        self.query_v = nn.Sequential(nn.Linear(D, D), nn.Tanh(), nn.Linear(D, D, bias=False))
        self.query_t = nn.Sequential(nn.Linear(D, D), nn.Tanh(), nn.Linear(D, D, bias=False))
        for name in ("gate_v", "gate_t", "gate_f", "gate_image_prefer", "gate_text_prefer", "gate_fusion_prefer"):
            setattr(self, name, nn.Sequential(nn.Linear(D, D), nn.Sigmoid()))
        self.image_complex_weight = nn.Parameter(torch.randn(1, D // 2 + 1, 2, dtype=torch.float32))
        self.text_complex_weight = nn.Parameter(torch.randn(1, D // 2 + 1, 2, dtype=torch.float32))
        self.fusion_complex_weight = nn.Parameter(torch.randn(1, D // 2 + 1, 2, dtype=torch.float32))
        self.hist = ranking.history_csr(user_item_dict, n_users, device)

    def pre_epoch_processing(self):
        pass

    # ---- graphs ---------------------------------------------------------------------------------------------------------------
    def get_adj_mat(self, edge_index):
        """:241-262: A[u, U + i] = the number of times (u, i) is listed, d = row sums, d^-1/2 A d^-1/2 in fp32 (1 / 0 -> 0)."""
        U, I = self.n_users, self.n_items
        e = torch.as_tensor(np.asarray(edge_index)).long()
        key, cnt = torch.unique(e[:, 0] * I + (e[:, 1] - U), return_counts=True)
        u, i, w = torch.div(key, I, rounding_mode="floor"), key % I, cnt.to(torch.float32)
        rowsum = torch.zeros(U + I, dtype=torch.float32).index_add_(0, u, w).index_add_(0, U + i, w)
        with np.errstate(divide="ignore"):
            d = np.power(rowsum.numpy(), -0.5)
        d[np.isinf(d)] = 0.
        d = torch.from_numpy(d)
        val = (d[u] * w) * d[U + i]
        adj = graph.coo_to_csr_coalesced(torch.cat([u, U + i]), torch.cat([U + i, u]), torch.cat([val, val]), U + I, U + I,
                                         symmetric=True).to(self.device)
        return adj, graph.coo_to_csr_coalesced(u, i, val, U, I).to(self.device)

    def max_pool_fusion(self, image_idx, image_val, text_idx, text_val):
        """:215-239: the union of the two kNN graphs' entries, the larger weight where both have one."""
        n = self.n_items
        key = torch.cat((image_idx[0] * n + image_idx[1], text_idx[0] * n + text_idx[1]))
        uniq, inverse = torch.unique(key, return_inverse=True)
        val = torch.full((uniq.numel(),), float('-inf'), device=key.device)
        val = val.scatter_reduce(0, inverse, torch.cat((image_val, text_val)), reduce="amax")
        return graph.coo_to_csr_coalesced(torch.div(uniq, n, rounding_mode="floor"), uniq % n, val, n, n).to(self.device)

    # ---- :272-292 -------------------------------------------------------------------------------------------------------------
    def spectrum_convolution(self, image_embeds, text_embeds):
        image_fft = torch.fft.rfft(image_embeds, dim=1, norm='ortho')
        text_fft = torch.fft.rfft(text_embeds, dim=1, norm='ortho')
        wi, wt, wf = (torch.view_as_complex(w) for w in (self.image_complex_weight, self.text_complex_weight, self.fusion_complex_weight))
        n = image_embeds.shape[1]
        return (torch.fft.irfft(image_fft * wi, n=n, dim=1, norm='ortho'), torch.fft.irfft(text_fft * wt, n=n, dim=1, norm='ortho'),
                torch.fft.irfft(text_fft * image_fft * wf, n=n, dim=1, norm='ortho'))

    @staticmethod
    def _lin(seq, x, k=0):
        return ops.linear(x, seq[k].weight, seq[k].bias)

    # ---- :294-375 -------------------------------------------------------------------------------------------------------------
    def forward(self, adj, train=False):
        image_feats = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        text_feats = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        image_conv, text_conv, fusion_conv = self.spectrum_convolution(image_feats, text_feats)
        ids = self.item_id_embedding.weight
        image_item_embeds = ids * torch.sigmoid(self._lin(self.gate_v, image_conv))
        text_item_embeds = ids * torch.sigmoid(self._lin(self.gate_t, text_conv))
        fusion_item_embeds = ids * torch.sigmoid(self._lin(self.gate_f, fusion_conv))

        ego = torch.cat([self.user_embedding.weight, ids], dim=0)
        content_embeds = ops.layer_mean_propagate(ego, adj, self.n_ui_layers)        # :308-315: mean of the L + 1 tables

        def side(item_embeds, item_adj):                                              # :317-343
            for _ in range(self.n_layers):
                item_embeds = sparse.mm(item_adj, item_embeds)
            return torch.cat([sparse.mm(self.R, item_embeds), item_embeds], dim=0)

        image_embeds = side(image_item_embeds, self.image_original_adj)
        text_embeds = side(text_item_embeds, self.text_original_adj)
        fusion_embeds = side(fusion_item_embeds, self.fusion_adj)

        fusion_att_v = ops.linear(torch.tanh(self._lin(self.query_v, fusion_embeds)), self.query_v[2].weight)
        fusion_att_t = ops.linear(torch.tanh(self._lin(self.query_t, fusion_embeds)), self.query_t[2].weight)
        agg_image_embeds = self.softmax(fusion_att_v) * image_embeds
        agg_text_embeds = self.softmax(fusion_att_t) * text_embeds
        image_prefer = self.dropout(torch.sigmoid(self._lin(self.gate_image_prefer, content_embeds)))
        text_prefer = self.dropout(torch.sigmoid(self._lin(self.gate_text_prefer, content_embeds)))
        fusion_prefer = self.dropout(torch.sigmoid(self._lin(self.gate_fusion_prefer, content_embeds)))
        side_embeds = torch.mean(torch.stack([image_prefer * agg_image_embeds, text_prefer * agg_text_embeds,
                                              fusion_prefer * fusion_embeds]), dim=0)
        all_embeds = content_embeds + side_embeds
        users, items = torch.split(all_embeds, [self.n_users, self.n_items], dim=0)
        self.result = torch.cat((users, items), dim=0)
        if train:
            return users, items, side_embeds, content_embeds
        return users, items

    # ---- :377-423 -------------------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items):
        pos_scores = torch.sum(torch.mul(users, pos_items), dim=1)
        neg_scores = torch.sum(torch.mul(users, neg_items), dim=1)
        regularizer = (1. / 2 * (users ** 2).sum() + 1. / 2 * (pos_items ** 2).sum() + 1. / 2 * (neg_items ** 2).sum()) / self.batch_size
        return -torch.mean(F.logsigmoid(pos_scores - neg_scores)), self.reg_weight * regularizer, 0.0

    def InfoNCE(self, view1, view2, temperature):
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
        pos_score = torch.exp((view1 * view2).sum(dim=-1) / temperature)
        ttl_score = torch.exp(torch.matmul(view1, view2.transpose(0, 1)) / temperature).sum(dim=1)
        return torch.mean(-torch.log(pos_score / ttl_score))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.n_users, neg_items - self.n_users
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        ua, ia, side_embeds, content_embeds = self.forward(self.norm_adj, train=True)
        mf, emb, reg = self.bpr_loss(ua[users], ia[pos_items], ia[neg_items])
        side_u, side_i = torch.split(side_embeds, [self.n_users, self.n_items], dim=0)
        content_u, content_i = torch.split(content_embeds, [self.n_users, self.n_items], dim=0)
        cl_loss = self.InfoNCE(side_i[pos_items], content_i[pos_items], 0.2) + self.InfoNCE(side_u[users], content_u[users], 0.2)
        return mf + emb + reg + self.cl_loss * cl_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:425-455: the tables of the last forward, history at 1e-6."""
        return ranking.gene_ranklist(self.result.detach(), self.n_users, self.n_items, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
