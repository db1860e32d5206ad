"""DDRec with the reference's surface (Model/DDRec.py:17-315) -- three LightGCN-style encoders over the user-item graph (ids,
visual, textual), the two modality encoders on a graph that is RE-FILTERED in every layer (an interaction stays only while
the current user and item rows score >= threshold), an item-item kNN graph on top, four InfoNCE terms between the views.

The reference forms a dense [U, I] score matrix per layer and branch only to read it at the E interactions (:110-113,
:197-204), builds a new edge list from the survivors and lets `BasicGCN.GCNConv` count its degrees (BasicGCN.py:60-79).
Here the graph's STRUCTURE never changes: one symmetric CSR over the distinct interactions, and per layer
  * the E scores as row-wise dot products of the gathered rows (no [U, I] matrix),
  * the survivors' degrees and  deg^-1/2[u] deg^-1/2[i]  per entry as a VALUE array over that CSR (0 on a filtered entry),
  * the propagate as the dynamic-values HIP SpMM (`sparse.DroppedAdj`, forward and backward) --
the same treatment as MMGCL's dropped graphs.  The id encoder is `ops.layer_mean_propagate` over the static values, the three
item-item products `chaorec_amd.sparse.mm`, every Linear `ops.linear` on the MFMA GEMM, the ranking
`ranking.gene_ranklist` over the concatenated tables of the last forward (:291-315).

Same constructor, parameters in the reference's creation order (the two frozen feature tables included: they are Parameters
without gradient there too).  Kept quirk: `final_i_g_embeddings` of the PREVIOUS forward gates the modality features of the
next one (:98-101) -- the first forward runs ungated."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


def knn_binary_graph(feats, topk, chunk=4096):
    """:71-93: cosine kNN (the row itself among them), binary adjacency, (1e-7 + row sum)^-1/2 on both ends.
    -> (idx [2, n topk], val)"""
    x = feats.detach().float()
    x = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    n = x.shape[0]
    knn_ind = torch.cat([torch.topk(x[s:s + chunk] @ x.T, topk, dim=-1)[1] for s in range(0, n, chunk)])
    row = torch.arange(n, device=x.device).repeat_interleave(topk)
    col = knn_ind.flatten()
    r_inv_sqrt = torch.pow(1e-7 + torch.bincount(row, minlength=n).to(torch.float32), -0.5)
    return torch.stack([row, col]), r_inv_sqrt[row] * r_inv_sqrt[col]


class DDRec(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, feat_E,
                 reg_weight, n_layers, ssl_temp, ssl_alpha, threshold, aggr_mode, device):
        super(DDRec, self).__init__()
        self.final_i_g_embeddings = self.i_g_embeddings = self.t_embedding = self.v_embedding = self.result = None
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.feat_E, self.reg_weight, self.n_layers = dim_E, feat_E, reg_weight, n_layers
        self.aggr_mode, self.device = aggr_mode, device
        self.mm_layers, self.knn_k, self.mm_image_weight = 1, 10, 0.5
        self.threshold, self.ssl_temp, self.ssl_alpha = threshold, ssl_temp, ssl_alpha

        self.user_embedding = nn.Embedding(num_user, dim_E)
        nn.init.xavier_normal_(self.user_embedding.weight)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_normal_(self.item_embedding.weight)
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=True)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=True)
        self.image_trs = nn.Linear(v_feat.shape[1], feat_E)
        nn.init.xavier_normal_(self.image_trs.weight)
        self.text_trs = nn.Linear(t_feat.shape[1], feat_E)
        nn.init.xavier_normal_(self.text_trs.weight)
        self.guide_image_trs = nn.Sequential(nn.Linear(feat_E, feat_E), nn.Sigmoid())
        self.guide_text_trs = nn.Sequential(nn.Linear(feat_E, feat_E), nn.Sigmoid())

        U, I = num_user, num_item
        self._pairs = sparse.PairStructure(edge_index, U, I, device)      # distinct interactions + one symmetric [N, N] structure
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self._lower, both = self._pairs.lower, self._pairs.csr
        both.val.copy_(self._values(self._ew))                      # the unfiltered graph: the id encoder's (:131-137)
        self.norm_adj = both
        self._structure = self._pairs.structure

        idx_v, val_v = knn_binary_graph(v_feat.to(device), self.knn_k)
        idx_t, val_t = knn_binary_graph(t_feat.to(device), self.knn_k)
        self.image_adj = graph.coo_to_csr_coalesced(idx_v[0], idx_v[1], val_v, I, I).to(device)
        self.text_adj = graph.coo_to_csr_coalesced(idx_t[0], idx_t[1], val_t, I, I).to(device)
        self.mm_adj = graph.add_scaled_coo((idx_v, val_v), self.mm_image_weight, (idx_t, val_t), 1.0 - self.mm_image_weight,
                                           I).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    # ---- graphs -------------------------------------------------------------------------------------------------------------
    def _values(self, w):
        """BasicGCN.py:67-71 for the pair weights w [n_edges] (multiplicity, 0 = filtered): degree = the weights at the node,
        deg^-1/2[u] deg^-1/2[i] per listed edge -- w times that per entry of the symmetric CSR (a node without a surviving
        edge has no entry to scale: its deg^-1/2 is taken as 0 instead of inf)."""
        U, N = self.num_user, self.num_user + self.num_item
        deg = torch.zeros(N, dtype=torch.float32, device=w.device).index_add_(0, self._eu, w).index_add_(0, U + self._ei, w)
        d = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
        val = w * (d[self._eu] * d[U + self._ei])
        return torch.cat([val, val[self._lower]])

    def filter_edges(self, ego):
        """:197-204 -> the weights of the interactions whose current rows score >= threshold (the score of an interaction is
        the dot product of its two rows: the one entry the reference reads from its [U, I] product)."""
        with torch.no_grad():
            sim = ops.edge_dot_raw(self._structure.entry_row, self._structure.col, ego, ego, self.n_edges)
            return torch.where(sim >= self.threshold, self._ew, torch.zeros_like(self._ew))

    def _filtered_encoder(self, item_table):
        """:103-118 / :120-135: L layers, each over the graph filtered by ITS input rows; the mean of the L + 1 tables."""
        ego = torch.cat((self.user_embedding.weight, item_table), dim=0)
        total = ego
        for _ in range(self.n_layers):
            val = self._values(self.filter_edges(ego))
            ego = sparse.mm(sparse.DroppedAdj(self._structure, val, val), ego)          # (symmetric: its own transpose)
            total = total + ego
        return torch.split(total / (self.n_layers + 1), [self.num_user, self.num_item], dim=0)

    @staticmethod
    def _lin(seq, x):
        return ops.linear(x, seq[0].weight, seq[0].bias)

    # ---- :95-195 ------------------------------------------------------------------------------------------------------------
    def forward(self):
        self.v_embedding = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        self.t_embedding = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        visual_tensor, text_tensor = self.v_embedding, self.t_embedding
        if self.final_i_g_embeddings is not None:
            item_embedding = self.final_i_g_embeddings.detach()
            visual_tensor = item_embedding * torch.sigmoid(self._lin(self.guide_image_trs, self.v_embedding))
            text_tensor = item_embedding * torch.sigmoid(self._lin(self.guide_text_trs, self.t_embedding))
        self.u_v_embeddings, i_v = self._filtered_encoder(visual_tensor)
        self.u_t_embeddings, i_t = self._filtered_encoder(text_tensor)
        ego = torch.cat((self.user_embedding.weight, self.item_embedding.weight), dim=0)
        self.u_g_embeddings, self.i_g_embeddings = torch.split(ops.layer_mean_propagate(ego, self.norm_adj, self.n_layers),
                                                               [self.num_user, self.num_item], dim=0)

        def lifted(h0):
            h = h0
            for _ in range(self.mm_layers):
                h = sparse.mm(self.mm_adj, h)
            return h0 + h

        self.final_i_g_embeddings = lifted(self.i_g_embeddings)
        self.i_v_embeddings, self.i_t_embeddings = lifted(i_v), lifted(i_t)
        self.i = torch.cat((self.final_i_g_embeddings, self.i_v_embeddings, self.i_t_embeddings), dim=1)
        self.u = torch.cat((self.u_g_embeddings, self.u_v_embeddings, self.u_t_embeddings), dim=1)
        self.result = torch.cat((self.u, self.i), dim=0)
        return self.result

    # ---- :206-289 -----------------------------------------------------------------------------------------------------------
    def main_bpr_loss(self, users, pos_items, neg_items, embedding):
        u, p, n = embedding[users], embedding[pos_items + self.num_user], embedding[neg_items + self.num_user]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items, embedding):
        u, p, n = embedding[users], embedding[self.num_user + pos_items], embedding[self.num_user + neg_items]
        return self.reg_weight * (torch.mean(u ** 2) + torch.mean(p ** 2) + torch.mean(n ** 2))

    def ssl_compute(self, embedded_s1, embedded_s2, users_or_pos_items):
        s1, s2 = F.normalize(embedded_s1[users_or_pos_items], dim=1), F.normalize(embedded_s2[users_or_pos_items], dim=1)
        pos_score = torch.sum(torch.mul(s1, s2), dim=1, keepdim=False)
        all_score = torch.mm(s1, s2.t())
        return -torch.log(torch.exp(pos_score / self.ssl_temp) / torch.exp(all_score / self.ssl_temp).sum(dim=1, keepdim=False)).mean()

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        embedding = self.forward()
        bpr_loss = self.main_bpr_loss(users, pos_items, neg_items, embedding)
        cl_loss = self.ssl_alpha * (self.ssl_compute(self.u_v_embeddings, self.u_g_embeddings, users)
                                    + self.ssl_compute(self.i_v_embeddings, self.final_i_g_embeddings, pos_items)
                                    + self.ssl_compute(self.u_t_embeddings, self.u_g_embeddings, users)
                                    + self.ssl_compute(self.i_t_embeddings, self.final_i_g_embeddings, pos_items))
        return bpr_loss + self.regularization_loss(users, pos_items, neg_items, embedding) + cl_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:291-315: the [N, 3 D] table of the last forward, history at 1e-6."""
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
