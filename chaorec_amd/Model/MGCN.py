"""MGCN with the reference's surface (Model/MGCN.py:70-354), compute on HIP kernels -- the first member of the
`torch.sparse.mm` model family carried over: every sparse product goes through sparse.mm / ops.spmm.

Same constructor, parameters and state_dict keys (user/item embeddings, trainable image/text feature tables, image_trs /
text_trs, query_common, the four gates -- created in the reference's order, so the same torch seed gives the same
weights), `forward()` -> (users, items, side, content), `bpr_loss`, `regularization_loss`, `InfoNCE`, `loss()`,
`gene_ranklist()`.

What changed underneath:
  * D^-1/2 A D^-1/2 and its user x item block R are built vectorised into CSRs in HBM (the reference assembles them
    through scipy dok/lil matrices, Model/MGCN.py:158-183); values follow the reference's fp32 arithmetic
    (np.power(rowsum, -0.5), (d_i * a_ij) * d_j);
  * the item-item kNN graphs (Model/MGCN.py:14-17,57-68,113-122) come from the scoring + top-K kernel: the [I, I]
    cosine matrix is never materialised and no Python list of I*k pairs is built;
  * the five torch.sparse.mm per forward (:229,:239-240,:245-246) are CSR SpMM launches, the user-item propagate with
    its layer mean is the fused LightGCN propagate; every nn.Linear is the f32 MFMA GEMM; BPR + L2 is the fused kernel.
The small dense remainder (tanh / sigmoid / softmax gates, InfoNCE on a [B, D] batch) stays in torch ops.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


def _sym_normalised_knn(feat, k, device):
    """build_sim + build_knn_normalized_graph(is_sparse=True, norm_type='sym') (Model/MGCN.py:14-17,57-68,20-33):
    top-k cosine neighbours (self included), weights = the similarities, deg = row sums of the weights,
    w' = deg^-1/2[row] * w * deg^-1/2[col] (inf -> 0).  Returns a graph.CSR [n, n] and its transpose."""
    emb = feat.to(device)
    norm = emb.div(torch.norm(emb, p=2, dim=-1, keepdim=True))
    d = norm.shape[1]
    d_pad = next(c for c in (8, 16, 32, 64, 128) if c >= d) if d <= 128 else (d + 63) // 64 * 64
    norm = F.pad(norm, (0, d_pad - d)).contiguous()          # zero columns leave the cosine unchanged
    ind, val = ops.score_topk(norm, norm, None, 0.0, k)
    n = norm.shape[0]
    row = torch.arange(n, device=device).unsqueeze(1).expand(-1, k).reshape(-1)
    col, w = ind.reshape(-1), val.reshape(-1)
    deg = torch.zeros(n, dtype=torch.float32, device=device).scatter_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float('inf'), 0)
    w = dis[row] * w * dis[col]
    csr = graph.coo_to_csr_coalesced(row, col, w, n, n, symmetric=False)
    csr.t()
    return csr


class MGCN(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, reg_weight,
                 n_layers, aggr_mode, ssl_temp, ssl_alpha, device):
        super(MGCN, self).__init__()
        self.result = None
        self.num_user = num_user
        self.num_item = num_item
        self.edge_index = edge_index
        self.user_item_dict = user_item_dict
        self.dim_E = dim_E
        self.n_layers = 1                 # reference: hard-coded (the constructor argument is ignored, :81-82)
        self.n_ui_layers = 2
        self.ssl_temp = ssl_temp
        self.ssl_alpha = ssl_alpha
        self.device = device
        self.v_feat = v_feat
        self.t_feat = t_feat
        self.reg_weight = reg_weight
        self.aggr_mode = aggr_mode
        self.sparse = True
        self.knn_k = 10

        self.user_embedding = nn.Embedding(self.num_user, self.dim_E)
        self.item_embedding = nn.Embedding(self.num_item, self.dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)

        self.norm_adj, self.R = self.get_adj_mat(np.asarray(edge_index))
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))

        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        self.image_original_adj = _sym_normalised_knn(self.image_embedding.weight.detach(), self.knn_k, device)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        self.text_original_adj = _sym_normalised_knn(self.text_embedding.weight.detach(), self.knn_k, device)
        # the trainable feature tables are read only through their projections (ops.linear below): optim.FusedAdam may
        # apply their rank-64 gradient gy W without materialising it (chaorec_adam_lowrank_f32)
        self.image_embedding.weight._chaorec_projected_only = True
        self.text_embedding.weight._chaorec_projected_only = True

        self.image_trs = nn.Linear(v_feat.shape[1], self.dim_E)
        self.text_trs = nn.Linear(t_feat.shape[1], self.dim_E)
        self.softmax = nn.Softmax(dim=-1)
        self.query_common = nn.Sequential(nn.Linear(self.dim_E, self.dim_E), nn.Tanh(),
                                          nn.Linear(self.dim_E, 1, bias=False))
        self.gate_v = nn.Sequential(nn.Linear(self.dim_E, self.dim_E), nn.Sigmoid())
        self.gate_t = nn.Sequential(nn.Linear(self.dim_E, self.dim_E), nn.Sigmoid())
        self.gate_image_prefer = nn.Sequential(nn.Linear(self.dim_E, self.dim_E), nn.Sigmoid())
        self.gate_text_prefer = nn.Sequential(nn.Linear(self.dim_E, self.dim_E), nn.Sigmoid())

    def get_adj_mat(self, edges):
        """Model/MGCN.py:158-183: A = [[0, R0], [R0^T, 0]] with R0 the interaction counts, D^-1/2 A D^-1/2 in fp32
        ((d_i * a) * d_j, empty rows -> 1e-16 before the power), and its user x item block.  -> (CSR [N,N], CSR [U,I])."""
        U, I, N = self.num_user, self.num_item, self.num_user + self.num_item
        u = edges[:, 0].astype(np.int64)
        i = edges[:, 1].astype(np.int64) - U
        key, cnt = np.unique(u * I + i, return_counts=True)         # coo -> lil sums repeated interactions
        u, i, a = key // I, key % I, cnt.astype(np.float32)
        rowsum = np.zeros(N, dtype=np.float32)
        np.add.at(rowsum, u, a)
        np.add.at(rowsum, U + i, a)
        rowsum[rowsum == 0.] = 1e-16
        d = np.power(rowsum, np.float32(-0.5)).astype(np.float32)
        d[np.isinf(d)] = 0.
        val = ((d[u] * a) * d[U + i]).astype(np.float32)
        val_t = ((d[U + i] * a) * d[u]).astype(np.float32)
        tu, ti, tv, tvt = (torch.from_numpy(x) for x in (u, i, val, val_t))
        full = graph.coo_to_csr_coalesced(torch.cat([tu, ti + U]), torch.cat([ti + U, tu]), torch.cat([tv, tvt]), N, N,
                                          symmetric=True)
        block = graph.coo_to_csr_coalesced(tu, ti, tv, U, I)
        block.t()
        return full.to(self.device), block.to(self.device)

    @staticmethod
    def _lin(seq, x, act=None):
        y = ops.linear(x, seq[0].weight, seq[0].bias)
        return act(y) if act is not None else y

    def forward(self):
        """Model/MGCN.py:214-268; side effect: self.result."""
        image_feats = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        text_feats = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        item_embeds, user_embeds = self.item_embedding.weight, self.user_embedding.weight
        image_item_embeds = item_embeds * self._lin(self.gate_v, image_feats, torch.sigmoid)
        text_item_embeds = item_embeds * self._lin(self.gate_t, text_feats, torch.sigmoid)

        # user-item view: mean of the ego table and n_ui_layers propagated copies (:224-233)
        ego = torch.cat([user_embeds, item_embeds], dim=0)
        content_embeds = ops.layer_mean_propagate(ego, self.norm_adj, self.n_ui_layers)

        # item-item view, then lifted to the users through R (:235-248)
        for _ in range(self.n_layers):
            image_item_embeds = sparse.mm(self.image_original_adj, image_item_embeds)
            text_item_embeds = sparse.mm(self.text_original_adj, text_item_embeds)
        image_embeds = torch.cat([sparse.mm(self.R, image_item_embeds), image_item_embeds], dim=0)
        text_embeds = torch.cat([sparse.mm(self.R, text_item_embeds), text_item_embeds], dim=0)

        # behaviour-aware fuser (:250-263)
        q = self.query_common
        att = torch.cat([ops.linear(torch.tanh(ops.linear(e, q[0].weight, q[0].bias)), q[2].weight)
                         for e in (image_embeds, text_embeds)], dim=-1)
        weight_common = self.softmax(att)
        common_embeds = weight_common[:, 0].unsqueeze(dim=1) * image_embeds + \
            weight_common[:, 1].unsqueeze(dim=1) * text_embeds
        image_prefer = self._lin(self.gate_image_prefer, content_embeds, torch.sigmoid)
        text_prefer = self._lin(self.gate_text_prefer, content_embeds, torch.sigmoid)
        sep_image_embeds = image_prefer * (image_embeds - common_embeds)
        sep_text_embeds = text_prefer * (text_embeds - common_embeds)
        side_embeds = (sep_image_embeds + sep_text_embeds + common_embeds) / 3

        all_embeds = content_embeds + side_embeds
        self.result = all_embeds
        users, items = torch.split(all_embeds, [self.num_user, self.num_item], dim=0)
        return users, items, side_embeds, content_embeds

    def _fused(self, users, pos_items, neg_items, u_g, i_g, reg):
        return ops.bpr_loss(u_g, i_g, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, reg)

    def bpr_loss(self, users, pos_items, neg_items, u_g, i_g):
        """Model/MGCN.py:270-281."""
        return self._fused(users, pos_items, neg_items, u_g, i_g, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items, u_g, i_g):
        """Model/MGCN.py:283-292."""
        return self._fused(users, pos_items, neg_items, u_g, i_g, self.reg_weight)[2]

    def InfoNCE(self, view1, view2):
        """Model/MGCN.py:294-301."""
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
        pos_score = torch.exp((view1 * view2).sum(dim=-1) / self.ssl_temp)
        ttl_score = torch.exp(ops.linear(view1, view2) / self.ssl_temp).sum(dim=1)
        return ops.mean_all(-torch.log(pos_score / ttl_score))

    def loss(self, users, pos_items, neg_items):
        """Model/MGCN.py:303-320."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        return self.loss_local(users, pos_items, neg_items)

    def loss_local(self, users, pos_items, neg_items):
        ua, ia, side_embeds, content_embeds = self.forward()
        bpr_reg = self._fused(users, pos_items, neg_items, ua.contiguous(), ia.contiguous(), self.reg_weight)[0]
        side_u, side_i = torch.split(side_embeds, [self.num_user, self.num_item], dim=0)
        content_u, content_i = torch.split(content_embeds, [self.num_user, self.num_item], dim=0)
        ssl_loss = self.InfoNCE(side_i[pos_items], content_i[pos_items]) + self.InfoNCE(side_u[users], content_u[users])
        return bpr_reg + self.ssl_alpha * ssl_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/MGCN.py:322-354 (mask value 1e-6)."""
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
