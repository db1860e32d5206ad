"""LGMRec with the reference's surface (Model/LGMRec.py:16-272) -- local graph embeddings (a LightGCN propagate of the id
tables + two propagations of every modality's projected features) plus global hypergraph embeddings (four learned
hyperedges per modality, Gumbel-softmax memberships), through the hot-path adapters alone: the id propagate is the fused
layer-mean SpMM chain (`ops.layer_mean_propagate`, :116-127), the modality propagations and the user <- item aggregations
over the raw interaction matrix are `chaorec_amd.sparse.mm` (:139-145,150,156), the feature projections run on the MFMA
GEMM (`ops.linear`), the ranking is `ranking.gene_ranklist` on a fresh forward (:245-272).  The hypergraph layer multiplies
[I, 4] / [U, 4] membership matrices (:23-31) and the hypergraph InfoNCE is a [B, U] matmul (:211-218): dense torch work.

Same constructor, parameters in the reference's creation order (user / item tables, then per modality the frozen feature
table, its projection and its hyperedge matrix).  Randomness: the four Gumbel draws and the four dropouts of a forward
come from torch's device generator; `gumbel_fn(logits) -> noise` and `drop_fn(x) -> scaled keep mask` replace them (the
golden test feeds the reference run's recorded draws)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


class HGNNLayer(nn.Module):
    """:16-31: item embeddings -> hyperedges -> items and users, n_hyper_layer times."""

    def __init__(self, n_hyper_layer):
        super().__init__()
        self.h_layer = n_hyper_layer

    def forward(self, i_hyper, u_hyper, embeds):
        u_ret, i_ret = None, embeds
        for _ in range(self.h_layer):
            lat = torch.mm(i_hyper.T, i_ret)
            i_ret = torch.mm(i_hyper, lat)
            u_ret = torch.mm(u_hyper, lat)
        return u_ret, i_ret


class LGMRec(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 reg_weight, n_layers, ssl_alpha, device):
        super(LGMRec, self).__init__()
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.reg_weight, self.device = dim_E, reg_weight, device
        self.n_mm_layer, self.n_ui_layers, self.n_hyper_layer, self.hyper_num = 2, n_layers, 1, 4
        self.keep_rate, self.tau, self.ssl_reg, self.alpha = 0.2, 0.2, ssl_alpha, 0.2
        self.cf_model = 'lightgcn'
        self.n_nodes = num_user + num_item
        self.hgnnLayer = HGNNLayer(self.n_hyper_layer)

        e = torch.as_tensor(np.asarray(edge_index)).long()
        u, i = e[:, 0], e[:, 1] - num_user
        # :60-66: the interaction matrix as given (a repeated interaction counts twice in torch.sparse.mm: coalesced = summed)
        self.adj = graph.coo_to_csr_coalesced(u, i, torch.ones(len(u)), num_user, num_item).to(device)
        # :98-113: binary A, degree + 1e-7, D^-1/2 A D^-1/2; :71: 1 / (degree + 1e-7) per node
        self.norm_adj = graph.binary_sym_norm_csr(u, i, num_user, num_item).to(device)
        key = torch.unique(u * num_item + i)
        deg = torch.zeros(num_user, dtype=torch.float64).index_add_(0, torch.div(key, num_item, rounding_mode="floor"),
                                                                      torch.ones(key.numel(), dtype=torch.float64))
        self.num_inters = (1.0 / (deg + 1e-7)).to(torch.float32).unsqueeze(1).to(device)          # [U, 1] (users are all it is read for)

        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.drop = nn.Dropout(p=1 - self.keep_rate)
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=True)
        self.item_image_trs = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(v_feat.shape[1], dim_E)))
        self.v_hyper = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(v_feat.shape[1], self.hyper_num)))
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=True)
        self.item_text_trs = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(t_feat.shape[1], dim_E)))
        self.t_hyper = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(t_feat.shape[1], self.hyper_num)))
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.gumbel_fn = self.drop_fn = None

    # ---- randomness -------------------------------------------------------------------------------------------------------
    def _gumbel_softmax(self, logits):
        """F.gumbel_softmax(logits, tau, dim=1, hard=False) (:152-153,159-160)."""
        g = self.gumbel_fn(logits) if self.gumbel_fn is not None else -torch.empty_like(logits).exponential_().log()
        return ((logits + g) / self.tau).softmax(dim=1)

    def _drop(self, x):
        return x * self.drop_fn(x) if self.drop_fn is not None else self.drop(x)

    # ---- :116-145 -----------------------------------------------------------------------------------------------------------
    def cge(self):
        ego = torch.cat((self.user_embedding.weight, self.item_embedding.weight), dim=0)
        if self.cf_model == 'mf':
            return ego
        return ops.layer_mean_propagate(ego, self.norm_adj, self.n_ui_layers)

    def mge(self, str='v'):
        feats, trs = (self.image_embedding.weight, self.item_image_trs) if str == 'v' else (self.text_embedding.weight, self.item_text_trs)
        item_feats = ops.linear(feats, trs.t().contiguous())
        user_feats = sparse.mm(self.adj, item_feats) * self.num_inters
        mge_feats = torch.concat([user_feats, item_feats], dim=0)
        for _ in range(self.n_mm_layer):
            mge_feats = sparse.mm(self.norm_adj, mge_feats)
        return mge_feats

    # ---- :147-191 -----------------------------------------------------------------------------------------------------------
    def forward(self):
        iv_hyper = ops.linear(self.image_embedding.weight, self.v_hyper.t().contiguous())
        uv_hyper = sparse.mm(self.adj, iv_hyper)
        iv_hyper, uv_hyper = self._gumbel_softmax(iv_hyper), self._gumbel_softmax(uv_hyper)
        it_hyper = ops.linear(self.text_embedding.weight, self.t_hyper.t().contiguous())
        ut_hyper = sparse.mm(self.adj, it_hyper)
        it_hyper, ut_hyper = self._gumbel_softmax(it_hyper), self._gumbel_softmax(ut_hyper)

        cge_embs = self.cge()
        v_feats, t_feats = self.mge('v'), self.mge('t')
        lge_embs = cge_embs + (F.normalize(v_feats) + F.normalize(t_feats))

        items = cge_embs[self.num_user:]
        uv_hyper_embs, iv_hyper_embs = self.hgnnLayer(self._drop(iv_hyper), self._drop(uv_hyper), items)
        ut_hyper_embs, it_hyper_embs = self.hgnnLayer(self._drop(it_hyper), self._drop(ut_hyper), items)
        ghe_embs = torch.concat([uv_hyper_embs, iv_hyper_embs], dim=0) + torch.concat([ut_hyper_embs, it_hyper_embs], dim=0)
        all_embs = lge_embs + self.alpha * F.normalize(ghe_embs)
        u_embs, i_embs = torch.split(all_embs, [self.num_user, self.num_item], dim=0)
        return u_embs, i_embs, [uv_hyper_embs, iv_hyper_embs, ut_hyper_embs, it_hyper_embs]

    # ---- :193-243 -----------------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        return ops.bpr_loss(user_emb.contiguous(), item_emb.contiguous(), users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items, u_g, i_g):
        return self.reg_weight * (ops.mean_all(u_g[users] ** 2) + ops.mean_all(i_g[pos_items] ** 2) + ops.mean_all(i_g[neg_items] ** 2))

    def ssl_triple_loss(self, emb1, emb2, all_emb):
        norm_emb1, norm_emb2, norm_all_emb = F.normalize(emb1), F.normalize(emb2), F.normalize(all_emb)
        pos_score = torch.exp(torch.mul(norm_emb1, norm_emb2).sum(dim=1) / self.tau)
        ttl_score = torch.exp(torch.matmul(norm_emb1, norm_all_emb.T) / self.tau).sum(dim=1)
        return -torch.log(pos_score / ttl_score).sum()

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        ua, ia, (uv_embs, iv_embs, ut_embs, it_embs) = self.forward()
        batch_hcl_loss = self.ssl_triple_loss(uv_embs[users], ut_embs[users], ut_embs) + \
            self.ssl_triple_loss(iv_embs[pos_items], it_embs[pos_items], it_embs)
        return self.bpr_loss(users, pos_items, neg_items, ua, ia) + self.ssl_reg * batch_hcl_loss + \
            self.regularization_loss(users, pos_items, neg_items, ua, ia)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:245-272: a fresh forward (its own Gumbel draws), history at 1e-6."""
        with torch.no_grad():
            u, i, _ = self.forward()
            self.result = torch.cat([u, i], 0)
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
