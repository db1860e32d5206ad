"""GUME with the reference's surface (Model/GUME.py:93-492) -- a user-item graph ENHANCED with item-item edges (the items that
are neighbours in BOTH modalities' cosine-kNN graphs), ids and modality-gated ids propagated over it, the modality views
propagated over their own kNN graphs and lifted to the users through the R block, a coarse / fine attribute separation, and
three InfoNCE terms (two of them between randomly perturbed views) -- through the hot-path adapters alone: the 3 x L_ui +
2 x L + 2 sparse products of a forward (:306-347) are the HIP SpMM (`ops.layer_mean_propagate` for the three layer means,
`chaorec_amd.sparse.mm` for the rest), every Linear -- the two [I, F] feature projections as well as the D x D gates -- is
`ops.linear` on the MFMA GEMM, the ranking is `ranking.gene_ranklist` over the table of the last forward (:467-492).

Same constructor, parameters in the reference's creation order.  The graphs are built once, vectorised on the device:
  * the two cosine-kNN graphs (knn_k = 10, symmetric normalisation over the kept weights: :37-40,77-87 -- SMORE's builder);
  * the modality intersection of :215-244 (a Python loop over 10 I index pairs there, cached as Data/<dataset>/gume_inter.json;
    that cache is read here when it exists -- the reference would --, and written only where its directory exists);
  * the enhanced adjacency [[0, R], [R^T, S]] with S the 0/1 intersection graph (NOT symmetric: v is near id in both
    modalities, id need not be near v), d = its row sums, (d^-1/2[r] a) d^-1/2[c] in fp32 (:268-296); repeated interactions
    add up (lil assignment of a coo matrix).
The perturbation of :446-457 draws `torch.rand_like` on the device (`noise_fn` replays stored draws in the golden test)."""
import json
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse
from .SMORE import knn_sym_graph


class GUME(nn.Module):
    def __init__(self, n_users, n_items, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 n_layers, n_ui_layers, um_loss, vt_loss, dataset_path, device):
        super(GUME, self).__init__()
        self.result = None
        self.n_users, self.n_items, self.user_item_dict = n_users, n_items, user_item_dict
        self.sparse = True
        self.bm_loss, self.um_loss, self.vt_loss = 0.01, um_loss, vt_loss
        self.reg_weight_1, self.reg_weight_2 = float(1e-05), 0.1
        self.bm_temp = self.um_temp = 0.2
        self.n_ui_layers, self.embedding_dim, self.knn_k, self.n_layers = n_ui_layers, dim_E, 10, n_layers
        self.dataset_path, self.device, self.batch_size = dataset_path, device, 1024
        self.noise_fn = None
        D = dim_E

        self.user_embedding = nn.Embedding(n_users, D)
        self.item_id_embedding = nn.Embedding(n_items, D)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        self.extended_image_user = nn.Embedding(n_users, D)
        nn.init.xavier_uniform_(self.extended_image_user.weight)
        self.extended_text_user = nn.Embedding(n_users, D)
        nn.init.xavier_uniform_(self.extended_text_user.weight)

        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        idx_v, val_v = knn_sym_graph(v_feat.to(device), self.knn_k)
        self.image_original_adj = graph.coo_to_csr_coalesced(idx_v[0], idx_v[1], val_v, n_items, n_items).to(device)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        idx_t, val_t = knn_sym_graph(t_feat.to(device), self.knn_k)
        self.text_original_adj = graph.coo_to_csr_coalesced(idx_t[0], idx_t[1], val_t, n_items, n_items).to(device)

        self.inter = self.find_inter(idx_v, idx_t)
        self.norm_adj, self.R = self.get_adj_mat(edge_index, self.inter)

        self.image_reduce_dim = nn.Linear(v_feat.shape[1], D)
        self.image_trans_dim = nn.Sequential(nn.Linear(D, D), nn.Sigmoid())
        self.image_space_trans = nn.Sequential(self.image_reduce_dim, self.image_trans_dim)
        self.text_reduce_dim = nn.Linear(t_feat.shape[1], D)
        self.text_trans_dim = nn.Sequential(nn.Linear(D, D), nn.Sigmoid())
        self.text_space_trans = nn.Sequential(self.text_reduce_dim, self.text_trans_dim)
        self.separate_coarse = nn.Sequential(nn.Linear(D, D), nn.Tanh(), nn.Linear(D, 1, bias=False))
        self.softmax = nn.Softmax(dim=-1)
        self.image_behavior = nn.Sequential(nn.Linear(D, D), nn.Sigmoid())
        self.text_behavior = nn.Sequential(nn.Linear(D, D), nn.Sigmoid())
        self.tau = 0.5
        self.hist = ranking.history_csr(user_item_dict, n_users, device)

    def pre_epoch_processing(self):
        pass

    # ---- graphs ---------------------------------------------------------------------------------------------------------------
    def find_inter(self, image_idx, text_idx):
        """:215-244 -> [2, n] (item, neighbour) pairs: neighbours of the item in BOTH kNN graphs, the item itself left out."""
        inter_file = os.path.join('Data', str(self.dataset_path), 'gume_inter.json')
        if os.path.exists(inter_file):
            with open(inter_file) as f:
                inter = json.load(f)
            pairs = [(int(k), int(v)) for k, vs in inter.items() for v in vs]
            return torch.tensor(pairs, dtype=torch.int64).reshape(-1, 2).t().contiguous()
        n = self.n_items
        key_v = image_idx[0] * n + image_idx[1]
        key_t = text_idx[0] * n + text_idx[1]
        both = key_v[torch.isin(key_v, key_t)]
        row, col = torch.div(both, n, rounding_mode="floor"), both % n
        keep = row != col
        pairs = torch.stack([row[keep], col[keep]]).cpu()
        if os.path.isdir(os.path.dirname(inter_file)):
            inter = {i: [] for i in range(n)}
            for r, c in pairs.t().tolist():
                inter[r].append(c)
            with open(inter_file, "w") as f:
                json.dump(inter, f)
        return pairs

    def get_adj_mat(self, edge_index, inter):
        """:246-296: the interaction counts and the 0/1 item-item block, normalised by the ROW sums on both sides."""
        U, I = self.n_users, self.n_items
        e = torch.as_tensor(np.asarray(edge_index)).long()
        key, cnt = torch.unique(e[:, 0] * I + (e[:, 1] - U), return_counts=True)
        u, i, w = torch.div(key, I, rounding_mode="floor"), key % I, cnt.to(torch.float32)
        skey = torch.unique(inter[0] * I + inter[1])                                  # (a coo matrix of ones: a pair listed once)
        si, sj = torch.div(skey, I, rounding_mode="floor"), skey % I
        ones = torch.ones(skey.numel(), dtype=torch.float32)
        rowsum = torch.zeros(U + I, dtype=torch.float32).index_add_(0, u, w).index_add_(0, U + i, w).index_add_(0, U + si, ones)
        with np.errstate(divide="ignore"):
            d = np.power(rowsum.numpy(), -0.5)
        d[np.isinf(d)] = 0.
        d = torch.from_numpy(d)
        val = (d[u] * w) * d[U + i]
        val_t = (d[U + i] * w) * d[u]
        val_s = (d[U + si] * ones) * d[U + sj]
        adj = graph.coo_to_csr_coalesced(torch.cat([u, U + i, U + si]), torch.cat([U + i, u, U + sj]),
                                         torch.cat([val, val_t, val_s]), U + I, U + I).to(self.device)
        return adj, graph.coo_to_csr_coalesced(u, i, val, U, I).to(self.device)

    # ---- :306-378 -------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _lin(seq, x, k=0):
        return ops.linear(x, seq[k].weight, seq[k].bias)

    def conv_ui(self, adj, user_embeds, item_embeds):
        return ops.layer_mean_propagate(torch.cat([user_embeds, item_embeds], dim=0), adj, self.n_ui_layers)

    def conv_ii(self, ii_adj, single_modal):
        for _ in range(self.n_layers):
            single_modal = sparse.mm(ii_adj, single_modal)
        return single_modal

    def forward(self, adj, train=False):
        ids = self.item_id_embedding.weight
        space = lambda table, reduce, trans: torch.sigmoid(self._lin(trans, ops.linear(table, reduce.weight, reduce.bias)))
        image_item_embeds = ids * space(self.image_embedding.weight, self.image_reduce_dim, self.image_trans_dim)
        text_item_embeds = ids * space(self.text_embedding.weight, self.text_reduce_dim, self.text_trans_dim)

        extended_id_embeds = self.conv_ui(adj, self.user_embedding.weight, ids)
        explicit_image_item = self.conv_ii(self.image_original_adj, image_item_embeds)
        explicit_image_embeds = torch.cat([sparse.mm(self.R, explicit_image_item), explicit_image_item], dim=0)
        extended_image_embeds = self.conv_ui(adj, self.extended_image_user.weight, explicit_image_item)
        explicit_text_item = self.conv_ii(self.text_original_adj, text_item_embeds)
        explicit_text_embeds = torch.cat([sparse.mm(self.R, explicit_text_item), explicit_text_item], dim=0)
        extended_text_embeds = self.conv_ui(adj, self.extended_text_user.weight, explicit_text_item)
        extended_it_embeds = (extended_image_embeds + extended_text_embeds) / 2

        coarse = lambda x: ops.linear(torch.tanh(self._lin(self.separate_coarse, x)), self.separate_coarse[2].weight)
        image_weights, text_weights = torch.split(
            self.softmax(torch.cat([coarse(explicit_image_embeds), coarse(explicit_text_embeds)], dim=-1)), 1, dim=-1)
        coarse_grained_embeds = image_weights * explicit_image_embeds + text_weights * explicit_text_embeds
        fine_grained_image = torch.sigmoid(self._lin(self.image_behavior, extended_id_embeds)) * (explicit_image_embeds - coarse_grained_embeds)
        fine_grained_text = torch.sigmoid(self._lin(self.text_behavior, extended_id_embeds)) * (explicit_text_embeds - coarse_grained_embeds)
        integration_embeds = (fine_grained_image + fine_grained_text + coarse_grained_embeds) / 3
        all_embeds = extended_id_embeds + integration_embeds
        self.result = all_embeds
        if train:
            return all_embeds, (integration_embeds, extended_id_embeds, extended_it_embeds), (explicit_image_embeds, explicit_text_embeds)
        return all_embeds

    # ---- :380-465 -------------------------------------------------------------------------------------------------------------
    def sq_sum(self, emb):
        return 1. / 2 * (emb ** 2).sum()

    def bpr_loss(self, users, pos_items, neg_items):
        pos_scores = torch.sum(torch.mul(users, pos_items), dim=1)
        neg_scores = torch.sum(torch.mul(users, neg_items), dim=1)
        regularizer = (self.sq_sum(users) + self.sq_sum(pos_items) + self.sq_sum(neg_items)) / self.batch_size
        return -torch.mean(F.logsigmoid(pos_scores - neg_scores)), self.reg_weight_1 * regularizer

    def InfoNCE(self, view1, view2, temperature):
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
        pos_score = torch.exp((view1 * view2).sum(dim=-1) / temperature)
        ttl_score = torch.exp(torch.matmul(view1, view2.transpose(0, 1)) / temperature).sum(dim=1)
        return torch.mean(-torch.log(pos_score / ttl_score))

    def cal_noise_loss(self, id, emb, temp):
        def add_perturbation(x):
            random_noise = self.noise_fn(x) if self.noise_fn is not None else torch.rand_like(x)
            return x + torch.sign(x) * F.normalize(random_noise, dim=-1) * 0.1
        return self.InfoNCE(add_perturbation(emb)[id], add_perturbation(emb)[id], temp)

    def align_vt(self, embed1, embed2):
        emb1_var, emb1_mean = torch.var(embed1), torch.mean(embed1)
        emb2_var, emb2_mean = torch.var(embed2), torch.mean(embed2)
        return (torch.abs(emb1_var - emb2_var) + torch.abs(emb1_mean - emb2_mean)).mean()

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.n_users, neg_items - self.n_users
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        U, I = self.n_users, self.n_items
        embeds_1, (integration_embeds, extended_id_embeds, extended_it_embeds), (explicit_image_embeds, explicit_text_embeds) = \
            self.forward(self.norm_adj, train=True)
        users_embeddings, items_embeddings = torch.split(embeds_1, [U, I], dim=0)
        vt_loss = self.vt_loss * self.align_vt(explicit_image_embeds, explicit_text_embeds)
        integration_users, integration_items = torch.split(integration_embeds, [U, I], dim=0)
        extended_id_user, extended_id_items = torch.split(extended_id_embeds, [U, I], dim=0)
        bpr_loss, reg_loss_1 = self.bpr_loss(users_embeddings[users], items_embeddings[pos_items], items_embeddings[neg_items])
        bm_loss = self.bm_loss * (self.InfoNCE(integration_users[users], extended_id_user[users], self.bm_temp)
                                  + self.InfoNCE(integration_items[pos_items], extended_id_items[pos_items], self.bm_temp))
        al_loss = vt_loss + bm_loss
        extended_it_user, extended_it_items = torch.split(extended_it_embeds, [U, I], dim=0)
        c_loss = self.InfoNCE(extended_it_user[users], integration_users[users], self.um_temp)
        noise_loss_1 = self.cal_noise_loss(users, integration_users, self.um_temp)
        noise_loss_2 = self.cal_noise_loss(users, extended_it_user, self.um_temp)
        um_loss = self.um_loss * (c_loss + noise_loss_1 + noise_loss_2)
        reg_loss = reg_loss_1 + self.reg_weight_2 * self.sq_sum(extended_it_items[pos_items]) / self.batch_size
        return bpr_loss + al_loss + um_loss + reg_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:467-492: the table of the last forward, history at 1e-6."""
        return ranking.gene_ranklist(self.result.detach(), self.n_users, self.n_items, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
