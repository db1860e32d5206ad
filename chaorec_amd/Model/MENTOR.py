"""MENTOR with the reference's surface (Model/MENTOR.py:17-459) -- seven two-hop GCN encoders over the user-item graph (visual,
textual, id, and two randomly perturbed copies of the visual and the textual one), their [N, 2 D] read-outs lifted over the
item-item kNN graph, a BPR term and four auxiliary ones (Gaussian alignment of the views, a feature-masking cosine that is
constant under `no_grad`, an InfoNCE between the two perturbed read-outs, L2 on the preferences).

Through the hot-path adapters: `Base_gcn` (:78-103: remove self loops, deg^-1/2[row] deg^-1/2[col], scatter-add) is ONE CSR
built at construction (`graph.lightgcn_csr`) and the 14 propagates of a forward are `chaorec_amd.sparse.mm`; the six item-graph
products are `sparse.mm` over the mixed kNN CSR; the encoders' MLPs -- [I, 4096] features included -- are `ops.linear` on the
MFMA GEMM with the leaky ReLU in the first product's epilogue; the ranking is `ranking.gene_ranklist` over the fused
[N, 2 D] table of the last forward (:435-459).

Same constructor, parameters in the reference's creation order and drawn the same way (numpy's global generator for the
preferences / id features / modality weights, torch's for the Linears).  The perturbations (`torch.rand_like`, :54-60) and the
feature mask (`F.dropout`, :381-382) draw on the device; `noise_fn` / `dropout_fn` replay stored draws in the golden test."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse
from .DDRec import knn_binary_graph


class GCN(torch.nn.Module):
    """:17-64.  preference | MLP_1(leaky_relu(MLP(features))) -> row-normalised -> x + A x + A A x."""

    def __init__(self, num_user, num_item, dim_E, aggr_mode, device=None, features=None):
        super(GCN, self).__init__()
        self.num_user, self.num_item, self.dim_feat, self.dim_E = num_user, num_item, features.size(1), dim_E
        self.aggr_mode, self.device = aggr_mode, device
        self.preference = nn.Parameter(nn.init.xavier_normal_(
            torch.tensor(np.random.randn(num_user, dim_E if dim_E else self.dim_feat), dtype=torch.float32, requires_grad=True),
            gain=1).to(device))
        if dim_E:
            self.MLP = nn.Linear(self.dim_feat, 4 * dim_E)
            self.MLP_1 = nn.Linear(4 * dim_E, dim_E)

    def forward(self, adj, features, perturbed=False, noise_fn=None):
        if self.dim_E:
            features = ops.linear(ops.linear(features, self.MLP.weight, self.MLP.bias, act=1), self.MLP_1.weight, self.MLP_1.bias)
        x = F.normalize(torch.cat((self.preference, features), dim=0))
        draw = noise_fn if noise_fn is not None else torch.rand_like
        h = sparse.mm(adj, x)
        if perturbed:
            h = h + torch.sign(h) * F.normalize(draw(h), dim=-1) * 0.1
        h_1 = sparse.mm(adj, h)
        if perturbed:
            h_1 = h_1 + torch.sign(h_1) * F.normalize(draw(h), dim=-1) * 0.1
        return x + h + h_1, self.preference


class MENTOR(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, mm_layers,
                 reg_weight, ssl_temp, dropout, align_weight, mask_weight_g, mask_weight_f, device):
        super(MENTOR, self).__init__()
        self.num_user, self.num_item, self.dim_E, self.reg_weight = num_user, num_item, dim_E, reg_weight
        self.dropout, self.temp, self.user_item_dict, self.device = dropout, ssl_temp, user_item_dict, device
        self.knn_k, self.mm_layers, self.mm_image_weight, self.aggr_mode = 10, mm_layers, 0.5, 'add'
        self.align_weight, self.mask_weight_g, self.mask_weight_f = align_weight, mask_weight_g, mask_weight_f
        self.mlp = nn.Linear(2 * dim_E, 2 * dim_E)
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=False)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=False)
        self.register_buffer("v_feat", v_feat.clone(), persistent=False)
        self.register_buffer("t_feat", t_feat.clone(), persistent=False)
        idx_v, val_v = knn_binary_graph(v_feat.to(device), self.knn_k)
        idx_t, val_t = knn_binary_graph(t_feat.to(device), self.knn_k)
        self.mm_adj = graph.add_scaled_coo((idx_v, val_v), self.mm_image_weight, (idx_t, val_t), 1.0 - self.mm_image_weight,
                                           num_item).to(device)
        self.graph = graph.lightgcn_csr(edge_index, num_user + num_item).to(device)
        self.weight_u = nn.Parameter(nn.init.xavier_normal_(
            torch.tensor(np.random.randn(num_user, 2, 1), dtype=torch.float32, requires_grad=True)))
        self.weight_u.data = F.softmax(self.weight_u, dim=1)
        mk = lambda feats: GCN(num_user, num_item, dim_E, self.aggr_mode, device=device, features=feats)
        self.v_gcn, self.v_gcn_n1, self.v_gcn_n2 = mk(v_feat), mk(v_feat), mk(v_feat)
        self.t_gcn, self.t_gcn_n1, self.t_gcn_n2 = mk(t_feat), mk(t_feat), mk(t_feat)
        self.id_feat = nn.Parameter(nn.init.xavier_normal_(
            torch.tensor(np.random.randn(num_item, dim_E), dtype=torch.float32, requires_grad=True), gain=1).to(device))
        self.id_gcn = mk(self.id_feat)
        self.result_embed = self.result_embed_guide = self.result_embed_v = self.result_embed_t = None
        self.result_embed_n1 = self.result_embed_n2 = None
        self.noise_fn = self.dropout_fn = None
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def InfoNCE(self, view1, view2, temp):
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
        pos_score = torch.exp((view1 * view2).sum(dim=-1) / temp)
        ttl_score = torch.exp(torch.matmul(view1, view2.transpose(0, 1)) / temp).sum(dim=1)
        return torch.mean(-torch.log(pos_score / ttl_score))

    def buildItemGraph(self, h):
        for _ in range(self.mm_layers):
            h = sparse.mm(self.mm_adj, h)
        return h

    def fit_Gaussian_dis(self):
        out = []
        for t in (self.result_embed, self.result_embed_guide, self.result_embed_v, self.result_embed_t):
            out += [torch.var(t), torch.mean(t)]
        return tuple(out)

    def forward(self):
        """:161-260."""
        U, g, nf = self.num_user, self.graph, self.noise_fn
        v_rep, self.v_preference = self.v_gcn(g, self.v_feat)
        t_rep, self.t_preference = self.t_gcn(g, self.t_feat)
        id_rep, self.id_preference = self.id_gcn(g, self.id_feat)
        v_n1, _ = self.v_gcn_n1(g, self.v_feat, perturbed=True, noise_fn=nf)
        t_n1, _ = self.t_gcn_n1(g, self.t_feat, perturbed=True, noise_fn=nf)
        v_n2, _ = self.v_gcn_n2(g, self.v_feat, perturbed=True, noise_fn=nf)
        t_n2, _ = self.t_gcn_n2(g, self.t_feat, perturbed=True, noise_fn=nf)
        w0, w1 = self.weight_u[:, 0], self.weight_u[:, 1]                      # [U, 1] each
        weighted = lambda v, t: torch.cat((w0 * v[:U], w1 * t[:U]), dim=1)
        lifted = lambda items: items + self.buildItemGraph(items)

        self.user_rep, self.item_rep = weighted(v_rep, t_rep), lifted(torch.cat((v_rep[U:], t_rep[U:]), dim=1))
        self.result_embed = torch.cat((self.user_rep, self.item_rep), dim=0)
        self.guide_user_rep, self.guide_item_rep = torch.cat((id_rep[:U], id_rep[:U]), dim=1), lifted(torch.cat((id_rep[U:], id_rep[U:]), dim=1))
        self.result_embed_guide = torch.cat((self.guide_user_rep, self.guide_item_rep), dim=0)
        self.v_user_rep, self.v_item_rep = torch.cat((v_rep[:U], v_rep[:U]), dim=1), lifted(torch.cat((v_rep[U:], v_rep[U:]), dim=1))
        self.result_embed_v = torch.cat((self.v_user_rep, self.v_item_rep), dim=0)
        self.t_user_rep, self.t_item_rep = torch.cat((t_rep[:U], t_rep[:U]), dim=1), lifted(torch.cat((t_rep[U:], t_rep[U:]), dim=1))
        self.result_embed_t = torch.cat((self.t_user_rep, self.t_item_rep), dim=0)
        self.user_rep_n1, self.item_rep_n1 = weighted(v_n1, t_n1), lifted(torch.cat((v_n1[U:], t_n1[U:]), dim=1))
        self.result_embed_n1 = torch.cat((self.user_rep_n1, self.item_rep_n1), dim=0)
        self.user_rep_n2, self.item_rep_n2 = weighted(v_n2, t_n2), lifted(torch.cat((v_n2[U:], t_n2[U:]), dim=1))
        self.result_embed_n2 = torch.cat((self.user_rep_n2, self.item_rep_n2), dim=0)
        self.v_rep, self.t_rep, self.id_rep = v_rep.unsqueeze(2), t_rep.unsqueeze(2), id_rep.unsqueeze(2)

    # ---- :262-433 -------------------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items):
        u, p, n = self.result_embed[users], self.result_embed[self.num_user + pos_items], self.result_embed[self.num_user + neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users):
        reg_loss = self.reg_weight * ((self.v_preference[users] ** 2).mean() + (self.t_preference[users] ** 2).mean())
        return reg_loss + self.reg_weight * (self.weight_u ** 2).mean()

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        self.forward()
        bpr_loss, reg_loss = self.bpr_loss(users, pos_items, neg_items), self.regularization_loss(users)
        with torch.no_grad():                   # (:374-383: the whole feature-masking term is a constant of the step)
            drop = self.dropout_fn if self.dropout_fn is not None else (lambda x, p: F.dropout(x, p))
            u_temp2 = ops.linear(self.user_rep.detach(), self.mlp.weight, self.mlp.bias)
            i_temp2 = ops.linear(self.item_rep.detach(), self.mlp.weight, self.mlp.bias)
            u_temp, i_temp = drop(self.user_rep.detach(), self.dropout), drop(self.item_rep.detach(), self.dropout)
        mask_loss_u = 1 - F.cosine_similarity(u_temp, u_temp2).mean()
        mask_loss_i = 1 - F.cosine_similarity(i_temp, i_temp2).mean()
        mask_f_loss = self.mask_weight_f * (mask_loss_i + mask_loss_u)
        r_var, r_mean, g_var, g_mean, v_var, v_mean, t_var, t_mean = self.fit_Gaussian_dis()
        pair = lambda a_var, a_mean, b_var, b_mean: (torch.abs(a_var - b_var) + torch.abs(a_mean - b_mean)).mean()
        align_loss = (pair(g_var, g_mean, r_var, r_mean) + pair(g_var, g_mean, v_var, v_mean) + pair(g_var, g_mean, t_var, t_mean)
                      + pair(r_var, r_mean, v_var, v_mean) + pair(r_var, r_mean, t_var, t_mean) + pair(v_var, v_mean, t_var, t_mean))
        align_loss = align_loss * self.align_weight
        U = self.num_user
        mask_g_loss = (self.InfoNCE(self.result_embed_n1[:U], self.result_embed_n2[:U], self.temp)
                       + self.InfoNCE(self.result_embed_n1[U:], self.result_embed_n2[U:], self.temp)) * self.mask_weight_g
        return bpr_loss + reg_loss + align_loss + mask_f_loss + mask_g_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:435-459: the fused [N, 2 D] table of the last forward, history at 1e-6."""
        return ranking.gene_ranklist(self.result_embed.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
