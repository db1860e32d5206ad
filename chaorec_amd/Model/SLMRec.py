"""SLMRec with the reference's surface (Model/SLMRec.py:15-206) -- a MULTI-MODAL member of the `torch.sparse.mm` family
through the adapter alone: three LightGCN-style propagations per step (id, visual, textual item tables over
the same normalised graph) are `chaorec_amd.sparse.mm`, the ranking is the shared `ranking.gene_ranklist`; the Linears and
the InfoNCE losses are the reference's own torch expressions.

Same constructor, parameters (created AND initialised in the reference's order, :36-77: same seed, same weights),
`compute_graph`, `forward`, `fac`, `loss`, `gene_ranklist` (the table of the last training forward, mask 1e-6).
The adjacency is the reference's (:79-96): both directions of every interaction, weight deg^-1/2[row] * deg^-1/2[col] with
deg counted over rows AND columns of the doubled list (twice the node degree), coalesced."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ranking, sparse


class SLMRec(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, n_layers, ssl_temp, ssl_alpha,
                 device):
        super(SLMRec, self).__init__()
        self.num_user, self.num_item, self.dim_E, self.n_layers = num_user, num_item, dim_E, n_layers
        self.ssl_temp, self.temp, self.ssl_alpha = ssl_temp, 0.2, ssl_alpha
        self.num_nodes = num_user + num_item
        self.ssl_task = "FAC"
        self.infonce_criterion = nn.CrossEntropyLoss()
        self.device, self.user_item_dict = device, user_item_dict
        self.v_feat, self.t_feat = v_feat, t_feat
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_normal_(self.user_embedding.weight)
        nn.init.xavier_normal_(self.item_embedding.weight)
        n_modal = 0
        if self.v_feat is not None:
            self.v_feat = F.normalize(self.v_feat, dim=1)
            self.v_dense = nn.Linear(self.v_feat.shape[1], dim_E)
            nn.init.xavier_uniform_(self.v_dense.weight)
            n_modal += 1
        if self.t_feat is not None:
            self.t_feat = F.normalize(self.t_feat, dim=1)
            self.t_dense = nn.Linear(self.t_feat.shape[1], dim_E)
            nn.init.xavier_uniform_(self.t_dense.weight)
            n_modal += 1
        self.item_feat_dim = dim_E * (n_modal + 1)
        self.embedding_item_after_GCN = nn.Linear(self.item_feat_dim, dim_E)
        self.embedding_user_after_GCN = nn.Linear(self.item_feat_dim, dim_E)
        nn.init.xavier_uniform_(self.embedding_item_after_GCN.weight)
        nn.init.xavier_uniform_(self.embedding_user_after_GCN.weight)
        # the propagate graph: both directions, weights from the degrees of the doubled list
        e = torch.as_tensor(edge_index).long().t().contiguous()
        both = torch.cat((e, e[[1, 0]]), dim=1)
        inv = torch.bincount(both.reshape(-1)).pow(-0.5)
        adj = torch.sparse_coo_tensor(both, inv[both[0]] * inv[both[1]], (self.num_nodes, self.num_nodes))
        self.norm_adj = sparse.from_torch_sparse(adj).to(device)
        if self.ssl_task == "FAC":
            names = ("g_i_iv", "g_v_iv", "g_iv_iva", "g_a_iva", "g_iva_ivat", "g_t_ivat")
            for name in names:
                setattr(self, name, nn.Linear(dim_E, dim_E // 2 if name in ("g_iva_ivat", "g_t_ivat") else dim_E))
            for name in names:
                nn.init.xavier_uniform_(getattr(self, name).weight)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.result = None

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        dev = next(out.parameters()).device
        for name in ("v_feat", "t_feat"):           # (plain tensors in the reference: moved with the model here)
            t = getattr(out, name)
            if t is not None:
                setattr(out, name, t.to(dev))
        return out

    def compute_graph(self, u_emb, i_emb):
        """:98-107: the mean over [x, A x, .., A^L x]."""
        x = torch.cat([u_emb, i_emb])
        layers = [x]
        for _ in range(self.n_layers):
            x = sparse.mm(self.norm_adj, x)
            layers.append(x)
        return torch.mean(torch.stack(layers, dim=1), dim=1)

    def forward(self):
        users_emb, items_emb = self.user_embedding.weight, self.item_embedding.weight
        split = [self.num_user, self.num_item]
        self.i_emb = self.compute_graph(users_emb, items_emb)
        self.i_emb_u, self.i_emb_i = torch.split(self.i_emb, split)
        parts_u, parts_i = [self.i_emb_u], [self.i_emb_i]
        if self.v_feat is not None:
            self.v_dense_emb = self.v_dense(self.v_feat)
            self.v_emb = self.compute_graph(users_emb, self.v_dense_emb)
            self.v_emb_u, self.v_emb_i = torch.split(self.v_emb, split)
            parts_u.append(self.v_emb_u)
            parts_i.append(self.v_emb_i)
        if self.t_feat is not None:
            self.t_dense_emb = self.t_dense(self.t_feat)
            self.t_emb = self.compute_graph(users_emb, self.t_dense_emb)
            self.t_emb_u, self.t_emb_i = torch.split(self.t_emb, split)
            parts_u.append(self.t_emb_u)
            parts_i.append(self.t_emb_i)
        user = self.embedding_user_after_GCN(torch.cat(parts_u, dim=1))
        item = self.embedding_item_after_GCN(torch.cat(parts_i, dim=1))
        self.result = torch.cat((user, item), dim=0)
        return user, item

    def _infonce(self, a, b):
        logits = torch.mm(a, b.T) / self.ssl_temp
        return self.infonce_criterion(logits, torch.arange(a.shape[0], device=a.device))

    def fac(self, idx):
        """:137-156: the id view against the visual one, their fusion against the textual one."""
        x_i_iv = self.g_i_iv(self.i_emb_i[idx])
        v_loss = self._infonce(x_i_iv, self.g_v_iv(self.v_emb_i[idx]))
        x_iva_ivat = self.g_iva_ivat(self.g_iv_iva(x_i_iv))
        return v_loss + self._infonce(x_iva_ivat, self.g_t_ivat(self.t_emb_i[idx]))

    def loss(self, users, pos_items, neg_items):
        pos_items = pos_items - self.num_user
        users, pos_items = users.to(self.device), pos_items.to(self.device)
        user_tensor, item_tensor = self.forward()
        main_loss = self._infonce(F.normalize(user_tensor[users], dim=1), F.normalize(item_tensor[pos_items], dim=1))
        return main_loss + self.ssl_alpha * self.fac(pos_items)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:181-206 (mask value 1e-6, the table of the last training forward)."""
        return ranking.gene_ranklist(self.result.detach(), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
