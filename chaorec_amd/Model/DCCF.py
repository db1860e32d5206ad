"""DCCF with the reference's surface (Model/DCCF.py:17-290) -- per layer four views of every node: the LightGCN propagate,
a soft assignment to K learned intents, and two ADAPTIVELY re-weighted propagates whose edge weights are the cosine of the
edge's two endpoint rows (in the graph view and in the intent view), so the sparse operand's VALUES carry gradient.

Through the hot-path adapters: the propagate is `chaorec_amd.sparse.mm` over the binary D^-1/2 A D^-1/2; the two adaptive
products are the dynamic-values HIP SpMM over ONE fixed structure (`sparse.DroppedAdj`) with the weight array as a
differentiable input -- d weight[e] = <gy[h_e], x[t_e]>, what torch.sparse.mm's backward gives a sparse operand -- instead of a
new sparse tensor per layer and view (:106-118); the intent read-outs are two MFMA GEMMs each (`ops.linear`); the ranking is
`ranking.gene_ranklist` over the layer-summed tables of the last forward (:266-290).

Kept quirk: the adaptive adjacency lists every interaction ONCE, head = user, tail = item (:37-38,112-113): only user rows
aggregate through it, item rows of the two augmented views are zero.  A repeated interaction is two entries of the
reference's uncoalesced tensor: its weight counts twice here."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


class DCCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, ssl_temp,
                 ssl_alpha, n_intents, cen_reg, device):
        super(DCCF, self).__init__()
        self.ua_embedding = self.ia_embedding = None
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.reg_weight, self.cen_reg, self.n_layers, self.device = dim_E, reg_weight, cen_reg, n_layers, device
        self.ssl_temp, self.ssl_alpha, self.n_intents = ssl_temp, ssl_alpha, n_intents
        U, I = num_user, num_item
        e = torch.as_tensor(np.asarray(edge_index)).long()
        self.norm_adj_mat = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - U, U, I).to(device)
        self._pairs = sparse.PairStructure(edge_index, U, I, device)      # distinct interactions + one symmetric [N, N] structure
        self._eu, self._ei, self._ew, self.n_edges = self._pairs.eu, self._pairs.ei, self._pairs.ew, self._pairs.n
        self._lower, self._structure = self._pairs.lower, self._pairs.structure
        self._zeros = torch.zeros(self.n_edges, dtype=torch.float32, device=device)

        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_normal_(self.user_embedding.weight)
        nn.init.xavier_normal_(self.item_embedding.weight)
        _user_intent = torch.empty(dim_E, n_intents)
        nn.init.xavier_normal_(_user_intent)
        self.user_intent = torch.nn.Parameter(_user_intent, requires_grad=True)
        _item_intent = torch.empty(dim_E, n_intents)
        nn.init.xavier_normal_(_item_intent)
        self.item_intent = torch.nn.Parameter(_item_intent, requires_grad=True)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)

    def _adaptive_mask(self, table):
        """:106-118 -> the adjacency whose (user, item) entries weigh (cos(table[u], table[U + i]) + 1) / 2."""
        unit = F.normalize(table)                                   # (row-normalising the table = normalising every gathered row)
        val = self._ew * ((ops.edge_dot(self._structure, unit, unit, self.n_edges) + 1) / 2)
        return sparse.DroppedAdj(self._structure, torch.cat([val, self._zeros]), torch.cat([self._zeros, val[self._lower]]))

    def _intent(self, x, intent):
        return ops.linear(torch.softmax(ops.linear(x, intent.t().contiguous()), dim=1), intent)

    def forward(self):
        """:120-165."""
        all_embeddings = [torch.concat([self.user_embedding.weight, self.item_embedding.weight], dim=0)]
        gnn_embeddings, int_embeddings, gaa_embeddings, iaa_embeddings = [], [], [], []
        for i in range(self.n_layers):
            x = all_embeddings[i]
            gnn = sparse.mm(self.norm_adj_mat, x)
            u_x, i_x = torch.split(x, [self.num_user, self.num_item], 0)
            intent = torch.concat([self._intent(u_x, self.user_intent), self._intent(i_x, self.item_intent)], dim=0)
            gaa = sparse.mm(self._adaptive_mask(gnn), x)
            iaa = sparse.mm(self._adaptive_mask(intent), x)
            gnn_embeddings.append(gnn)
            int_embeddings.append(intent)
            gaa_embeddings.append(gaa)
            iaa_embeddings.append(iaa)
            all_embeddings.append(gnn + intent + gaa + iaa + x)
        total = torch.sum(torch.stack(all_embeddings, dim=1), dim=1, keepdim=False)
        self.ua_embedding, self.ia_embedding = torch.split(total, [self.num_user, self.num_item], 0)
        return gnn_embeddings, int_embeddings, gaa_embeddings, iaa_embeddings

    def cal_ssl_loss(self, users, items, gnn_emb, int_emb, gaa_emb, iaa_emb):
        """:167-213."""
        def cal_loss(emb1, emb2):
            pos_score = torch.exp(torch.sum(emb1 * emb2, dim=1) / self.ssl_temp)
            neg_score = torch.sum(torch.exp(torch.mm(emb1, emb2.T) / self.ssl_temp), dim=1)
            return torch.sum(-torch.log(pos_score / (neg_score + 1e-8) + 1e-8)) / pos_score.shape[0]

        cl_loss = 0.0
        for i in range(len(gnn_emb)):
            views = []
            for emb in (gnn_emb[i], int_emb[i], gaa_emb[i], iaa_emb[i]):
                u, it = torch.split(emb, [self.num_user, self.num_item], 0)
                views.append((F.normalize(u[users], dim=1), F.normalize(it[items], dim=1)))
            for side in (0, 1):
                for other in (1, 2, 3):
                    cl_loss = cl_loss + cal_loss(views[0][side], views[other][side])
        return cl_loss

    def bpr_loss(self, users, pos_items, neg_items):
        u, p, n = self.ua_embedding[users], self.ia_embedding[pos_items], self.ia_embedding[neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items):
        return self.reg_weight * (torch.mean(self.user_embedding.weight[users] ** 2) + torch.mean(self.item_embedding.weight[pos_items] ** 2)
                                  + torch.mean(self.item_embedding.weight[neg_items] ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        views = self.forward()
        cen_loss = self.cen_reg * (self.user_intent.norm(2).pow(2) + self.item_intent.norm(2).pow(2))
        ssl_loss = self.ssl_alpha * self.cal_ssl_loss(users, pos_items, *views)
        return self.bpr_loss(users, pos_items, neg_items) + self.regularization_loss(users, pos_items, neg_items) + ssl_loss + cen_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:266-290: the layer-summed tables of the last forward, history at 1e-6."""
        res = torch.cat((self.ua_embedding.detach(), self.ia_embedding.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
