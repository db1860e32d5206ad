"""Plain-PyTorch CPU restatement of the reference's LightGCN train step and gene_ranklist.

TEST INFRASTRUCTURE ONLY (see oracle/chaorec_oracle.c).  This is what bench.py times as
`cpu_baseline` (kind "port"): the same eager op sequence the reference executes on its CPU path
-- degree + two gathers per conv call, index_select -> multiply -> scatter_add_, autograd
backward, torch.optim.Adam, dense matmul + per-user python mask loop + torch.topk -- with the
third-party propagate spelled out in torch ops (oracle/pyg_standin.py explains why).
Validated against the reference goldens in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn as nn


class TorchRefLightGCN(nn.Module):
    """Model/LightGCN.py:49-162 restated without torch_geometric."""

    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers):
        super().__init__()
        self.num_user, self.num_item = num_user, num_item
        self.user_item_dict = user_item_dict
        self.reg_weight, self.n_layers = reg_weight, n_layers
        e = torch.tensor(np.asarray(edge_index)).t().contiguous()
        self.edge_index = torch.cat((e, e[[1, 0]]), dim=1)                      # :63-64
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.result = None

    def conv(self, x):
        edge_index = self.edge_index.long()                                     # :29
        row, col = edge_index
        deg = torch.zeros(x.size(0), dtype=x.dtype).scatter_add_(0, row, torch.ones(row.numel(), dtype=x.dtype))
        deg_inv_sqrt = deg.pow(-0.5)                                            # :36-38
        norm = deg_inv_sqrt[row] * deg_inv_sqrt[col]
        msg = norm.view(-1, 1) * x.index_select(0, row)                         # :43
        return torch.zeros_like(x).scatter_add_(0, col.view(-1, 1).expand_as(msg), msg)

    def forward(self):
        x = torch.cat((self.user_embedding.weight, self.item_embedding.weight), dim=0)
        embs = [x]
        for _ in range(self.n_layers):
            x = self.conv(x)
            embs.append(x)
        w = 1.0 / len(embs)
        final = torch.zeros_like(embs[0])
        for e in embs:
            final = final + w * e
        self.result = final
        return final

    def loss(self, users, pos_items, neg_items):
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        emb = self.forward()
        u = emb[users]
        p = emb[self.num_user + pos_items]
        n = emb[self.num_user + neg_items]
        pos_scores = torch.sum(u * p, dim=1)
        neg_scores = torch.sum(u * n, dim=1)
        bpr = -torch.mean(torch.log(torch.sigmoid(pos_scores - neg_scores) + 1e-5))
        reg = self.reg_weight * (torch.mean(u ** 2) + torch.mean(p ** 2) + torch.mean(n ** 2))
        return bpr + reg

    def gene_ranklist(self, topk=50):
        user_tensor = self.result[:self.num_user].detach()
        item_tensor = self.result[self.num_user:self.num_user + self.num_item].detach()
        score_matrix = torch.matmul(user_tensor, item_tensor.t())
        for row, col in self.user_item_dict.items():                            # :150-152
            col = torch.LongTensor(list(col)) - self.num_user
            score_matrix[row][col] = 1e-6
        _, idx = torch.topk(score_matrix, topk)
        return idx + self.num_user


def reference_sampler_step(user_item_dict, all_items_tuple, user, rng):
    """dataload.py:74-79 restated: random.sample(set,1) copies the set to a sequence first (O(I))."""
    while True:
        neg = rng.sample(all_items_tuple, 1)[0]
        if neg not in user_item_dict[user]:
            return neg


def freedom_reference_loss(model, users, pos_items, neg_items):
    """Model/FREEDOM.py:164-217 restated in plain torch ops (torch.sparse.mm, F.linear, F.logsigmoid) on the
    SAME parameters / graphs as a chaorec_amd FREEDOM instance: an independent check of the fused path at
    full dataset sizes.  Returns (loss, result)."""
    import torch.nn.functional as F

    def to_sparse(csr):
        rp = csr.rowptr
        rows = torch.repeat_interleave(torch.arange(csr.n_rows, device=rp.device), rp[1:] - rp[:-1])
        return torch.sparse_coo_tensor(torch.stack([rows, csr.col.long()]), csr.val, (csr.n_rows, csr.n_cols))

    adj, mm = to_sparse(model.masked_adj), to_sparse(model.mm_adj)
    U = model.num_user
    h = model.item_embedding.weight
    for _ in range(model.mm_layers):
        h = torch.sparse.mm(mm, h)
    ego = torch.cat((model.user_embedding.weight, model.item_embedding.weight), 0)
    alls = [ego]
    for _ in range(model.n_layers):
        ego = torch.sparse.mm(adj, ego)
        alls.append(ego)
    allm = torch.stack(alls, 1).mean(1)
    ug, ig = allm[:U], allm[U:] + h
    pos, neg = pos_items - U, neg_items - U

    def bpr(u, p, n):
        return -torch.mean(F.logsigmoid((u * p).sum(1) - (u * n).sum(1)))

    mf = bpr(ug[users], ig[pos], ig[neg])
    tf = F.linear(model.text_embedding.weight, model.text_trs.weight, model.text_trs.bias)
    vf = F.linear(model.image_embedding.weight, model.image_trs.weight, model.image_trs.bias)
    loss = mf + model.reg_weight * (bpr(ug[users], tf[pos], tf[neg]) + bpr(ug[users], vf[pos], vf[neg]))
    return loss, torch.cat((ug, ig), 0)


def mmgcn_reference_forward(model):
    """Model/MMGCN.py:96-143,176-186 restated in plain torch ops on a chaorec_amd MMGCN instance."""
    import torch.nn.functional as F
    csr = model.graph
    rp = csr.rowptr
    rows = torch.repeat_interleave(torch.arange(csr.n_rows, device=rp.device), rp[1:] - rp[:-1])
    A = torch.sparse_coo_tensor(torch.stack([rows, csr.col.long()]), csr.val, (csr.n_rows, csr.n_cols))

    def branch(g, feat):
        temp = F.linear(feat, g.MLP.weight, g.MLP.bias) if g.dim_latent else feat
        x = F.normalize(torch.cat((g.preference, temp), 0))
        for k in (1, 2, 3, 4):
            conv, lin, gl = getattr(g, f"conv_embed_{k}"), getattr(g, f"linear_layer{k}"), getattr(g, f"g_layer{k}")
            hh = F.leaky_relu(torch.sparse.mm(A, F.linear(x, conv.lin.weight, conv.lin.bias)))
            u_hat = F.leaky_relu(F.linear(x, lin.weight, lin.bias)) + model.id_embedding
            x = F.leaky_relu(F.linear(torch.cat((hh, u_hat), 1), gl.weight, gl.bias))
        return x

    return (branch(model.v_gcn, model.v_feat) + branch(model.t_gcn, model.t_feat)) / 2
