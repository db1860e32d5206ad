"""LightGT with the reference's surface (Model/LightGT.py:17-410) -- a LightGCN propagate whose per-layer prefix means feed,
as attention biases, two small transformer encoders (visual / textual) over every sample's HISTORY SEQUENCE (the user token
plus up to 50 / 20 of the user's items, dataload.py:89-101,109-143), and a score that mixes the graph and the two modal views.

Hot-path pieces through the adapters: the propagate is `chaorec_amd.sparse.mm` on the HIP SpMM (every layer once; the L + 1
prefix means the reference forms by stacking are running sums here, :181-207), the two feature projections over all items run
on the MFMA GEMM (`ops.linear`), and the evaluation is ONE `ranking.gene_ranklist` over concatenated tables: w1 u.i + w2 (v_out.v
+ t_out.t) = [w1 u | w2 v_out | w2 t_out] . [i | v | t], history at 1e-5 (:369-410) -- the reference materialises a
[2000, I] score matrix per eval batch.  The encoders (single-head attention over <= 51 tokens, LayerNorm, Linears on [B, 51, D])
are dense torch work on the batch and stay torch.

The history sequences are drawn on the device (`dataload.history_sequences`: a uniformly random subset of src_len items when
the history is longer, all of it otherwise; only the user token's output is read and the attention has no positional term, so
the order the reference's `random.shuffle` gives the kept items does not matter).

Same constructor, parameters in the reference's creation order and under its names (the unused template layers
`v_encoder_layer` / `t_encoder_layer` included: the encoders hold deep copies of them, :139)."""
import copy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


class MultiheadAttention(nn.Module):
    """:17-89 (one head; the query is scaled by head_dim^-1/2 / 100)."""

    def __init__(self, embed_dim, nheads=1, dropout=0.):
        super().__init__()
        self.embed_dim, self.nheads, self.dropout = embed_dim, nheads, dropout
        self.head_dim = embed_dim
        self.q_in_proj = nn.Linear(embed_dim, embed_dim)
        self.k_in_proj = nn.Linear(embed_dim, embed_dim)
        self.v_in_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)

    def forward(self, query, key, value, key_padding_mask=None, attn_mask=None):
        tgt_len, batch_size, embed_dim = query.size()
        nheads = self.nheads
        head_dim = embed_dim // nheads
        q = (self.q_in_proj(query) * (float(head_dim) ** -0.5)) / 100
        k, v = self.k_in_proj(key), self.v_in_proj(value)
        q = q.contiguous().view(tgt_len, batch_size * nheads, head_dim).transpose(0, 1)
        k = k.contiguous().view(-1, batch_size * nheads, head_dim).transpose(0, 1)
        v = v.contiguous().view(-1, batch_size * nheads, head_dim).transpose(0, 1)
        src_len = k.size(1)
        w = torch.bmm(q, k.transpose(1, 2)).view(batch_size, nheads, tgt_len, src_len)
        if key_padding_mask is not None:
            w = w.masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2), float('-inf'))
        w = torch.softmax(w.view(batch_size * nheads, tgt_len, src_len), dim=-1)
        w = torch.dropout(w, p=self.dropout, train=self.training)
        out = torch.bmm(w, v).transpose(0, 1).contiguous().view(tgt_len, batch_size, embed_dim)
        return self.out_proj(out)


class TransformerEncoderLayer(nn.Module):
    """:91-131: attention + LayerNorm (no residual, no feed-forward)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.nhead = nhead
        self.self_attn = nn.ModuleList([MultiheadAttention(d_model, dropout=dropout) for _ in range(nhead)])
        self.norm1 = nn.LayerNorm(d_model)

    def forward(self, query, key, value, src_mask=None, src_key_padding_mask=None):
        if self.nhead != 1:
            src2 = torch.sum(torch.stack([m(query, key, value, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
                                          for m in self.self_attn], dim=-1), dim=-1)
        else:
            src2 = self.self_attn[0](query, key, value, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
        return self.norm1(src2)


class TransformerEncoder(nn.Module):
    """:133-154: layer i attends with (output + src[i]) as query and key, output as value."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers, self.norm = num_layers, norm

    def forward(self, input, src, mask=None, src_key_padding_mask=None):
        output = input
        for i in range(self.num_layers):
            output = self.layers[i](output + src[i], output + src[i], output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
        return self.norm(output) if self.norm is not None else output


class LightGCN(nn.Module):
    """:156-207: the tables, the mean of the first n_layers + 1 of them, and for every i < transformer_layers the mean of
    E_0 .. E_{i+1}."""

    def __init__(self, user_num, item_num, graph_csr, transformer_layers, latent_dim=64, n_layers=3):
        super().__init__()
        self.user_num, self.item_num, self.graph = user_num, item_num, graph_csr
        self.transformer_layers, self.latent_dim, self.n_layers = transformer_layers, latent_dim, n_layers
        self.user_emb = nn.Embedding(user_num, latent_dim)
        nn.init.xavier_normal_(self.user_emb.weight)
        self.item_emb = nn.Embedding(item_num, latent_dim)
        nn.init.xavier_normal_(self.item_emb.weight)

    def forward(self):
        x = torch.cat([self.user_emb.weight, self.item_emb.weight])
        run, n_in = x, 1                       # running sums of the layers: the stacked means of the reference
        main = x if self.n_layers == 0 else None
        means = []
        for layer in range(self.transformer_layers):
            x = sparse.mm(self.graph, x)
            run, n_in = run + x, n_in + 1
            means.append(run / n_in)
            if layer + 1 == self.n_layers:
                main = means[-1]
        if main is None:                       # (n_layers > transformer_layers: every propagated layer is in it)
            main = means[-1] if means else x
        split = lambda t: torch.split(t, [self.user_num, self.item_num])
        users, items = split(main)
        um, im = zip(*(split(m) for m in means)) if means else ((), ())
        return users, items, list(um), list(im)


class LightGT(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 reg_weight, n_layers, device):
        super(LightGT, self).__init__()
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.reg_weight, self.device = dim_E, reg_weight, device
        self.lightgcn_layers = self.transformer_layers = n_layers
        self.score_weight1 = 0.05
        self.score_weight2 = 1 - self.score_weight1
        self.src_len, self.nhead = 20, 1
        self.register_buffer("weight", torch.tensor([[1.], [-1.]]), persistent=False)
        for name, feat in (("v_feat", v_feat), ("t_feat", t_feat)):
            if feat is not None:
                self.register_buffer(name, F.normalize(feat), persistent=False)
            else:
                setattr(self, name, None)
        e = torch.as_tensor(np.asarray(edge_index)).long()
        self.norm_adj_mat = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)   # :263-303
        self.lightgcn = LightGCN(num_user, num_item, self.norm_adj_mat, self.transformer_layers, dim_E, self.lightgcn_layers)
        self.user_exp = nn.Parameter(torch.rand(num_user, dim_E))
        nn.init.xavier_normal_(self.user_exp)
        for tag, feat in (("v", self.v_feat), ("t", self.t_feat)):
            if feat is None:
                continue
            setattr(self, tag + "_mlp", nn.Linear(dim_E, dim_E))
            setattr(self, tag + "_linear", nn.Linear(feat.size(1), dim_E))
            layer = TransformerEncoderLayer(d_model=dim_E, nhead=self.nhead)
            setattr(self, tag + "_encoder_layer", layer)
            setattr(self, tag + "_encoder", TransformerEncoder(layer, num_layers=self.transformer_layers))
            setattr(self, tag + "_dense", nn.Linear(dim_E, dim_E))
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.eval_chunk = 2000                  # users per evaluation batch (main.py:199)

    def _modal(self, tag, feat, users, user_item, mask, users_mean, items_mean):
        """:306-334 for one modality -> (projected item table, the user tokens' encoder output)."""
        mlp, lin, enc, dense = (getattr(self, tag + s) for s in ("_mlp", "_linear", "_encoder", "_dense"))
        src = []
        for i in range(self.transformer_layers):
            temp = items_mean[i][user_item].detach()
            temp[:, 0] = users_mean[i][users].detach()
            src.append(torch.sigmoid(mlp(temp).transpose(0, 1)))
        table = ops.linear(feat, lin.weight, lin.bias)
        x_in = table[user_item]
        x_in[:, 0] = self.user_exp[users]
        out = enc(x_in.transpose(0, 1), src, src_key_padding_mask=mask).transpose(0, 1)[:, 0]
        return table, F.leaky_relu(dense(out))

    def forward(self, users, user_item, mask):
        user_emb, item_emb, users_mean, items_mean = self.lightgcn()
        v = t = v_out = t_out = None
        if self.v_feat is not None:
            v, v_out = self._modal("v", self.v_feat, users, user_item, mask, users_mean, items_mean)
        if self.t_feat is not None:
            t, t_out = self._modal("t", self.t_feat, users, user_item, mask, users_mean, items_mean)
        return user_emb, item_emb, v, t, v_out, t_out

    def loss(self, users, items, mask, user_item):
        """:336-367.  users [B, 2] (the user twice), items [B, 2] (pos, neg; GLOBAL ids), mask [B, 51] (True = padding),
        user_item [B, 51] (local item ids, -1 in the user token's place)."""
        dev = self.device
        users, items, mask, user_item = users.to(dev), items.to(dev), mask.to(dev), user_item.to(dev)
        user_emb, item_emb, v, t, v_out, t_out = self.forward(users[:, 0], user_item, mask)
        users = users.view(-1)
        items = items - self.num_user
        pos_items, neg_items = items[:, 0].view(-1), items[:, 1].view(-1)
        items = items.view(-1)
        score1 = torch.sum(user_emb[users] * item_emb[items], dim=1).view(-1, 2)
        score2_1 = torch.sum(v_out * v[pos_items], dim=1).view(-1, 1) + torch.sum(t_out * t[pos_items], dim=1).view(-1, 1)
        score2_2 = torch.sum(v_out * v[neg_items], dim=1).view(-1, 1) + torch.sum(t_out * t[neg_items], dim=1).view(-1, 1)
        score = self.score_weight1 * score1 + self.score_weight2 * torch.cat((score2_1, score2_2), dim=1)
        loss = -torch.mean(torch.log(torch.sigmoid(torch.matmul(score, self.weight))))
        reg_loss = self.reg_weight * ((user_emb ** 2).mean() + (item_emb ** 2).mean())
        return loss + reg_loss

    def user_tables(self, eval_batches):
        """The users' side of get_score_matrix (:369-376) for every user: [w1 u | w2 v_out | w2 t_out] and the items'
        [i | v | t].  eval_batches yields (users [n], user_item [n, src_len + 1], mask) covering the users in order."""
        rows, items = [], None
        for users, user_item, mask in eval_batches:
            users = users.view(-1).to(self.device)
            user_emb, item_emb, v, t, v_out, t_out = self.forward(users, user_item.to(self.device), mask.to(self.device))
            rows.append(torch.cat((self.score_weight1 * user_emb[users], self.score_weight2 * v_out, self.score_weight2 * t_out), 1))
            if items is None:
                items = torch.cat((item_emb, v, t), 1)
        return torch.cat(rows, 0), items

    def gene_ranklist(self, eval_dataloader=None, step=2000, topk=50, to_cpu=True):
        """:378-410.  eval_dataloader: the reference's DataLoader(EvalDataset, 2000) or None -- then the sequences are drawn
        on the device (dataload.history_sequences, src_len 20)."""
        from .. import dataload
        if eval_dataloader is None:
            eval_dataloader = dataload.device_eval_batches(self.hist, self.num_user, self.src_len, self.eval_chunk, self.device)
        with torch.no_grad():
            users, items = self.user_tables(eval_dataloader)
        return ranking.gene_ranklist(users, self.num_user, self.num_item, self.hist, 1e-5, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self), items=items)

    full_sort_predict = gene_ranklist
