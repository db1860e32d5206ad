"""MCLN with the reference's surface (Model/MCLN.py:17-352) -- the one model that reads the sampler's SECOND negative
(dataload.py:81-84: `int_items`, the "intervention" item of its counterfactual attention), through the hot-path adapters:
the LightGCN-style propagation is `chaorec_amd.sparse.mm` on the HIP SpMM, the two modality projections
(image_trs / text_trs over all items, :237-238) are `ops.linear` on the MFMA GEMM, the ranking -- the sum of three dot
products, :330-331 -- is ONE `ranking.gene_ranklist` over the concatenated [id | visual | textual] tables.  The
counterfactual attention over the batch (B x B softmax, LayerNorms, feed-forward: :148-228) is dense torch work on
[B, 3 D] tensors and stays torch.

Same constructor, parameters (created AND initialised in the reference's order, :40-84: same seed, same weights), `forward`
(the [B, B] score matrix, :285), `loss(users, pos_items, neg_items, int_items)` (:287-322), `gene_ranklist` (mask 1e-6, the
tables of the last training forward)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


class MCLN(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, reg_weight, n_layers, n_mca,
                 device):
        super(MCLN, self).__init__()
        self.num_user, self.num_item, self.dim_E = num_user, num_item, dim_E
        self.user_item_dict, self.reg_weight, self.n_layers, self.n_mca, self.device = user_item_dict, reg_weight, n_layers, n_mca, device
        e = torch.as_tensor(edge_index).long()
        # :90-128: binary A, degree + 1e-7, D^-1/2 A D^-1/2 (scipy in the reference)
        self.norm_adj_mat = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)
        W = 3 * dim_E
        self.image_embedding = nn.Embedding.from_pretrained(v_feat, freeze=True)
        self.text_embedding = nn.Embedding.from_pretrained(t_feat, freeze=True)
        self.image_trs = nn.Linear(v_feat.shape[1], dim_E)
        nn.init.xavier_normal_(self.image_trs.weight)
        self.text_trs = nn.Linear(t_feat.shape[1], dim_E)
        nn.init.xavier_normal_(self.text_trs.weight)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        nn.init.xavier_normal_(self.user_embedding.weight)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_normal_(self.item_embedding.weight)
        self.user_embedding_v = nn.Embedding(num_user, dim_E)
        self.user_embedding_t = nn.Embedding(num_user, dim_E)
        nn.init.xavier_normal_(self.user_embedding_v.weight)
        nn.init.xavier_normal_(self.user_embedding_t.weight)
        self.fc_pos = nn.Linear(W, dim_E)
        self.fc_neg = nn.Linear(W, dim_E)
        self.relu = nn.ReLU()
        for name in ("V1", "K1", "Q1", "K_int", "Q_int", "cfl1"):          # counterfactual layer 1
            setattr(self, name, nn.Linear(W, W, bias=False))
        self.ln1 = nn.LayerNorm(W)
        for name in ("V2", "K2", "Q2", "cfl2"):                            # counterfactual layer 2
            setattr(self, name, nn.Linear(W, W, bias=False))
        self.ln2 = nn.LayerNorm(W)
        self.inner_layer = nn.Linear(W, 4 * W)
        self.output_layer = nn.Linear(4 * W, W)
        self.layer_norm = nn.LayerNorm(W)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.ua_embeddings = self.ia_embeddings = self.visual = self.textual = None

    # ---- :130-145 ------------------------------------------------------------------------------------------------
    def _create_norm_embed(self):
        x = torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)
        layers = [x]
        for _ in range(self.n_layers):
            x = sparse.mm(self.norm_adj_mat, x)
            layers.append(x)
        mean = torch.mean(torch.stack(layers, dim=1), dim=1)
        return torch.split(mean, [self.num_user, self.num_item], dim=0)

    # ---- :148-228: attention over the batch, optionally minus the intervention's attention scores -----------------
    def _attend(self, x, q, k, v, out, norm, x_int=None):
        scale = math.sqrt(3 * self.dim_E)
        score = torch.matmul(q(x), k(x).transpose(-2, -1)) / scale
        if x_int is not None:
            score = score - torch.matmul(self.Q_int(x_int), self.K_int(x_int).transpose(-2, -1)) / scale
        return norm(out(torch.matmul(F.softmax(score, dim=-1), v(x))) + x)

    def feed_forward_layer(self, inputs, activation=F.relu):
        return self.layer_norm(self.output_layer(activation(self.inner_layer(inputs))) + inputs)

    def causal_difference_1(self, cd_inputs_embedding, cd_inputs_embedding_int):
        x = cd_inputs_embedding
        for _ in range(self.n_mca):     # (:152-165: the intervention side is never updated between the layers)
            x = self.feed_forward_layer(self._attend(x, self.Q1, self.K1, self.V1, self.cfl1, self.ln1, cd_inputs_embedding_int))
        return x

    def causal_difference_2(self, cd_inputs_embedding):
        x = cd_inputs_embedding
        for _ in range(self.n_mca):
            x = self.feed_forward_layer(self._attend(x, self.Q2, self.K2, self.V2, self.cfl2, self.ln2))
        return x

    # ---- :236-285 ------------------------------------------------------------------------------------------------
    def forward(self, users, pos_items, neg_items, int_items):
        self.visual = ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        self.textual = ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        self.ua_embeddings, self.ia_embeddings = self._create_norm_embed()
        self.u_g_embeddings = self.ua_embeddings[users]
        self.u_g_embeddings_pre = self.user_embedding(users)
        self.u_g_embeddings_v = self.user_embedding_v(users)
        self.u_g_embeddings_t = self.user_embedding_t(users)
        cat = {}
        for tag, ids in (("pos", pos_items), ("neg", neg_items), ("int", int_items)):
            g, v, t = self.ia_embeddings[ids], self.visual[ids], self.textual[ids]
            setattr(self, f"{tag}_i_g_embeddings", g)
            setattr(self, f"{tag}_i_g_embeddings_pre", self.item_embedding(ids))
            setattr(self, f"{tag}_i_g_embeddings_v", v)
            setattr(self, f"{tag}_i_g_embeddings_t", t)
            cat[tag] = torch.cat([g, v, t], dim=1)
        self.pos_outputs_embeddings = self.causal_difference_1(cat["pos"], cat["int"])
        self.neg_outputs_embeddings = self.causal_difference_2(cat["neg"])
        self.pos_i_g_embeddings_m = self.relu(self.fc_pos(self.pos_outputs_embeddings))
        self.neg_i_g_embeddings_m = self.relu(self.fc_neg(self.neg_outputs_embeddings))
        return (torch.matmul(self.u_g_embeddings, self.pos_i_g_embeddings.t())
                + torch.matmul(self.u_g_embeddings_v, self.pos_i_g_embeddings_v.t())
                + torch.matmul(self.u_g_embeddings_t, self.pos_i_g_embeddings_t.t())
                + torch.matmul(self.u_g_embeddings, self.pos_i_g_embeddings_m.t()))

    # ---- :287-322 ------------------------------------------------------------------------------------------------
    def loss(self, users, pos_items, neg_items, int_items):
        dev = self.device
        users = users.to(dev)
        pos_items, neg_items, int_items = ((t - self.num_user).to(dev) for t in (pos_items, neg_items, int_items))
        self.forward(users, pos_items, neg_items, int_items)
        u = self.u_g_embeddings
        mf_loss, squares = 0.0, torch.sum(self.u_g_embeddings_pre.pow(2))
        for suffix in ("", "_v", "_t", "_m"):         # the id, visual, textual and counterfactual views against the SAME user rows
            p, n = getattr(self, "pos_i_g_embeddings" + suffix), getattr(self, "neg_i_g_embeddings" + suffix)
            mf_loss = mf_loss + torch.mean(F.softplus(-(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1))))
            if suffix == "":
                p, n = self.pos_i_g_embeddings_pre, self.neg_i_g_embeddings_pre      # (the L2 term of the id view is on the raw rows)
            squares = squares + torch.sum(p.pow(2)) + torch.sum(n.pow(2))
        # (:316-320: normalised copies nobody reads afterwards)
        self.user_embed = F.normalize(self.u_g_embeddings_pre, p=2, dim=1)
        self.item_embed = F.normalize(self.pos_i_g_embeddings_pre, p=2, dim=1)
        self.item_embed_v = F.normalize(self.pos_i_g_embeddings_v, p=2, dim=1)
        self.item_embed_t = F.normalize(self.pos_i_g_embeddings_t, p=2, dim=1)
        return mf_loss + self.reg_weight * squares

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:324-352: user . item + user_v . visual + user_t . textual = ONE dot product of the concatenated rows (3 D wide),
        history at 1e-6, the tables of the last training forward."""
        users = torch.cat((self.ua_embeddings, self.user_embedding_v.weight, self.user_embedding_t.weight), dim=1).detach()
        items = torch.cat((self.ia_embeddings, self.visual, self.textual), dim=1).detach()
        return ranking.gene_ranklist(users, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self), items=items)

    full_sort_predict = gene_ranklist
