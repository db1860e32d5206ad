"""POWERec with the reference's surface (Model/POWERec.py:17-282) -- prompt-tuned LayerGCN branches for the id, visual and
textual view of the items, through the hot-path adapters alone: every branch is LayerGCN's propagate (`chaorec_amd.sparse.mm`
on the HIP SpMM + the fused cosine re-weighting launch, four layers, ego row included in the sum: :41-54), the item side of
a branch is `tanh(Linear(features))` on the MFMA GEMM (:32,40), the per-epoch pruning is LayerGCN's
(`pre_epoch_processing`, :155-176: the same alternation of a degree-sensitive and a uniform draw without replacement), the
ranking is ONE `ranking.gene_ranklist` over the concatenated [id | visual | textual] tables of a fresh forward on the
unpruned graph (:257-282).  The weak-modality negative of the loss (:196-231) is elementwise work on [B, 3 D] batch rows
and stays torch.

Same constructor, parameters created in the reference's order (user / item tables, the three prompts, then the branches'
Linears: same seed, same weights, same `named_parameters()` names)."""
import torch
from torch import nn

from .. import ops, ranking, sparse
from .LayerGCN import LayerGCN


class _Branch(nn.Module):
    """Model/POWERec.py:17-55 (the file's own `LayerGCN`): user rows = the shared user table + the summed prompt, item rows =
    tanh(Linear(item features)); four propagations, each re-weighted by its rows' cosine to the ego rows; the sum of all
    five tables."""
    n_layers = 4

    def __init__(self, num_user, num_item, user_fea, item_fea, emb_size, prompt_embedding):
        super().__init__()
        self.num_user, self.num_item = num_user, num_item
        self.user_fea = user_fea                     # (the model's own Parameters: shared objects, listed once)
        self.prompt_embedding = prompt_embedding
        if isinstance(item_fea, nn.Parameter):
            self.item_fea = item_fea
        else:
            self.register_buffer("item_fea", item_fea, persistent=False)
        self.mlp = nn.Sequential(nn.Linear(item_fea.shape[1], emb_size), nn.Tanh())

    def forward(self, adj):
        user_embd = self.user_fea + torch.sum(self.prompt_embedding, 0)[None, :]
        item_embd = torch.tanh(ops.linear(self.item_fea, self.mlp[0].weight, self.mlp[0].bias))
        ego = torch.cat((user_embd, item_embd), dim=0)
        x, total = ego, ego
        for _ in range(self.n_layers):
            x = ops.row_cosine_scale(sparse.mm(adj, x), ego)
            total = total + x
        return torch.split(total, [self.num_user, self.num_item])


class POWERec(LayerGCN):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, reg_weight,
                 n_layers, prompt_num, neg_weight, dropout, device):
        # user / item tables, normalised adjacency, history, pruning weights: LayerGCN's (:85-86,89,108 in the same order)
        super().__init__(num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, dropout, device)
        self.result = None
        self.prompt_num, self.neg_weight, self.num_modal = prompt_num, neg_weight, 3
        self.id_prompt = nn.Parameter(nn.init.xavier_uniform_(torch.empty(prompt_num, dim_E)))
        self.v_prompt = nn.Parameter(nn.init.xavier_uniform_(torch.empty(prompt_num, dim_E)))
        self.t_prompt = nn.Parameter(nn.init.xavier_uniform_(torch.empty(prompt_num, dim_E)))
        self.id_model = _Branch(num_user, num_item, self.user_embeddings, self.item_embeddings, dim_E, self.id_prompt)
        self.v_model = _Branch(num_user, num_item, self.user_embeddings, v_feat, dim_E, self.v_prompt)
        self.t_model = _Branch(num_user, num_item, self.user_embeddings, t_feat, dim_E, self.t_prompt)

    def forward(self, adj):
        """:178-187."""
        user_id, item_id = self.id_model(adj)
        user_v, item_v = self.v_model(adj)
        user_t, item_t = self.t_model(adj)
        return torch.cat([user_id, user_v, user_t], 1), torch.cat([item_id, item_v, item_t], 1)

    def find_weak_modality(self, user_e, pos_e, neg_e):
        """:216-231: per batch row the modality whose (pos - neg) score is the smallest, as a 0/1 mask over its D columns."""
        pos_score_ = torch.mul(user_e, pos_e).view(-1, self.num_modal, self.dim_E).sum(dim=-1)
        neg_score_ = torch.mul(user_e, neg_e).view(-1, self.num_modal, self.dim_E).sum(dim=-1)
        modality_indicator = (pos_score_ - neg_score_).softmax(-1).detach()
        weak_modality = (modality_indicator == modality_indicator.min(dim=-1, keepdim=True)[0]).to(dtype=torch.float32)
        weak_modality = torch.tile(weak_modality.view(-1, self.num_modal, 1), [1, 1, self.dim_E])
        return weak_modality.view(-1, self.num_modal * self.dim_E), modality_indicator

    def bpr_loss(self, users, pos_items, neg_items, u_g, i_g):
        """:189-214: BPR + neg_weight * BPR against the positive with its weakest modality swapped for the negative's."""
        user_embeddings, pos_e, neg_e = u_g[users], i_g[pos_items], i_g[neg_items]
        pos_scores = torch.sum(user_embeddings * pos_e, dim=1)
        neg_scores = torch.sum(user_embeddings * neg_e, dim=1)
        bpr_loss = -torch.mean(torch.log(torch.sigmoid(pos_scores - neg_scores) + 1e-5))
        weak_modality, _ = self.find_weak_modality(user_embeddings, pos_e, neg_e)
        fake_neg_e = (1 - weak_modality) * pos_e + weak_modality * neg_e
        fake_neg_scores = torch.mul(user_embeddings, fake_neg_e).sum(1)
        weak_loss = -torch.mean(torch.log(torch.sigmoid(pos_scores - fake_neg_scores) + 1e-5))
        return bpr_loss + self.neg_weight * weak_loss

    def regularization_loss(self, users, pos_items, neg_items, u_g, i_g):
        """:233-243: on the PROPAGATED rows (LayerGCN regularises the ego tables)."""
        return self.reg_weight * (ops.mean_all(u_g[users] ** 2) + ops.mean_all(i_g[pos_items] ** 2) + ops.mean_all(i_g[neg_items] ** 2))

    def loss_local(self, users, pos_items, neg_items):
        """:245-255 (ids already local and on the device: LayerGCN.loss does that part)."""
        u_g, i_g = self.forward(self.masked_adj)
        return self.bpr_loss(users, pos_items, neg_items, u_g, i_g) + self.regularization_loss(users, pos_items, neg_items, u_g, i_g)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:257-282: fresh forward on the unpruned graph, history at 1e-6."""
        with torch.no_grad():
            u, i = self.forward(self.norm_adj_matrix)
            self.result = torch.cat([u, i], 0)
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
