"""VBPR (reference Model/VBPR.py, SURVEY 8(f).2: the models whose whole step is the BPR kernel + the shared ranking).

gamma_u [U, E + 64]; gamma_i = [item_embedding [I, E] | item_linear(v_feat) [I, 64]] with the visual features a trainable
table (Model/VBPR.py:34: from_pretrained(..., freeze=False)).  The step is one projection, one concatenation and the fused
BPR(+L2) kernel over the joined [U + I, E + 64] table (the same loss form as LightGCN's: log(sigmoid + 1e-5), means of the
squared rows); evaluation is the shared full-rank top-K over the stale `result` of the last forward, as in the reference.
The feature table is read only through its projection: optim.FusedAdam applies its rank-64 gradient without materialising
it (chaorec_adam_lowrank_f32) -- only the batch rows of gy are non-zero here.
"""
import torch
import torch.nn as nn

from .. import graph, ops, ranking


class VBPR(nn.Module):
    def __init__(self, num_user, num_item, user_item_dict, v_feat, embedding_dim, feature_embedding, reg_weight, device):
        super(VBPR, self).__init__()
        self.result = None
        self.user_item_dict = user_item_dict
        self.num_user = num_user
        self.num_item = num_item
        self.device = device
        self.visual_embedding = 64                    # (Model/VBPR.py:23: fixed, `feature_embedding` is not used)
        self.user_embedding = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_user, embedding_dim + self.visual_embedding)))
        self.item_embedding = nn.Parameter(nn.init.xavier_uniform_(torch.empty(num_item, embedding_dim)))
        self.v_feat = nn.Embedding.from_pretrained(v_feat, freeze=False)
        self.item_linear = nn.Linear(v_feat.shape[1], self.visual_embedding)
        nn.init.xavier_uniform_(self.item_linear.weight)
        self.reg_weight = reg_weight
        self.v_feat.weight._chaorec_projected_only = True
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))

    def forward(self):
        """Model/VBPR.py:40-46."""
        visual_embeddings = ops.linear(self.v_feat.weight, self.item_linear.weight, self.item_linear.bias)
        item_embeddings = torch.cat((self.item_embedding, visual_embeddings), dim=-1)
        self.result = torch.cat((self.user_embedding, item_embeddings), dim=0)
        return self.result

    def _fused(self, users, pos_items, neg_items, embeddings):
        return ops.bpr_loss(embeddings, None, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, self.reg_weight,
                            item_offset=self.num_user)

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        """Model/VBPR.py:48-61 (local item ids)."""
        return self._fused(users, pos_items, neg_items, embeddings)[1]

    def regularization_loss(self, users, pos_items, neg_items, embeddings):
        """Model/VBPR.py:63-72."""
        return self._fused(users, pos_items, neg_items, embeddings)[2]

    def loss(self, users, pos_items, neg_items):
        """Model/VBPR.py:74-86."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        return self._fused(users, pos_items, neg_items, self.forward())[0]

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/VBPR.py:88-113 (mask value 1e-6, stale self.result)."""
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
