"""BPRMF (reference Model/BPR.py): matrix factorisation with an item bias, trained with the BPR kernel and
ranked with the shared full-rank top-K.

score(u, i) = <U_u, V_i> + b_i (Model/BPR.py:35-51).  The fused kernel computes row dot products, so the bias rides in
the tables: user rows [U_u | 1 | 0 0 0], item rows [V_i | b_i | 0 0 0] (width dim_E + 4: the kernel's float4 lanes), built
per step by one concatenation each -- autograd slices the gradient of the joined row back into the three parameters.
The reference's regulariser (Model/BPR.py:62, kept as written: the NEGATIVE rows enter un-squared) is three row gathers
in torch; its ranking ignores the bias (Model/BPR.py:71-77), and so does this one.
"""
import torch
import torch.nn as nn

from .. import graph, ops, ranking


class BPRMF(nn.Module):
    def __init__(self, num_user, num_item, user_item_dict, dim_E, reg_weight, device):
        super(BPRMF, self).__init__()
        self.user_item_dict = user_item_dict
        self.num_user = num_user
        self.num_item = num_item
        self.device = device
        self.item_bias = nn.Embedding(num_item, 1)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        self.reg_weight = reg_weight
        nn.init.zeros_(self.item_bias.weight)
        nn.init.xavier_normal_(self.user_embedding.weight)
        nn.init.xavier_normal_(self.item_embedding.weight)
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))
        self._pad = None

    def _tables(self):
        U, I = self.num_user, self.num_item
        dev = self.user_embedding.weight.device
        if self._pad is None or self._pad[0].device != dev:
            ones = torch.zeros(U, 4, device=dev)
            ones[:, 0] = 1.0
            self._pad = (ones, torch.zeros(I, 3, device=dev))
        tab_u = torch.cat((self.user_embedding.weight, self._pad[0]), 1)
        tab_i = torch.cat((self.item_embedding.weight, self.item_bias.weight, self._pad[1]), 1)
        return tab_u, tab_i

    def forward(self, users, pos_items, neg_items):
        """Model/BPR.py:35-51 (local item ids) -> (positive scores, negative scores)."""
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        u = self.user_embedding(users)
        pos = torch.sum(u * self.item_embedding(pos_items), dim=1) + self.item_bias(pos_items).squeeze(-1)
        neg = torch.sum(u * self.item_embedding(neg_items), dim=1) + self.item_bias(neg_items).squeeze(-1)
        return pos, neg

    def loss(self, users, pos_items, neg_items):
        """Model/BPR.py:53-67."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        tab_u, tab_i = self._tables()
        bpr_loss = ops.bpr_loss(tab_u, tab_i, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID, 0.0)[0]
        reg_loss = (self.user_embedding(users) ** 2).mean() + (self.item_embedding(pos_items) ** 2).mean() + \
            (self.item_embedding(neg_items)).mean()
        return bpr_loss + reg_loss * self.reg_weight

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/BPR.py:69-93 (mask value 1e-6; scores WITHOUT the item bias, current tables)."""
        result = torch.cat((self.user_embedding.weight, self.item_embedding.weight), 0).detach()
        return ranking.gene_ranklist(result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
