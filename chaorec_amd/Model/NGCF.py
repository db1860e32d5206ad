"""NGCF with the reference's surface (Model/NGCF.py:20-195), compute on HIP kernels.

Same constructor, `forward()`, `bpr_loss()`, `regularization_loss()`, `loss()`, `gene_ranklist()`, same parameters
(`user_embedding.weight`, `item_embedding.weight`, `conv_layers.{l}.W1.weight`, `conv_layers.{l}.W2.weight`) created
and initialised in the reference's order (same torch seed -> same weights).

What changed underneath.  The reference evaluates, per conv call, for every edge e = (row -> col)
    m_e = norm_e * (W1 x[row] + W2 (x[row] * x[col]))        and        out[col] = leaky_relu_0.2(sum_e m_e)
(message()/update(), Model/NGCF.py:60-84: x_j is lifted by edge_index[0], x_i by edge_index[1], aggregation at
edge_index[1]) -- two [E, D] x [D, D] products and four [E, D] intermediates per layer.  Both W's are linear and x[col]
is constant inside a destination's sum, so
    out[c] = leaky_relu_0.2( W1 s_c + W2 (s_c * x_c) ),      s = A x,   A = D^-1/2 (keep*(edges) + I) D^-1/2
which is ONE SpMM over the CSR in HBM, one elementwise product and two [N, D] x [D, D] MFMA GEMMs (the second
accumulates into the first with the activation in its epilogue).  Same real-number result; fp32 rounding differs by
re-association, so parity with the reference is to tolerance (tests/test_gpu_models.py), not bit-wise.

Edge dropout (drop = 'all', p = args.dropout, applied in training AND evaluation forwards, as in the reference):
the CSR structure is fixed; chaorec_edge_dropout_norm rewrites only its value array each call (dropped entry -> 0,
kept entry -> renormalised weight, degrees recounted on the kept list).  The mask comes from a counter generator on
the device, so the whole step is hipGraph-capturable; it follows dropout_adj's law, not torch's generator stream.
"""
import torch
import torch.nn as nn

from .. import graph, ops, ranking


class NGCFConv(nn.Module):
    def __init__(self, in_channels, out_channels, isdrop, dropout, aggr='add', **kwargs):
        super(NGCFConv, self).__init__()
        self.drop = isdrop
        self.message_dropout = dropout
        self.node_dropout = dropout
        self.aggr = aggr
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.W1 = nn.Linear(in_channels, out_channels, bias=False)
        self.W2 = nn.Linear(in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.W1.weight)
        nn.init.xavier_uniform_(self.W2.weight)

    def drops_edges(self):
        """Model/NGCF.py:39, operator precedence included: `drop == "message" or (drop == "all" and p > 0)`."""
        return self.drop == "message" or (self.drop == "all" and self.message_dropout > 0)

    def forward(self, x, structure, seed=0, step_dev=None, salt=0, keep=None):
        if self.drops_edges() and (self.message_dropout > 0 or keep is not None):
            val, val_t = ops.edge_dropout_norm(structure, self.message_dropout, seed, 0, step_dev, salt, keep)
            s = ops.spmm_values(structure, val, val_t, x)
        else:
            s = ops.spmm(structure, x)
        return ops.ngcf_layer(s, x, self.W1.weight, self.W2.weight)


class NGCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, dropout, n_layers, aggr_mode,
                 device):
        super(NGCF, self).__init__()
        self.result = None
        self.device = device
        self.num_user = num_user
        self.num_item = num_item
        self.aggr_mode = aggr_mode
        self.user_item_dict = user_item_dict
        self.reg_weight = reg_weight
        self.dim_embedding = dim_E
        self.message_dropout = dropout
        self.node_dropout = dropout
        self.drop = 'all'
        self.edge_index = graph.bidirectional_edge_index(edge_index)
        self.graph = graph.ngcf_structure(edge_index, num_user + num_item).to(device)
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))

        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.conv_layers = nn.ModuleList([NGCFConv(dim_E, dim_E, self.drop, self.message_dropout, aggr=aggr_mode)
                                          for _ in range(n_layers)])
        # dropout stream: keyed by torch's seed (torch.manual_seed governs it, nothing is drawn from the generator),
        # one device counter per forward so that a captured step draws a fresh mask on every replay
        self.drop_seed = int(torch.initial_seed()) & (2**63 - 1)
        self._drop_calls = torch.zeros(1, dtype=torch.int64, device=device)
        self.forced_keep = None       # tests: list of per-layer uint8 masks in CSR entry order

    def forward(self):
        """Model/NGCF.py:114-127; result = SUM of the layer outputs (the reference's comment says mean, the code sums)."""
        x = torch.cat((self.user_embedding.weight, self.item_embedding.weight), dim=0)
        out = x
        for l, conv in enumerate(self.conv_layers):
            keep = self.forced_keep[l] if self.forced_keep is not None else None
            x = conv(x, self.graph, self.drop_seed, self._drop_calls, l, keep)
            out = out + x
        if self.message_dropout > 0:
            self._drop_calls += 1
        self.result = out
        return self.result

    def _fused(self, users, pos_items, neg_items, embeddings):
        return ops.bpr_loss(embeddings, None, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS,
                            self.reg_weight, item_offset=self.num_user)

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        """Model/NGCF.py:129-142."""
        return ops.bpr_loss(embeddings, None, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0,
                            item_offset=self.num_user)[0]

    def regularization_loss(self, users, pos_items, neg_items, embeddings):
        """Model/NGCF.py:144-154."""
        return self._fused(users, pos_items, neg_items, embeddings)[2]

    def loss(self, users, pos_items, neg_items):
        """Model/NGCF.py:156-168."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        return self._fused(users, pos_items, neg_items, self.forward())[0]

    def loss_local(self, users, pos_items, neg_items):
        return self._fused(users, pos_items, neg_items, self.forward())[0]

    def loss_drawn(self, edges, B, seed, step, step_dev=None, advance=False, perm=None, perm_pos=None):
        """loss_local() with the batch drawn inside the fused BPR forward (see LightGCN.loss_drawn)."""
        out, users, pos, neg = ops.bpr_loss_drawn(self.forward(), None, edges, self.hist, B, self.num_user,
                                                  self.num_item, seed, step, ops.VARIANT_LOG_SIGMOID_EPS,
                                                  self.reg_weight, item_offset=self.num_user, step_dev=step_dev,
                                                  advance=advance, perm=perm, perm_pos=perm_pos)
        self.batch = (users, pos, neg)
        return out[0]

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/NGCF.py:170-195 (mask value 1e-6)."""
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
