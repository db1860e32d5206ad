"""LightGCN with the reference's surface (Model/LightGCN.py:49-162), compute on HIP kernels.

Same constructor, `forward()`, `bpr_loss()`, `regularization_loss()`, `loss()` and
`gene_ranklist()`; same parameters (`user_embedding.weight`, `item_embedding.weight`), same
initialisation calls in the same order (so the same torch seed gives the same weights).
What changed underneath:
  * the graph is normalised ONCE into a CSR in HBM instead of degree()+gathers every forward
    (reference: Model/LightGCN.py:36-38 inside every conv call);
  * the L propagate layers and the layer mean are L fused SpMM launches (ops.layer_mean_propagate);
  * BPR + L2 is one fused kernel pair; gene_ranklist never materialises the [U, I] matrix.
"""
import torch
import torch.nn as nn

from .. import graph, ops, ranking


class _JoinTables(torch.autograd.Function):
    """torch.cat((user_embedding.weight, item_embedding.weight)) (Model/LightGCN.py:77-78) without the copy: both
    weights are views of ONE [N, D] buffer, the joined table is that buffer and the gradient splits into two views."""

    @staticmethod
    def forward(ctx, user_w, item_w, flat):
        ctx.n_user = user_w.shape[0]
        return flat.detach()

    @staticmethod
    def backward(ctx, G):
        G = G.contiguous()
        return G[:ctx.n_user], G[ctx.n_user:], None


class LightGCN(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, aggr_mode,
                 device):
        super(LightGCN, self).__init__()
        self.result = None
        self.device = device
        self.num_user = num_user
        self.num_item = num_item
        self.aggr_mode = aggr_mode
        self.user_item_dict = user_item_dict
        self.reg_weight = reg_weight
        self.dim_embedding = dim_E
        self.n_layers = n_layers
        # reference keeps the raw bidirectional edge list (Model/LightGCN.py:63-64); kept for callers
        # that read it, the kernels use the CSR below
        # (an edge list handed over as a CUDA tensor -- config 5: 2e8 interactions generated on the device -- is laid
        #  out there; the raw [2, 2E] int64 copy, 6.4 GB at that size and read by nothing here, is not made)
        on_device = torch.is_tensor(edge_index) and edge_index.is_cuda
        self.edge_index = None if on_device else graph.bidirectional_edge_index(edge_index)
        self.graph = graph.lightgcn_csr(edge_index, num_user + num_item).to(device)
        # (user_item_dict=None: the history is derived from the edge list, vectorised -- graphs with millions of
        #  users, where a python dict of lists is the slowest thing in the constructor)
        rowptr, col = (graph.user_hist_csr(user_item_dict, num_user) if user_item_dict is not None
                       else graph.user_hist_csr_from_edges(edge_index, num_user))
        self.hist = (rowptr.to(device), col.to(device))

        # (a graph that was generated on the device -- BASELINE configs[4]: 12 M rows x 128 -- gets its tables there as well:
        #  xavier on the host took 6.7 of the 15.5 s a rank spent building that configuration.  A graph handed over as a host
        #  array keeps the host initialisation: the reference's seed then gives the reference's weights.)
        emb_dev = torch.device(device) if on_device else None
        self.user_embedding = nn.Embedding(num_user, dim_E, device=emb_dev)
        self.item_embedding = nn.Embedding(num_item, dim_E, device=emb_dev)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self._flat = None
        self._join_tables()

    def _join_tables(self):
        """Re-home the two embedding tables as views of one contiguous [N, D] buffer (same Parameters, same
        state_dict): the propagate then reads the joined table in place instead of concatenating every step."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        flat = torch.cat((uw.data, iw.data), 0)
        uw.data, iw.data = flat[:self.num_user], flat[self.num_user:]
        self._flat = flat

    def _apply(self, fn, *args, **kwargs):     # .to(device) / .float() / ... replace the Parameters' storage
        out = super()._apply(fn, *args, **kwargs)
        self._join_tables()
        return out

    def forward(self):
        """Model/LightGCN.py:76-95; side effect: self.result (read later by gene_ranklist)."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        if uw.data_ptr() != self._flat.data_ptr() or iw.data_ptr() != self._flat[self.num_user:].data_ptr():
            self._join_tables()                # someone re-assigned a weight's storage (e.g. weight.data = ...)
        x = _JoinTables.apply(uw, iw, self._flat)
        self.result = ops.layer_mean_propagate(x, self.graph, self.n_layers)
        return self.result

    def _fused(self, users, pos_items, neg_items, embeddings):
        return ops.bpr_loss(embeddings, None, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS,
                            self.reg_weight, item_offset=self.num_user)

    def bpr_loss(self, users, pos_items, neg_items, embeddings):
        """Model/LightGCN.py:97-110 (items are LOCAL ids here, as in the reference)."""
        return ops.bpr_loss(embeddings, None, users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0,
                            item_offset=self.num_user)[0]

    def regularization_loss(self, users, pos_items, neg_items, embeddings):
        """Model/LightGCN.py:112-121."""
        return self._fused(users, pos_items, neg_items, embeddings)[2]

    def loss(self, users, pos_items, neg_items):
        """Model/LightGCN.py:123-135: global item ids in, full-graph forward, BPR + L2 out."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        embeddings = self.forward()
        return self._fused(users, pos_items, neg_items, embeddings)[0]

    def loss_local(self, users, pos_items, neg_items):
        """loss() for batches that already hold LOCAL item ids on the device (ops.draw_batch): same arithmetic,
        without the id shift and the host->device copies."""
        return self._fused(users, pos_items, neg_items, self.forward())[0]

    def loss_drawn(self, edges, B, seed, step, step_dev=None, advance=False, perm=None, perm_pos=None):
        """loss_local() with the batch drawn inside the fused BPR forward (ops.bpr_loss_drawn): `edges` is the int64
        [E, 2] training list on the device (global item ids); the ids follow ops.draw_batch for the same
        (seed, step, step_dev).  Keeps self.batch = (users, pos, neg) for callers that want to look."""
        out, users, pos, neg = ops.bpr_loss_drawn(self.forward(), None, edges, self.hist, B, self.num_user,
                                                  self.num_item, seed, step, ops.VARIANT_LOG_SIGMOID_EPS,
                                                  self.reg_weight, item_offset=self.num_user, step_dev=step_dev,
                                                  advance=advance, perm=perm, perm_pos=perm_pos)
        self.batch = (users, pos, neg)
        return out[0]

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/LightGCN.py:137-162 -> LongTensor [U, topk] of GLOBAL item ids on the CPU.
        Uses the stale self.result of the last training forward, as the reference does."""
        if self.result is None:
            raise RuntimeError("LightGCN.gene_ranklist: no propagated table -- the last training step was a light one "
                               "(optim.FusedLightGCNStep with light_forward: only its batch's rows were computed).  Run the "
                               "step before an evaluation with full_result=True (FusedLightGCNStep.run does), or call "
                               "forward() first")
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    # north_star names full_sort_predict(); the reference method is gene_ranklist
    full_sort_predict = gene_ranklist
