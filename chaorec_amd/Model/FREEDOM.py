"""FREEDOM with the reference's surface (Model/FREEDOM.py:16-244), compute on HIP kernels.

Same constructor arguments, parameters (user/item embeddings, trainable image/text feature
tables, image_trs / text_trs Linears), `pre_epoch_processing()`, `forward(adj)`, `bpr_loss`,
`loss()` and `gene_ranklist()`.  Reference quirks kept on purpose:
  Q5  main.py passes args.lambda_coeff in the mm_image_weight slot -- the constructor takes what it is given;
  Q6  get_norm_adj_mat counts degrees on the already bidirectional list (doubled degrees);
  Q4  gene_ranklist ranks the stale self.result of the last training forward (pruned graph).
What changed underneath: every sparse product is the CSR SpMM kernel, the kNN graph is built by
the scoring+top-K kernel (no [I, I] similarity matrix), the three BPR terms are the fused BPR
kernel, the modality projections run on the f32 MFMA GEMM, the per-epoch pruning draws its weighted sample
without replacement with the device sampler (chaorec_weighted_sample_keep) instead of torch.multinomial.
"""
import torch
import torch.nn.functional as F  # noqa: F401  (kept: reference module namespace)
from torch import nn

from .. import graph, ops, ranking
from .LightGCN import _JoinTables


class FREEDOM(nn.Module):
    prunes_in_place = True     # pre_epoch_processing() rewrites masked_adj's arrays, it never re-allocates them

    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E, dim_feat, reg_weight,
                 dropout, n_layers, mm_layers, ii_topk, mm_image_weight, device):
        super(FREEDOM, self).__init__()
        self.result = None
        self.num_user = num_user
        self.num_item = num_item
        self.n_nodes = self.num_user + self.num_item
        self.user_item_dict = user_item_dict
        self.dim_E = dim_E
        self.dim_feat = dim_feat
        self.reg_weight = reg_weight
        self.n_layers = n_layers
        self.mm_layers = mm_layers
        self.mm_image_weight = mm_image_weight
        self.dropout = dropout
        self.knn_k = ii_topk
        self.v_feat = v_feat
        self.t_feat = t_feat
        self.device = device

        self.edge_index_clone = torch.as_tensor(edge_index).t().contiguous().long()          # [2, E] u -> i
        self.edge_index = torch.cat((self.edge_index_clone, self.edge_index_clone[[1, 0]]), dim=1)

        self.norm_adj = self.get_norm_adj_mat(self.edge_index).to(self.device)
        self.masked_adj, self.mm_adj = None, None

        self.edge_indices, self.edge_values = self.get_edge_info(self.edge_index_clone)
        self.edge_indices, self.edge_values = self.edge_indices.to(self.device), self.edge_values.to(self.device)
        self._prune_seed, self._prune_calls = int(torch.initial_seed()) & (2**63 - 1), 0

        self.user_embedding = nn.Embedding(self.num_user, self.dim_E)
        self.item_embedding = nn.Embedding(self.num_item, self.dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)

        self._flat = None
        self._join_tables()

        self.image_embedding = nn.Embedding.from_pretrained(self.v_feat, freeze=False)
        self.text_embedding = nn.Embedding.from_pretrained(self.t_feat, freeze=False)
        self.image_trs = nn.Linear(self.v_feat.shape[1], self.dim_feat)
        self.text_trs = nn.Linear(self.t_feat.shape[1], self.dim_feat)
        # the feature tables are read only through ops.linear_rows (the batch rows of their projection): an optimizer
        # that knows how (optim.FusedAdam) may update them without ever forming their dense [I, K] gradient
        self.image_embedding.weight._chaorec_projected_only = True
        self.text_embedding.weight._chaorec_projected_only = True
        self._batch_idx = self._loss_w = None

        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))

        image_adj = text_adj = None
        if self.v_feat is not None:
            image_adj = self.get_knn_adj_mat(self.image_embedding.weight.detach())
            self.mm_adj = graph.coo_to_csr_coalesced(image_adj[0][0], image_adj[0][1], image_adj[1], num_item,
                                                     num_item).to(self.device)
        if self.t_feat is not None:
            text_adj = self.get_knn_adj_mat(self.text_embedding.weight.detach())
            self.mm_adj = graph.coo_to_csr_coalesced(text_adj[0][0], text_adj[0][1], text_adj[1], num_item,
                                                     num_item).to(self.device)
        if self.v_feat is not None and self.t_feat is not None:
            # Model/FREEDOM.py:69: mm_adj = w * image_adj + (1 - w) * text_adj (sparse add = union of entries)
            self.mm_adj = graph.add_scaled_coo(image_adj, self.mm_image_weight, text_adj, 1.0 - self.mm_image_weight,
                                               self.num_item).to(self.device)

    # ---- the two id-embedding tables as views of ONE [N, D] buffer (as in Model/LightGCN.py here) -----------------------
    def _join_tables(self):
        """torch.cat((user_embedding.weight, item_embedding.weight)) (Model/FREEDOM.py:165) without the per-step copy
        and without the gradient split in the backward: same Parameters, same state_dict, one Adam launch for both."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        flat = torch.cat((uw.data, iw.data), 0)
        uw.data, iw.data = flat[:self.num_user], flat[self.num_user:]
        self._flat = flat

    def _apply(self, fn, *args, **kwargs):     # .to(device) / .float() / ... replace the Parameters' storage
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_flat", None) is not None:
            self._join_tables()
        return out

    @property
    def result(self):
        """[N, D] users then items (read by gene_ranklist, stale as in the reference): concatenated when it is READ -- once
        per evaluation, not once per training step.  Never cached: under a captured step the two halves are the hipGraph's
        static output buffers, rewritten by every replay without this Python code running again."""
        if self._result_parts is not None:
            return torch.cat(self._result_parts, dim=0)
        return self._result_set

    @result.setter
    def result(self, value):
        self._result_set, self._result_parts = value, None

    # ---- graph construction (host, once) -------------------------------------------------
    def get_norm_adj_mat(self, edge_index):
        """Model/FREEDOM.py:73-83 (Q6: bincount over cat(row, col) of the bidirectional list)."""
        dst, src = edge_index.long()
        # (Q6: the degree is counted over BOTH endpoint columns of the already bidirectional list -- every node twice)
        inv = torch.bincount(torch.cat([dst, src])).pow(-0.5)
        # torch.sparse.mm(adj, x): out[dst] += v * x[src]; entries coalesced (column-ascending per row)
        return graph.coo_to_csr_coalesced(dst, src, inv[dst] * inv[src], self.n_nodes, self.n_nodes, symmetric=True)

    def _normalize_adj_m(self, indices, adj_size):
        """Model/FREEDOM.py:85-99 (kept for callers of the reference's name): graph.inv_sqrt_degree_edge_weights."""
        return graph.inv_sqrt_degree_edge_weights(indices[0], indices[1], adj_size[0], adj_size[1])

    def get_edge_info(self, edge_index):
        """Model/FREEDOM.py:101-108 -> (interactions [2, E] with LOCAL item ids, their normalised weights [E])."""
        pairs = torch.stack((edge_index[0], edge_index[1] - self.num_user)).long().cpu()
        return pairs, graph.inv_sqrt_degree_edge_weights(pairs[0], pairs[1], self.num_user, self.num_item)

    def get_knn_adj_mat(self, mm_embeddings):
        """Model/FREEDOM.py:111-126: cosine kNN (self included) -> D^-1/2 A D^-1/2 with row-sum degrees.
        The [I, I] similarity matrix is never materialised: scoring + top-K are one kernel."""
        dev = self.device
        emb = mm_embeddings.to(dev)
        context_norm = emb.div(torch.norm(emb, p=2, dim=-1, keepdim=True))
        d = context_norm.shape[1]
        d_pad = next(c for c in (8, 16, 32, 64, 128) if c >= d) if d <= 128 else (d + 63) // 64 * 64
        context_norm = F.pad(context_norm, (0, d_pad - d)).contiguous()   # zero columns leave the cosine unchanged
        knn_ind, _ = ops.score_topk(context_norm, context_norm, None, 0.0, self.knn_k)
        n = context_norm.shape[0]
        rows = torch.arange(n, device=dev).unsqueeze(1).expand(-1, self.knn_k).reshape(-1)
        cols = knn_ind.reshape(-1)
        return self.compute_normalized_laplacian(torch.stack((rows, cols), 0).cpu(), (n, n))

    def compute_normalized_laplacian(self, indices, adj_size):
        """Model/FREEDOM.py:128-138 -> (indices [2, nnz], values [nnz]) on the host."""
        return indices, graph.out_degree_normalised_weights(indices[0], indices[1], adj_size[0])

    # ---- per-epoch pruning ----------------------------------------------------------------
    def pre_epoch_processing(self):
        """Model/FREEDOM.py:143-162: every epoch trains on a degree-sensitive sample of (1 - dropout) E interactions,
        re-normalised on the sample and symmetrised."""
        if self.dropout <= .0:
            self.masked_adj = self.norm_adj
            return
        n_keep = int(self.edge_values.size(0) * (1. - self.dropout))
        # torch.multinomial(edge_values, n_keep) (:151) as a keep mask from the device sampler: the same law (weighted,
        # without replacement), any edge count (multinomial stops at 2^24 categories), and a function of (torch seed,
        # epoch) only.  The reference uses the drawn set, never its order (the graph is coalesced).
        mask = ops.weighted_sample_keep(self.edge_values, n_keep, self._prune_seed, step=self._prune_calls)
        self._prune_calls += 1
        self._set_masked_adj(self.edge_indices[:, mask.bool()])

    def _set_masked_adj(self, kept):
        """The pruned propagation graph of the interactions `kept` [2, E'] (users, LOCAL item ids)."""
        w = graph.inv_sqrt_degree_edge_weights(kept[0], kept[1], self.num_user, self.num_item)
        new = graph.symmetric_bipartite_csr(kept[0], kept[1], w, self.num_user, self.num_item).to(self.device)
        # every epoch keeps the same number of edges: rewrite the pruned graph in place, so that a captured training
        # step (which holds the addresses of these arrays) trains on the new graph at its next replay
        cur = self.masked_adj
        if cur is None or cur is self.norm_adj:
            self.masked_adj = new
        elif not cur.update_from(new):
            # another entry count (a train.npy with repeated interactions coalesces to fewer entries; a key tie in the
            # race select): the arrays cannot be rewritten in place.  Without a captured step nothing holds their
            # addresses: rebind.  With one, the caller has to re-capture (train_and_evaluate does: graph_generation).
            self.masked_adj = new
            self.graph_generation = getattr(self, "graph_generation", 0) + 1

    # ---- hot path ---------------------------------------------------------------------------
    def forward(self, adj):
        """Model/FREEDOM.py:164-183."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        if uw.data_ptr() != self._flat.data_ptr() or iw.data_ptr() != self._flat[self.num_user:].data_ptr():
            self._join_tables()                # someone re-assigned a weight's storage (e.g. weight.data = ...)
        ego_embeddings = _JoinTables.apply(uw, iw, self._flat)
        all_embeddings = ops.layer_mean_propagate(ego_embeddings, adj, self.n_layers)
        u_g_embeddings, i_g_embeddings = ops.split_rows(all_embeddings, self.num_user)
        h = self.item_embedding.weight
        if self.mm_layers == 0:
            i_g_embeddings = i_g_embeddings + h
        for i in range(self.mm_layers):
            if i + 1 < self.mm_layers:
                h = ops.spmm(self.mm_adj, h)
            else:  # last item-item layer fused with `i_g_embeddings + h`
                i_g_embeddings = ops.spmm_add(self.mm_adj, h, i_g_embeddings)
        self._result_parts = (u_g_embeddings.detach(), i_g_embeddings.detach())
        return u_g_embeddings, i_g_embeddings

    def bpr_loss(self, users, pos_items, neg_items):
        """Model/FREEDOM.py:185-192 on already gathered rows (API parity; the training path below uses
        the fused gather+loss kernel and never materialises the gathers)."""
        B = users.shape[0]
        idx = torch.arange(B, device=users.device)
        tab_i = torch.cat((pos_items, neg_items), 0)
        return ops.bpr_loss(users, tab_i, idx, idx, idx + B, ops.VARIANT_LOGSIGMOID, 0.0)[0]

    def loss(self, users, pos_items, neg_items):
        """Model/FREEDOM.py:194-217."""
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        if pos_items.is_cuda and pos_items.dim() == 1 and pos_items.numel() == neg_items.numel():
            pos_items, neg_items, rows = ops.shift_cat(pos_items, neg_items, self.num_user)   # :195-196 + the batch's row list
        else:
            pos_items = pos_items - self.num_user
            neg_items = neg_items - self.num_user
            rows = torch.cat((pos_items, neg_items), 0)

        ua_embeddings, ia_embeddings = self.forward(self.masked_adj)
        # Model/FREEDOM.py:208-213 project the WHOLE feature table and read the batch rows of the result; a row of a
        # Linear depends on that row alone, so only the rows of the batch are projected (ops.linear_rows).  The three
        # BPR terms -- total = mf + reg_weight * (text + image), :203-215 -- share the user table and the batch's users:
        # one autograd node (ops.bpr_loss_multi)
        B = users.shape[0]
        if self._batch_idx is None or self._batch_idx[0].shape[0] != B or self._batch_idx[0].device != users.device:
            idx = torch.arange(B, device=users.device)
            self._batch_idx = (idx, idx + B)
        idx, idx_neg = self._batch_idx
        terms, weights, gathered = [(ia_embeddings, pos_items, neg_items)], [1.0], [None]
        if self.t_feat is not None:
            text_rows = ops.linear_rows(self.text_embedding.weight, rows, self.text_trs.weight, self.text_trs.bias)
            terms.append((text_rows, idx, idx_neg))
            weights.append(self.reg_weight)
            gathered.append((rows, self.num_item))
        if self.v_feat is not None:
            image_rows = ops.linear_rows(self.image_embedding.weight, rows, self.image_trs.weight, self.image_trs.bias)
            terms.append((image_rows, idx, idx_neg))
            weights.append(self.reg_weight)
            gathered.append((rows, self.num_item))
        if self._loss_w is None or self._loss_w.device != users.device or self._loss_w.numel() != len(weights):
            self._loss_w = torch.tensor(weights, dtype=torch.float32, device=users.device)
        return ops.bpr_loss_multi(ua_embeddings, users, ops.VARIANT_LOGSIGMOID, terms, self._loss_w, gathered=gathered)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/FREEDOM.py:219-244 (mask value 1e-6, stale self.result)."""
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
