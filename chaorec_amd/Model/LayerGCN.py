"""LayerGCN with the reference's surface (Model/LayerGCN.py:18-219), compute on HIP kernels -- a second member of the
`torch.sparse.mm` model family: LightGCN's propagate, each layer re-weighted by its cosine similarity
to the ego table, on a graph that is re-pruned every epoch (alternating degree-sensitive and uniform pruning).

Same constructor, parameters (`user_embeddings`, `item_embeddings`, plain nn.Parameters created in the reference's
order), `pre_epoch_processing()`, `forward()` over `self.forward_adj`, `bpr_loss`, `regularization_loss`, `loss()`,
`gene_ranklist()` (which, unlike LightGCN's, runs a fresh forward on the UNPRUNED graph, :196-219).

What changed underneath: the normalised adjacency is built vectorised into a CSR in HBM (the reference fills a scipy
dok matrix entry by entry, :55-75; its fp64 degree arithmetic and final fp32 cast are kept); torch.sparse.mm is the
CSR SpMM kernel; both prunings draw their sample without replacement on the device (chaorec_weighted_sample_keep,
equal weights for the uniform one) and rewrite the pruned CSR in place, so the captured training step follows it;
the cosine re-weighting of every layer (a dozen elementwise / reduction launches forward, two dozen backward in
torch) is one fused row kernel each way (chaorec_row_cosine_scale_*); BPR is the fused kernel; scoring + mask +
top-K never materialise [U, I].
"""
import numpy as np
import torch
from torch import nn

from .. import graph, ops, ranking, sparse


class LayerGCN(nn.Module):
    prunes_in_place = True     # pre_epoch_processing() rewrites masked_adj's arrays, it never re-allocates them

    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, dropout,
                 device):
        super(LayerGCN, self).__init__()
        self.num_user = num_user
        self.num_item = num_item
        self.edge_index = edge_index
        self.user_item_dict = user_item_dict
        self.dim_E = dim_E
        self.reg_weight = reg_weight
        self.n_layers = n_layers
        self.dropout = dropout
        self.device = device
        self.n_nodes = num_user + num_item

        edges = np.asarray(edge_index)
        self._u = torch.from_numpy(edges[:, 0].astype(np.int64))
        self._i = torch.from_numpy(edges[:, 1].astype(np.int64) - num_user)

        self.user_embeddings = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.num_user, self.dim_E)))
        self.item_embeddings = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.num_item, self.dim_E)))

        self.norm_adj_matrix = self.get_norm_adj_mat().to(self.device)
        rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
        self.hist = (rowptr.to(device), col.to(device))

        self.masked_adj = None
        self.forward_adj = None
        self.pruning_random = False
        self.edge_indices, self.edge_values = self.get_edge_info()
        self.edge_indices, self.edge_values = self.edge_indices.to(device), self.edge_values.to(device)
        self._uniform = torch.ones_like(self.edge_values)
        self._prune_seed, self._prune_calls = int(torch.initial_seed()) & (2**63 - 1), 0

    def get_norm_adj_mat(self):
        """Model/LayerGCN.py:55-75: binary A (repeated interactions count once), degree = distinct neighbours + 1e-7,
        D^-1/2 A D^-1/2 evaluated in fp64 and stored as fp32."""
        U, I, N = self.num_user, self.num_item, self.n_nodes
        key = torch.unique(self._u * I + self._i)
        u, i = torch.div(key, I, rounding_mode="floor"), key % I
        deg = torch.zeros(N, dtype=torch.float64)
        deg.index_add_(0, u, torch.ones(u.numel(), dtype=torch.float64))
        deg.index_add_(0, U + i, torch.ones(i.numel(), dtype=torch.float64))
        d = torch.from_numpy(np.power(deg.numpy() + 1e-7, -0.5))
        val = ((d[u] * 1.0) * d[U + i]).to(torch.float32)
        return graph.coo_to_csr_coalesced(torch.cat([u, U + i]), torch.cat([U + i, u]), torch.cat([val, val]), N, N,
                                          symmetric=True)

    def get_edge_info(self):
        """Model/LayerGCN.py:77-82: the interaction list as given (repeats kept) and its pruning weights."""
        edges = torch.stack([self._u, self._i]).type(torch.LongTensor)
        return edges, self._normalize_adj_m(edges, torch.Size((self.num_user, self.num_item)))

    def _normalize_adj_m(self, indices, adj_size):
        """Model/LayerGCN.py:84-93: (1e-7 + row count)^-1/2 * (1e-7 + column count)^-1/2 per entry, fp32."""
        one = torch.ones(indices.shape[1], dtype=torch.float32, device=indices.device)
        row_sum = 1e-7 + torch.zeros(adj_size[0], dtype=torch.float32, device=indices.device).index_add_(0, indices[0], one)
        col_sum = 1e-7 + torch.zeros(adj_size[1], dtype=torch.float32, device=indices.device).index_add_(0, indices[1], one)
        return torch.pow(row_sum, -0.5)[indices[0]] * torch.pow(col_sum, -0.5)[indices[1]]

    def pre_epoch_processing(self):
        """Model/LayerGCN.py:95-112: every other epoch a degree-sensitive sample (torch.multinomial), in between a
        uniform one (random.sample) -- both as a device keep mask; the reference only uses the drawn set."""
        if self.dropout <= .0:
            self.masked_adj = self.norm_adj_matrix
            return
        keep_len = int(self.edge_values.size(0) * (1. - self.dropout))
        weights = self._uniform if self.pruning_random else self.edge_values
        keep = ops.weighted_sample_keep(weights, keep_len, self._prune_seed, step=self._prune_calls)
        self._prune_calls += 1
        self.pruning_random = True ^ self.pruning_random
        self._set_masked_adj(self.edge_indices[:, keep.bool()])

    def _set_masked_adj(self, keep_indices):
        keep_values = self._normalize_adj_m(keep_indices, torch.Size((self.num_user, self.num_item)))
        all_values = torch.cat((keep_values, keep_values))
        keep_indices = keep_indices.clone()
        keep_indices[1] += self.num_user
        all_indices = torch.cat((keep_indices, torch.flip(keep_indices, [0])), 1)
        new = graph.coo_to_csr_coalesced(all_indices[0], all_indices[1], all_values, self.n_nodes, self.n_nodes,
                                         symmetric=True).to(self.device)
        cur = self.masked_adj
        if cur is None or cur is self.norm_adj_matrix:
            self.masked_adj = new
        elif not cur.update_from(new):
            # another entry count (a train.npy with repeated interactions coalesces to fewer entries; a key tie in the
            # race select): the arrays cannot be rewritten in place.  Without a captured step nothing holds their
            # addresses: rebind.  With one, the caller has to re-capture (train_and_evaluate does: graph_generation).
            self.masked_adj = new
            self.graph_generation = getattr(self, "graph_generation", 0) + 1

    def get_ego_embeddings(self):
        return torch.cat([self.user_embeddings, self.item_embeddings], 0)

    def forward(self):
        """Model/LayerGCN.py:118-132."""
        ego_embeddings = self.get_ego_embeddings()
        all_embeddings = ego_embeddings
        total = None
        for _ in range(self.n_layers):
            all_embeddings = sparse.mm(self.forward_adj, all_embeddings)
            # _weights = F.cosine_similarity(all, ego, dim=-1); all = einsum('a,ab->ab', _weights, all): one launch
            all_embeddings = ops.row_cosine_scale(all_embeddings, ego_embeddings)
            total = all_embeddings if total is None else total + all_embeddings
        return torch.split(total, [self.num_user, self.num_item])

    def bpr_loss(self, users, pos_items, neg_items, user_all_embeddings, item_all_embeddings):
        """Model/LayerGCN.py:134-145."""
        return ops.bpr_loss(user_all_embeddings.contiguous(), item_all_embeddings.contiguous(), users, pos_items,
                            neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items):
        """Model/LayerGCN.py:147-155: on the EGO tables (LightGCN regularises the propagated ones)."""
        return self.reg_weight * (ops.mean_all(self.user_embeddings[users] ** 2)
                                  + ops.mean_all(self.item_embeddings[pos_items] ** 2)
                                  + ops.mean_all(self.item_embeddings[neg_items] ** 2))

    def loss(self, users, pos_items, neg_items):
        """Model/LayerGCN.py:157-169."""
        pos_items = pos_items - self.num_user
        neg_items = neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        return self.loss_local(users, pos_items, neg_items)

    def loss_local(self, users, pos_items, neg_items):
        self.forward_adj = self.masked_adj
        user_all_embeddings, item_all_embeddings = self.forward()
        return self.bpr_loss(users, pos_items, neg_items, user_all_embeddings, item_all_embeddings) + \
            self.regularization_loss(users, pos_items, neg_items)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """Model/LayerGCN.py:196-219: fresh forward on the unpruned graph, mask value 1e-6."""
        self.forward_adj = self.norm_adj_matrix
        with torch.no_grad():
            u, i = self.forward()
            self.result = torch.cat([u, i], 0)
        return ranking.gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
