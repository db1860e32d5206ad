"""FKAN_GCF with the reference's surface (Model/FKAN_GCF.py:17-241, kanlayer.py:14-37) -- NGCF's bi-interaction layer with the
second weight matrix replaced by a Fourier KAN layer: out = E + L E + KAN((L E) * E), leaky-relu, dropout, row-normalised,
all layers concatenated -- through the hot-path adapters alone: `L E` is `chaorec_amd.sparse.mm` on the HIP SpMM (node
dropout = the family's `sparse.sparse_dropout`: same structure, another value array), the KAN layer's einsum over (input,
frequency) is ONE MFMA GEMM over the [N, 2 D G] cos / sin features (`ops.linear`), BPR is the fused kernel, the ranking is
`ranking.gene_ranklist` over the tables of the last training forward (:222-241).

Same constructor, parameters in the reference's creation order (the layers' Fourier coefficients, then the user / item
tables).  Quirk kept: the message dropout is a fresh `nn.Dropout` per call (:176), so it is active in eval mode too."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import graph, ops, ranking, sparse


class NaiveFourierKANLayer(nn.Module):
    """kanlayer.py:14-37: y[b, j] = sum_i sum_k cos(k x[b, i]) C[0, j, i, k] + sin(k x[b, i]) C[1, j, i, k], k = 1 .. gridsize."""

    def __init__(self, inputdim, outdim, gridsize=300):
        super().__init__()
        self.gridsize, self.inputdim, self.outdim = gridsize, inputdim, outdim
        self.fouriercoeffs = nn.Parameter(torch.randn(2, outdim, inputdim, gridsize) / (np.sqrt(inputdim) * np.sqrt(gridsize)))

    def forward(self, x):
        k = torch.arange(1, self.gridsize + 1, device=x.device, dtype=x.dtype)
        kx = x.unsqueeze(-1) * k                                                       # [N, I, G]
        feats = torch.cat([torch.cos(kx).flatten(1), torch.sin(kx).flatten(1)], dim=1)      # [N, 2 I G]
        w = torch.cat([self.fouriercoeffs[0].flatten(1), self.fouriercoeffs[1].flatten(1)], dim=1)   # [J, 2 I G]
        return ops.linear(feats, w)


class FourierGNNLayer(nn.Module):
    """:17-36."""

    def __init__(self, in_dim, out_dim, grid_size):
        super().__init__()
        self.in_dim, self.out_dim, self.grid_size = in_dim, out_dim, grid_size
        self.interActTransform = NaiveFourierKANLayer(in_dim, out_dim, grid_size)

    def forward(self, lap_matrix, eye_matrix, features):
        x = sparse.mm(lap_matrix, features)
        return features + x + self.interActTransform(torch.mul(x, features))


class FKAN_GCF(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers,
                 node_dropout, message_dropout, grid_size, device):
        super(FKAN_GCF, self).__init__()
        self.num_user, self.num_item, self.user_item_dict = num_user, num_item, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers = dim_E, reg_weight, n_layers
        self.node_dropout, self.message_dropout, self.grid_size, self.device = node_dropout, message_dropout, grid_size, device
        self.hidden_size_list = [dim_E] * n_layers
        e = torch.as_tensor(np.asarray(edge_index)).long()
        self.norm_adj_matrix = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)   # :89-117
        self.eye_matrix = None                                         # (:119-128: built there, never read by the layer)
        self.GNNlayers = torch.nn.ModuleList()
        for input_size, output_size in zip(self.hidden_size_list[:-1], self.hidden_size_list[1:]):
            self.GNNlayers.append(FourierGNNLayer(input_size, output_size, grid_size))
        self.user_embedding = nn.Embedding(num_embeddings=num_user, embedding_dim=dim_E)
        self.item_embedding = nn.Embedding(num_embeddings=num_item, embedding_dim=dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.user_emb_final = self.item_emb_final = None

    def get_ego_embeddings(self):
        return torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)

    def forward(self):
        """:164-183."""
        A_hat = self.norm_adj_matrix
        if self.node_dropout != 0 and self.training:
            A_hat = sparse.sparse_dropout(self.norm_adj_matrix, self.node_dropout)      # :38-55
        all_embeddings = self.get_ego_embeddings()
        embeddings_list = [all_embeddings]
        for gnn in self.GNNlayers:
            all_embeddings = gnn(A_hat, self.eye_matrix, all_embeddings)
            all_embeddings = F.leaky_relu(all_embeddings, negative_slope=0.2)
            all_embeddings = F.dropout(all_embeddings, self.message_dropout, training=True)
            all_embeddings = F.normalize(all_embeddings, p=2, dim=1)
            embeddings_list += [all_embeddings]
        return torch.split(torch.cat(embeddings_list, dim=1), [self.num_user, self.num_item])

    def bpr_loss(self, users, pos_items, neg_items, user_emb, item_emb):
        return ops.bpr_loss(user_emb.contiguous(), item_emb.contiguous(), users, pos_items, neg_items, ops.VARIANT_LOG_SIGMOID_EPS, 0.0)[0]

    def regularization_loss(self, users, pos_items, neg_items):
        """:198-207: on the ego rows."""
        return self.reg_weight * (ops.mean_all(self.user_embedding.weight[users] ** 2) + ops.mean_all(self.item_embedding.weight[pos_items] ** 2)
                                  + ops.mean_all(self.item_embedding.weight[neg_items] ** 2))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        user_emb, item_emb = self.forward()
        self.user_emb_final, self.item_emb_final = user_emb, item_emb
        return self.bpr_loss(users, pos_items, neg_items, user_emb, item_emb) + self.regularization_loss(users, pos_items, neg_items)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:222-241: the tables of the last training forward, history at 1e-6."""
        result = torch.cat([self.user_emb_final.detach(), self.item_emb_final.detach()], 0)
        return ranking.gene_ranklist(result, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
