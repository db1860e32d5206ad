"""NCL with the reference's surface (Model/NCL.py:17-315) -- `torch.sparse.mm` family, no per-model kernel work (SURVEY
8(f).1): LightGCN's propagate through `chaorec_amd.sparse.mm`, the shared ranking, the reference's own torch expressions
for the structure-contrastive (layer 0 vs layer 2) and prototype-contrastive terms.

Same constructor, parameters and attribute names (`user_centroids`, `user_2cluster`, ... read by ProtoNCE_loss;
`restore_user_e` / `restore_item_e` read by gene_ranklist).  Differences: the adjacency is built vectorised
(graph.binary_sym_norm_csr); `e_step()` -- the per-epoch k-means, faiss on the CPU in the reference (:61-95) -- is a
seeded Lloyd iteration in torch on the embeddings' device (faiss is not in this image; the clustering is random in the
reference too, so only its contract is kept: L2-normalised centroids [k, D] and a nearest-centroid index per node)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ranking, sparse


class NCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, aggr_mode, ssl_temp,
                 ssl_reg, device):
        super(NCL, self).__init__()
        self.diag = None
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers, self.aggr_mode = dim_E, reg_weight, n_layers, aggr_mode
        self.ssl_temp, self.ssl_reg, self.device = ssl_temp, ssl_reg, device
        self.hyper_layers, self.alpha, self.proto_reg, self.k = 1, 1, 1e-7, 200          # (:33-36)
        self.user_embedding = nn.Embedding(num_user, dim_E)
        self.item_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self.restore_user_e = self.restore_item_e = None
        e = torch.as_tensor(edge_index).long()
        self.norm_adj_mat = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.user_centroids = self.user_2cluster = self.item_centroids = self.item_2cluster = None
        self._kmeans_seed = 0

    # ---- per-epoch clustering (train_and_evaluate calls e_step() before each epoch of NCL) ----------------------
    def e_step(self):
        self.user_centroids, self.user_2cluster = self.run_kmeans(self.user_embedding.weight.detach())
        self.item_centroids, self.item_2cluster = self.run_kmeans(self.item_embedding.weight.detach())

    def run_kmeans(self, x, iters=20):
        """-> (L2-normalised centroids [k, D], nearest centroid per row [n]) -- Model/NCL.py:67-95's contract."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        k = min(self.k, x.shape[0])
        g = torch.Generator(device="cpu").manual_seed(self._kmeans_seed)
        self._kmeans_seed += 1
        c = x[torch.randperm(x.shape[0], generator=g)[:k].to(x.device)].clone()
        for _ in range(iters):
            a = torch.cdist(x, c).argmin(1)
            s = torch.zeros_like(c).index_add_(0, a, x)
            n = torch.bincount(a, minlength=k).to(x.dtype).unsqueeze(1)
            c = torch.where(n > 0, s / n.clamp(min=1), c)
        return F.normalize(c, p=2, dim=1), torch.cdist(x, c).argmin(1)

    # ---- hot path -------------------------------------------------------------------------------------------------
    def get_ego_embeddings(self):
        return torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)

    def forward(self):
        """:145-156: max(L, 2 * hyper_layers) propagates, the LightGCN mean over layers 0..L, and the list of all layers."""
        x = self.get_ego_embeddings()
        layers = [x]
        for _ in range(max(self.n_layers, self.hyper_layers * 2)):
            x = sparse.mm(self.norm_adj_mat, x)
            layers.append(x)
        mean = torch.mean(torch.stack(layers[:self.n_layers + 1], dim=1), dim=1)
        u, i = torch.split(mean, [self.num_user, self.num_item])
        return u, i, layers

    def _proto_nce(self, emb_all, ids, centroids, node2cluster):
        z = F.normalize(emb_all[ids])
        pos = torch.exp(torch.mul(z, centroids[node2cluster[ids]]).sum(dim=1) / self.ssl_temp)
        ttl = torch.exp(torch.matmul(z, centroids.transpose(0, 1)) / self.ssl_temp).sum(dim=1)
        return -torch.log(pos / ttl).sum()

    def ProtoNCE_loss(self, node_embedding, user, item):
        if self.user_2cluster is None or self.item_2cluster is None:
            raise RuntimeError("user_2cluster or item_2cluster is None. Please ensure e_step is called before ProtoNCE_loss.")
        ue, ie = torch.split(node_embedding, [self.num_user, self.num_item])
        return self.proto_reg * (self._proto_nce(ue, user, self.user_centroids, self.user_2cluster)
                                 + self._proto_nce(ie, item, self.item_centroids, self.item_2cluster))

    def _layer_nce(self, cur_all, prev_all, ids):
        a, b = F.normalize(cur_all[ids]), F.normalize(prev_all[ids])
        pos = torch.exp(torch.mul(a, b).sum(dim=1) / self.ssl_temp)
        ttl = torch.exp(torch.matmul(a, F.normalize(prev_all).transpose(0, 1)) / self.ssl_temp).sum(dim=1)
        return -torch.log(pos / ttl).sum()

    def ssl_layer_loss(self, current_embedding, previous_embedding, user, item):
        cu, ci = torch.split(current_embedding, [self.num_user, self.num_item])
        pu, pi = torch.split(previous_embedding, [self.num_user, self.num_item])
        return self.ssl_reg * (self._layer_nce(cu, pu, user) + self.alpha * self._layer_nce(ci, pi, item))

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        self.restore_user_e, self.restore_item_e, layers = self.forward()
        center, context = layers[0], layers[self.hyper_layers * 2]
        ssl_loss = self.ssl_layer_loss(context, center, users, pos_items)
        proto_loss = self.ProtoNCE_loss(center, users, pos_items)
        u, p, n = self.restore_user_e[users], self.restore_item_e[pos_items], self.restore_item_e[neg_items]
        bpr_loss = -torch.mean(torch.log(torch.sigmoid(torch.mul(u, p).sum(dim=1) - torch.mul(u, n).sum(dim=1)) + 1e-5))
        reg_loss = self.reg_weight * (torch.mean(self.user_embedding(users) ** 2) + torch.mean(self.item_embedding(pos_items) ** 2)
                                      + torch.mean(self.item_embedding(neg_items) ** 2))
        return bpr_loss + reg_loss + ssl_loss + proto_loss

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:290-315 (mask value 1e-6, the embeddings of the last training forward)."""
        res = torch.cat((self.restore_user_e.detach(), self.restore_item_e.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
