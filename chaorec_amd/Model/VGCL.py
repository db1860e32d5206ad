"""VGCL with the reference's surface (Model/VGCL.py:18-321) -- `torch.sparse.mm` family, no per-model kernel work (SURVEY
8(f).1): the L propagates of the variational graph encoder through `chaorec_amd.sparse.mm` (the mean runs over layers
1..L: the ego table is NOT part of it, :135-141), the log-std projection on the MFMA GEMM (`ops.linear`), the shared
ranking over the FIRST noised view of the last forward (:301-321); node-level InfoNCE, cluster-level contrast over the
batch and the KL term are the reference's own torch expressions on [B, B] / [N, D].

Same constructor, parameters in the reference's creation order, same attributes (`user_emb`, `user_emb_sub1/2`, `mean`,
`std`, `user_2cluster` -- a [n, 1] column, as faiss returns it: the cluster mask compares it with its transpose, :224-229).
The training loop calls forward() and e_step() before every batch's loss() (train_and_evaluate.py:116-125), loss() itself
runs no forward.  Differences: the adjacency is built vectorised (graph.binary_sym_norm_csr); `e_step()` -- faiss k-means on
the CPU there (:89-118) -- is NCL's seeded Lloyd iteration in torch on the embeddings' device (faiss is not in this image and
the clustering is random there too: its contract is kept); the two Gaussian draws come from the device generator
(`noise_fn` replays stored draws in the golden test)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import graph, ops, ranking, sparse


class VGCL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, dim_E, reg_weight, n_layers, ssl_temp,
                 ssl_alpha, device):
        super(VGCL, self).__init__()
        self.num_user, self.num_item = num_user, num_item
        self.edge_index, self.user_item_dict = edge_index, user_item_dict
        self.dim_E, self.reg_weight, self.n_layers, self.device = dim_E, reg_weight, n_layers, device
        self.alpha, self.beta = ssl_alpha, 1
        self.temp_node, self.temp_cluster = ssl_temp, 0.7 * ssl_temp
        self.num_user_cluster = self.num_item_cluster = 50
        self.user_embedding = nn.Embedding(num_embeddings=num_user, embedding_dim=dim_E)
        self.item_embedding = nn.Embedding(num_embeddings=num_item, embedding_dim=dim_E)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        e = torch.as_tensor(edge_index).long()
        self.adj_matrix = graph.binary_sym_norm_csr(e[:, 0], e[:, 1] - num_user, num_user, num_item).to(device)
        self.eps_weight = nn.Parameter(torch.randn(dim_E, dim_E))
        nn.init.xavier_uniform_(self.eps_weight)
        self.eps_bias = nn.Parameter(torch.zeros(dim_E))
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.user_emb = self.item_emb = self.mean = self.std = None
        self.user_centroids = self.user_2cluster = self.item_centroids = self.item_2cluster = None
        self.noise_fn = None
        self._kmeans_seed = 0

    # ---- per-batch clustering (:89-118) -----------------------------------------------------------------------------------
    def e_step(self):
        self.user_centroids, self.user_2cluster = self.run_kmeans(self.user_emb.detach(), self.num_user_cluster)
        self.item_centroids, self.item_2cluster = self.run_kmeans(self.item_emb.detach(), self.num_item_cluster)

    def run_kmeans(self, x, num_cluster, iters=20):
        """-> (L2-normalised centroids [k, D], nearest centroid per row as a [n, 1] column)."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        k = min(num_cluster, x.shape[0])
        g = torch.Generator(device="cpu").manual_seed(self._kmeans_seed)
        self._kmeans_seed += 1
        c = x[torch.randperm(x.shape[0], generator=g)[:k].to(x.device)].clone()
        for _ in range(iters):
            a = torch.cdist(x, c).argmin(1)
            s = torch.zeros_like(c).index_add_(0, a, x)
            n = torch.bincount(a, minlength=k).to(x.dtype).unsqueeze(1)
            c = torch.where(n > 0, s / n.clamp(min=1), c)
        return F.normalize(c, p=2, dim=1), torch.cdist(x, c).argmin(1).unsqueeze(1)

    # ---- hot path (:120-152) ----------------------------------------------------------------------------------------------
    def graph_encoder(self):
        ego_emb = torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)
        all_emb = []
        for _ in range(self.n_layers):
            ego_emb = sparse.mm(self.adj_matrix, ego_emb)
            all_emb.append(ego_emb)
        mean = torch.mean(torch.stack(all_emb), dim=0)
        std = torch.exp(ops.linear(mean, self.eps_weight.t().contiguous(), self.eps_bias))
        draw = self.noise_fn if self.noise_fn is not None else torch.randn_like
        noise1, noise2 = draw(std), draw(std)
        return mean + 0.01 * std * noise1, mean + 0.01 * std * noise2, mean, std

    def forward(self):
        noised_emb1, noised_emb2, self.mean, self.std = self.graph_encoder()
        self.user_emb, self.item_emb = torch.split(noised_emb1, [self.num_user, self.num_item], dim=0)
        self.user_emb_sub1, self.item_emb_sub1 = self.user_emb, self.item_emb
        self.user_emb_sub2, self.item_emb_sub2 = torch.split(noised_emb2, [self.num_user, self.num_item], dim=0)

    # ---- losses (:154-280) ------------------------------------------------------------------------------------------------
    def bpr_loss(self, users, pos_items, neg_items):
        u, p, n = self.user_emb[users], self.item_emb[pos_items], self.item_emb[neg_items]
        return -torch.mean(torch.log(torch.sigmoid(torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)) + 1e-5))

    def regularization_loss(self, users, pos_items, neg_items):
        return self.reg_weight * (torch.mean(self.user_embedding.weight[users] ** 2) + torch.mean(self.item_embedding.weight[pos_items] ** 2)
                                  + torch.mean(self.item_embedding.weight[neg_items] ** 2))

    def _node_nce(self, a, b):
        a, b = F.normalize(a, p=2, dim=1), F.normalize(b, p=2, dim=1)
        pos = torch.exp((a * b).sum(dim=1) / self.temp_node)
        ttl = torch.exp(torch.matmul(a, b.T) / self.temp_node).sum(dim=1)
        return -torch.mean(torch.log(pos / ttl))

    def compute_cl_loss_node(self, users, pos_items):
        return self.alpha * (self._node_nce(self.user_emb_sub1[users], self.user_emb_sub2[users])
                             + self._node_nce(self.item_emb_sub1[pos_items], self.item_emb_sub2[pos_items]))

    def _cluster_nce(self, ids, node2cluster, emb1, emb2):
        cluster_id = F.embedding(ids, node2cluster)                                 # [B, 1]
        mask = (cluster_id == cluster_id.transpose(0, 1)).float()
        a, b = F.normalize(F.embedding(ids, emb1), p=2, dim=1), F.normalize(F.embedding(ids, emb2), p=2, dim=1)
        logit = torch.matmul(a, b.transpose(0, 1)) / self.temp_cluster
        exp_logit = torch.exp(logit - logit.max(dim=1, keepdim=True).values)
        probs = (exp_logit / exp_logit.sum(dim=1, keepdim=True) * mask).sum(dim=1) / mask.sum(dim=1)
        return -torch.mean(torch.log(probs))

    def compute_cl_loss_cluster(self, users, pos_items):
        if self.user_2cluster is None or self.item_2cluster is None:
            raise RuntimeError("user_2cluster or item_2cluster is None. Please ensure e_step is called before the loss.")
        return self.alpha * (self._cluster_nce(users, self.user_2cluster, self.user_emb_sub1, self.user_emb_sub2)
                             + self._cluster_nce(pos_items, self.item_2cluster, self.item_emb_sub1, self.item_emb_sub2))

    def kl_regularizer(self, mean, std):
        regu_loss = -0.5 * (1 + 2 * std - mean.pow(2) - std.exp().pow(2))
        return regu_loss.sum(1).mean() / 1024

    def loss(self, users, pos_items, neg_items):
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        return (self.bpr_loss(users, pos_items, neg_items) + self.regularization_loss(users, pos_items, neg_items)
                + self.compute_cl_loss_node(users, pos_items) + self.compute_cl_loss_cluster(users, pos_items)
                + self.kl_regularizer(self.mean, self.std) * self.beta)

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:301-321: the first noised view of the last forward, history at 1e-6."""
        res = torch.cat((self.user_emb.detach(), self.item_emb.detach()), 0)
        return ranking.gene_ranklist(res, self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
