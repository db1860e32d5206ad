"""MMSSL with the reference's surface (Model/MMSSL.py:21-657) -- a generator / discriminator pair: the GENERATOR propagates
projected modality features and the ids over row-normalised user-item graphs (one shared, two modality-specific ones that the
step itself rewires from its top-k predictions), mixes the two modality views with a small multi-head attention and runs
LightGCN-style layers on top; the DISCRIMINATOR (a dense MLP over [2 B, I] score rows) is trained against it with a gradient
penalty.

On the hot path here: every product with a graph (:302-315, :341-347 -- eight per modality layer and two per interaction layer,
over rectangular [U, I] / [I, U] operands) is `chaorec_amd.sparse.mm` on the HIP SpMM, the feature projections are `ops.linear`
on the MFMA GEMM, the ranking is `ranking.gene_ranklist` over a fresh forward's tables (:626-657).  The discriminator stays
plain torch modules -- its gradient penalty differentiates THROUGH a backward (`create_graph=True`, :205-213), which needs
double-differentiable layers -- and so do the [B, I] score rows it is fed (library GEMMs).  What the reference does on the host
per batch is done on the device: the batch users' interaction rows (scipy slicing + todense + a copy there, :447,512) are
scattered from the history CSR, the rewired modality graphs (Python lists + scipy, :559-600) are built from the top-k index
tensors.

Kept quirks: the modality graphs are rebuilt from what the PREVIOUS batch collected and the lists are emptied in the same
branch, so with T = 1 every batch after the second one propagates the modality ids over EMPTY graphs (:559-587); with fewer
than 10 000 items int(I * m_topk_rate) is 0 and nothing is ever collected; `model.parameters()` -- what the loop hands its
AdamW -- includes the discriminator.  Hooks: `uniform_fn(shape)` / `alpha_fn(n)` replay the two host draws of loss_D."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import autograd, nn

from .. import graph, ops, ranking, sparse


class Discriminator(nn.Module):
    """:21-44."""

    def __init__(self, dim):
        super(Discriminator, self).__init__()
        self.G_drop1, self.G_drop2 = 0.31, 0.5
        self.net = nn.Sequential(
            nn.Linear(dim, int(dim / 4)), nn.LeakyReLU(True), nn.BatchNorm1d(int(dim / 4)), nn.Dropout(self.G_drop1),
            nn.Linear(int(dim / 4), int(dim / 8)), nn.LeakyReLU(True), nn.BatchNorm1d(int(dim / 8)), nn.Dropout(self.G_drop2),
            nn.Linear(int(dim / 8), 1), nn.Sigmoid())

    def forward(self, x):
        return (100 * self.net(x.float())).view(-1)


def _row_mean_graph(rows, cols, n_rows, n_cols, device):
    """csr_norm(mean_flag=True) (:176-190) of the count matrix of the listed (row, col) pairs: (row sum + 1e-8)^-1/2 on the
    rows only.  -> graph.CSR [n_rows, n_cols], or None for an empty list (an all-zero operand)."""
    if rows.numel() == 0:
        return None
    key, cnt = torch.unique(rows.long() * n_cols + cols.long(), return_counts=True)
    r, c, w = torch.div(key, n_cols, rounding_mode="floor"), key % n_cols, cnt.to(torch.float32)
    rowsum = torch.zeros(n_rows, dtype=torch.float32, device=w.device).index_add_(0, r, w)
    d = torch.pow(rowsum + 1e-8, -0.5)
    return graph.coo_to_csr_coalesced(r, c, d[r] * w, n_rows, n_cols).to(device)


class MMSSL(nn.Module):
    def __init__(self, num_user, num_item, edge_index, user_item_dict, v_feat, t_feat, dim_E,
                 reg_weight, ssl_alpha, ssl_temp, G_rate, mmlayer, device):
        super(MMSSL, self).__init__()
        self.num_user, self.num_item, self.dim_E = num_user, num_item, dim_E
        self.weight_size = [dim_E] + [64] * mmlayer
        self.n_ui_layers = mmlayer
        self.device, self.mmlayer, self.user_item_dict = device, mmlayer, user_item_dict
        self.reg_weight, self.tau, self.feat_reg_decay = reg_weight, ssl_temp, 1e-5
        self.gene_u, self.gene_real, self.gene_fake = None, None, {}
        self.log_log_scale, self.real_data_tau, self.ui_pre_scale = 0.00001, 0.005, 100
        self.gp_rate, self.T, self.m_topk_rate = 1, 1, 0.0001
        self.cl_rate, self.G_rate = ssl_alpha, G_rate

        e = torch.as_tensor(np.asarray(edge_index)).long()
        u, i = e[:, 0], e[:, 1] - num_user
        self.ui_graph = _row_mean_graph(u, i, num_user, num_item, device)
        self.iu_graph = _row_mean_graph(i, u, num_item, num_user, device)
        self.image_ui_graph = self.text_ui_graph = self.ui_graph
        self.image_iu_graph = self.text_iu_graph = self.iu_graph
        self.image_ui_index, self.text_ui_index = {'x': [], 'y': []}, {'x': [], 'y': []}
        # the raw count rows (ui_graph_raw, :80): a CSR of the distinct pairs with their multiplicities, read per batch
        key, cnt = torch.unique(u * num_item + i, return_counts=True)
        self._raw = graph.coo_to_csr_coalesced(torch.div(key, num_item, rounding_mode="floor"), key % num_item, cnt.to(torch.float32),
                                               num_user, num_item).to(device)

        # (drawn on the HOST generator like every other model here -- the reference moves D to its device before the kaiming
        #  draw, :96-97 --, then moved)
        self.D = Discriminator(num_item)
        self.D.apply(self.weights_init)
        self.D = self.D.to(device)
        self.image_trans = nn.Linear(v_feat.shape[1], dim_E)
        self.text_trans = nn.Linear(t_feat.shape[1], dim_E)
        nn.init.xavier_uniform_(self.image_trans.weight)
        nn.init.xavier_uniform_(self.text_trans.weight)
        self.encoder = nn.ModuleDict()
        self.encoder['image_encoder'], self.encoder['text_encoder'] = self.image_trans, self.text_trans
        self.common_trans = nn.Linear(dim_E, dim_E)
        nn.init.xavier_uniform_(self.common_trans.weight)
        self.align = nn.ModuleDict()
        self.align['common_trans'] = self.common_trans
        self.user_id_embedding = nn.Embedding(num_user, dim_E)
        self.item_id_embedding = nn.Embedding(num_item, dim_E)
        nn.init.xavier_uniform_(self.user_id_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        self.register_buffer("image_feats", v_feat.clone(), persistent=False)
        self.register_buffer("text_feats", t_feat.clone(), persistent=False)
        self.softmax, self.act, self.sigmoid = nn.Softmax(dim=-1), nn.Sigmoid(), nn.Sigmoid()
        self.dropout = nn.Dropout(p=0.2)
        self.batch_norm = nn.BatchNorm1d(dim_E)
        self.head_num = 4
        initializer = nn.init.xavier_uniform_
        self.weight_dict = nn.ParameterDict({
            'w_q': nn.Parameter(initializer(torch.empty([dim_E, dim_E]))),
            'w_k': nn.Parameter(initializer(torch.empty([dim_E, dim_E]))),
            'w_v': nn.Parameter(initializer(torch.empty([dim_E, dim_E]))),
            'w_self_attention_item': nn.Parameter(initializer(torch.empty([dim_E, dim_E]))),
            'w_self_attention_user': nn.Parameter(initializer(torch.empty([dim_E, dim_E]))),
            'w_self_attention_cat': nn.Parameter(initializer(torch.empty([self.head_num * dim_E, dim_E]))),
        })
        self.embedding_dict = {'user': {}, 'item': {}}
        self.sparse, self.model_cat_rate, self.id_cat_rate = 1, 0.55, 0.36
        self.hist = ranking.history_csr(user_item_dict, num_user, device)
        self.uniform_fn = self.alpha_fn = None

    # ---- helpers --------------------------------------------------------------------------------------------------------------
    def mm(self, x, y):
        """:160-164."""
        return sparse.mm(x, y)

    def _mm(self, g, n_rows, y):
        """mm with an operand that may be EMPTY (a modality graph rewired from emptied lists: None): all-zero rows."""
        return y.new_zeros((n_rows, y.shape[1])) if g is None else self.mm(g, y)

    def sim(self, z1, z2):
        return torch.mm(F.normalize(z1), F.normalize(z2).t())

    def weights_init(self, m):
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight)
            m.bias.data.fill_(0)

    def gradient_penalty(self, D, xr, xf):
        """:193-215."""
        LAMBDA = 0.3
        xf, xr = xf.detach(), xr.detach()
        alpha = self.alpha_fn(xr.size(0)).to(xr.device) if self.alpha_fn is not None else torch.rand(xr.size(0), 1).to(xr.device)
        alpha = alpha.expand_as(xr)
        interpolates = alpha * xr + ((1 - alpha) * xf)
        interpolates.requires_grad_()
        disc_interpolates = D(interpolates)
        gradients = autograd.grad(outputs=disc_interpolates, inputs=interpolates, grad_outputs=torch.ones_like(disc_interpolates),
                                  create_graph=True, retain_graph=True, only_inputs=True)[0]
        return ((gradients.norm(2, dim=1) - 1) ** 2).mean() * LAMBDA

    def multi_head_self_attention(self, trans_w, embedding_t_1, embedding_t):
        """:247-287 (the normalised value of :286 is computed and dropped there; Z goes out as it is)."""
        q = torch.stack([embedding_t[k] for k in embedding_t.keys()], dim=0)
        v = k = torch.stack([embedding_t_1[key] for key in embedding_t_1.keys()], dim=0)
        beh, N, d_h = q.shape[0], q.shape[1], self.dim_E / self.head_num
        Q = torch.matmul(q, trans_w['w_q']).reshape(beh, N, self.head_num, int(d_h)).permute(2, 0, 1, 3)
        K = torch.matmul(k, trans_w['w_k']).reshape(beh, N, self.head_num, int(d_h)).permute(2, 0, 1, 3)
        Q, K, V = torch.unsqueeze(Q, 2), torch.unsqueeze(K, 1), torch.unsqueeze(v, 1)
        att = torch.sum(torch.mul(Q, K) / torch.sqrt(torch.tensor(d_h)), dim=-1)
        att = F.softmax(torch.unsqueeze(att, dim=-1), dim=2)
        Z = torch.sum(torch.mul(att, V), dim=2)
        Z = torch.cat([value for value in Z], -1)
        Z = torch.matmul(Z, self.weight_dict['w_self_attention_cat'])
        return Z, att.detach()

    # ---- :289-365 -------------------------------------------------------------------------------------------------------------
    def forward(self, ui_graph, iu_graph, image_ui_graph, image_iu_graph, text_ui_graph, text_iu_graph):
        U, I = self.num_user, self.num_item
        image_feats = image_item_feats = self.dropout(ops.linear(self.image_feats, self.image_trans.weight, self.image_trans.bias))
        text_feats = text_item_feats = self.dropout(ops.linear(self.text_feats, self.text_trans.weight, self.text_trans.bias))
        image_user_id = text_user_id = image_item_id = text_item_id = image_user_feats = text_user_feats = None
        for _ in range(self.mmlayer):
            image_user_feats = self._mm(ui_graph, U, image_feats)
            image_item_feats = self._mm(iu_graph, I, image_user_feats)
            image_user_id = self._mm(image_ui_graph, U, self.item_id_embedding.weight)
            image_item_id = self._mm(image_iu_graph, I, self.user_id_embedding.weight)
            text_user_feats = self._mm(ui_graph, U, text_feats)
            text_item_feats = self._mm(iu_graph, I, text_user_feats)
            text_user_id = self._mm(text_ui_graph, U, self.item_id_embedding.weight)
            text_item_id = self._mm(text_iu_graph, I, self.user_id_embedding.weight)
        self.embedding_dict['user']['image'], self.embedding_dict['user']['text'] = image_user_id, text_user_id
        self.embedding_dict['item']['image'], self.embedding_dict['item']['text'] = image_item_id, text_item_id
        user_z, _ = self.multi_head_self_attention(self.weight_dict, self.embedding_dict['user'], self.embedding_dict['user'])
        item_z, _ = self.multi_head_self_attention(self.weight_dict, self.embedding_dict['item'], self.embedding_dict['item'])
        u_g = self.user_id_embedding.weight + self.id_cat_rate * F.normalize(user_z.mean(0), p=2, dim=1)
        i_g = self.item_id_embedding.weight + self.id_cat_rate * F.normalize(item_z.mean(0), p=2, dim=1)
        user_emb_list, item_emb_list = [u_g], [i_g]
        for i in range(self.n_ui_layers):
            if i == (self.n_ui_layers - 1):
                u_g = self.softmax(self._mm(ui_graph, U, i_g))
                i_g = self.softmax(self._mm(iu_graph, I, u_g))
            else:
                u_g = self._mm(ui_graph, U, i_g)
                i_g = self._mm(iu_graph, I, u_g)
            user_emb_list.append(u_g)
            item_emb_list.append(i_g)
        u_g = torch.mean(torch.stack(user_emb_list), dim=0)
        i_g = torch.mean(torch.stack(item_emb_list), dim=0)
        u_g = u_g + self.model_cat_rate * F.normalize(image_user_feats, p=2, dim=1) + self.model_cat_rate * F.normalize(text_user_feats, p=2, dim=1)
        i_g = i_g + self.model_cat_rate * F.normalize(image_item_feats, p=2, dim=1) + self.model_cat_rate * F.normalize(text_item_feats, p=2, dim=1)
        return (u_g, i_g, image_item_feats, text_item_feats, image_user_feats, text_user_feats, u_g, i_g,
                image_user_id, text_user_id, image_item_id, text_item_id)

    def _forward_now(self):
        return self.forward(self.ui_graph, self.iu_graph, self.image_ui_graph, self.image_iu_graph, self.text_ui_graph, self.text_iu_graph)

    # ---- losses ---------------------------------------------------------------------------------------------------------------
    def batched_contrastive_loss(self, z1, z2, batch_size=1024):
        """:367-413."""
        num_nodes = z1.size(0)
        num_batches = (num_nodes - 1) // batch_size + 1
        f = lambda x: torch.exp(x / self.tau)
        losses = []
        for i in range(num_batches):
            lo, hi = i * batch_size, (i + 1) * batch_size
            refl_sim = torch.cat([f(self.sim(z1[lo:hi], z1[j * batch_size:(j + 1) * batch_size])) for j in range(num_batches)], dim=-1)
            between_sim = torch.cat([f(self.sim(z1[lo:hi], z2[j * batch_size:(j + 1) * batch_size])) for j in range(num_batches)], dim=-1)
            losses.append(-torch.log(between_sim[:, lo:hi].diag() / (refl_sim.sum(1) + between_sim.sum(1) - refl_sim[:, lo:hi].diag()) + 1e-8))
        return torch.cat(losses).mean()

    def feat_reg_loss_calculation(self, g_item_image, g_item_text, g_user_image, g_user_text):
        feat_reg = 1. / 2 * (g_item_image ** 2).sum() + 1. / 2 * (g_item_text ** 2).sum() \
            + 1. / 2 * (g_user_image ** 2).sum() + 1. / 2 * (g_user_text ** 2).sum()
        return self.feat_reg_decay * (feat_reg / self.num_item)

    def _raw_rows(self, users):
        """The batch users' rows of the count matrix as a dense [B, I] tensor (:447,512), scattered on the device."""
        rp, col, val = self._raw.rowptr, self._raw.col.long(), self._raw.val
        start, cnt = rp[users], rp[users + 1] - rp[users]
        which = torch.repeat_interleave(torch.arange(users.numel(), device=users.device), cnt)
        pos = torch.arange(which.numel(), device=users.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt) + start[which]
        out = torch.zeros((users.numel(), self.num_item), dtype=torch.float32, device=users.device)
        out[which, col[pos]] = val[pos]
        return out

    def u_sim_calculation(self, users, user_final, item_final):
        """:443-465."""
        u_ui = self._raw_rows(users)
        sim_gt = torch.multiply(torch.mm(user_final[users], item_final.T), (1 - u_ui))
        return F.normalize(sim_gt, p=2, dim=1)

    def bpr_loss(self, users, pos_items, neg_items):
        pos_scores = torch.sum(torch.mul(users, pos_items), dim=1)
        neg_scores = torch.sum(torch.mul(users, neg_items), dim=1)
        regularizer = (1. / 2 * (users ** 2).sum() + 1. / 2 * (pos_items ** 2).sum() + 1. / 2 * (neg_items ** 2).sum()) / 1024
        return -torch.mean(F.logsigmoid(pos_scores - neg_scores)), self.reg_weight * regularizer, 0.0

    def loss_D(self, users, pos_items, neg_items):
        """:490-527."""
        users = users.to(self.device)
        with torch.no_grad():
            ua, ia, image_item_embeds, text_item_embeds, image_user_embeds, text_user_embeds, *_ = self._forward_now()
        ui_u_sim_detach = self.u_sim_calculation(users, ua, ia).detach()
        image_u_sim_detach = self.u_sim_calculation(users, image_user_embeds, image_item_embeds).detach()
        text_u_sim_detach = self.u_sim_calculation(users, text_user_embeds, text_item_embeds).detach()
        inputf = torch.cat((image_u_sim_detach, text_u_sim_detach), dim=0)
        lossf = self.D(inputf).mean()
        u_ui = self._raw_rows(users)
        shape = (u_ui.shape[0], u_ui.shape[1])
        noise = self.uniform_fn(shape).to(self.device) if self.uniform_fn is not None else torch.empty(shape, dtype=torch.float32).uniform_(0, 1).to(self.device)
        u_ui = F.softmax(u_ui - self.log_log_scale * torch.log(-torch.log(noise + 1e-8) + 1e-8) / self.real_data_tau, dim=1)
        u_ui += ui_u_sim_detach * self.ui_pre_scale
        u_ui = F.normalize(u_ui, dim=1)
        inputr = torch.cat((u_ui, u_ui), dim=0)
        lossr = -(self.D(inputr).mean())
        gp = self.gradient_penalty(self.D, inputr, inputf.detach())
        return lossr + lossf + self.gp_rate * gp

    def loss(self, users, pos_items, neg_items, idx):
        """:529-624."""
        pos_items, neg_items = pos_items - self.num_user, neg_items - self.num_user
        users, pos_items, neg_items = users.to(self.device), pos_items.to(self.device), neg_items.to(self.device)
        (G_ua, G_ia, G_image_item_embeds, G_text_item_embeds, G_image_user_embeds, G_text_user_embeds, G_user_emb, _,
         G_image_user_id, G_text_user_id, _, _) = self._forward_now()
        G_batch_mf_loss, G_batch_emb_loss, G_batch_reg_loss = self.bpr_loss(G_ua[users], G_ia[pos_items], G_ia[neg_items])
        G_image_u_sim = self.u_sim_calculation(users, G_image_user_embeds, G_image_item_embeds)
        G_text_u_sim = self.u_sim_calculation(users, G_text_user_embeds, G_text_item_embeds)
        U, I, k = self.num_user, self.num_item, int(self.num_item * self.m_topk_rate)
        if idx % self.T == 0 and idx != 0:
            cat = lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=torch.int64, device=self.device)
            ix, iy = cat(self.image_ui_index['x']), cat(self.image_ui_index['y'])
            tx, ty = cat(self.text_ui_index['x']), cat(self.text_ui_index['y'])
            self.image_ui_graph, self.image_iu_graph = _row_mean_graph(ix, iy, U, I, self.device), _row_mean_graph(iy, ix, I, U, self.device)
            self.text_ui_graph, self.text_iu_graph = _row_mean_graph(tx, ty, U, I, self.device), _row_mean_graph(ty, tx, I, U, self.device)
            self.image_ui_index, self.text_ui_index = {'x': [], 'y': []}, {'x': [], 'y': []}
        else:
            for sim, store in ((G_image_u_sim.detach(), self.image_ui_index), (G_text_u_sim.detach(), self.text_ui_index)):
                _, ui_id = torch.topk(sim, k, dim=-1)
                store['x'].append(users.repeat(1, k).view(-1))             # (:593-594: the batch's user column tiled k times, in that order)
                store['y'].append(ui_id.reshape(-1))
        feat_emb_loss = self.feat_reg_loss_calculation(G_image_item_embeds, G_text_item_embeds, G_image_user_embeds, G_text_user_embeds)
        batch_contrastive_loss = self.batched_contrastive_loss(G_image_user_id[users], G_user_emb[users]) \
            + self.batched_contrastive_loss(G_text_user_id[users], G_user_emb[users])
        G_lossf = -(self.D(torch.cat((G_image_u_sim, G_text_u_sim), dim=0)).mean())
        return G_batch_mf_loss + G_batch_emb_loss + G_batch_reg_loss + feat_emb_loss + self.cl_rate * batch_contrastive_loss + self.G_rate * G_lossf

    def gene_ranklist(self, topk=50, to_cpu=True):
        """:626-657: a fresh forward's tables, history at 1e-6."""
        with torch.no_grad():
            ua, ia, *rest = self._forward_now()
        return ranking.gene_ranklist(torch.cat((ua, ia), 0), self.num_user, self.num_item, self.hist, 1e-6, topk, to_cpu=to_cpu,
                                      state=ranking.state_of(self))

    full_sort_predict = gene_ranklist
