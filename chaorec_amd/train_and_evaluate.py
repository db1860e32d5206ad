"""Train / evaluate loop for the hot-path models (reference train_and_evaluate.py:25-48, :516-520, :523-686).

Kept: the generic (users, pos, neg) branch, the MMGCN (user_tensor, item_tensor) branch, FREEDOM's
pre_epoch_processing hook, per-epoch gene_ranklist + val/test metrics at K in topk, early stopping on test
Recall@max(topk) with patience 20, the same log lines.  Changed: the per-batch `loss.item()` host sync
(reference :48) becomes one device-side accumulation and a single .item() per epoch."""
import logging
import os

import torch

from .utils import EarlyStopping, EvalLists, gene_metrics, gene_metrics_device

MMGCN_STYLE = ("MMGCN", "GRCN")
PRE_EPOCH = ("FREEDOM", "LayerGCN", "POWERec")     # reference train_and_evaluate.py:554
E_STEP = ("NCL", "VGCL")                # reference train_and_evaluate.py:107-114, :116-125
BUILD_FIRST = ("LATTICE", "MICRO")                # reference train_and_evaluate.py:96-103: the first batch of an epoch rebuilds the item graphs
# eager launches: clustering with host decisions inside the step (NCL, VGCL), an operand whose structure is rebuilt per epoch
# (LATTICE, MICRO), launches that hipGraph capture refuses (hipErrorStreamCaptureUnsupported inside the per-step view construction: SGL, MMGCL), a HOST draw per step (SelfCF's dropout
# rate is np.random.random(), Model/SelfCF.py:55: a capture would freeze it), host-built sequence batches (LightGT).
# SimGCL / XSimGCL / SLMRec draw only with rand_like on the device generator, which a captured step advances per replay:
# they are captured (round 5: 2 x the eager epoch rate at baby size, tools/capture_family_probe.py).
NO_CAPTURE = ("NCL", "VGCL", "LATTICE", "MICRO", "SGL", "SelfCF", "MMGCL", "LightGT", "MMSSL")    # (MMSSL: two optimizers per batch, its own loop)


def _train_epoch_in_launch(model, loader, optimizer, graphed):
    """One epoch with the batches drawn INSIDE the fused BPR forward (models with loss_drawn): the loader only lays
    down the epoch's permutation; every full batch is one argument-less graph replay, the short last batch (the
    reference's DataLoader keeps it, drop_last=False) one eager step on the same counters."""
    loader.begin_epoch()
    E, B = loader.edges.shape[0], loader.batch_size
    sum_loss = None
    if getattr(graphed, "fused", False):
        # optim.FusedLightGCNStep: several steps per replay, the per-batch losses summed on the device by the step itself
        graphed.loss_accum.zero_()
        # (a step with the light forward leaves the whole propagated table behind only when asked to: the epoch's last
        # step does, unless the short last batch below -- an ordinary forward -- comes after it)
        graphed.run(E // B, full_last=(E % B == 0))
        sum_loss = graphed.loss_accum[0].clone()
    else:
        for _ in range(E // B):
            d = graphed()
            sum_loss = d.clone() if sum_loss is None else sum_loss.add_(d)
    tail = E - (E // B) * B
    if tail:
        optimizer.zero_grad()
        loss = model.loss_drawn(loader.edges, tail, loader.seed, 0, step_dev=loader.step_dev, advance=True,
                                perm=loader.perm, perm_pos=loader.perm_pos)
        loss.backward()
        optimizer.step()
        d = loss.detach()
        sum_loss = d.clone() if sum_loss is None else sum_loss.add_(d)
    loader.global_step += len(loader)
    return float(sum_loss.item()) if sum_loss is not None else 0.0


def _train_epoch_mmssl(model, train_loader, optimizer):
    """reference train_and_evaluate.py:49-71: per batch a discriminator step (Adam, lr 3e-4, betas (0.5, 0.9)) and a generator step
    (AdamW over ALL of model.parameters(), the discriminator's included, at the run's learning rate); both optimizers are
    created anew every epoch there (their moments restart), the optimizer main() built is not used."""
    optim_D = torch.optim.Adam(model.D.parameters(), lr=3e-4, betas=(0.5, 0.9))
    optimizer_D = torch.optim.AdamW([{'params': model.parameters()}], lr=optimizer.param_groups[0]['lr'])
    sum_loss = None
    for idx, (users, pos_items, neg_items) in enumerate(train_loader):
        optim_D.zero_grad()
        loss_D = model.loss_D(users, pos_items, neg_items)
        loss_D.backward()
        optim_D.step()
        optimizer_D.zero_grad()
        batch_loss = model.loss(users, pos_items, neg_items, idx)
        batch_loss.backward(retain_graph=False)
        optimizer_D.step()
        d = (loss_D + batch_loss).detach()
        sum_loss = d.clone() if sum_loss is None else sum_loss.add_(d)
    return float(sum_loss.item()) if sum_loss is not None else 0.0


def train(model, train_loader, optimizer, model_name="LightGCN", graphed=None):
    """One epoch.  `graphed` (optim.GraphedTrainStep) replays the captured step for full-size batches."""
    model.train()
    if model_name == "MMSSL":
        return _train_epoch_mmssl(model, train_loader, optimizer)
    if graphed is not None and getattr(graphed, "draws_in_launch", False):
        return _train_epoch_in_launch(model, train_loader, optimizer, graphed)
    sum_loss = None
    build_item_graph = True
    for batch in train_loader:
        if graphed is not None:
            d = graphed(*batch)
        else:
            optimizer.zero_grad()
            if model_name in E_STEP:
                if model_name == "VGCL":
                    model.forward()          # (train_and_evaluate.py:120: VGCL clusters the noised view of THIS forward; its loss() runs none)
                model.e_step()               # NCL clusters its embeddings before EVERY batch (train_and_evaluate.py:107-114)
            if model_name in BUILD_FIRST:
                loss = model.loss(*batch, build_item_graph=build_item_graph)
                build_item_graph = False
            else:
                loss = model.loss(*batch)
            loss.backward()
            optimizer.step()
            d = loss.detach()
        sum_loss = d.clone() if sum_loss is None else sum_loss.add_(d)
    return float(sum_loss.item()) if sum_loss is not None else 0.0


def evaluate(model, data, ranklist, topk):
    """train_and_evaluate.py:655-659.  `data` may be an EvalLists and `ranklist` a device tensor: the metrics are then
    computed in HBM (chaorec_rank_metrics_f64), otherwise by the host restatement."""
    model.eval()
    with torch.no_grad():
        if isinstance(data, EvalLists):
            return gene_metrics_device(data, ranklist, topk)
        return gene_metrics(data, ranklist, topk)


def _log_metrics(title, metrics):
    logging.info(title)
    for k, m in metrics.items():
        logging.info(f"{k}: {' | '.join(f'{name}: {value:.5f}' for name, value in m.items())}")


def _capture_step(model, train_loader, optimizer, model_name):
    """The launch-bound inner loop (train_and_evaluate.py:43-48) as one hipGraph replay per batch: possible when the
    batches already live in HBM with a fixed shape, the optimizer's step counter is on the device and the graph
    graph arrays keep their addresses between epochs (FREEDOM re-prunes its adjacency every epoch IN PLACE, so its
    step is captured once, after the first pre_epoch_processing())."""
    from .dataload import DeviceBatchSampler
    from .optim import FusedAdam, GraphedTrainStep
    if not isinstance(train_loader, DeviceBatchSampler) or not isinstance(optimizer, FusedAdam):
        return None
    if model_name in NO_CAPTURE and model_name not in os.environ.get("CHAOREC_TRY_CAPTURE", "").split(","):
        return None
    if model_name in PRE_EPOCH and not getattr(model, "prunes_in_place", False):
        return None          # a graph that is re-allocated every epoch cannot sit behind captured addresses
    if len(train_loader) < 2:
        return None
    if hasattr(model, "loss_drawn") and model_name not in MMGCN_STYLE and model_name not in PRE_EPOCH:
        # the batch is drawn by the fused BPR forward itself: a replay takes no inputs at all.  The capture's warm-up
        # steps draw full batches from the epoch permutation: the edge list must hold them (the kernel clamps a read
        # past the end, but those would not be the sampler's batches), and the loader's counters / generator are put
        # back afterwards so that a captured run draws exactly what an eager one draws.
        E, B = train_loader.edges.shape[0], train_loader.batch_size
        if E < 4 * B:
            return None
        gen_state = train_loader.gen.get_state()
        train_loader.begin_epoch()
        saved = [t.clone() for t in (train_loader.step_dev, train_loader.perm_pos, train_loader.perm)]
        from .Model import LightGCN
        if type(model) is LightGCN and model.n_layers >= 1 and len(optimizer.param_groups) == 1:
            from .optim import FusedLightGCNStep
            acc = torch.zeros(1, dtype=torch.float32, device=train_loader.edges.device)
            # steps per replay: a divisor of the steps that run as whole replays.  With the light forward (by size: large
            # graphs) FusedLightGCNStep.run(E // B, full_last=True) runs E // B - 1 light steps and one full step, so the
            # divisor is taken of E // B - 1 (a divisor of E // B left k - 1 steps per epoch to single replays)
            light = FusedLightGCNStep.frontier_modes(model.num_user + model.num_item, model.n_layers,
                                                     model.user_embedding.weight.shape[1])[2]
            n_rep = E // B - 1 if (light and E % B == 0) else E // B
            k = next((c for c in (11, 10, 8, 7, 5, 4, 3, 2) if n_rep % c == 0), 1)
            g = FusedLightGCNStep(model, optimizer, batch_size=B, edges=train_loader.edges, seed=train_loader.seed,
                                  step_dev=train_loader.step_dev, perm=train_loader.perm, perm_pos=train_loader.perm_pos,
                                  loss_accum=acc, steps_per_replay=k)
            g.fused = True
        else:
            g = GraphedTrainStep(model, optimizer, batch_fn=lambda: (), loss_fn=train_loader.drawn_loss_fn(model))
        with torch.no_grad():
            for dst, src in zip((train_loader.step_dev, train_loader.perm_pos, train_loader.perm), saved):
                dst.copy_(src)
        train_loader.gen.set_state(gen_state)
        g.draws_in_launch = True
        return g
    example = next(iter(train_loader))
    return GraphedTrainStep(model, optimizer, example_batch=example)


def train_and_evaluate(model, train_loader, val_data, test_data, optimizer, epochs, model_name="LightGCN",
                       topk=(5, 10, 20), patience=20, graph=True, history=None):
    """history (optional list): receives one dict per epoch -- epoch, loss (the summed batch losses, what the reference
    logs as "Epoch n, Loss"), val and test metrics -- the numbers the log lines carry, for callers that compare runs."""
    model.train()
    graphed, capture_pending = None, bool(graph)
    early_stopping = EarlyStopping(patience=patience, verbose=True)
    topk = [int(k) for k in topk]
    # evaluation stays on the device when the model can hand the rank list over in HBM
    # (gene_ranklist returns 50 columns, as the reference's does: larger cut-offs take the host path, which truncates)
    on_device = getattr(model, "device", None) is not None and torch.device(model.device).type == "cuda" and max(topk) <= 50
    if on_device:
        val_data, test_data = EvalLists(val_data, model.device), EvalLists(test_data, model.device)
    for epoch in range(epochs):
        if model_name in PRE_EPOCH:
            model.pre_epoch_processing()
            if graphed is not None and getattr(model, "graph_generation", 0) != getattr(graphed, "_graph_generation", 0):
                graphed, capture_pending = None, bool(graph)     # the pruned graph was re-allocated: capture again
        if capture_pending:
            graphed, capture_pending = _capture_step(model, train_loader, optimizer, model_name), False
        loss = train(model, train_loader, optimizer, model_name, graphed)
        logging.info("Epoch {}, Loss: {:.5f}".format(epoch + 1, loss))

        model.eval()
        if hasattr(optimizer, "flush"):
            # lazily updated feature rows (optim.FusedAdam lazy_rows) are brought up to the current step at every epoch's end:
            # whoever reads a claimed table between epochs -- model.state_dict(), a checkpoint, a per-epoch kNN rebuild -- sees
            # current rows (one launch per claimed table; the rows' values are the eager ones bit for bit either way)
            optimizer.flush()
        rank_list = model.gene_ranklist(to_cpu=False) if on_device else model.gene_ranklist()
        val_metrics = evaluate(model, val_data, rank_list, topk)
        test_metrics = evaluate(model, test_data, rank_list, topk)
        _log_metrics('Validation Metrics:', val_metrics)
        _log_metrics('Test Metrics:', test_metrics)
        if history is not None:
            history.append({"epoch": epoch + 1, "loss": loss, "val": val_metrics, "test": test_metrics})

        recall = test_metrics[max(topk)]['recall']
        early_stopping(recall, test_metrics)
        if early_stopping.early_stop:
            print("Early stopping")
            break

    if hasattr(optimizer, "flush"):
        optimizer.flush()          # lazily updated feature rows (optim.FusedAdam lazy_rows): current before anyone reads them
    best_metrics = early_stopping.best_metrics
    _log_metrics('Best Test Metrics:', best_metrics)
    return best_metrics
