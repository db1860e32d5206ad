#!/usr/bin/env python3
"""Step / epoch timing of the models on dataset-shaped synthetic graphs (BASELINE configs 2-4 shapes)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph, dataload
from chaorec_amd.Model import LightGCN, FREEDOM, MMGCN, NGCF, MGCN, LayerGCN, BPRMF, VBPR
from chaorec_amd.optim import FusedAdam
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["LightGCN:sports", "FREEDOM:clothing", "MMGCN:microlens"]
for spec in which:
    name, ds = spec.split(":")
    U, I, E = DATASET_SHAPES[ds]
    edges = synthetic_interactions(U, I, E, seed=42)
    uid = graph.user_item_dict_from_edges(edges)
    v_feat, t_feat = dataload.synthetic_features(I, ds)
    torch.manual_seed(0)
    t0 = time.time()
    if name == "LightGCN":
        m = LightGCN(U, I, edges, uid, 64, 1e-3, 3, "add", dev)
    elif name == "NGCF":
        m = NGCF(U, I, edges, uid, 64, 1e-3, 0.2, 3, "add", dev)
    elif name == "LayerGCN":
        m = LayerGCN(U, I, edges, uid, 64, 1e-3, 3, 0.1, dev)
    elif name == "MGCN":
        m = MGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, 2, "add", 0.2, 0.01, dev)
    elif name == "BPR":
        m = BPRMF(U, I, uid, 64, 1e-3, dev)
    elif name == "VBPR":
        m = VBPR(U, I, uid, v_feat, 64, 64, 1e-3, dev)
    elif name == "FREEDOM":
        m = FREEDOM(U, I, edges, uid, v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev)
    else:
        m = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev)
    m = m.to(dev)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    opt = FusedAdam(m.parameters(), lr=1e-3)
    sampler = dataload.DeviceBatchSampler(U, I, uid, edges, 1024, dev, name)
    if name in ("FREEDOM", "LayerGCN"):
        m.pre_epoch_processing()
    it = iter(sampler)
    batches = [next(it) for _ in range(12)]
    def step(b):
        opt.zero_grad()
        loss = m.loss(*b)
        loss.backward()
        opt.step()
    for b in batches[:2]:
        step(b)
    torch.cuda.synchronize()
    t0 = time.time()
    for b in batches[2:]:
        step(b)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 10 * 1e3
    t0 = time.time(); m.gene_ranklist(); torch.cuda.synchronize(); t_rank = time.time() - t0
    if getattr(opt, "_claimed", None):
        print(f"          feature tables claimed by FusedAdam (no dense [I,K] gradient), lazy_rows={opt.lazy_rows}")
    print(f"{name:9s} {ds:10s} feat=({v_feat.shape[1]},{t_feat.shape[1]}) build {t_build:6.2f} s  step {ms:8.2f} ms  "
          f"epoch({len(sampler)} batches) {ms * len(sampler) / 1e3:6.2f} s  gene_ranklist {t_rank * 1e3:7.2f} ms")
    # the captured step on its own (what train_and_evaluate replays per batch): zero_grad + loss + backward + Adam
    from chaorec_amd.optim import GraphedTrainStep
    if name != "LightGCN":
        gstep = GraphedTrainStep(m, opt, example_batch=batches[0])
        for b in batches[:3]:
            gstep(*b)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(100):
            gstep(*batches[i % len(batches)])
        torch.cuda.synchronize()
        print(f"          captured step (hipGraph replay, batch copied in): {(time.time() - t0) / 100 * 1e3:7.3f} ms")
    # the real loop of train_and_evaluate.train_and_evaluate: one epoch incl. evaluation and device metrics
    from chaorec_amd import train_and_evaluate as tae
    from chaorec_amd.synthetic import synthetic_eval_lists
    val, test = synthetic_eval_lists(U, I, edges, seed=1), synthetic_eval_lists(U, I, edges, seed=2)
    import logging
    logging.disable(logging.CRITICAL)
    def run(epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        tae.train_and_evaluate(m, sampler, val, test, opt, epochs, model_name=name, topk=(5, 10, 20), patience=100)
        torch.cuda.synchronize()
        return time.time() - t0
    t1, t6 = run(1), run(6)        # both include one capture: the difference is five steady-state epochs
    print(f"          train_and_evaluate: {(t6 - t1) / 5:6.3f} s per epoch (captured step + gene_ranklist + val/test metrics on "
          f"the device); first epoch incl. capture {t1:6.3f} s")
