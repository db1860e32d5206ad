"""--model MMGCN / FREEDOM, single and sharded."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import MIN_TIMED_S, flush_c_stdout, init_ranks  # noqa: F401


def measure_model(args, name, world, rank, dev, sharded, backend, steps=None, warmup=None, dataset=None):
    """MMGCN / FREEDOM: the model's train step (zero_grad -> loss -> backward -> [gradient exchange] -> FusedAdam,
    one captured hipGraph, batch drawn on the device) and gene_ranklist on the REAL interaction graph of its BASELINE
    config (microlens / clothing) with the seeded synthetic modality features.  Not sharded: the single-process model
    class.  Sharded: dist.ShardedMMGCN / dist.ShardedFREEDOM, weak scaling -- rank g owns one copy of the dataset's
    users over the shared item set, like the LightGCN path.  `value`: directed-edge messages per second through the
    step's SpMM launches (sum of nnz over every propagate, forward and backward, all ranks).  -> the record (dict)."""
    import torch.distributed as dist
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    dataset = dataset or {"MMGCN": "microlens", "FREEDOM": "clothing"}[name]
    from chaorec_amd import _lib, dataload, graph, ops
    from chaorec_amd import dist as cdist
    from chaorec_amd.Model import FREEDOM, MMGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    _lib.ensure_built()
    _lib.load()
    B = args.batch
    packed = dataload.packed_interactions(dataset)
    U1, I, edges1 = packed["num_user"], packed["num_item"], np.asarray(packed["train"], dtype=np.int64)
    U = U1 * world
    edges_all = np.concatenate([np.stack([edges1[:, 0] + k * U1, edges1[:, 1] - U1 + U], 1) for k in range(world)], 0)
    v_feat, t_feat = dataload.synthetic_features(I, dataset)
    torch.manual_seed(42)                       # every rank builds the same whole model, then keeps its shard of it
    t0 = time.perf_counter()
    uid = graph.user_item_dict_from_edges(edges_all)
    if name == "MMGCN":
        full = MMGCN(U, I, edges_all.astype(np.int32), uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
    else:
        full = FREEDOM(U, I, edges_all.astype(np.int32), uid, v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev).to(dev)
    bounds = [k * U1 for k in range(world + 1)]
    if not sharded:
        model, U_g, u0 = full, U, 0
        local = np.stack([edges_all[:, 0], edges_all[:, 1]], 1)                       # [user, item + U]
    elif name == "MMGCN":
        shard = cdist.UserShard(edges_all, U, I, world, rank, dev, self_loops=True)
        assert shard.bounds == bounds, (shard.bounds, bounds)
        model, U_g, u0 = cdist.ShardedMMGCN(full, shard, dev), shard.num_user_local, shard.u0
        local = shard.local_edges.astype(np.int64)                                    # [local user, item + U_g]
    else:
        model = cdist.ShardedFREEDOM(full, bounds, world, rank, dev)
        U_g, u0 = model.num_user, model.u0
        local = np.stack([model.local_edges[:, 0] - u0, model.local_edges[:, 1] - U + U_g], 1)
    if sharded:
        del full
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    if hasattr(model, "pre_epoch_processing"):
        model.pre_epoch_processing()            # FREEDOM: this epoch's pruned graph (the step below trains on it)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    edges_dev = torch.from_numpy(local).to(dev)
    hist = model.hist
    counter = torch.zeros(1, dtype=torch.int64, device=dev)

    def draw():
        counter.add_(1)
        # item ids as the reference's dataset hands them over (dataload.py:74-88): GLOBAL (item + num_user), added in the draw
        # launch; the sharded FREEDOM takes local ones
        glob = name == "MMGCN" or not sharded
        u, pos, neg = ops.draw_batch(edges_dev, hist, B, U_g, I, 42 + rank, 0, step_dev=counter, item_offset=U_g if glob else 0)
        if name == "MMGCN":                     # Model/MMGCN.py:188-202: [B, 2] user / item tensors indexing the joined table
            return torch.stack((u, u), 1), torch.stack((pos, neg), 1)
        return u, pos, neg                      # (FREEDOM.loss shifts global ids itself)

    sync = model.sync_grads if sharded else None
    # exchange bytes of one step on this rank, and the step's SpMM work: one eager step with the calls counted
    xbytes, nnz_step = [0], [0]

    spmm_orig = ops.spmm_raw

    def spmm_counting(csr, x, *a, **k):
        nnz_step[0] += csr.nnz
        return spmm_orig(csr, x, *a, **k)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = model.loss(*draw())
        loss.backward()
        if sync is not None:
            sync()
        opt.step()
        return loss.detach()

    eager_step()                                # warm-up: lazily built schedules, Adam state, communicators
    ops.spmm_raw = spmm_counting
    before = dict(cdist.STATS)
    eager_step()
    ops.spmm_raw = spmm_orig
    xbytes[0] = cdist.STATS["bytes"] - before["bytes"]
    n_exchanges = cdist.STATS["exchanges"] - before["exchanges"]
    torch.cuda.synchronize()
    use_graph = not args.no_graph and (not sharded or (backend == "nccl" and os.environ.get("CHAOREC_DIST_GRAPH", "1") == "1"))
    graphed = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, batch_fn=draw, after_backward=sync)
        except Exception as exc:      # noqa: BLE001 -- "launch eagerly", never a wrong result
            print(f"[bench rank {rank}] hipGraph capture of the {name} step failed ({exc!r}); eager launches", file=sys.stderr)
        if sharded:
            ok = torch.tensor([1.0 if graphed is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                graphed = None
    step = graphed if graphed is not None else eager_step

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    blocks = []
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        blocks.append(time.perf_counter() - t0)
        stop = sum(blocks) >= MIN_TIMED_S or len(blocks) >= 64
        if sharded:                             # (the ranks leave the loop together: every block ends in a barrier)
            flag = torch.tensor([1.0 if stop else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            stop = float(flag.item()) > 0.0
        if stop:
            break
    dt = float(np.median(blocks))
    t = torch.tensor([dt, float(nnz_step[0]), float(U_g)], device=dev, dtype=torch.float64)
    if sharded:
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t)
        dt, nnz_all, n_scored = float(tm[0]), float(t[1]), float(t[2])
    else:
        nnz_all, n_scored = float(t[1]), float(t[2])
    model.gene_ranklist()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        model.gene_ranklist()
    torch.cuda.synchronize()
    rank_ms = (time.perf_counter() - t1) / 3 * 1e3
    out = {
        "metric": "GCN edges/sec + full-rank users-scored/sec, dim=64",
        "value": nnz_all / (dt / steps), "unit": "directed-edge messages/s (every SpMM launch of the train step, fwd+bwd)",
        "users_scored_per_s_incl_d2h": n_scored / (rank_ms * 1e-3),
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "timed_blocks": {"blocks_of_steps": len(blocks), "ms_per_step_min": min(blocks) / steps * 1e3,
                         "ms_per_step_max": max(blocks) / steps * 1e3},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "real interactions, synthetic modality features, random-init weights",
        "config": {"workload": f"{name} train step on the real {dataset} graph" +
                               (f" replicated by user rows: rank g owns the {U1} users as users g*{U1}.. over the same {I} items"
                                if sharded else "") + f" (U={U1}x{world}, I={I}), features {tuple(v_feat.shape[1:])} / "
                               f"{tuple(t_feat.shape[1:])}, dim=64, batch={B}x{world}; gene_ranklist top-50 (to the CPU)",
                   "model_class": type(model).__name__, "spmm_nnz_per_step_all_ranks": nnz_all,
                   "exchange_bytes_per_step_per_rank": xbytes[0], "collectives_per_step": n_exchanges,
                   "collectives_forced_on_one_rank": bool(sharded and world == 1 and cdist._FORCE_COLLECTIVES),
                   "gene_ranklist_ms_incl_d2h_wall": rank_ms,
                   "launch": "captured hipGraph per step" if graphed is not None else "eager launches",
                   "parallelism": (f"user-row shards x{world}; exchanges by {cdist.exchange_mode_used()} over {backend}"
                                   if sharded else "single GPU"),
                   "multi_rank_rccl_measured": bool(sharded and world > 1 and backend == "nccl"
                                                    and torch.cuda.device_count() >= world),
                   "host_build_seconds": build_s},
    }
    if rank == 0:
        try:
            from .model_roofline import model_roofline
            rows = None
            if name == "FREEDOM":
                b = draw()
                rows = int(torch.unique(torch.cat((b[1], b[2]))).numel())
            out["roofline"] = model_roofline(name, model if not sharded else getattr(model, "full", model), rows)
        except Exception as exc:      # noqa: BLE001 -- a roofline annotation must not take the measured record with it
            out["roofline"] = {"error": repr(exc)[:200]}
    del model, opt, graphed, step
    torch.cuda.empty_cache()
    return out


def main_model(args, world, rank, local_rank, force_sharded):
    """--model MMGCN / FREEDOM as the headline of the line (measure_model)."""
    sharded = world > 1 or force_sharded
    dev, backend = init_ranks(local_rank, sharded)
    dataset = args.dataset if args.dataset != "sports" else None
    out = measure_model(args, args.model, world, rank, dev, sharded, backend, dataset=dataset)
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        from .record import emit
        out.setdefault("cpu_baseline", {"value": None, "reason": "--model runs carry no CPU leg (the LightGCN line does)"})
        emit(out)
