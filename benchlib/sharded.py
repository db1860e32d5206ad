"""N > 1 (or one rank with the N > 1 path forced): the user-row-sharded LightGCN step over RCCL."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, flush_c_stdout, init_ranks, spmm_kernel_name  # noqa: F401
from .models import measure_model  # noqa: F401
from .probe import captured_all_reduce_is_exact, probe_mark, probe_node  # noqa: F401
from .record import emit
from .single import CHAIN_TIMING_NOTE, performed_value, scoring_roofline, time_spmm_chain  # noqa: F401


def measure_sharded_lightgcn(args, dataset, D, steps, warmup, world, rank, dev, backend, use_graph, probe_mode=False):
    """One user-sharded LightGCN measurement (weak scaling: rank g owns one copy of the dataset's users over the shared
    item set; dist.FusedShardedLightGCNStep, joined or split launches by item-table size): timed steps between
    barriers (max over ranks), the SpMM roofline from the step's own launches, cold ranking of every rank's users.
    -> dict (identical on every rank)."""
    import torch.distributed as dist
    from chaorec_amd import ops
    from chaorec_amd import dist as cdist
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    L, B, reg = args.n_layers, args.batch, 1e-3
    # the sharded step: "fused" (dist.FusedShardedLightGCNStep: joined-graph propagates, Adam in the last propagate's
    # epilogue, no autograd) or "autograd" (round 2's path: loss_local -> backward -> FusedAdam under GraphedTrainStep)
    step_kind = os.environ.get("CHAOREC_DIST_STEP", "fused")
    if args.torch_adam or L < 1:
        step_kind = "autograd"
    t_build = time.perf_counter()
    job = cdist.build_weak_scaling_job(dataset, world, rank, D, L, reg, dev, seed=42, synthetic=args.synthetic)
    model, edges, U, I, U1 = job["model"], job["local_edges"], job["num_user_local"], job["I"], job["U1"]
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build
    E = len(edges)
    e_dir = 2 * E
    opt = torch.optim.Adam(model.parameters(), lr=1e-3) if args.torch_adam else FusedAdam(model.parameters(), lr=1e-3)
    edges_dev = torch.from_numpy(edges.astype(np.int64)).to(dev)
    loss_sum = torch.zeros((), device=dev)
    batch_counter = torch.zeros(1, dtype=torch.int64, device=dev)   # device-resident: advances inside the graph

    # --- first contact: before a step trusts an exchange mode on this node, the mode sums a random buffer of the step's
    # own size and is compared with dist.all_reduce (eagerly and replayed from a hipGraph); large buffers also get the
    # modes timed against each other and `auto` takes the fastest that passed (dist.calibrate_exchange)
    item_bytes = cdist.padded_rows(I) * D * 4
    calibration = None
    if cdist._active(None):
        big = item_bytes >= cdist.AUTO_BIG_BYTES
        asked = cdist.exchange_mode()
        cands = ("allreduce", "rs_ag", "p2p") if (asked == "auto" and big) else \
            (() if asked in ("auto", "allreduce") else (asked if asked != "direct" else "rs_ag",))
        if cands:
            calibration = cdist.calibrate_exchange(I, D, dev, captured=use_graph and backend == "nccl", candidates=cands)
            if probe_mode and "p2p" in cands:
                probe_mark("p2p", calibration.get("p2p", {}).get("ok", False))

    def draw(i=None):
        """One batch in ONE launch (chaorec_draw_batch): B training edges of this rank picked uniformly + one sampled
        negative each, LOCAL item ids.  i=None: graph-capturable form, the batch index comes from the device counter."""
        if i is None:
            batch_counter.add_(1)
            return ops.draw_batch(edges_dev, model.hist, B, model.num_user, I, 42 + rank, 0, step_dev=batch_counter)
        return ops.draw_batch(edges_dev, model.hist, B, model.num_user, I, 42 + rank, 1_000_000 + i)

    graphed = None
    fused = None
    fused_loss = torch.zeros(1, device=dev)     # sum of this rank's batch losses, accumulated inside the step
    spr = 1 if E > 5_000_000 else args.steps_per_replay       # (a config-5-shard step is ~30 ms: nothing to gain from k-step replays)
    if step_kind == "fused":
        def make_fused(capture):
            return cdist.FusedShardedLightGCNStep(model, opt, batch_size=B, edges=edges_dev, seed=42 + rank,
                                                  step_dev=batch_counter, capture=capture, loss_accum=fused_loss,
                                                  steps_per_replay=spr)
        if use_graph:
            try:
                fused = make_fused(True)
            except Exception as exc:      # noqa: BLE001
                print(f"[bench rank {rank}] hipGraph capture of the fused sharded step failed ({exc!r}); eager launches",
                      file=sys.stderr)
                fused = None
            ok = torch.tensor([1.0 if fused is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                fused = None
        if fused is None:
            use_graph = False
            fused = make_fused(False)
        graphed = fused if use_graph else None          # (what the launch-mode fields below report)
    elif use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, batch_fn=draw, loss_fn=model.loss_local)
        except Exception as exc:      # noqa: BLE001 -- any capture failure means "launch eagerly", never a wrong result
            print(f"[bench rank {rank}] hipGraph capture of the sharded step failed ({exc!r}); eager launches",
                  file=sys.stderr)
            graphed = None
        torch.cuda.synchronize()
        ok = torch.tensor([1.0 if graphed is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            graphed = None
        if graphed is not None:
            # first replays of a graph holding RCCL kernels, under a watchdog: a launch mode that cannot make progress
            # must end the job with a message, not sit on the GPUs until an outer timeout
            import threading
            done = threading.Event()

            def watchdog():
                if not done.wait(float(os.environ.get("CHAOREC_GRAPH_WATCHDOG_S", "120"))):
                    print(f"[bench rank {rank}] captured sharded step did not complete; rerun with "
                          f"CHAOREC_DIST_GRAPH=0", file=sys.stderr, flush=True)
                    os._exit(17)

            threading.Thread(target=watchdog, daemon=True).start()
            for _ in range(2):
                graphed()
            torch.cuda.synchronize()
            done.set()

    n_loss = [0]

    def step(i, force_eager=False):
        n_loss[0] += 1
        if fused is not None:
            if force_eager:           # (the SpMM-recording pass: the same launches, issued eagerly)
                fused._launch()
            else:
                fused(single=True)
            return
        if graphed is not None and not force_eager:
            graphed()                 # sampling + loss + backward + Adam: one hipGraph replay, no inputs
            loss = graphed.static_loss
        else:
            opt.zero_grad(set_to_none=True)
            loss = model.loss_local(*draw(i))
            loss.backward()
            opt.step()
            loss = loss.detach()
        loss_sum.add_(loss)   # the reference's per-batch loss.item() sync is kept off the device path

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def run_steps(first, n):
        if fused is not None:         # whole k-step replays, single-step replays for the remainder
            n_loss[0] += n
            fused.run(n, full_last=False)      # (steps INSIDE an epoch: the full-result step is timed separately below)
            return
        for i in range(n):
            step(first + i)

    run_steps(0, warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(warmup, steps)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / steps * 1e3
    t = torch.tensor([float(e_dir), float(U)], device=dev, dtype=torch.float64)
    dist.all_reduce(t)
    e_dir_all, n_scored = int(t[0].item()), float(t[1].item())
    msgs_per_step_all = 2 * L * e_dir_all
    value = msgs_per_step_all / (dt / steps)
    loss_mean = (float(fused_loss.item()) / world if fused is not None else float(loss_sum.item())) / max(n_loss[0], 1)

    if probe_mode:
        return dict(graphed=graphed is not None)

    # --- exposed communication: the same step with the exchanges switched off (every rank computes on its own partial
    # sums: wrong numbers, same launches) -- what the exchanges cost the step beyond what the launches hide
    exposed = None
    if fused is not None and cdist._active(None) and not probe_mode:
        saved = fused._save_state()
        real_exchange, real_frontier = fused._exchange, fused._exchange_frontier
        fused._exchange = lambda buf: cdist._Pending(None)
        fused._exchange_frontier = lambda buf, bits, cap=None: cdist._Pending(None)
        try:
            for _ in range(2):
                fused._launch()
            barrier()
            t0 = time.perf_counter()
            n_dry = max(3, min(steps, 10))
            for _ in range(n_dry):
                fused._launch()
            barrier()
            dry = (time.perf_counter() - t0) / n_dry * 1e3
        finally:
            fused._exchange, fused._exchange_frontier = real_exchange, real_frontier
            fused._restore_state(saved)
        for _ in range(2):
            fused._launch()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_dry):
            fused._launch()
        barrier()
        wet = (time.perf_counter() - t0) / n_dry * 1e3
        t = torch.tensor([dry, wet], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed = {"eager_ms_per_step_with_exchanges": float(t[1]), "eager_ms_per_step_without_exchanges": float(t[0]),
                   "exposed_exchange_ms_per_step": float(t[1] - t[0]), "exchanges_per_step": 2 * L + 1,
                   "bytes_per_exchange": item_bytes,
                   "note": "both eager (same launches, the exchanges replaced by nothing in the second run): the difference "
                           "is what the 2L+1 exchanges cost beyond what the SpMM launches hide"}
        fused._restore_state(saved)

    # --- SpMM roofline: the step's own SpMM calls, recorded in one eager step and replayed IN THE STEP'S ORDER
    calls = []
    orig = ops.spmm_raw

    def recording_spmm(csr, x, *a, **k):
        out = orig(csr, x, *a, **k)
        calls.append((csr, x, a, dict(k)))
        return out

    ops.spmm_raw = recording_spmm
    step(warmup + steps, force_eager=True)
    ops.spmm_raw = orig
    torch.cuda.synchronize()
    timed = []
    for csr, x, a, k in calls:
        timed.append((lambda csr=csr, x=x, a=a, k=k: orig(csr, x, *a, **k), csr, x.shape[1]))
    avg_spmm_ms, model_bytes, compulsory = time_spmm_chain(timed)
    achieved = model_bytes / (avg_spmm_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": spmm_kernel_name(D), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": model_bytes,
                "avg_launch_us": avg_spmm_ms * 1e3, "compulsory_bytes_per_launch": compulsory,
                "launches_per_step": len(calls), "timing": CHAIN_TIMING_NOTE,
                "note": "rank 0's shard: " + ("the two row blocks of every layer as separate launches (split step)"
                                              if getattr(fused, "split", False) else
                                              "one launch per layer over the rank's joined graph [[0, B_g], [B_g^T, 0]]")}

    # --- work PERFORMED by a light step: source rows gathered by its SpMM-family launches, counted over one eager step of this
    # rank (dense launches: every entry; list launches: the entries of the listed rows; gated launches: the entries whose
    # source row is flagged), summed over the ranks -- what the sub-record's `value` divides by the step time
    performed = None
    if fused is not None and getattr(fused, "light", False):
        gathered = [0]

        def _deg_sum(csr, lst, n):
            rows = lst[:int(n.item())].to(torch.int64)
            return int((csr.rowptr[rows + 1] - csr.rowptr[rows]).sum().item())

        def _flagged(csr, bits):
            w = bits.to(torch.int64) & 0xFFFFFFFF
            c = csr.col.to(torch.int64)
            return int(((w[c >> 5] >> (c & 31)) & 1).sum().item())

        class _Counting:
            def __init__(self, inner):
                self._inner = inner

            def __getattr__(self, name):
                fn = getattr(self._inner, name)
                if name in ("spmm", "spmm_mean", "spmm_adam"):
                    def dense(csr, *a, **k):
                        gathered[0] += csr.nnz
                        return fn(csr, *a, **k)
                    return dense
                if name == "spmm_rowlist":
                    def listed(csr, x, y, lst, n, *a, **k):
                        gathered[0] += _deg_sum(csr, lst, n)
                        return fn(csr, x, y, lst, n, *a, **k)
                    return listed
                if name == "spmm_rowsparse":
                    def gated(csr, x, y, *a, **k):
                        gathered[0] += _flagged(csr, k["src_bits"]) if k.get("src_bits") is not None else csr.nnz
                        return fn(csr, x, y, *a, **k)
                    return gated
                return fn

        saved = fused._save_state()
        inner = fused.K
        fused.K = _Counting(inner)
        try:
            fused._launch()
            torch.cuda.synchronize()
        finally:
            fused.K = inner
            fused._restore_state(saved)
        t = torch.tensor([float(gathered[0])], device=dev, dtype=torch.float64)
        dist.all_reduce(t)
        performed = {"messages_gathered_per_step": int(t.item()), "value_performed": float(t.item()) / (dt / steps),
                     "what": "source rows gathered by one light step's SpMM-family launches on every rank (one eager step counted)"}

    forward_note = None
    if fused is not None and getattr(fused, "light", False):
        # the timed steps were LIGHT ones (steps inside an epoch); the step that precedes an evaluation computes every row
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            n_loss[0] += 1
            fused(single=True, full_result=True)
        barrier()
        full_ms = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(full_ms, op=dist.ReduceOp.MAX)
        forward_note = {"timed_steps": "light", "ms_per_step_full_result": float(full_ms.item()), "full_steps_per_epoch": 1,
                        "what": "dist.FusedShardedLightGCNStep with the light forward: the batch drawn first, the last two "
                                "forward layers over the frontier's row lists (item partials through frontier buffers and "
                                "frontier exchanges); the step before an evaluation is a full one.  The record's `value` counts the "
                                "source rows the light steps gather, value_reference_equivalent the reference step's 2 L E_dir"}
    elif fused is not None:
        fused(single=True)                  # (the recording pass above ran eager launches; leave a complete result behind)

    # --- full-rank evaluation: every rank ranks its own users against the replicated item table, no exchange --------
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3 if E > 5_000_000 else 5)]
    st = {}
    with torch.no_grad():
        ru, ri = model.result_u.detach(), model.result_i.detach()
        ops.score_topk(ru, ri, model.hist, 1e-6, 50, id_offset=model.shard.num_user_global)
        for s_, e_ in ev:
            s_.record()
            ops.score_topk(ru, ri, model.hist, 1e-6, 50, id_offset=model.shard.num_user_global)
            e_.record()
        torch.cuda.synchronize()
        ops.score_topk(ru, ri, model.hist, 1e-6, 50, id_offset=model.shard.num_user_global, stats=st)
    score_ms = float(np.median([s_.elapsed_time(e_) for s_, e_ in ev]))
    t = torch.tensor([score_ms], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    score_ms = float(t.item())
    tf = 2.0 * n_scored * I * D / (score_ms * 1e-3) / 1e12
    launch = ("captured hipGraph per step" if graphed is not None else "eager launches") + \
        (f", fused sharded step (dist.FusedShardedLightGCNStep, {'split' if fused.split else 'joined'} launches: "
         f"{'4L+5' if fused.split else '2L+5'} launches, 2L+1 exchanges"
         f"{'; the first two backward propagates over the batch frontier only (row-sparse)' if fused.sparse_bwd else ''}"
         f"{'; light forward' if getattr(fused, 'light', False) else ''}"
         f"; {fused.steps_per_replay} steps per replay)"
         if fused is not None else ", autograd step")
    res = dict(dataset=dataset, data=job["data"], U1=U1, I=I, D=D, L=L, B=B, world=world, e_dir_all=e_dir_all,
               ms_per_step=ms_per_step, value=value, msgs_per_step=msgs_per_step_all, loss_mean=loss_mean, launch=launch,
               graphed=graphed is not None, fused=fused is not None, split=bool(getattr(fused, "split", False)),
               roofline=roofline, score_ms=score_ms, score_tf=tf, score_st=st, n_scored=n_scored, build_s=build_s,
               exchange=cdist.exchange_mode_used(), exchange_bytes=item_bytes, calibration=calibration, exposed=exposed,
               table_mb=(U + I) * D * 4 / 1e6, forward=forward_note, performed=performed,
               frontier_exchanges=cdist.STATS.get("frontier_exchanges", 0),
               frontier_caps=({"batch_items_rows": getattr(fused, "_cap0", None), "n1_items_rows": getattr(fused, "_cap1", None),
                               "item_rows": I, "bytes_per_compact_exchange": {
                                   "batch_items": (getattr(fused, "_cap0", 0) or 0) * D * 4,
                                   "n1_items": (getattr(fused, "_cap1", 0) or 0) * D * 4, "dense": item_bytes},
                               "capture_attempts": getattr(fused, "capture_attempts", None)}
                              if fused is not None and getattr(fused, "sparse_bwd", False) else None))
    del fused, graphed, model, opt, job, edges_dev, calls, timed
    torch.cuda.empty_cache()
    return res


def main_sharded(args, world, rank, local_rank, force_sharded):
    """N > 1 (weak scaling): rank g owns one copy of the dataset's users over the shared item set, the item partials
    of every layer are summed over RCCL (chaorec_amd/dist.py; CHAOREC_DIST_EXCHANGE picks the collective, `auto` by
    size after a first-contact calibration on this node).  The line carries the same sub-records as the N = 1 line:
    `hbm_regime` (config5_shard per rank, D = 128: at N = 8 that IS BASELINE configs[4]) and `models` (MMGCN/microlens
    = configs[3], FREEDOM/clothing = configs[2], user-sharded)."""
    backend = os.environ.get("CHAOREC_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    probe = None
    want_p2p = backend == "nccl" and os.environ.get("CHAOREC_DIST_EXCHANGE", "auto") in ("auto", "p2p") \
        and "p2p" not in os.environ.get("CHAOREC_DIST_VETO", "")
    if (backend == "nccl" and not args.probe_graph and not args.no_graph and not args.torch_adam
            and os.environ.get("CHAOREC_DIST_GRAPH") is None):
        probe = probe_node(args, world, want_p2p)          # before anything here initialises the GPU
        if want_p2p and not probe["p2p"]:
            os.environ["CHAOREC_DIST_VETO"] = ",".join(filter(None, [os.environ.get("CHAOREC_DIST_VETO", ""), "p2p"]))
    dev, backend = init_ranks(local_rank)
    import torch.distributed as dist

    from chaorec_amd import _lib
    from chaorec_amd import dist as cdist
    _lib.ensure_built()
    _lib.load()
    if args.probe_graph:
        ok = captured_all_reduce_is_exact(dev, world, rank)
        probe_mark("allreduce_replay", ok)
        if not ok:
            print(f"[bench probe rank {rank}] a captured all-reduce returned stale sums on replay", file=sys.stderr, flush=True)
            sys.exit(4)
        if "p2p" not in cdist._VETOED and not args.no_hbm_regime:
            # the p2p exchange at the size of the hbm_regime sub-record (2 M items x 128: 1 GB), eager and replayed
            from chaorec_amd.synthetic import DATASET_SHAPES
            tbl = cdist.calibrate_exchange(DATASET_SHAPES["config5_shard"][1], 128, dev, captured=True, candidates=("p2p",))
            probe_mark("p2p", tbl.get("p2p", {}).get("ok", False))

    # the whole zero_grad -> loss -> backward -> Adam sequence as ONE captured hipGraph, RCCL calls included
    # (CHAOREC_DIST_GRAPH=0 keeps it eager).  Every rank must run the same launch mode.
    use_graph = (not args.no_graph and not args.torch_adam and backend == "nccl"
                 and os.environ.get("CHAOREC_DIST_GRAPH", "1") == "1")
    if probe is not None:
        flag = torch.tensor([1.0 if (use_graph and probe["graph"]) else 0.0, 1.0 if "p2p" not in cdist._VETOED else 0.0],
                            device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_graph = float(flag[0].item()) > 0.0
        if float(flag[1].item()) == 0.0:
            cdist.veto("p2p")

    head = measure_sharded_lightgcn(args, args.dataset, args.dim, args.steps, args.warmup, world, rank, dev, backend,
                                    use_graph, probe_mode=args.probe_graph)
    if args.probe_graph:
        probe_mark("step_graph", head["graphed"])
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if head["graphed"] else 3)
    D, L, B, U1, I = head["D"], head["L"], head["B"], head["U1"], head["I"]
    measured_rccl = bool(world > 1 and backend == "nccl" and int(os.environ.get("CHAOREC_BENCH_VISIBLE_GPUS", str(world))) >= world
                         and torch.cuda.device_count() >= world)
    out = {
        "metric": f"GCN edges/sec + full-rank users-scored/sec, dim={D}",
        **performed_value(head), "unit": "directed-edge messages/s (fwd+bwd SpMM of the train step)",
        "users_scored_per_s": head["n_scored"] / (head["score_ms"] * 1e-3),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": head["data"],
        "multi_rank_rccl_measured": measured_rccl,
        "config": {"workload": f"LightGCN train step, {args.dataset} graph replicated by user rows: rank g owns the {U1} "
                               f"users of the {head['data']} graph as users g*{U1}.. over the same {I} items "
                               f"(U={U1}x{world}, I={I}, E_dir={head['e_dir_all']}), dim={D}, n_layers={L}, batch={B}x{world}; "
                               f"gene_ranklist top-50 over all users (cold thresholds)",
                   "messages_per_step": head["msgs_per_step"], "gene_ranklist_ms": head["score_ms"],
                   "launch": head["launch"],
                   "optimizer": "torch.optim.Adam" if args.torch_adam else
                   ("Adam in the last user-row SpMM's epilogue + one fused launch on the replicated item rows"
                    if head["fused"] else "FusedAdam (chaorec_adam_step_f32)"),
                   "parallelism": f"user-row shards x{world}; item partials ({head['exchange_bytes'] / 1e6:.1f} MB) summed per "
                                  f"layer by {head['exchange']} over {backend}",
                   "ranks_share_devices": torch.cuda.device_count() < world,
                   "node_probe": probe, "exchange_calibration": head["calibration"],
                   "exposed_communication": head["exposed"], "host_build_seconds": head["build_s"]},
        "roofline": head["roofline"], "roofline_scoring": scoring_roofline(head),
        "loss_mean": head["loss_mean"],
        **({"forward": head["forward"], "frontier_exchanges_issued": head["frontier_exchanges"],
            "frontier_capacities": head["frontier_caps"]} if head.get("forward") else {}),
    }

    # (the second dominant kernel family inside the object the driver keeps, as in the N = 1 line)
    sr = out["roofline_scoring"]
    out["roofline"]["scoring"] = {"bound": "mfma", "kernel": sr["kernel"], "achieved": sr["achieved"], "peak": sr["peak"],
                                  "unit": sr["unit"], "frac": sr["frac"], "sweep_only_frac": sr.get("sweep_only_frac"),
                                  "gene_ranklist_ms": head["score_ms"]}
    # cpu_baseline: rank 0 times the reference op sequence (oracle/torch_ref.py) on ONE rank's share of the workload --
    # the N = 1 workload, which is what a weak-scaling rank owns -- while the other ranks wait at the next barrier
    if args.no_cpu_baseline:
        out["cpu_baseline"] = {"value": None, "reason": "--no-cpu-baseline"}
    elif rank == 0:
        try:
            from .common import load_graph
            from .cpu import cpu_baseline
            edges_cpu, U_c, I_c, _ = load_graph(args.dataset, args.synthetic)
            out["cpu_baseline"] = cpu_baseline(np.asarray(edges_cpu), U_c, I_c, D, L, B, 1e-3, args.cpu_seconds)
            out["cpu_baseline"]["sample"] = "ONE rank's share (the N = 1 workload): " + out["cpu_baseline"]["sample"]
        except Exception as exc:      # noqa: BLE001 -- the baseline must not take the measured headline with it
            out["cpu_baseline"] = {"value": None, "reason": repr(exc)[:200]}
    dist.barrier()

    # Sub-records, under a watchdog: a collective that cannot make progress in a sub-record must not take the headline
    # numbers (measured above) with it -- rank 0 then prints the line, naming the unfinished ones, and every rank ends
    # with a non-zero exit code so that launchers and CI see the hang.
    import threading
    finished = threading.Event()
    wanted = ([] if (args.no_hbm_regime or args.dataset in ("config5_shard", "config5")) else ["hbm_regime"]) + \
        ([] if args.no_models else ["models.MMGCN", "models.FREEDOM"])

    def unfinished():
        return [w for w in wanted if (w not in out if "." not in w else w.split(".")[1] not in out.get("models", {}))]

    def give_up():
        if not finished.wait(float(os.environ.get("CHAOREC_SUBRECORD_TIMEOUT_S", "900"))):
            if rank == 0:
                out["subrecords_timed_out"] = unfinished()
                emit(out)
            os._exit(21)

    threading.Thread(target=give_up, daemon=True).start()

    def all_ok(ok):
        t = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) > 0.0

    if not args.no_hbm_regime and args.dataset not in ("config5_shard", "config5"):
        try:
            h = measure_sharded_lightgcn(args, "config5_shard", 128, args.hbm_steps, 3, world, rank, dev, backend, use_graph)
            out["hbm_regime"] = {
                "workload": f"BASELINE configs[4] at {world} rank(s): every rank owns one GPU's share of the synthetic "
                            f"bipartite graph (U={h['U1']} per rank x {world}, I={h['I']}, E_dir={h['e_dir_all']}), dim=128, "
                            f"n_layers={h['L']}, batch={h['B']}x{world} (per-rank table {h['table_mb']:.0f} MB; item partial "
                            f"{h['exchange_bytes'] / 1e6:.0f} MB per exchange)" +
                            (" -- at 8 ranks this IS configs[4]" if world == 8 else ""),
                "data": h["data"], "steps": args.hbm_steps, "ms_per_step": h["ms_per_step"], **performed_value(h),
                "unit": "directed-edge messages/s", "launch": h["launch"], "exchange": h["exchange"],
                "exchange_calibration": h["calibration"], "exposed_communication": h["exposed"],
                "roofline": h["roofline"], "gene_ranklist_ms_cold": h["score_ms"],
                "users_scored_per_s_cold": h["n_scored"] / (h["score_ms"] * 1e-3),
                "roofline_scoring": scoring_roofline(h), "loss_mean": h["loss_mean"], "host_build_seconds": h["build_s"],
                **({"forward": h["forward"], "frontier_exchanges_issued": h["frontier_exchanges"],
                    "frontier_capacities": h["frontier_caps"]} if h.get("forward") else {}),
            }
            del h
        except Exception as exc:      # noqa: BLE001
            out["hbm_regime"] = {"error": repr(exc)[:300]}
        torch.cuda.empty_cache()
    if not args.no_models:
        out["models"] = {}
        for name in ("MMGCN", "FREEDOM"):
            try:
                out["models"][name] = measure_model(args, name, world, rank, dev, True, backend,
                                                    steps=min(args.steps, 20), warmup=min(args.warmup, 5))
            except Exception as exc:      # noqa: BLE001
                out["models"][name] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()
    finished.set()
    dist.barrier()
    cdist.destroy_side_groups()
    dist.destroy_process_group()
    cdist.P2PExchange.forget_all()
    if rank == 0:
        emit(out)
