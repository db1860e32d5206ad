#!/usr/bin/env python3
"""bench.py -- the hot path's headline numbers on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LightGCN on the REAL Amazon-sports interaction graph (U=28940, I=15207,
E=158554 -> 317108 directed edges; the reference's Data/sports/train.npy travels with the repository as a packed
fixture, tests/golden/sports_interactions.npz), dim=64, n_layers=3, batch 1024, embeddings xavier-initialised under
seed 42 (random-init weights: there are no checkpoints).  --synthetic swaps in a seeded graph of the same shape.

A "step" is one reference training iteration (train_and_evaluate.py:43-48): negative sampling,
model.loss() = full-graph 3-layer propagate + BPR + L2, loss.backward(), Adam step.  `value` is
directed-edge messages per second through that step, counting the forward AND backward SpMM
(2 * L * E_dir per step), with all inputs resident in HBM.  users_scored_per_s times
gene_ranklist() (scoring + mask + top-50 for every user).

At N > 1 the graph shards by user rows (weak scaling: every rank owns a sports-sized user shard
over the same item set), items replicated, item-row partial sums all-reduced over RCCL per layer.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MIN_TIMED_S = 0.05                   # the timed region is repeated in blocks of --steps until this much was timed
STEADY_EVALS = 6                     # carried-threshold evaluations (one per epoch) before the timed one
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md matrix-core table
TRAINED_STEPS = 5000            # ~32 epochs of the sports-sized graph: embeddings in a trained state
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32 dense peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--dataset", default="sports")
    p.add_argument("--n-layers", type=int, default=3)
    p.add_argument("--dim", type=int, default=64)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-trained-state", action="store_true",
                   help="time gene_ranklist only on the embeddings the timed steps leave behind (profiling runs)")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph step")
    p.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused HIP Adam")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    p.add_argument("--unfused", action="store_true",
                   help="N=1: the autograd step (loss_drawn -> backward -> FusedAdam) instead of FusedLightGCNStep")
    p.add_argument("--steps-per-replay", type=int, default=10,
                   help="training steps captured back to back in one hipGraph (N=1 fused step)")
    p.add_argument("--synthetic", action="store_true", help="dataset-shaped synthetic graph instead of the real one")
    p.add_argument("--no-hbm-regime", action="store_true", help="skip the config-5-shard sub-record (N=1)")
    p.add_argument("--hbm-steps", type=int, default=10, help="timed steps of the config-5-shard sub-record")
    p.add_argument("--no-full-config5", action="store_true",
                   help="skip the sub-record of BASELINE configs[4] WHOLE on this GPU (10 M x 2 M, 4e8 directed edges: ~40 s)")
    p.add_argument("--no-models", action="store_true", help="skip the MMGCN / FREEDOM sub-records (N > 1: the sharded models)")
    p.add_argument("--full-steps", type=int, default=10, help="timed steps of the configs[4]-whole sub-record (N=1)")
    p.add_argument("--model", default="LightGCN", choices=["LightGCN", "MMGCN", "FREEDOM"],
                   help="LightGCN: the headline workload.  MMGCN (BASELINE configs[3], microlens) / FREEDOM (configs[2], "
                        "clothing): the model's captured train step + gene_ranklist, user-sharded at --gpus N > 1")
    p.add_argument("--spmm-only", action="store_true",
                   help="N=1: stop after the timed steps and the SpMM roofline (no ranking): the command of the --pmc passes "
                        "at config 5, where a counter-collecting run of the 5 PFLOP ranking takes tens of minutes")
    p.add_argument("--probe-graph", action="store_true", help=argparse.SUPPRESS)   # child mode of probe_sharded_graph()
    p.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)   # child mode of the launcher's CPU test
    return p.parse_args()


def probe_path():
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"chaorec_probe_{os.environ.get('MASTER_PORT', '29511')}.json")


def probe_node(args, world, want_p2p):
    """What can this node's launch stack do?  Asked in a CHILD job (one child per rank, its own rendezvous port) before
    this process touches the GPU, so that a mode that hangs or faults costs a bounded wait, not the measurement:
      stage `allreduce_replay`  an all-reduce captured in a hipGraph returns fresh sums on every replay
      stage `p2p`               (want_p2p) the hand-written peer-to-peer exchange (csrc/exchange.hip: peer kernels'
                                writes read through IPC mappings after a stream-ordered barrier) equals dist.all_reduce,
                                eagerly and replayed, at the sizes this run will exchange -- its FIRST contact with
                                real xGMI links happens here, in a process whose death costs nothing
      stage `step_graph`        the fused sharded step captures, replays and trains a few steps
    The child job's rank 0 rewrites a small JSON file after every stage; a stage that was entered and never finished
    counts as failed.  -> dict(graph=bool, p2p=bool).  A child that died in the p2p stage (a fault in a pull kernel
    cannot be caught in-process) is followed by a second child job with p2p vetoed, for the remaining stages.
    CHAOREC_DIST_GRAPH=0/1 skips the probe."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29511")) + 17
    path = probe_path()
    rank = os.environ.get("RANK", "0")

    def run(veto_p2p):
        if rank == "0" and os.path.exists(path):
            os.remove(path)
        env = dict(os.environ, MASTER_PORT=str(port + (5 if veto_p2p else 0)), CHAOREC_DIST_GRAPH="1",
                   CHAOREC_GRAPH_WATCHDOG_S="60", CHAOREC_PROBE_FILE=path,
                   TORCHELASTIC_USE_AGENT_STORE="False")     # the children rendezvous among themselves, not at the agent
        if veto_p2p or not want_p2p:
            env["CHAOREC_DIST_VETO"] = ",".join(filter(None, [env.get("CHAOREC_DIST_VETO", ""), "p2p"]))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", "3", "--warmup", "1",
               "--dataset", args.dataset, "--dim", str(args.dim), "--n-layers", str(args.n_layers), "--batch",
               str(args.batch), "--no-cpu-baseline", "--no-trained-state", "--probe-graph"] + \
              (["--synthetic"] if args.synthetic else []) + (["--no-hbm-regime"] if args.no_hbm_regime else [])
        try:
            rc = subprocess.run(cmd, env=env, timeout=float(os.environ.get("CHAOREC_PROBE_TIMEOUT_S", "300")),
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        except subprocess.TimeoutExpired:
            rc = -1
        time.sleep(1.0)                    # (every rank's child has ended or been ended: the file is final)
        try:
            st = json.load(open(path))
        except Exception:      # noqa: BLE001
            st = {}
        return rc, st

    rc, st = run(False)
    res = dict(graph=bool(st.get("step_graph")), p2p=bool(st.get("p2p")) and want_p2p, first_rc=rc, stages=st)
    if want_p2p and not st.get("p2p") and "step_graph" not in st:
        rc2, st2 = run(True)               # the p2p stage took the child job down: the other stages without it
        res.update(graph=bool(st2.get("step_graph")), second_rc=rc2, stages_second=st2)
    if not res["graph"] or (want_p2p and not res["p2p"]):
        print(f"[bench rank {rank}] node probe: {res}", file=sys.stderr, flush=True)
    return res


def probe_mark(stage, ok):
    """Child side of probe_node(): rank 0 records a finished stage."""
    path = os.environ.get("CHAOREC_PROBE_FILE")
    if not path or os.environ.get("RANK", "0") != "0":
        return
    try:
        st = json.load(open(path))
    except Exception:      # noqa: BLE001
        st = {}
    st[stage] = bool(ok)
    with open(path + ".tmp", "w") as f:
        json.dump(st, f)
    os.replace(path + ".tmp", path)


def captured_all_reduce_is_exact(dev, world, rank):
    """Probe-mode check: an all-reduce captured in a hipGraph must return the sum of what the ranks hold AT REPLAY
    TIME, on every replay (a graph node that only acts on the first replay -- as memset nodes do on this stack,
    DESIGN 3.5 -- would time perfectly and train on stale sums)."""
    import torch.distributed as dist
    t = torch.zeros(1 << 20, device=dev)                 # 4 MB, the size of the item partials at sports scale
    src = torch.zeros_like(t)
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        t.copy_(src)
        dist.all_reduce(t)                               # eager first: communicator set-up happens outside capture
    cur.wait_stream(side)
    g = torch.cuda.CUDAGraph()
    from chaorec_amd.dist import settle_before_capture
    settle_before_capture()                              # (device idle, RCCL's watchdog has retired the eager all-reduce)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (RCCL's watchdog thread polls events meanwhile)
        t.copy_(src)
        dist.all_reduce(t)
    ok = True
    for r in range(3):
        src.fill_(float((rank + 1) * (r + 1)))
        g.replay()
        torch.cuda.synchronize()
        ok = ok and bool((t == float((r + 1) * world * (world + 1) // 2)).all())
    return ok


def spmm_model_bytes(nnz, n_rows, D):
    """no-reuse CSR model, fp32: per nonzero a D-float source row + 4 B col + 4 B val;
    per output row a D-float store + 8 B row pointer."""
    return nnz * (4 * D + 8) + n_rows * (4 * D + 8)


def cpu_baseline(edges, U, I, D, L, B, reg, budget_s):
    """The reference CPU path restated in plain torch (oracle/torch_ref.py), timed on this box's host cores on a
    bounded number of steps.  Best of a small sweep over torch's intra-op thread count, capped at 64 (the box has far
    more cores than a 0.3 M-edge scatter can use: all of them is slower than a few); `cores` = the thread count of the
    best run, the one `value` is quoted from."""
    from oracle.torch_ref import TorchRefLightGCN
    from chaorec_amd.graph import user_item_dict_from_edges
    ncpu = os.cpu_count() or 8
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})   # (all 256 threads: 20 s per step, never the best)
    uid = user_item_dict_from_edges(edges)
    rng = np.random.default_rng(0)
    E = len(edges)
    old_threads = torch.get_num_threads()

    def run(threads, budget):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        m = TorchRefLightGCN(U, I, edges, uid, D, reg, L)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            b = rng.integers(0, E, B)
            u, p = torch.from_numpy(edges[b, 0].astype(np.int64)), torch.from_numpy(edges[b, 1].astype(np.int64))
            n = torch.from_numpy(rng.integers(U, U + I, B))
            opt.zero_grad()
            loss = m.loss(u, p, n)
            loss.backward()
            opt.step()

        step()  # warm-up
        t0 = time.perf_counter()
        n_steps = 0
        while n_steps < 3 or (time.perf_counter() - t0 < budget and n_steps < 200):
            step()
            n_steps += 1
        return (time.perf_counter() - t0) / n_steps, n_steps, m

    share = budget_s * 0.6 / len(cands)
    tried = {}
    best = None
    for t in cands:
        dt, n_steps, m = run(t, share)
        tried[t] = dt * 1e3
        if best is None or dt < best[0]:
            best = (dt, n_steps, t, m)
    dt, n_steps, threads, m = best
    torch.set_num_threads(threads)
    t1 = time.perf_counter()
    with torch.no_grad():
        m.gene_ranklist()
    t_rank = time.perf_counter() - t1
    torch.set_num_threads(old_threads)
    e_dir = 2 * E
    return {
        "value": 2 * L * e_dir / dt, "unit": "directed-edge messages/s", "cores": threads, "kind": "port",
        "sample": f"{n_steps} train steps of the same workload ({dt * 1e3:.1f} ms/step) + 1 gene_ranklist "
                  f"({t_rank:.2f} s) with oracle/torch_ref.py (reference op sequence in plain torch, CPU); best of "
                  f"torch threads {cands} on {ncpu} host cores",
        "ms_per_step": dt * 1e3, "users_scored_per_s": U / t_rank,
        "ms_per_step_by_threads": {str(k): round(v, 1) for k, v in tried.items()},
    }


def load_graph(dataset, synthetic=False):
    """-> (edges int32 [E,2] with global item ids, U, I, 'real' | 'synthetic').  The reference's Data/<dataset> files
    travel with the repository as packed fixtures (tests/golden/<dataset>_interactions.npz); config5_shard (one GPU's
    share of BASELINE configs[4]) is synthetic by definition."""
    from chaorec_amd import dataload
    from chaorec_amd.synthetic import DATASET_SHAPES, DEVICE_BUILT, synthetic_interactions, synthetic_interactions_torch
    packed = None if synthetic else dataload.packed_interactions(dataset)
    if packed is not None:
        return packed["train"], packed["num_user"], packed["num_item"], "real"
    U, I, E = DATASET_SHAPES[dataset]
    if dataset in DEVICE_BUILT:      # BASELINE configs[4] whole: generated and laid out on the GPU (an int32 [E, 2] CUDA tensor)
        return synthetic_interactions_torch(U, I, E, seed=42, device="cuda"), U, I, "synthetic"
    return synthetic_interactions(U, I, E, seed=42), U, I, "synthetic"


def spmm_source_hash():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "chaorec_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest()


def time_spmm_calls(ops, calls, reps=20, passes=5):
    """HIP events on the launch stream around back-to-back re-launches of recorded SpMM calls: (median pass average
    in ms per launch, model bytes per launch, compulsory bytes per launch).  A single launch bracketed by events
    from Python mostly times the host; a saturated queue times the kernel."""
    pass_avg, tot_bytes, tot_comp, tot_launch = [], 0.0, 0.0, 0
    for _ in range(passes):
        pass_ms, pass_launch = 0.0, 0
        for fn, csr, D in calls:
            fn()                                     # warm
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            pass_ms += s.elapsed_time(e)
            pass_launch += reps
            tot_launch += reps
            tot_bytes += reps * spmm_model_bytes(csr.nnz, csr.n_rows, D)
            tot_comp += reps * (2 * csr.n_rows * 4 * D + csr.nnz * 8)
        pass_avg.append(pass_ms / pass_launch)
    return float(np.median(pass_avg)), tot_bytes / tot_launch, tot_comp / tot_launch


CHAIN_TIMING_NOTE = ("HIP events (on the launch stream) around replays of a hipGraph that holds the step's own SpMM launches "
                     "IN THE STEP'S ORDER -- every launch gathers from what the previous one wrote, as in the step, and the "
                     "kernel-to-kernel boundaries of the step are inside the figure: avg_launch_us = elapsed / launches.  "
                     "(Relaunching ONE call back to back, the method of rounds 1-3, re-reads a source table the previous "
                     "launch left in the caches and came out 3-6 % faster than the same kernel inside the step.)")


def time_spmm_chain(calls, min_pass_ms=10.0, passes=5):
    """calls: [(fn, csr, D)] in the step's order.  -> (ms per launch: median over `passes` of elapsed / launches, model
    bytes per launch, compulsory bytes per launch).  The chain is captured once and replayed (no host between the
    launches, like the step's own graph); if the capture fails the launches are issued eagerly, back to back."""
    for fn, _, _ in calls:
        fn()
    torch.cuda.synchronize()
    graph = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for fn, _, _ in calls:
                fn()
        torch.cuda.current_stream().wait_stream(side)
        import torch.distributed as _td
        graph = torch.cuda.CUDAGraph()
        if _td.is_initialized():
            from chaorec_amd.dist import settle_before_capture
            settle_before_capture()
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if _td.is_initialized() else "global"):
            for fn, _, _ in calls:
                fn()
    except Exception:      # noqa: BLE001
        graph = None
    torch.cuda.synchronize()

    def once():
        if graph is not None:
            graph.replay()
        else:
            for fn, _, _ in calls:
                fn()

    def timed(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            once()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e)

    once()
    reps = max(1, min(200, int(min_pass_ms / max(timed(1), 1e-3)) + 1))
    per_launch = [timed(reps) / (reps * len(calls)) for _ in range(passes)]
    tot_bytes = sum(spmm_model_bytes(csr.nnz, csr.n_rows, D) for _, csr, D in calls) / len(calls)
    tot_comp = sum(2 * csr.n_rows * 4 * D + csr.nnz * 8 for _, csr, D in calls) / len(calls)
    return float(np.median(per_launch)), tot_bytes, tot_comp


def light_step_accounting(ops, stepper, csr, U, I, D, L, edges_dev, hist, expand_n1, times_ms, names, whole_ms):
    """Every SpMM-family launch of ONE light step (optim.FusedLightGCNStep, large graphs) with the work it PERFORMS: rows
    computed, source rows gathered (= directed-edge messages formed), algorithmic bytes and their share of the 8 TB/s
    HBM peak -- for the batch the step's buffers hold (one real batch: R0 = its 3 B rows, N1 = R0 and its neighbours).
    Algorithmic bytes extend SURVEY 8(d)'s no-reuse CSR model to partial launches: per entry READ 8 B (col, val), per
    source row GATHERED 4 D, per row WRITTEN 4 D + 8 (the row and its pointer); the Adam epilogue adds eight passes over
    the table (z = G read and cleared; parameter, both moments read and written); the layer-mean epilogue L + 1 term
    reads per listed row.  The graph is symmetric, so the entries of a gated launch whose SOURCE is flagged are counted
    as the flagged rows' degrees."""
    N, nnz = csr.n_rows, csr.nnz
    B = stepper.B if hasattr(stepper, "B") else 1024
    ops.batch_rows(stepper.ids, stepper.bits[0], U, stepper._list0, stepper._list0_n, edges=edges_dev, hist=hist,
                   num_user=U, num_item=I, seed=4242, step=7)
    expand_n1()
    torch.cuda.synchronize()
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).to(torch.int64)
    r0 = stepper._list0[:int(stepper._list0_n.item())].to(torch.int64)
    n1 = stepper._row_list[:int(stepper._list_n.item())].to(torch.int64)
    n_r0, n_n1 = int(r0.numel()), int(n1.numel())
    deg_r0, deg_n1 = int(deg[r0].sum().item()), int(deg[n1].sum().item())
    row, ent, src = 4 * D + 8, 8, 4 * D
    dense = nnz * (ent + src) + N * row
    per = [("forward layer 1, every row (dense plain launch)", times_ms["dense"], dense, N, nnz)]
    for name, ms in zip(names, times_ms["sparse"]):
        if name.startswith("forward layer L-1"):
            per.append((name, ms, deg_n1 * (ent + src) + n_n1 * row, n_n1, deg_n1))
        elif name.startswith("forward layer L over R0"):
            per.append((name, ms, deg_r0 * (ent + src) + n_r0 * (row + (L + 1) * 4 * D), n_r0, deg_r0))
        elif name.startswith("backward propagate 1 over N1"):
            per.append((name, ms, deg_n1 * ent + deg_r0 * src + n_n1 * row + n_r0 * 4 * D, n_n1, deg_r0))
        else:                                   # every row written, the gathers gated by N1's bitmap
            per.append((name, ms, nnz * ent + deg_n1 * src + N * row + n_r0 * 4 * D, N, deg_n1))
    per.append(("backward propagate 3, every row, Adam epilogue (dense launch + 8 table passes)", times_ms["adam"],
                dense + 8 * N * 4 * D, N, nnz))
    out = [{"launch": n, "us": ms * 1e3, "rows_computed": rows, "source_rows_gathered": g, "algorithmic_bytes": float(by),
            "GBps": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS} for n, ms, by, rows, g in per]
    worst = min(out, key=lambda o: o["frac"])
    total_us = sum(o["us"] for o in out)
    return {"launches": out, "sum_us": total_us, "replayed_together_us": whole_ms * 1e3,
            "sum_over_replayed_together": total_us / (whole_ms * 1e3),
            "messages_gathered_per_step": int(sum(o["source_rows_gathered"] for o in out)),
            "frontier": {"R0_rows": n_r0, "N1_rows": n_n1, "graph_rows": N, "R0_entries": deg_r0, "N1_entries": deg_n1,
                         "graph_entries": nnz},
            "lowest_frac": {"launch": worst["launch"], "frac": worst["frac"]},
            "note": "each launch timed alone as a replayed one-launch hipGraph over the buffers one real batch leaves (R0, N1 "
                    "re-made here from the sampler's batch of seed 4242 / step 7); `replayed_together_us` is the six in the "
                    "step's order as ONE graph"}


def spmm_kernel_name(D, adam=False, rowsparse=False):
    """<LPR, CPL, ADAM, SP> as rocprofv3 prints the instantiation."""
    d4, lpr = D // 4, 1
    while lpr < min(d4, 64):
        lpr *= 2
    return (f"spmm_csr_ordered_kernel<{lpr}, {max(1, (d4 + 63) // 64)}, {'true' if adam else 'false'}, "
            f"{'true' if rowsparse else 'false'}>")


def measure_single_gpu(args, dataset, D, steps, warmup, dev, trained_steps, reps_rank=5, synthetic=False):
    """One GPU, unsharded LightGCN: the timed training steps, the SpMM roofline, gene_ranklist.  -> dict."""
    from chaorec_amd import ops, ranking
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep, GraphedTrainStep
    L, B, reg = args.n_layers, args.batch, 1e-3
    t_build = time.perf_counter()
    edges, U, I, data_kind = load_graph(dataset, synthetic)
    torch.cuda.synchronize()
    build_s = {"edge_list_s": time.perf_counter() - t_build}
    E = len(edges)
    e_dir = 2 * E
    torch.manual_seed(42)
    t_build = time.perf_counter()
    model = LightGCN(U, I, edges, None, D, reg, L, "add", dev).to(dev)
    torch.cuda.synchronize()
    build_s["model_csr_history_tables_s"] = time.perf_counter() - t_build
    t_build = time.perf_counter()
    model.graph.schedule(D)                     # (the SpMM row descriptors, built on the host from a copy of the CSR)
    torch.cuda.synchronize()
    build_s["spmm_schedule_s"] = time.perf_counter() - t_build
    opt = torch.optim.Adam(model.parameters(), lr=1e-3) if args.torch_adam else FusedAdam(model.parameters(), lr=1e-3)
    edges_dev = edges.to(torch.int64) if torch.is_tensor(edges) else torch.from_numpy(edges.astype(np.int64)).to(dev)
    if torch.is_tensor(edges):
        edges = None                            # (the int32 device copy is not needed any more; no CPU baseline at this size)
    loss_sum = torch.zeros(1, device=dev)
    batch_counter = torch.zeros(1, dtype=torch.int64, device=dev)   # device-resident: advances inside the graph
    fused = not args.unfused and not args.torch_adam and L >= 1
    spr = 1 if E > 50_000_000 else args.steps_per_replay     # (a config-5 step is ~0.25 s: nothing to gain from k-step replays)
    n_loss = [0]
    if fused:
        # 2L+2 launches per step, no autograd, no optimizer launch (optim.FusedLightGCNStep); --no-graph launches the
        # same kernels eagerly
        stepper = FusedLightGCNStep(model, opt, batch_size=B, edges=edges_dev, seed=42, step_dev=batch_counter,
                                    loss_accum=loss_sum, capture=not args.no_graph, steps_per_replay=spr)
        launch = ((f"captured hipGraph, {stepper.steps_per_replay} steps per replay" if not args.no_graph
                   else "eager launches") + ", fused step (2L+1 kernels per step + one loss-bookkeeping launch per replay)")

        def run_steps(n, full_last=True):   # whole replays of steps_per_replay steps, single-step replays for the remainder
            n_loss[0] += n
            stepper.run(n, full_last=full_last)
    else:
        acc0 = torch.zeros((), device=dev)

        def drawn_loss():
            loss = model.loss_drawn(edges_dev, B, 42, 0, step_dev=batch_counter, advance=True)
            acc0.add_(loss.detach())
            return loss

        graphed = None
        if not args.no_graph and not args.torch_adam:
            graphed = GraphedTrainStep(model, opt, batch_fn=lambda: (), loss_fn=drawn_loss)
            acc0.zero_()
        launch = ("captured hipGraph per step" if graphed is not None else "eager launches") + ", autograd step"

        def run_steps(n, full_last=True):
            for _ in range(n):
                n_loss[0] += 1
                if graphed is not None:
                    graphed()
                    continue
                opt.zero_grad(set_to_none=True)
                loss = drawn_loss()
                loss.backward()
                opt.step()

    run_steps(warmup, full_last=False)
    torch.cuda.synchronize()
    loss_sum.zero_()
    if not fused:
        acc0.zero_()
    n_loss[0] = 0
    # The timed region is a block of EXACTLY `steps` steps between two synchronisations.  A block of the driver's 20
    # sports steps is 2.5 ms (two graph replays): too short to quote alone, so the block is repeated until >= 50 ms
    # have been timed and the MEDIAN block is the one reported; every block's ms/step is in `ms_per_step_blocks`.
    # A step built with the LIGHT forward (large graphs: optim.FusedLightGCNStep.light) computes the propagated table in the
    # rows its loss reads; the timed steps are consecutive steps INSIDE an epoch, as the training loop runs them -- the one
    # step per epoch that precedes the evaluation and leaves the whole table behind is timed separately below.
    blocks = []
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(steps, full_last=False)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
        if sum(blocks) >= MIN_TIMED_S or len(blocks) >= 64:
            break
    dt = float(np.median(blocks))
    ms_per_step = dt / steps * 1e3
    forward_note = None
    if fused and stepper.light:
        n_full = 3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_full):
            n_loss[0] += 1
            stepper(full_result=True)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) / n_full * 1e3
        forward_note = {"timed_steps": "light", "ms_per_step_full_result": full_ms, "full_steps_per_epoch": 1,
                        "steps_per_epoch": max(E // B, 1),
                        "what": "a light step draws its batch first and runs the last two forward propagates over the row lists of "
                                "N1 (the batch rows' 1-hop image) and R0 (the batch rows) only: loss, gradient and updated tables are "
                                "the full step's bit for bit (tests/test_gpu_round4.py); the ONE step of an epoch that precedes the "
                                "evaluation computes every row (model.result for gene_ranklist, the reference's stale-result quirk) "
                                "and costs ms_per_step_full_result.  `value` keeps counting the reference step's 2 L E_dir messages "
                                "per step (the work of Model/LightGCN.py's step that this step replaces), not the smaller number of "
                                "rows a light step gathers"}
    loss_mean = (float(loss_sum.item()) if fused else float(acc0.item())) / max(n_loss[0], 1)
    msgs_per_step = 2 * L * e_dir

    # --- SpMM roofline: the step's own SpMM launches (same graph, operands, epilogues) replayed IN THE STEP'S ORDER -----
    csr = model.graph
    N = csr.n_rows
    w = 1.0 / (L + 1)
    x0 = model._flat.detach()
    if fused and L >= 3 and os.environ.get("CHAOREC_BENCH_CHAIN_BUFFERS", "step") == "step":
        # the step's OWN layer / gradient buffers (between steps they hold nothing anyone reads; G is all-zero by the
        # step's contract): the replayed launches then touch exactly the memory the step's launches touch
        b0, b1, fin, G = stepper.fbuf[0], stepper.fbuf[1], stepper.final, stepper.G
    else:
        b0, b1, fin, G = (torch.empty_like(x0) for _ in range(4))
        G.zero_()
    use_mean = L <= ops.mean_terms_limit(D)
    adam_call = None
    plain, whole, sparse_calls, src = [], [], [], x0
    xs = [x0]
    # The fused step runs some of its propagates over ROW LISTS / with gated gathers (optim.FusedLightGCNStep: the batch
    # gradient G has 3 B non-zero rows R0, its 1-hop image N1 is a part of the graph; a light step also restricts its last two
    # FORWARD propagates to N1 / R0): they are replayed as the step issues them, over the G, bitmaps and lists ONE real batch
    # leaves behind -- and they are not `plain` launches of the dense kernel (their model bytes are not the dense kernel's:
    # the roofline below is the dense launches').
    sparse_bwd = bool(fused and getattr(stepper, "sparse_bwd", False) and G is stepper.G)
    light = bool(sparse_bwd and getattr(stepper, "light", False))
    if light:
        ops.batch_rows(stepper.ids, stepper.bits[0], U, stepper._list0, stepper._list0_n, edges=edges_dev, hist=model.hist,
                       num_user=U, num_item=I, seed=4242, step=7)
        ops.bpr_fwd_bwd(stepper.final, U, G, B, ops.VARIANT_LOG_SIGMOID_EPS, reg, stepper.coef, stepper.ws, stepper.ids,
                        num_user=U, num_item=I)
    elif sparse_bwd:
        ops.bpr_fwd_bwd(stepper.final, U, G, B, ops.VARIANT_LOG_SIGMOID_EPS, reg, stepper.coef, stepper.ws, stepper.ids,
                        edges=edges_dev, hist=model.hist, num_user=U, num_item=I, seed=4242, step=7, row_bits=stepper.bits[0])

    def expand_n1():
        return ops.expand_row_bits(csr, stepper.bits[0], stepper.bits[1], stepper._row_list, stepper._list_n)

    if light:
        for l in range(L - 2):                       # dense layers 1 .. L-2
            y = b0 if l % 2 == 0 else b1
            plain.append((lambda src=src, y=y: ops.spmm_raw(csr, src, y=y), csr, D))
            src = y
            xs.append(y)
        whole += plain
        y = b0 if (L - 2) % 2 == 0 else b1
        sparse_calls.append(("forward layer L-1 over N1's row list (expansion of R0 included)", (lambda src=src, y=y: (
            expand_n1(), ops.spmm_rowlist_raw(csr, src, y, stepper._row_list, stepper._list_n, long_rows=stepper._long)), csr, D)))
        xs.append(y)
        sparse_calls.append(("forward layer L over R0's row list + layer mean", (lambda xs=list(xs): ops.spmm_rowlist_raw(
            csr, xs[-1], None, stepper._list0, stepper._list0_n, mean_out=fin, mean_terms=xs, mean_w=w, long_rows=stepper._long), csr, D)))
        whole += [c for _, c in sparse_calls]
    else:
        for l in range(L - 1 if use_mean else L):    # forward propagates (ops.forward_layers)
            y = b0 if l % 2 == 0 else b1
            if use_mean:
                plain.append((lambda src=src, y=y: ops.spmm_raw(csr, src, y=y), csr, D))
            else:
                last = l == L - 1
                plain.append((lambda src=src, y=y, l=l, last=last: ops.spmm_raw(
                    csr, src, y=None if last else y, acc=fin, acc_init=x0 if l == 0 else None, acc_w=w, want_y=not last), csr, D))
            src = y
            xs.append(y)
        whole += plain
        if use_mean:                                 # the last forward propagate with the whole layer mean in its epilogue
            whole.append((lambda: ops.spmm_mean_raw(csr, xs[-1], xs, w, fin), csr, D))
    n_epilogue = (1 if (use_mean and not light) else 0)
    # backward: g_l = A g_{l+1} + w G
    g, alpha = G, w
    for l in range(L - 1):
        y = b0 if l % 2 == 0 else b1
        if sparse_bwd and l == 0 and L >= 3:
            sparse_calls.append(("backward propagate 1 over N1's row list" + ("" if light else " (expansion of R0 included)"),
                                 (lambda g=g, y=y, alpha=alpha: (
                                     None if light else expand_n1(),
                                     ops.spmm_rowlist_raw(csr, g, y, stepper._row_list, stepper._list_n, alpha=alpha, z=G, beta=w,
                                                          src_bits=stepper.bits[0], z_bits=stepper.bits[0],
                                                          long_rows=stepper._long)), csr, D)))
            whole.append(sparse_calls[-1][1])
        elif sparse_bwd and l < 2:
            sparse_calls.append(("backward propagate %d, every row, gathers gated by the source's bitmap" % (l + 1),
                                 (lambda g=g, y=y, alpha=alpha, l=l: ops.spmm_rowsparse_raw(
                                     csr, g, y, alpha=alpha, z=G, beta=w, src_bits=stepper.bits[l], z_bits=stepper.bits[0]), csr, D)))
            whole.append(sparse_calls[-1][1])
        else:
            plain.append((lambda g=g, y=y, alpha=alpha: ops.spmm_raw(csr, g, y=y, alpha=alpha, z=G, beta=w), csr, D))
            whole.append(plain[-1])
        g, alpha = y, 1.0
    if fused and D <= 256:                           # the last backward propagate with the Adam epilogue, on copies
        pc, mc, vc = x0.clone(), torch.zeros_like(x0), torch.zeros_like(x0)
        bc = torch.tensor([0.1, 0.0316], device=dev)
        adam_call = (lambda: ops.spmm_adam_raw(csr, g, pc, mc, vc, bc, 1e-3, (0.9, 0.999), 1e-8, 0.0, alpha=alpha, z=G,
                                               beta=w, clear_z=False), csr, D)
        whole.append(adam_call)
        n_epilogue += 1
    heavy_graph = csr.nnz > 50_000_000
    avg_spmm_ms, model_bytes, compulsory = time_spmm_chain(plain, passes=3 if heavy_graph else 5)
    whole_ms, _, _ = time_spmm_chain(whole, passes=3 if heavy_graph else 5)
    whole_ms *= len(whole)                           # all SpMM-family launches of ONE step, boundaries included
    n_plain, n_whole = len(plain), len(whole)
    sparse_ms = None
    light_launches = None
    if sparse_calls:
        sparse_each = [time_spmm_chain([c], passes=3 if heavy_graph else 5)[0] for _, c in sparse_calls]
        sparse_ms = float(np.mean(sparse_each))
        if light and adam_call is not None:
            light_launches = light_step_accounting(
                ops, stepper, csr, U, I, D, L, edges_dev, model.hist, expand_n1,
                times_ms=dict(dense=avg_spmm_ms, sparse=sparse_each, adam=time_spmm_chain([adam_call], passes=3)[0]),
                names=[n for n, _ in sparse_calls], whole_ms=whole_ms)
        G.zero_()                                    # (the step's contract: all-zero between steps, bitmaps clear)
        stepper._bits_all.zero_()
    del b0, b1, fin, G, plain, whole
    achieved = model_bytes / (avg_spmm_ms * 1e-3) / 1e9
    table_mb = N * D * 4 / 1e6
    traffic = kernel_only_us = None
    tpath = os.path.join(ROOT, "profiles", f"spmm_traffic_{dataset}_d{D}.json")
    traffic_note = "no PMC file for this workload under profiles/"
    if os.path.exists(tpath):
        # PMC traffic is collected by separate rocprofv3 --pmc passes (tools/collect_profiles.py), not in this run: it
        # is only quoted when the file was measured on the spmm.hip this run was built from
        try:
            tj = json.load(open(tpath))
            if tj.get("spmm_hip_sha256") == spmm_source_hash():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_note = "from " + os.path.relpath(tpath, ROOT) + " (same spmm.hip)"
                kernel_only_us = tj.get("kernel_avg_us_rocprofv3")
            else:
                traffic_note = os.path.relpath(tpath, ROOT) + " was measured on a different spmm.hip: dropped"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": spmm_kernel_name(D), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": model_bytes,
                "avg_launch_us": avg_spmm_ms * 1e3, "compulsory_bytes_per_launch": compulsory,
                "launches_per_step": n_plain, "timing": CHAIN_TIMING_NOTE,
                "spmm_launches_of_one_step": {"launches": n_whole, "us": whole_ms * 1e3,
                                              "share_of_ms_per_step": whole_ms / ms_per_step,
                                              "what": f"the step's {n_whole} SpMM-family launches ({n_plain} dense plain"
                                                      + (f" + {len(sparse_calls)} over row lists / with gated gathers" if sparse_calls else "")
                                                      + f" + {n_epilogue} with the layer-mean / Adam epilogue) replayed in order as one hipGraph"
                                                      + ("; a LIGHT step (forward restricted to the rows the loss reads)" if light else "")},
                "note": ("embedding table (%.1f MB) is Infinity-Cache resident at this config: the fraction is against "
                         "the HBM peak but the bytes are served on-die (SURVEY 8(d) reporting rule)" % table_mb)
                if table_mb < 256 else
                ("embedding table %.0f MB, beyond the 256 MiB Infinity Cache: HBM-bound regime; `achieved` counts the "
                 "no-reuse CSR model bytes, `traffic` (when present) the measured FETCH_SIZE+WRITE_SIZE bytes" % table_mb)}
    if sparse_ms is not None:
        roofline["rowsparse_launches"] = {
            "kernels": [spmm_kernel_name(D, rowsparse=True), "spmm_rowlist_kernel"], "per_step": len(sparse_calls),
            "each_us": {name: t * 1e3 for (name, _), t in zip(sparse_calls, sparse_each)},
            "note": "propagates whose operands or results live in the batch's frontier (R0 = the 3 B batch rows, N1 = their 1-hop image): "
                    "over the LIST of N1's / R0's rows (chaorec_expand_row_bits + chaorec_spmm_csr_rowlist_f32) or, where every row has "
                    "to be written, as the ordinary launch with its gathers gated by the source's bitmap "
                    "(chaorec_spmm_csr_rowsparse_f32) -- the same sums bit for bit; timed over the G, bitmaps and lists one real batch "
                    "left"}
    if light_launches is not None:
        roofline["light_step_launches"] = light_launches
    if traffic:
        roofline["traffic_GBps"] = traffic / (avg_spmm_ms * 1e-3) / 1e9
    if kernel_only_us:
        # the same kernel's average duration in the rocprofv3 kernel trace of the same command (begin -> end of the kernel,
        # no launch boundary), from the committed profile this run's spmm.hip was measured with -- NOT measured in this run
        roofline["kernel_only_rocprofv3"] = {"avg_us": kernel_only_us,
                                             "frac": model_bytes / (kernel_only_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                             "source": os.path.relpath(tpath, ROOT),
                                             "launch_boundary_us": avg_spmm_ms * 1e3 - kernel_only_us}
    if getattr(args, "spmm_only", False):
        return dict(spmm_only=True, dataset=dataset, data=data_kind, U=U, I=I, E=E, e_dir=e_dir, D=D, L=L, B=B,
                    ms_per_step=ms_per_step, value=msgs_per_step / (dt / steps), msgs_per_step=msgs_per_step,
                    loss_mean=loss_mean, launch=launch, roofline=roofline, build_s=build_s)

    # --- full-rank evaluation ---------------------------------------------------------------------------------
    # Two states of the same call.  COLD: no thresholds carried (the first evaluation of a run): sampled thresholds,
    # timed on ops.score_topk.  STEADY: the evaluation loop itself (train_and_evaluate.py:655-659 ranks once per epoch)
    # through the PRODUCT ENTRY, model.gene_ranklist(to_cpu=False) -- ranking.RankState decides hints / light mode /
    # back-off exactly as it does in a training run, nothing of it is re-implemented here: STEADY_EVALS epochs of
    # training each followed by its evaluation, then `reps_rank` more epochs whose evaluations are the timed calls
    # (HIP events around the call; the median is reported, every call's time and queue lengths are in the line).
    epoch_steps = max(E // B, 1)

    def time_calls(fn, n):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in ev:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        return float(np.median([s.elapsed_time(e) for s, e in ev]))

    def sweep_alone(ue, ie):
        """pack + the bf16 sweep over all users of one workspace-sized user range, timed alone: the FRONT phase of a call
        with carried thresholds (chaorec_score_topk_hinted_f32, CHAOREC_SCORE_FRONT), the thresholds being those a cold
        call over the same range just left.  -> {users, ms, TFLOP/s, frac} or None where the call takes no prefilter."""
        from chaorec_amd import _lib
        lib = _lib.load()
        u = min(U, 524288)
        if D not in (64, 128) or I < 4096 or lib.chaorec_score_topk_workspace_bytes(u, I, 50, D) > (24 << 30):
            return None
        sub, hsub = ue[:u].contiguous(), (model.hist[0][:u + 1], model.hist[1])
        hint = torch.empty(u, dtype=torch.float32, device=dev)
        ops.score_topk(sub, ie, hsub, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False)
        nb = lib.chaorec_score_topk_workspace_bytes(u, I, 50, D)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        idx = torch.empty((u, 50), dtype=torch.int64, device=dev)
        val = torch.empty((u, 50), dtype=torch.float32, device=dev)
        ms = time_calls(lambda: ops._score_call(lib, sub, ie, hsub, 1e-6, 50, U, 0, hint, True, 110, False, None, idx, val, ws,
                                                nb, phase=ops.SCORE_FRONT), 3 if heavy else 5)
        tf = 2.0 * u * I * D / (ms * 1e-3) / 1e12
        # the whole call with those carried thresholds (no sampling pass, ~2.2 K candidates per user): what an evaluation costs
        # when the tables did not move since the previous one -- the floor of a steady-state call, NOT a measured epoch-to-epoch
        # call (sports' headline is one; an epoch of this graph is minutes)
        ms_c = time_calls(lambda: ops.score_topk(sub, ie, hsub, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True), 3)
        tf_c = 2.0 * u * I * D / (ms_c * 1e-3) / 1e12
        return {"users": u, "ms": ms, "TFLOPs": tf, "frac": tf / BF16_MFMA_PEAK_TFLOPS,
                "what": "pack + score_sweep_bf16_kernel over this many users (thresholds carried from a cold call on the same "
                        "tables), the call's FRONT phase timed alone with HIP events",
                "carried_thresholds_same_tables": {"users": u, "ms": ms_c, "frac": tf_c / BF16_MFMA_PEAK_TFLOPS,
                                                   "what": "whole call, thresholds carried from a call on the SAME tables (floor of "
                                                           "a steady-state call; not the headline)"}}

    def time_ranklist(with_steady):
        res = model.result.detach()
        ue, ie = res[:U], res[U:U + I]
        st, out = {}, {}
        with torch.no_grad():
            out["cold_ms"] = time_calls(lambda: ops.score_topk(ue, ie, model.hist, 1e-6, 50, id_offset=U), reps_rank)
            ops.score_topk(ue, ie, model.hist, 1e-6, 50, id_offset=U, stats=st)
            out["cold_st"] = st
            out["sweep_alone"] = sweep_alone(ue, ie)
            if with_steady:
                state = ranking.state_of(model)
                model.gene_ranklist(to_cpu=False)            # the run's first evaluation: leaves thresholds behind
                for _ in range(STEADY_EVALS):                # epochs of training, each followed by its evaluation
                    run_steps(epoch_steps)
                    model.gene_ranklist(to_cpu=False)
                calls = []
                for _ in range(max(reps_rank, 5)):
                    run_steps(epoch_steps)                   # (queued ahead of the call: the events see the device time)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    model.gene_ranklist(to_cpu=False)
                    e.record()
                    torch.cuda.synchronize()
                    calls.append({"ms": s.elapsed_time(e), "hinted": bool(state.last_hinted),
                                  "light": bool(state.last_light),
                                  "queues_retry_exact_wide_retry2exact": state.counters.tolist()})
                out["steady_ms"] = float(np.median([c["ms"] for c in calls]))
                st2 = {"through": "model.gene_ranklist(to_cpu=False) (ranking.RankState decides hints / light mode)",
                       "timed_calls": calls}
                out["steady_st"] = st2
            # the reference contract: a LongTensor on the CPU (Model/LightGCN.py:162) -- wall time incl. the D2H copy,
            # through the model's own gene_ranklist (carried thresholds, as the evaluation loop calls it)
            out["host_ms"] = None
            if not heavy:
                model.gene_ranklist()
                t1 = time.perf_counter()
                for _ in range(3):
                    model.gene_ranklist()
                out["host_ms"] = (time.perf_counter() - t1) / 3 * 1e3
        return out

    steps_done = warmup + steps * len(blocks)
    # (BASELINE configs[4] whole is 5 PFLOP per ranking call -- seconds: one timed call, and no 4 GB rank list on the host)
    heavy = 2.0 * U * I * D > 1e15
    if heavy:
        reps_rank = 1
    if fused:
        run_steps(1)                        # (the chain above wrote into the step's buffers; a full step leaves model.result)
        steps_done += 1
    early = time_ranklist(False)
    extra = trained_steps - steps_done - (STEADY_EVALS + 1) * epoch_steps
    if extra > 0 and (extra + (STEADY_EVALS + 1) * epoch_steps) * ms_per_step < 10_000:
        run_steps(extra)
        rk = time_ranklist(True)
        state = (f"after {trained_steps} training steps ({extra + (STEADY_EVALS + 1) * epoch_steps} of them untimed, past the measured "
                 f"ones); steady = evaluation number {STEADY_EVALS + 2} of a run that evaluates once per epoch ({epoch_steps} steps): "
                 f"thresholds carried from the evaluation one epoch earlier")
    else:
        # (no trained state within the run's budget -- an epoch of the config-5 shard is 24 k steps --: the cold call
        #  only; thresholds carried across the first steps of training are stale by construction, ranking.RankState
        #  backs off from them)
        rk = early if heavy else time_ranklist(False)
        state = f"after {steps_done} training steps; cold thresholds"
    score_ms = rk.get("steady_ms", rk["cold_ms"])
    early_ms, early_st = early["cold_ms"], early["cold_st"]
    st, host_ms = rk.get("steady_st", rk["cold_st"]), rk["host_ms"]
    tf = 2.0 * U * I * D / (score_ms * 1e-3) / 1e12
    performed = None
    if light_launches is not None:
        performed = {"messages_gathered_per_step": light_launches["messages_gathered_per_step"],
                     "value_performed": light_launches["messages_gathered_per_step"] / (dt / steps),
                     "what": "`value` counts the REFERENCE step's 2 L E_dir directed-edge messages per step (the work of "
                             "Model/LightGCN.py's step that this step replaces, bit for bit); value_performed counts the source "
                             "rows a light step actually gathers in its six SpMM-family launches"}
    return dict(dataset=dataset, data=data_kind, U=U, I=I, E=E, e_dir=e_dir, D=D, L=L, B=B, ms_per_step=ms_per_step,
                value=msgs_per_step / (dt / steps), msgs_per_step=msgs_per_step, loss_mean=loss_mean, launch=launch,
                performed=performed, roofline=roofline, score_ms=score_ms, score_state=state, early_ms=early_ms, early_st=early_st,
                cold_ms=rk["cold_ms"], cold_st=rk["cold_st"], steady="steady_ms" in rk,
                score_st=st, score_tf=tf, sweep_alone=rk.get("sweep_alone") or early.get("sweep_alone"),
                host_rank_ms=host_ms, edges=edges, reg=reg, table_mb=table_mb,
                build_s=build_s, blocks_ms_per_step=[b / steps * 1e3 for b in blocks], forward=forward_note)


def scoring_roofline(r):
    sorted_tbl = r["I"] >= 524288 and os.environ.get("CHAOREC_PF_CLS_MIN_ITEMS", "") in ("",)     # (score_topk.hip: use_sorted_table)
    return {"bound": "mfma", "kernel": f"score_sweep_bf16_kernel<{r['D']},{3 if r['D'] <= 64 else 2},{'true' if sorted_tbl else 'false'}> "
                                       f"(+ {'norm-class sort, ' if sorted_tbl else ''}pack, sample, select/re-score)",
            "achieved": r["score_tf"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": r["score_tf"] / BF16_MFMA_PEAK_TFLOPS, "frac_of_f32_mfma_peak": r["score_tf"] / F32_MFMA_PEAK_TFLOPS,
            "sweep_only_frac": (r.get("sweep_alone") or {}).get("frac"), "sweep_alone": r.get("sweep_alone"),
            "prefilter": r["score_st"],
            "note": "2*U*I*D over the whole gene_ranklist call.  The [U,I] sweep runs on the bf16 MFMA pipe "
                    "(v_mfma_f32_32x32x16_bf16, 2.5 PF dense peak) as a prefilter with a proven error bound, the top-K "
                    "is ranked on exact fp32 re-scores (bit-identical to the fp32 route); see DESIGN.md 3.3"}


def main_single(args, dev):
    from chaorec_amd import _lib
    _lib.ensure_built()
    _lib.load()
    D = args.dim
    r = measure_single_gpu(args, args.dataset, D, args.steps, args.warmup, dev,
                           0 if args.no_trained_state else TRAINED_STEPS, synthetic=args.synthetic)
    U, I = r["U"], r["I"]
    if r.get("spmm_only"):
        print(json.dumps({"metric": f"GCN edges/sec, dim={D} (--spmm-only: no ranking)", "value": r["value"],
                          "unit": "directed-edge messages/s (fwd+bwd SpMM of the train step)", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "dtype": "f32",
                          "data": r["data"], "config": {"workload": f"LightGCN train step on the {r['data']} {args.dataset} "
                                                                    f"graph (U={U}, I={I}, E_dir={r['e_dir']}), dim={D}",
                                                        "launch": r["launch"], "host_build_seconds": r["build_s"]},
                          "roofline": r["roofline"], "loss_mean": r["loss_mean"]}), flush=True)
        return
    out = {
        "metric": f"GCN edges/sec + full-rank users-scored/sec, dim={D}",
        "value": r["value"], "unit": "directed-edge messages/s (fwd+bwd SpMM of the train step)",
        "users_scored_per_s": U / (r["score_ms"] * 1e-3),
        "users_scored_per_s_cold": U / (r["cold_ms"] * 1e-3),
        "users_scored_per_s_incl_d2h": U / (r["host_rank_ms"] * 1e-3) if r["host_rank_ms"] else None,
        "users_scored_per_s_right_after_timed_steps": U / (r["early_ms"] * 1e-3),
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "timed_blocks": {"blocks_of_steps": len(r["blocks_ms_per_step"]), "ms_per_step_min": min(r["blocks_ms_per_step"]),
                         "ms_per_step_median": r["ms_per_step"], "ms_per_step_max": max(r["blocks_ms_per_step"]),
                         "note": f"the --steps block repeated until >= {MIN_TIMED_S * 1e3:.0f} ms were timed; value and "
                                 "ms_per_step are the median block"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": r["data"],
        "config": {"workload": f"LightGCN train step on the {'real' if r['data'] == 'real' else 'synthetic'} "
                               f"{args.dataset} graph (U={U}, I={I}, E_dir={r['e_dir']}), dim={D}, n_layers={r['L']}, "
                               f"batch={r['B']}; gene_ranklist top-50 over all users",
                   "messages_per_step": r["msgs_per_step"], "gene_ranklist_ms": r["score_ms"],
                   "gene_ranklist_mode": ("steady state: per-user thresholds carried from the evaluation one epoch earlier"
                                          if r["steady"] else "cold: sampled thresholds"),
                   "gene_ranklist_ms_cold": r["cold_ms"], "prefilter_cold": r["cold_st"],
                   "gene_ranklist_ms_incl_d2h_wall": r["host_rank_ms"],
                   "gene_ranklist_state": r["score_state"], "gene_ranklist_ms_right_after_timed_steps": r["early_ms"],
                   "prefilter_right_after_timed_steps": r["early_st"], "launch": r["launch"],
                   "optimizer": "torch.optim.Adam" if args.torch_adam else
                   ("Adam in the last backward SpMM's epilogue (chaorec_spmm_csr_adam_f32)" if "fused" in r["launch"]
                    else "FusedAdam (chaorec_adam_step_f32)"),
                   "parallelism": "single GPU", "host_build_seconds": r["build_s"]},
        "roofline": r["roofline"], "roofline_scoring": scoring_roofline(r), "loss_mean": r["loss_mean"],
        **({"forward": r["forward"]} if r.get("forward") else {}),
        **(r["performed"] if r.get("performed") else {}),
    }
    # (the second dominant kernel family inside the object the driver keeps: the all-items scoring's share of the MFMA peak)
    sr = out["roofline_scoring"]
    out["roofline"]["scoring"] = {"bound": "mfma", "kernel": sr["kernel"], "achieved": sr["achieved"], "peak": sr["peak"],
                                  "unit": sr["unit"], "frac": sr["frac"], "sweep_only_frac": sr.get("sweep_only_frac"),
                                  "gene_ranklist_ms": r["score_ms"]}
    edges, reg = r["edges"], r["reg"]
    # --- the HBM-bound regime in the same run: one GPU's share of BASELINE configs[4] -----------------------------
    if not args.no_hbm_regime and args.dataset not in ("config5_shard", "config5"):
        del r
        torch.cuda.empty_cache()
        h = measure_single_gpu(args, "config5_shard", 128, args.hbm_steps, 3, dev, 0, reps_rank=3)
        out["hbm_regime"] = {
            "workload": f"one GPU's share of BASELINE configs[4]: synthetic bipartite graph U={h['U']}, I={h['I']}, "
                        f"E_dir={h['e_dir']}, dim=128, n_layers={h['L']}, batch={h['B']} (embedding table "
                        f"{h['table_mb']:.0f} MB = {h['table_mb'] / 268.4:.1f}x the Infinity Cache)",
            "data": h["data"], "steps": args.hbm_steps, "ms_per_step": h["ms_per_step"], "value": h["value"],
            "unit": "directed-edge messages/s", "roofline": h["roofline"],
            "gene_ranklist_ms": h["score_ms"], "users_scored_per_s": h["U"] / (h["score_ms"] * 1e-3),
            "gene_ranklist_mode": "steady state" if h["steady"] else "cold: sampled thresholds",
            "gene_ranklist_ms_cold": h["cold_ms"],
            "roofline_scoring": scoring_roofline(h), "loss_mean": h["loss_mean"],
            **({"forward": h["forward"]} if h.get("forward") else {}),
            **(h["performed"] if h.get("performed") else {}),
        }
        del h
        torch.cuda.empty_cache()
        if not args.no_full_config5 and torch.cuda.get_device_properties(dev).total_memory > 200 * (1 << 30):
            # ... and the whole of configs[4] on this one GPU: the N = 1 anchor of that config's scaling curve
            f = measure_single_gpu(args, "config5", 128, args.full_steps, 2, dev, 0, reps_rank=1)
            out["config5_whole_on_one_gpu"] = {
                "workload": f"BASELINE configs[4] whole: synthetic bipartite graph U={f['U']}, I={f['I']}, E_dir={f['e_dir']}, "
                            f"dim=128, n_layers={f['L']}, batch={f['B']} (embedding table {f['table_mb']:.0f} MB; generated "
                            f"and laid out on the device)",
                "data": f["data"], "steps": args.full_steps, "ms_per_step": f["ms_per_step"], "value": f["value"],
                "unit": "directed-edge messages/s", "roofline": f["roofline"], "host_build_seconds": f["build_s"],
                "gene_ranklist_ms_cold": f["cold_ms"], "users_scored_per_s_cold": f["U"] / (f["cold_ms"] * 1e-3),
                "roofline_scoring": scoring_roofline(f), "loss_mean": f["loss_mean"],
                **({"forward": f["forward"]} if f.get("forward") else {}),
                **(f["performed"] if f.get("performed") else {}),
            }
            del f
            torch.cuda.empty_cache()
    if not args.no_models and args.dataset == "sports":
        # BASELINE configs[3] / [2] in the driver-run line: the models' captured train step + gene_ranklist, single process
        out["models"] = {}
        for name in ("MMGCN", "FREEDOM"):
            try:
                m = measure_model(args, name, 1, 0, dev, False, None, steps=200, warmup=20)
                out["models"][name] = {k: m[k] for k in ("ms_per_step", "timed_blocks", "value", "unit", "steps", "warmup", "data",
                                                         "users_scored_per_s_incl_d2h", "config") if k in m}
            except Exception as exc:      # noqa: BLE001 -- a sub-record must not take the headline with it
                out["models"][name] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()
    if not args.no_cpu_baseline and edges is not None:
        out["cpu_baseline"] = cpu_baseline(edges, U, I, D, args.n_layers, args.batch, reg, args.cpu_seconds)
    print(json.dumps(out), flush=True)


def init_ranks(local_rank, sharded=True):
    """This rank's device + the process group (RCCL = backend "nccl"; CHAOREC_DIST_BACKEND=gloo for ranks that share a
    device).  -> (dev, backend)."""
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    local_rank %= torch.cuda.device_count()     # (lets a 1-GPU box exercise the N>1 code path with gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("CHAOREC_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    return dev, backend


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
                                "frontier exchanges); the step before an evaluation is a full one.  `value` keeps counting the "
                                "reference step's 2 L E_dir messages per step"}
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
               table_mb=(U + I) * D * 4 / 1e6, forward=forward_note,
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
        "value": head["value"], "unit": "directed-edge messages/s (fwd+bwd SpMM of the train step)",
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

    # Sub-records, under a watchdog: a collective that cannot make progress in a sub-record must not take the headline
    # numbers (measured above) with it -- rank 0 then prints the line without the unfinished ones and the job ends.
    import threading
    finished = threading.Event()

    def give_up():
        if not finished.wait(float(os.environ.get("CHAOREC_SUBRECORD_TIMEOUT_S", "900"))):
            if rank == 0:
                out.setdefault("hbm_regime", {"error": "sub-records did not finish in time"})
                flush_c_stdout()
                print(json.dumps(out), flush=True)
            os._exit(0 if rank == 0 else 0)

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
                "data": h["data"], "steps": args.hbm_steps, "ms_per_step": h["ms_per_step"], "value": h["value"],
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
        flush_c_stdout()
        print(json.dumps(out), flush=True)


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
        flush_c_stdout()
        print(json.dumps(out), flush=True)


def flush_c_stdout():
    """RCCL writes its version banner through C stdio, which would otherwise drain at exit, AFTER the result: flush C stdout
    first so that the JSON object is the last line (stdout only -- an fflush(NULL) from here hung under rocprofv3)."""
    import ctypes
    libc = ctypes.CDLL(None)
    try:
        libc.fflush(ctypes.c_void_p.in_dll(libc, "stdout"))
    except (ValueError, OSError):
        pass


def visible_gpu_count(sysfs_root="/sys/class/kfd/kfd/topology/nodes"):
    """Devices this process could use, WITHOUT touching the GPU runtime (the launcher must not initialise it: it starts
    children, and torch.cuda.device_count() goes through hipGetDeviceCount -- an HSA init -- on ROCm).  The kernel driver's
    own topology: one directory per node under /sys/class/kfd/kfd/topology/nodes, a GPU is a node whose `properties` show
    simd_count > 0 (CPUs have 0).  A visibility list in the environment (ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES) caps the
    count.  No readable topology (a box without the driver, this build container) means 0."""
    n = 0
    try:
        for node in sorted(os.listdir(sysfs_root)):
            try:
                with open(os.path.join(sysfs_root, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            except OSError:
                continue                      # (a node this user may not read: not a device it can use)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one child process per GPU with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what torch.distributed.run would hand them -- BEFORE this
    process makes any GPU call (a process that initialised the GPU must not exec another program; this one only counts
    devices and waits).  Rank 0's stdout is relayed line by line (its last line is the JSON result), the other ranks'
    stdout goes to stderr.  Any rank ending non-zero ends the job: the others are terminated BY PID and the launcher
    exits with that code.  On a box with fewer devices than ranks the ranks share devices (LOCAL_RANK modulo the count)
    and, unless CHAOREC_DIST_BACKEND says otherwise, exchange through gloo: RCCL wants one device per rank -- the line
    then says `multi_rank_rccl_measured: false`."""
    import signal
    import subprocess
    import threading
    n = args.gpus
    n_dev = visible_gpu_count()
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(os.environ.get("MASTER_PORT") or _free_port()), CHAOREC_BENCH_SELF_LAUNCHED="1",
               CHAOREC_BENCH_VISIBLE_GPUS=str(n_dev))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL / shared CUDA tensors across processes
    if n_dev < n and "CHAOREC_DIST_BACKEND" not in env:
        env["CHAOREC_DIST_BACKEND"] = "gloo"
        print(f"[bench launcher] {n} ranks on {n_dev} visible GPU(s): ranks share devices, exchanges over gloo "
              f"(not an RCCL measurement)", file=sys.stderr, flush=True)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    deadline = time.time() + float(os.environ.get("CHAOREC_BENCH_TIMEOUT_S", "3000"))
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 128 - bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print("[bench launcher] timed out", file=sys.stderr, flush=True)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                       # (only ever the exact processes started above)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except (ProcessLookupError, PermissionError):
                    pass
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
        t.join(timeout=5)
    if rc:
        print(f"[bench launcher] a rank ended with {rc}", file=sys.stderr, flush=True)
    return rc


def launch_selftest(world, rank):
    """Child mode of the launcher's CPU test (tests/test_host_logic.py): rendezvous over gloo, sum the ranks, rank 0
    prints one JSON line.  CHAOREC_BENCH_SELFTEST_FAIL_RANK makes that rank exit 7 first (failure propagation)."""
    import torch.distributed as dist
    if os.environ.get("CHAOREC_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "sum": float(t.item()),
                          "local_rank": int(os.environ["LOCAL_RANK"]),
                          "self_launched": os.environ.get("CHAOREC_BENCH_SELF_LAUNCHED") == "1"}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_selftest:
        return launch_selftest(world, rank)
    # CHAOREC_FORCE_SHARDED=1: run the N>1 code path (sharded model, RCCL calls, graph capture of them) on one rank
    force_sharded = world == 1 and os.environ.get("CHAOREC_FORCE_SHARDED", "0") == "1"
    if args.model != "LightGCN":
        return main_model(args, world, rank, local_rank, force_sharded)
    if world == 1 and not force_sharded:
        assert torch.cuda.is_available(), "bench.py needs the MI355X"
        torch.cuda.set_device(0)
        return main_single(args, torch.device("cuda", 0))
    return main_sharded(args, world, rank, local_rank, force_sharded)


if __name__ == "__main__":
    main()
