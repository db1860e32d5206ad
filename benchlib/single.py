"""N = 1: the sports step, its per-launch rooflines, the ranking call, the config-5 sub-records."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import BF16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, MIN_TIMED_S, ROOT, STEADY_EVALS, TRAINED_STEPS, load_graph, spmm_kernel_name, spmm_model_bytes, spmm_source_hash  # noqa: F401
from .cpu import cpu_baseline  # noqa: F401
from .models import measure_model  # noqa: F401
from .record import emit


def performed_value(m):
    """A sub-record's `value` is the work its step PERFORMS (source rows gathered per second by the step's SpMM-family
    launches); the reference step's 2 L E_dir messages over the same time go beside it as value_reference_equivalent.
    (A full step performs the reference's messages: both are then the same number.)"""
    p = m.get("performed")
    if p:
        return {"value": p["value_performed"], "value_reference_equivalent": m["value"],
                "value_is": "source rows gathered per second by the light step's SpMM-family launches (work performed); "
                            "value_reference_equivalent divides the REFERENCE step's 2 L E_dir messages by the same time",
                "messages_gathered_per_step": p["messages_gathered_per_step"]}
    return {"value": m["value"], "value_reference_equivalent": m["value"]}


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
                                "and costs ms_per_step_full_result.  The record's `value` counts the source rows a light step gathers; "
                                "value_reference_equivalent the reference step's 2 L E_dir messages over the same time"}
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
                # (the MEDIAN launch of the captured run when the file has it: the stats average pools eager warm-up launches)
                kernel_only_us = tj.get("kernel_median_us_rocprofv3") or tj.get("kernel_avg_us_rocprofv3")
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
                     "what": "value_performed counts the source rows a light step actually gathers in its six SpMM-family "
                             "launches; the reference step's 2 L E_dir messages per step are the work of Model/LightGCN.py's step "
                             "that this step replaces bit for bit"}
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
                          "roofline": r["roofline"], "loss_mean": r["loss_mean"]}), flush=True)   # (a profiling mode, not the record)
        return
    out = {
        "metric": f"GCN edges/sec + full-rank users-scored/sec, dim={D}",
        **performed_value(r), "unit": "directed-edge messages/s (fwd+bwd SpMM of the train step)",
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
            "data": h["data"], "steps": args.hbm_steps, "ms_per_step": h["ms_per_step"], **performed_value(h),
            "unit": "directed-edge messages/s", "roofline": h["roofline"],
            "gene_ranklist_ms": h["score_ms"], "users_scored_per_s": h["U"] / (h["score_ms"] * 1e-3),
            "gene_ranklist_mode": "steady state" if h["steady"] else "cold: sampled thresholds",
            "gene_ranklist_ms_cold": h["cold_ms"],
            "roofline_scoring": scoring_roofline(h), "loss_mean": h["loss_mean"],
            **({"forward": h["forward"]} if h.get("forward") else {}),
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
                "data": f["data"], "steps": args.full_steps, "ms_per_step": f["ms_per_step"], **performed_value(f),
                "unit": "directed-edge messages/s", "roofline": f["roofline"], "host_build_seconds": f["build_s"],
                "gene_ranklist_ms_cold": f["cold_ms"], "users_scored_per_s_cold": f["U"] / (f["cold_ms"] * 1e-3),
                "roofline_scoring": scoring_roofline(f), "loss_mean": f["loss_mean"],
                **({"forward": f["forward"]} if f.get("forward") else {}),
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
                                                         "users_scored_per_s_incl_d2h", "config", "roofline") if k in m}
            except Exception as exc:      # noqa: BLE001 -- a sub-record must not take the headline with it
                out["models"][name] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()
    if not args.no_cpu_baseline and edges is not None:
        out["cpu_baseline"] = cpu_baseline(edges, U, I, D, args.n_layers, args.batch, reg, args.cpu_seconds)
    else:
        out["cpu_baseline"] = {"value": None, "reason": "--no-cpu-baseline" if args.no_cpu_baseline else
                               "no host copy of this graph's edge list (device-built synthetic graph)"}
    emit(out)
