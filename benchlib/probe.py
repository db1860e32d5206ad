"""First contact with a node at N > 1: which exchange mode / graph capture survives here (child processes, marker files)."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import ROOT


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
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(args.gpus), "--steps", "3", "--warmup", "1",
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
