"""Stress harness for the two-stream MMGCN steps: the divergence that showed once in ~10 runs inside the GPU
suite and never alone.

    python tools/stream_stress.py --model sharded --variants one,two,two_eager,two_rec --trials 25 --load busy

For every variant a FRESH worker process runs `--trials` trials of: build the model from the same seed, capture (or not) its
train step, run six steps, snapshot every parameter after every step.  Trial results are compared with the reference (the
one-stream captured step, run first in the same worker): a trial DIVERGES when a parameter differs from the reference beyond
the noise of the BPR backward's atomic adds (share of entries off by > 1e-5 above 1e-3, or a median difference above 1e-6 --
the assertion of tests/test_gpu_round4.py).  For a divergent trial the first step at which it happened and the tensors that
differ are printed.  `--load busy` keeps a second PROCESS launching GEMMs on the same GPU meanwhile (the condition under which
the divergence showed); `--load idle` only holds a second HIP context with 8 GB allocated.

Variants (the worker applies them through the environment / module switches); a `_atomic` suffix runs the variant AND its
reference with CHAOREC_BPR_ORDERED=0 (the fp32 atomic row adds of the BPR backward, the round-5 code path):
  one        one stream, captured (control: is the step reproducible at all?)
  two        two streams, captured (the mode under test)
  two_eager  two streams, eager launches (no hipGraph)
Per variant: `inexact` = trials with ANY bit of any parameter different from the reference at any step, `divergent` = trials
beyond the tolerance of the round-5 test.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_main(kind):
    import torch
    dev = torch.device("cuda:0")
    hold = torch.empty(8 << 30, dtype=torch.uint8, device=dev)          # noqa: F841 -- a second context with memory in use
    if kind == "idle":
        while True:
            time.sleep(1.0)
    a = torch.randn(4096, 4096, device=dev)
    b = torch.randn(4096, 4096, device=dev)
    while True:
        for _ in range(50):
            torch.mm(a, b)
        torch.cuda.synchronize()


def divergent(got, ref):
    import numpy as np
    off = []
    for n, r in ref.items():
        d = np.abs(got[n] - r)
        if not ((d > 1e-5).mean() <= 1e-3 and np.median(d) <= 1e-6):
            off.append((n, float(d.max()), float((d > 1e-5).mean())))
    return off


def worker_main(args):
    variant = args.worker
    if variant.endswith("_atomic"):
        os.environ["CHAOREC_BPR_ORDERED"] = "0"
        variant = variant[:-len("_atomic")]
    two = variant != "one"
    eager = variant == "two_eager"
    import numpy as np
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from chaorec_amd import graph, ops
    from chaorec_amd.Model import MMGCN
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    from chaorec_amd.synthetic import synthetic_interactions
    import importlib
    mm = importlib.import_module("chaorec_amd.Model.MMGCN")
    sharded = args.model == "sharded"
    if sharded:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(args.port)
        os.environ["CHAOREC_FORCE_COLLECTIVES"] = "1"
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from chaorec_amd import dist as cdist
    U, I, E, B = 6000, 2500, 40000, 512
    edges = synthetic_interactions(U, I, E, seed=3)
    uid = graph.user_item_dict_from_edges(edges)
    g = torch.Generator().manual_seed(4)
    v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 256, generator=g)

    def run(streams, capture):
        os.environ["CHAOREC_DIST_MMGCN_STREAMS"] = "1" if streams else "0"
        mm.BRANCH_STREAMS = bool(streams) if not sharded else True
        torch.manual_seed(21)
        full = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
        if sharded:
            shard = cdist.UserShard(edges, U, I, 1, 0, dev, self_loops=True)
            m = cdist.ShardedMMGCN(full, shard, dev)
            del full
            edges_dev = torch.from_numpy(shard.local_edges.astype(np.int64)).to(dev)
        else:
            m = full
            edges_dev = torch.from_numpy(np.stack([edges[:, 0], edges[:, 1]], 1).astype(np.int64)).to(dev)
        opt = FusedAdam(m.parameters(), lr=1e-3)
        # (GraphedTrainStep's three warm-up steps move the device batch counter on and do not put it back: an eager run
        #  starts where the captured replays start, so that both see the same batches)
        counter = torch.full((1,), 0 if capture else 3, dtype=torch.int64, device=dev)

        def draw():
            counter.add_(1)
            u, pos, neg = ops.draw_batch(edges_dev, m.hist, B, U, I, 42, 0, step_dev=counter, item_offset=U)
            return torch.stack((u, u), 1), torch.stack((pos, neg), 1)

        after = m.sync_grads if sharded else None
        snaps = []
        if capture:
            step = GraphedTrainStep(m, opt, batch_fn=draw, after_backward=after)
        else:
            def step():
                opt.zero_grad(set_to_none=True)
                loss = m.loss(*draw())
                loss.backward()
                if after is not None:
                    after()
                opt.step()
        for _ in range(6):
            step()
            snaps.append({n: p.detach().clone() for n, p in m.named_parameters()})
        torch.cuda.synchronize()
        del step, opt, m
        return [{n: t.cpu().numpy() for n, t in s.items()} for s in snaps]

    ref = run(False, True)
    bad = []
    inexact = 0
    t0 = time.time()
    for trial in range(args.trials):
        got = run(two, not eager)
        inexact += int(any(not np.array_equal(got[k][n], ref[k][n]) for k in range(6) for n in ref[k]))
        for k in range(6):
            off = divergent(got[k], ref[k])
            if off:
                bad.append({"trial": trial, "first_step": k + 1, "tensors": [o[0] for o in off][:8], "n_tensors": len(off),
                            "max": max(o[1] for o in off), "share_max": max(o[2] for o in off)})
                break
    print("RESULT " + json.dumps({"model": args.model, "variant": args.worker, "trials": args.trials, "inexact": inexact, "divergent": len(bad),
                                  "seconds": round(time.time() - t0, 1), "bad": bad[:6]}), flush=True)
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="sharded", choices=["sharded", "unsharded"])
    p.add_argument("--variants", default="one,two")
    p.add_argument("--trials", type=int, default=20)
    p.add_argument("--load", default="busy", choices=["none", "idle", "busy"])
    p.add_argument("--rounds", type=int, default=1, help="fresh worker processes per variant")
    p.add_argument("--worker", default=None, help=argparse.SUPPRESS)
    p.add_argument("--load-child", default=None, help=argparse.SUPPRESS)
    p.add_argument("--port", type=int, default=29621, help=argparse.SUPPRESS)
    args = p.parse_args()
    if args.load_child:
        return load_main(args.load_child)
    if args.worker:
        return worker_main(args)
    # (this launcher never touches the GPU: it only starts children and collects their lines)
    load = None
    if args.load != "none":
        load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load-child", args.load], start_new_session=True)
        time.sleep(8.0)
    port = 29621
    try:
        for variant in args.variants.split(","):
            for r in range(args.rounds):
                port += 1
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", variant, "--model", args.model,
                                      "--trials", str(args.trials), "--port", str(port)], capture_output=True, text=True,
                                     timeout=1500)
                lines = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
                print(lines[-1] if lines else f"RESULT {{\"variant\": \"{variant}\", \"error\": {json.dumps(out.stderr[-600:])}}}",
                      flush=True)
    finally:
        if load is not None:
            load.terminate()
            try:
                load.wait(timeout=10)
            except subprocess.TimeoutExpired:
                load.kill()


if __name__ == "__main__":
    main()
