#!/usr/bin/env python3
"""Which combinations of HIP streams, RCCL collectives and hipGraph capture survive on this stack?  (the sharded
MMGCN with its two modality branches on two streams core-dumps when its exchanges go through RCCL inside a captured step.)
Every variant runs in its own child process (a crash is an answer), on a 1-rank RCCL group:

    python3 tools/rccl_streams_repro.py            -> one line per variant: ok / rc
"""
import os
import subprocess
import sys

VARIANTS = ["eager_main", "eager_side", "eager_comm", "capture_main", "capture_side", "capture_comm", "capture_comm_async",
            "capture_two_issuers", "capture_side_async", "capture_two_issuers_async", "capture_autograd_side",
            "capture_autograd_side_async", "eager_autograd_side_async",
            # the side stream's collectives on a process group (an RCCL communicator) of their own
            "capture_two_groups_side", "capture_two_groups_side_async", "capture_autograd_two_groups_side",
            "capture_autograd_two_groups_side_async"]


def child(variant):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g2 = dist.new_group(backend="nccl") if "two_groups" in variant else None
    a, b = torch.ones(1 << 20, device=dev), torch.ones(1 << 20, device=dev)
    w = torch.randn(1024, 1024, device=dev)
    side, comm = torch.cuda.Stream(), torch.cuda.Stream()

    def work(x):
        for _ in range(4):
            x = (x.view(1024, 1024) @ w).view(-1) * 1e-3
        return x

    def exch(t, issuer, async_op, group=None):
        cur = torch.cuda.current_stream()
        if group is None and g2 is not None and cur == side:
            group = g2                          # (the side stream's collectives on their own communicator)
        if issuer is None or issuer == cur:
            h = dist.all_reduce(t, async_op=async_op, group=group)
            if async_op:
                h.wait()
            return
        issuer.wait_stream(cur)
        with torch.cuda.stream(issuer):
            h = dist.all_reduce(t, async_op=async_op, group=group)
            if async_op:
                h.wait()
        cur.wait_stream(issuer)

    class Exch(torch.autograd.Function):          # an exchange in the forward AND in the backward (a sharded propagate)
        @staticmethod
        def forward(ctx, x, async_op):
            ctx.async_op = async_op
            y = x.clone()
            exch(y, None, async_op)
            return y

        @staticmethod
        def backward(ctx, g):
            g = g.contiguous().clone()
            exch(g, None, ctx.async_op)
            return g, None

    wp = torch.nn.Parameter(torch.randn(1024, 1024, device=dev) * 1e-2)

    def autograd_body(async_op):
        cur = torch.cuda.current_stream()
        wp.grad = None
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            xv = Exch.apply((a.view(1024, 1024) @ wp), async_op)
            xv = (xv @ wp).sum()
        xt = Exch.apply((b.view(1024, 1024) @ wp), async_op)
        xt = (xt @ wp).sum()
        cur.wait_stream(side)
        loss = xv + xt
        loss.backward()
        return wp.grad

    def body():
        cur = torch.cuda.current_stream()
        if "autograd" in variant:
            return autograd_body(variant.endswith("async"))
        if variant.endswith("main"):
            x = work(a)
            exch(x, None, False)
            return work(x)
        issuer_side = comm if "comm" in variant else (None if variant.endswith("side") else None)
        async_op = variant.endswith("async")
        side.wait_stream(cur)
        if "comm" in variant:
            comm.wait_stream(cur)
        with torch.cuda.stream(side):            # branch V on the side stream, its exchange from side / comm
            xv = work(a)
            exch(xv, comm if "comm" in variant else None, async_op)
            xv = work(xv)
        xt = work(b)                             # branch T on the current stream
        if "two_issuers" in variant or "side" in variant:
            exch(xt, None, async_op)             # ... its exchange from the CURRENT stream: two issuing streams
        else:
            exch(xt, comm, async_op)
        xt = work(xt)
        cur.wait_stream(side)
        if "comm" in variant:
            cur.wait_stream(comm)
        return xv + xt

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            out = body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    if variant.startswith("capture"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    print("value", float(out.sum()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    for i, v in enumerate(VARIANTS):
        env = dict(os.environ, MASTER_PORT=str(29611 + i))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v], env=env, capture_output=True, text=True,
                               timeout=180)
            tail = (r.stderr.strip().splitlines() or [""])[-1][:160]
            print(f"{v:22s} rc={r.returncode:4d} {'ok' if r.returncode == 0 else tail}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{v:22s} timed out (180 s)", flush=True)
