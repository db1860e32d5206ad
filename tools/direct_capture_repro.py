"""Which node of the `direct` exchange (all-to-all + local column sum + all-gather, chaorec_amd/dist.py) breaks hipGraph
capture on a 1-rank RCCL group?  Every variant runs in a child process (a crash is an answer, not the end of the run).

    python3 tools/direct_capture_repro.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["a2a", "a2a_async", "a2a_colsum", "a2a_ag", "a2a_async_colsum_ag_async", "ag_only", "rs_ag", "p2p_copy"]


def child(variant):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from chaorec_amd import ops, _lib
    _lib.ensure_built()
    world = 1
    I, D = 15208, 64
    buf = torch.randn(I, D, device=dev)
    recv = torch.empty_like(buf)
    chunk = torch.empty(I // world, D, device=dev)

    def body():
        if variant == "a2a":
            dist.all_to_all_single(recv, buf)
        elif variant == "a2a_async":
            dist.all_to_all_single(recv, buf, async_op=True).wait()
        elif variant == "a2a_colsum":
            dist.all_to_all_single(recv, buf)
            chunk.copy_(ops.col_sum(recv.view(world, -1)).view(I // world, D))
        elif variant == "a2a_ag":
            dist.all_to_all_single(recv, buf)
            dist.all_gather_into_tensor(buf, recv[:I // world])
        elif variant == "a2a_async_colsum_ag_async":
            dist.all_to_all_single(recv, buf, async_op=True).wait()
            c = ops.col_sum(recv.view(world, -1)).view(I // world, D)
            dist.all_gather_into_tensor(buf, c, async_op=True).wait()
        elif variant == "ag_only":
            dist.all_gather_into_tensor(buf, chunk)
        elif variant == "rs_ag":
            dist.reduce_scatter_tensor(chunk, buf)
            dist.all_gather_into_tensor(buf, chunk)
        elif variant == "p2p_copy":
            recv.copy_(buf)          # what a hand-written exchange is made of: plain stream copies + kernels
            chunk.copy_(ops.col_sum(recv.view(world, -1)).view(I // world, D))
            buf[:I // world].copy_(chunk)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()                        # eager first: communicators are set up outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("OK", variant, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for i, v in enumerate(VARIANTS):
        env = dict(os.environ, MASTER_PORT=str(29540 + i))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], env=env, capture_output=True, text=True, timeout=240)
            tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
            print(f"{v:28s} rc={r.returncode:4d}  {' | '.join(tail)[-300:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{v:28s} TIMEOUT", flush=True)
