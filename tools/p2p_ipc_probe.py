"""Can two processes map each other's device buffers (HIP IPC through torch's storage sharing) on this stack?  Two ranks
on ONE GPU, gloo for the handle exchange and the barriers: each rank fills a buffer, opens the peer's, reads it."""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    mine = torch.full((1 << 20,), float(rank + 1), device=dev)
    handle = mine.untyped_storage()._share_cuda_()
    handles = [None] * world
    dist.all_gather_object(handles, (handle, mine.storage_offset(), tuple(mine.shape)))
    peers = []
    for r in range(world):
        if r == rank:
            peers.append(mine)
            continue
        h, off, shape = handles[r]
        st = torch.UntypedStorage._new_shared_cuda(*h)
        peers.append(torch.empty(0, dtype=torch.float32, device=dev).set_(st, off, shape))
    torch.cuda.synchronize()
    dist.barrier()
    got = [float(p.sum()) / p.numel() for p in peers]
    print(f"rank {rank}: means of the ranks' buffers as seen here: {got}", flush=True)
    # the peer writes, this rank reads again after a barrier
    mine.fill_(10.0 * (rank + 1))
    torch.cuda.synchronize()
    dist.barrier()
    got = [float(p[12345]) for p in peers]
    print(f"rank {rank}: after the owners rewrote them: {got}", flush=True)
    dist.barrier()
    del peers
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
