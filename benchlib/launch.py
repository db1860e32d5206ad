"""bench.py --gpus N without a launcher: count GPUs without touching HIP, start the ranks, collect rank 0's line."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import ROOT


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
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e,
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
    if rank == 0 and os.environ.get("CHAOREC_BENCH_SELFTEST_EMIT"):
        # the record path of the real bench (benchlib/record.py) on a canned result: detail -> file + stderr, compact line last
        from .record import emit
        emit(json.load(open(os.environ["CHAOREC_BENCH_SELFTEST_EMIT"])))
    elif rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "sum": float(t.item()),
                          "local_rank": int(os.environ["LOCAL_RANK"]),
                          "self_launched": os.environ.get("CHAOREC_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
