"""Constants, the graph loader and small helpers shared by the bench modules (bench.py is the entry point)."""
import json
import os
import sys
import time

import numpy as np
import torch


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


MIN_TIMED_S = 0.05                   # the timed region is repeated in blocks of --steps until this much was timed


STEADY_EVALS = 6                     # carried-threshold evaluations (one per epoch) before the timed one


BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md matrix-core table


TRAINED_STEPS = 5000            # ~32 epochs of the sports-sized graph: embeddings in a trained state


F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32 dense peak


def spmm_model_bytes(nnz, n_rows, D):
    """no-reuse CSR model, fp32: per nonzero a D-float source row + 4 B col + 4 B val;
    per output row a D-float store + 8 B row pointer."""
    return nnz * (4 * D + 8) + n_rows * (4 * D + 8)


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


def spmm_kernel_name(D, adam=False, rowsparse=False):
    """<LPR, CPL, ADAM, SP> as rocprofv3 prints the instantiation."""
    d4, lpr = D // 4, 1
    while lpr < min(d4, 64):
        lpr *= 2
    return (f"spmm_csr_ordered_kernel<{lpr}, {max(1, (d4 + 63) // 64)}, {'true' if adam else 'false'}, "
            f"{'true' if rowsparse else 'false'}>")


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


def flush_c_stdout():
    """RCCL writes its version banner through C stdio, which would otherwise drain at exit, AFTER the result: flush C stdout
    first so that the JSON object is the last line (stdout only -- an fflush(NULL) from here hung under rocprofv3)."""
    import ctypes
    libc = ctypes.CDLL(None)
    try:
        libc.fflush(ctypes.c_void_p.in_dll(libc, "stdout"))
    except (ValueError, OSError):
        pass
