#!/usr/bin/env python3
"""Time chaorec_spmm_csr_f32 from several builds of spmm.hip in one process (interleaved rounds,
cdna_hip_programming.md rule 24).  Usage: python tools/spmm_variants.py build/variants/*.so"""
import ctypes
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph
from chaorec_amd.synthetic import synthetic_interactions, DATASET_SHAPES

dataset = os.environ.get("DATASET", "sports")
D = int(os.environ.get("DIM", "64"))
U, I, E = DATASET_SHAPES[dataset]
edges = synthetic_interactions(U, I, E, seed=42)
N = U + I
dev = torch.device("cuda:0")
A = graph.lightgcn_csr(edges, N).to(dev)
x = torch.randn(N, D, device=dev)
y = torch.empty_like(x)
P = ctypes.c_void_p
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    f = lib.chaorec_spmm_csr_f32
    f.restype = ctypes.c_int
    f.argtypes = [P, P, P, P, P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, P, ctypes.c_float,
                  P, P, ctypes.c_float, P, ctypes.c_int32, P]
    lib.chaorec_spmm_rows_per_wave.restype = ctypes.c_int
    lib.chaorec_spmm_rows_per_wave.argtypes = [ctypes.c_int32]
    libs.append((os.path.basename(path), lib))
order = A.schedule(D)
st = torch.cuda.current_stream().cuda_stream


def run(lib, use_order=True):
    rc = lib.chaorec_spmm_csr_f32(A.rowptr.data_ptr(), A.col.data_ptr(), A.val.data_ptr(), x.data_ptr(), y.data_ptr(),
                                  N, N, D, 1.0, None, 0.0, None, None, 0.0, order.data_ptr() if use_order else None, 0, st)
    assert rc == 0


ref = None
res = {}
for rnd in range(5):
    for name, lib in libs:
        for uo in (True, False):
            for _ in range(5):
                run(lib, uo)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                run(lib, uo)
            e.record()
            torch.cuda.synchronize()
            res.setdefault((name, uo), []).append(s.elapsed_time(e) / 50 * 1e3)
            if ref is None:
                ref = y.clone()
            assert torch.equal(ref, y), name
nnz = A.nnz
mb = (nnz * (4 * D + 8) + N * (4 * D + 8)) / 1e6
for k, v in res.items():
    print(f"{k[0]:40s} order={k[1]!s:5s} median {np.median(v):7.1f} us  min {min(v):7.1f} us  -> {mb / np.median(v) * 1e-3 * 1e3:6.0f} GB/s (model {mb:.1f} MB)")
