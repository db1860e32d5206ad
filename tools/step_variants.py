#!/usr/bin/env python3
"""Time the LightGCN forward+backward SpMM sequence (with epilogues) for several builds, interleaved."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph, _lib
from chaorec_amd.synthetic import synthetic_interactions, DATASET_SHAPES
U, I, E = DATASET_SHAPES[os.environ.get("DATASET", "sports")]
D = 64
dev = torch.device("cuda:0")
edges = synthetic_interactions(U, I, E, seed=42)
N = U + I
A = graph.lightgcn_csr(edges, N).to(dev)
order = A.schedule(D)
x0 = torch.randn(N, D, device=dev); y1 = torch.empty_like(x0); y2 = torch.empty_like(x0); y3 = torch.empty_like(x0)
fin = torch.empty_like(x0); G = torch.randn(N, D, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
def seq(lib):
    f = lib.chaorec_spmm_csr_f32
    def call(x, y, alpha, z, beta, acc, init, w):
        rc = f(A.rowptr.data_ptr(), A.col.data_ptr(), A.val.data_ptr(), x.data_ptr(), y.data_ptr() if y is not None else None,
               N, N, D, alpha, z.data_ptr() if z is not None else None, beta, acc.data_ptr() if acc is not None else None,
               init.data_ptr() if init is not None else None, w, order.data_ptr(), 0, st)
        assert rc == 0
    w = 0.25
    call(x0, y1, 1.0, None, 0.0, fin, x0, w); call(y1, y2, 1.0, None, 0.0, fin, None, w); call(y2, y3, 1.0, None, 0.0, fin, None, w)
    call(G, y1, w, G, w, None, None, 0.0); call(y1, y2, 1.0, G, w, None, None, 0.0); call(y2, y3, 1.0, G, w, None, None, 0.0)
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.chaorec_spmm_csr_f32.restype, lib.chaorec_spmm_csr_f32.argtypes = _lib.SIGNATURES["chaorec_spmm_csr_f32"]
    libs.append((os.path.basename(path), lib))
res, ref = {}, None
for rnd in range(5):
    for name, lib in libs:
        seq(lib); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): seq(lib)
        e.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(s.elapsed_time(e) / 20 / 6 * 1e3)
        chk = (fin.clone(), y3.clone())
        if ref is None: ref = chk
        assert torch.equal(ref[0], chk[0]) and torch.equal(ref[1], chk[1]), name
for k, v in res.items():
    print(f"{k:28s} median {np.median(v):6.2f} us per SpMM launch (6-launch fwd+bwd sequence)  min {min(v):6.2f}")
