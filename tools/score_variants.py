#!/usr/bin/env python3
"""A/B several builds of libchaorec_hip.so on the gene_ranklist kernel sequence, interleaved rounds in one process."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph, _lib
from chaorec_amd.synthetic import synthetic_interactions, DATASET_SHAPES
ds = os.environ.get("DATASET", "sports")
U, I, E = DATASET_SHAPES[ds]
dev = torch.device("cuda:0")
edges = synthetic_interactions(U, I, E, seed=42)
rp, col = (t.to(dev) for t in graph.user_hist_csr_from_edges(edges, U))
torch.manual_seed(0)
emb = torch.randn(U + I, 64, device=dev) * 0.1
ue, ie = emb[:U].contiguous(), emb[U:].contiguous()
idx = torch.empty((U, 50), dtype=torch.int64, device=dev); val = torch.empty((U, 50), device=dev)
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("chaorec_score_topk_workspace_bytes", "chaorec_score_topk_f32"):
        f = getattr(lib, name); f.restype, f.argtypes = _lib.SIGNATURES[name]
    libs.append((os.path.basename(path), lib))
nb = libs[0][1].chaorec_score_topk_workspace_bytes(U, I, 50, 64)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(lib):
    rc = lib.chaorec_score_topk_f32(ue.data_ptr(), ie.data_ptr(), U, I, 64, rp.data_ptr(), col.data_ptr(), 1e-6, 50, U,
                                    idx.data_ptr(), val.data_ptr(), ws.data_ptr(), nb, 0, st)
    assert rc == 0
res, ref = {}, None
for rnd in range(5):
    for name, lib in libs:
        run(lib); run(lib); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): run(lib)
        e.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(s.elapsed_time(e) / 5)
        if ref is None: ref = idx.clone()
        assert torch.equal(ref, idx), name
for k, v in res.items():
    print(f"{ds} {k:24s} median {np.median(v):.3f} ms  min {min(v):.3f} ms -> {U / np.median(v) / 1e3:.1f} M users/s")
