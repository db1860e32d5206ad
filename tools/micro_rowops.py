import sys, torch, time
sys.path.insert(0, "/root/repo")
from chaorec_amd import ops, graph
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N, D in ((47297, 64), (3_250_000, 128)):
    y = torch.randn(N, D, device=dev); e = torch.randn(N, D, device=dev); g = torch.randn(N, D, device=dev)
    lib = ops._lib.load()
    out = torch.empty_like(y); gy = torch.empty_like(y); ge = torch.empty_like(y)
    f = lambda: lib.chaorec_row_cosine_scale_fwd_f32(ops._ptr(y), ops._ptr(e), ops._ptr(out), None, N, D, ops._stream())
    b = lambda: lib.chaorec_row_cosine_scale_bwd_f32(ops._ptr(g), ops._ptr(y), ops._ptr(e), ops._ptr(gy), ops._ptr(ge), N, D, ops._stream())
    tf, tb = timeit(f), timeit(b)
    print(f"row_cosine N={N} D={D}: fwd {tf:.1f} us = {N*D*12/tf/1e6:.2f} TB/s   bwd {tb:.1f} us = {N*D*20/tb/1e6:.2f} TB/s")
U, I, E = DATASET_SHAPES["sports"]
edges = synthetic_interactions(U, I, E, seed=42)
s = graph.ngcf_structure(edges, U + I).to(dev)
t = timeit(lambda: ops.edge_dropout_norm(s, 0.2, 1, step=3))
print(f"edge_dropout_norm sports nnz={s.nnz}: {t:.1f} us (memset + 2 launches + 2 allocations)")
for n in (1 << 20, 1 << 25):
    w = torch.rand(n, device=dev) + 0.05
    t = timeit(lambda: ops.weighted_sample_keep(w, int(0.8 * n), 3, step=1), n=5)
    print(f"weighted_sample_keep n={n}: {t:.1f} us  ({7*4*n/t/1e6:.2f} TB/s over 7 passes of the weights)")
