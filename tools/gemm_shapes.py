#!/usr/bin/env python3
"""Which GEMMs a model's training step runs, and how long each takes (eager, every call bracketed by events):
python3 tools/gemm_shapes.py MMGCN:microlens"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import ops, graph, dataload
from chaorec_amd.Model import FREEDOM, MMGCN
from chaorec_amd.optim import FusedAdam
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
dev = torch.device("cuda:0")
name, ds = (sys.argv[1] if len(sys.argv) > 1 else "MMGCN:microlens").split(":")
U, I, E = DATASET_SHAPES[ds]
edges = synthetic_interactions(U, I, E, seed=42)
uid = graph.user_item_dict_from_edges(edges)
v_feat, t_feat = dataload.synthetic_features(I, ds)
torch.manual_seed(0)
if name == "FREEDOM":
    m = FREEDOM(U, I, edges, uid, v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev).to(dev)
    m.pre_epoch_processing()
else:
    m = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
sampler = dataload.DeviceBatchSampler(U, I, uid, edges, 1024, dev, name)
it = iter(sampler)
stats = collections.defaultdict(lambda: [0, 0.0])

def wrap(fn, tag, key):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        s = stats[(tag,) + key(*a, **k)]
        s[0] += 1
        s[1] += e0.elapsed_time(e1)
        return out
    return f

ops.gemm_raw = wrap(ops.gemm_raw, "f32", lambda A, B, transA=False, transB=False, **k: (tuple(A.shape), tuple(B.shape), transA, transB))
ops.gemm_nt_bf16x3 = wrap(ops.gemm_nt_bf16x3, "bf16x3", lambda x, w, **k: (tuple(x.shape), tuple(w.shape), False, True))
ops.col_sum = wrap(ops.col_sum, "colsum", lambda x: (tuple(x.shape), (), False, False))
for i in range(3):
    if i == 2:
        stats.clear()
    b = next(it)
    opt.zero_grad()
    m.loss(*b).backward()
    opt.step()
tot = sum(v[1] for v in stats.values())
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:7s} A{k[1]} B{k[2]} tA={int(k[3])} tB={int(k[4])}: {v[0]:3d} calls {v[1] * 1e3:8.1f} us total {v[1] / v[0] * 1e3:7.1f} us each")
print(f"all GEMM + colsum calls of one step: {tot * 1e3:.0f} us")
