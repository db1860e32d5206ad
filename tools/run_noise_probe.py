import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from chaorec_amd import graph
from chaorec_amd.Model import MMGCN
mm = sys.modules["chaorec_amd.Model.MMGCN"]
from chaorec_amd.optim import FusedAdam, GraphedTrainStep
from chaorec_amd.synthetic import synthetic_interactions
dev = torch.device("cuda:0")
U, I, E, B = 6000, 2500, 40000, 512
edges = synthetic_interactions(U, I, E, seed=3)
uid = graph.user_item_dict_from_edges(edges)
g = torch.Generator().manual_seed(4)
v_feat, t_feat = torch.randn(I, 128, generator=g), torch.randn(I, 256, generator=g)
rng = np.random.default_rng(9)
batches = []
for _ in range(6):
    sel = rng.choice(E, B, replace=False)
    u = torch.from_numpy(edges[sel, 0].astype(np.int64)); pos = torch.from_numpy(edges[sel, 1].astype(np.int64)); neg = torch.from_numpy(rng.integers(U, U + I, B))
    batches.append((torch.stack((u, u), 1).to(dev), torch.stack((pos, neg), 1).to(dev)))
def run(streams):
    mm.BRANCH_STREAMS = streams
    torch.manual_seed(21)
    m = MMGCN(U, I, edges, uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    step = GraphedTrainStep(m, opt, example_batch=batches[0])
    for b in batches: step(*b)
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in m.named_parameters()}
ref = run(False)
for rep, streams in enumerate((True, True, False, False, False, False, True, True)):
    got = run(streams)
    worst = max(((float(((got[n]-ref[n]).abs() > 1e-5).float().mean()), n) for n in ref))
    print("streams", streams, "worst share > 1e-5:", worst)
