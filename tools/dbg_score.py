import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from chaorec_amd import ops
from oracle import oracle
dev = torch.device('cuda:0')
for (U, I, D, K) in [(32, 2000, 32, 64), (32, 2000, 32, 57), (32, 2000, 32, 56), (32, 2000, 32, 52), (32, 2000, 32, 49), (32, 300, 32, 64), (32, 600, 32, 64), (32, 600, 32, 60)]:
    rng = np.random.default_rng(U * 7 + I)
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    want_i, want_v = oracle.score_topk(ue, ie, None, 1e-6, K, 0)
    gi, gv = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), None, 1e-6, K, precision=1)
    gi, gv = gi.cpu().numpy(), gv.cpu().numpy()
    bad = np.argwhere(gi != want_i)
    print((U, I, D, K), 'mismatches', len(bad), 'users', sorted(set(bad[:, 0])) if len(bad) else 0)
    for u in sorted(set(bad[:, 0]))[:3]:
        missing = sorted(set(want_i[u]) - set(gi[u]))
        print('   user', u, 'missing items', missing, 'tiles', [m // 32 for m in missing], 'ranks', [int(np.where(want_i[u] == m)[0][0]) for m in missing])
