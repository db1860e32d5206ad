"""Is torch.randperm on the GPU (ROCm build) a uniform permutation?  Local structure statistics against the CPU generator."""
import torch, numpy as np
n = 118551
dev = torch.device("cuda:0")
def stats(p):
    p = p.cpu().numpy().astype(np.int64)
    assert len(np.unique(p)) == n
    d = np.abs(np.diff(p))
    asc = (np.diff(p) > 0).mean()
    # within batches of 1024: how many DISTINCT "blocks of 128 consecutive source indices" does a batch touch (uniform: ~ 1024 * (1 - tiny))
    blocks = [len(np.unique(p[s:s + 1024] // 128)) for s in range(0, n - 1024, 1024)]
    # longest run of consecutive positions whose values are ascending
    runs, cur, best = 0, 1, 1
    for x in (np.diff(p) > 0):
        cur = cur + 1 if x else 1
        best = max(best, cur)
    return dict(mean_abs_diff=float(d.mean()) / n, ascending_share=float(asc), small_jumps_share=float((d < n / 100).mean()),
                distinct_blocks_per_batch=float(np.mean(blocks)), longest_ascending_run=int(best))
for seed in (100, 101, 102):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    print("cuda", seed, stats(torch.randperm(n, device=dev, generator=g)))
    print("cpu ", seed, stats(torch.randperm(n, generator=torch.Generator().manual_seed(seed))))
g = torch.Generator(device=dev); g.manual_seed(5)
a = torch.randperm(n, device=dev, generator=g); b = torch.randperm(n, device=dev, generator=g)
print("two consecutive cuda draws equal:", bool(torch.equal(a, b)), " share of fixed positions:", float((a == b).float().mean()))
print("expected: mean_abs_diff 0.333, ascending 0.5, small_jumps 0.0199, distinct_blocks ~%.0f, longest run ~8-9" % (927 * (1 - (1 - 1 / 927) ** 1024)))
