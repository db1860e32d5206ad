"""The cold ranking call at the config-5 shard / whole SHAPE on random tables (no graph build): user ranges under several
workspace budgets (the pipelined variant of DESIGN 7.12 needs profiles/r05_exp_score_pipeline.patch applied).  python3 tools/score_ranges_bench.py [users] [items] [dim]; prints ms per call and the
fraction of the 2.5 PF bf16 MFMA peak (2 U I D flops)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import ops  # noqa: E402


def main():
    U = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    I = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    a = (6.0 / (U + I + D)) ** 0.5
    ue = (torch.rand(U, D, generator=g, device=dev) * 2 - 1) * a
    ie = (torch.rand(I, D, generator=g, device=dev) * 2 - 1) * a
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 8
    col = (torch.arange(U * 8, device=dev) % 8 * (I // 8) + torch.arange(U * 8, device=dev) // 8 % (I // 8)).to(torch.int32)
    hist = (rowptr, col)
    flop = 2.0 * U * I * D
    for limit_gb, pipe in [(24, "0"), (24, "1"), (12, "1"), (48, "1"), (6, "1")]:
        os.environ["CHAOREC_SCORE_WS_LIMIT"] = str(limit_gb << 30)
        os.environ["CHAOREC_SCORE_PIPELINE"] = pipe
        st = {}
        ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U, stats=st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ms = min(ts)
        print(f"limit {limit_gb:3d} GiB pipeline {pipe}: {ms:8.1f} ms  frac {flop / ms / 1e9 / 2500:.3f}  ranges {st.get('user_chunks', 1)} "
              f"cand/user {st['candidates'] / U:.0f} reth {st['rethreshold_users']} exact {st['fallback_users']}", flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
