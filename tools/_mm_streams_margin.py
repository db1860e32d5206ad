"""How close test_sharded_mmgcn_two_streams_captured_with_rccl runs to its tolerances: the worst share of entries off by > 1e-5 and
the worst median difference over the parameters, for a few repetitions of the one-stream / two-stream pair."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch.multiprocessing as mp  # noqa: E402
from test_gpu_dist2 import _free_port  # noqa: E402
from test_gpu_round4 import _sharded_mmgcn_streams_worker  # noqa: E402

if __name__ == "__main__":
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        out = {}
        with tempfile.TemporaryDirectory() as tmp:
            for streams in (False, True):
                mp.spawn(_sharded_mmgcn_streams_worker, args=(1, _free_port(), tmp, streams), nprocs=1, join=True)
                out[streams] = dict(np.load(os.path.join(tmp, f"mm_streams_{int(streams)}.npz")))
        worst = max(((float((np.abs(out[True][n] - r) > 1e-5).mean()), float(np.median(np.abs(out[True][n] - r))), float(np.abs(out[True][n] - r).max()), n)
                     for n, r in out[False].items() if not n.startswith("__")))
        print(f"rep {rep}: worst share > 1e-5 = {worst[0]:.2e} (limit 1e-3), its median {worst[1]:.2e}, max {worst[2]:.2e}  [{worst[3]}]", flush=True)
