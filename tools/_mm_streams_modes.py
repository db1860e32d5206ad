"""Which mode of the sharded MMGCN step is not reproducible?  N runs of the one-stream and of the two-stream captured step (1 rank,
RCCL forced), each compared with the first run of its own mode: worst share of entries off by > 1e-5 per run."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for streams in (True, False):
        base, shares = None, []
        for rep in range(n):
            with tempfile.TemporaryDirectory() as tmp:
                mp.spawn(_sharded_mmgcn_streams_worker, args=(1, _free_port(), tmp, streams), nprocs=1, join=True)
                out = dict(np.load(os.path.join(tmp, f"mm_streams_{int(streams)}.npz")))
            if base is None:
                base = out
                continue
            w = max((float((np.abs(out[k] - r) > 1e-5).mean()), k) for k, r in base.items() if not k.startswith("__"))
            shares.append(w)
        print(f"streams={streams}: worst share > 1e-5 against the mode's first run, per run: " + " ".join(f"{s:.1e}" for s, _ in shares)
              + "   worst tensor: " + max(shares)[1], flush=True)
