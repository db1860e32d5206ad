"""Variants / stage cuts of the block-joint selection (score_blocksel.hpp) at bench.py's steady state.  Every variant is an
EXPERIMENT build (extra -D flags: its own library under csrc/exp/, never the product's).  With -DCHAOREC_BS_EXP,
hint_rank >= 1000 makes the selection return after a stage; everything else of the call runs as usual.

    python3 tools/bs_variants.py [train_steps=3000] [build-only]         (timing on the GPU box; `build-only` here, so
                                                                          that the libraries travel with the snapshot)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "3000"
build_only = "build-only" in sys.argv
VARIANTS = [   # (name, extra hipcc flags, [(label, TIMED_HINT_RANK)])
    ("block selection, stage cuts", "-DCHAOREC_PF_BLOCK=1 -DCHAOREC_BS_EXP=1",
     [("whole call", "100"), ("selection = launch only", "1001"), ("... + union popcounts / prefix sum", "1002"),
      ("... + item list + f32 MFMA re-score", "1003")]),
    ("block selection, rank batch 1", "-DCHAOREC_PF_BLOCK=1 -DCHAOREC_BS_RANK_BATCH=1", [("whole call", "100")]),
    ("block selection, rank batch 4", "-DCHAOREC_PF_BLOCK=1 -DCHAOREC_BS_RANK_BATCH=4", [("whole call", "100")]),
    ("per-user selection (product)", "-DCHAOREC_PF_BLOCK=0", [("whole call", "100")]),
]
for name, flags, runs in VARIANTS:
    env = dict(os.environ, CHAOREC_EXTRA_HIPCC_FLAGS=flags, EPOCH_APART="1")
    subprocess.check_call([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from chaorec_amd import _lib; _lib.build()" % ROOT],
                          env=env)
    if build_only:
        continue
    for label, rank in runs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_profile.py"), steps],
                           env=dict(env, TIMED_HINT_RANK=rank), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "epoch apart" in l]
        print(f"{name:28s} {label:40s} {line[-1] if line else r.stdout[-300:] + r.stderr[-300:]}", flush=True)
