#!/usr/bin/env python3
"""bench.py -- the hot path's headline numbers on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LightGCN on the REAL Amazon-sports interaction graph (U=28940, I=15207,
E=158554 -> 317108 directed edges; the reference's Data/sports/train.npy travels with the repository as a packed
fixture, tests/golden/sports_interactions.npz), dim=64, n_layers=3, batch 1024, embeddings xavier-initialised under
seed 42 (random-init weights: there are no checkpoints).  --synthetic swaps in a seeded graph of the same shape.

A "step" is one reference training iteration (train_and_evaluate.py:43-48): negative sampling,
model.loss() = full-graph 3-layer propagate + BPR + L2, loss.backward(), Adam step.  `value` is
directed-edge messages per second through that step, counting the forward AND backward SpMM
(2 * L * E_dir per step), with all inputs resident in HBM.  users_scored_per_s times
gene_ranklist() (scoring + mask + top-50 for every user).

At N > 1 the graph shards by user rows (weak scaling: every rank owns a sports-sized user shard
over the same item set), items replicated, item-row partial sums all-reduced over RCCL per layer.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.common import load_graph  # noqa: E402,F401  (tools/ import these through bench)
from benchlib.launch import launch_selftest, self_launch, visible_gpu_count  # noqa: E402,F401
from benchlib.models import main_model  # noqa: E402
from benchlib.sharded import main_sharded  # noqa: E402
from benchlib.single import main_single  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--dataset", default="sports")
    p.add_argument("--n-layers", type=int, default=3)
    p.add_argument("--dim", type=int, default=64)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-trained-state", action="store_true",
                   help="time gene_ranklist only on the embeddings the timed steps leave behind (profiling runs)")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph step")
    p.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused HIP Adam")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    p.add_argument("--unfused", action="store_true",
                   help="N=1: the autograd step (loss_drawn -> backward -> FusedAdam) instead of FusedLightGCNStep")
    p.add_argument("--steps-per-replay", type=int, default=10,
                   help="training steps captured back to back in one hipGraph (N=1 fused step)")
    p.add_argument("--synthetic", action="store_true", help="dataset-shaped synthetic graph instead of the real one")
    p.add_argument("--no-hbm-regime", action="store_true", help="skip the config-5-shard sub-record (N=1)")
    p.add_argument("--hbm-steps", type=int, default=10, help="timed steps of the config-5-shard sub-record")
    p.add_argument("--no-full-config5", action="store_true",
                   help="skip the sub-record of BASELINE configs[4] WHOLE on this GPU (10 M x 2 M, 4e8 directed edges: ~40 s)")
    p.add_argument("--no-models", action="store_true", help="skip the MMGCN / FREEDOM sub-records (N > 1: the sharded models)")
    p.add_argument("--full-steps", type=int, default=10, help="timed steps of the configs[4]-whole sub-record (N=1)")
    p.add_argument("--model", default="LightGCN", choices=["LightGCN", "MMGCN", "FREEDOM"],
                   help="LightGCN: the headline workload.  MMGCN (BASELINE configs[3], microlens) / FREEDOM (configs[2], "
                        "clothing): the model's captured train step + gene_ranklist, user-sharded at --gpus N > 1")
    p.add_argument("--spmm-only", action="store_true",
                   help="N=1: stop after the timed steps and the SpMM roofline (no ranking): the command of the --pmc passes "
                        "at config 5, where a counter-collecting run of the 5 PFLOP ranking takes tens of minutes")
    p.add_argument("--probe-graph", action="store_true", help=argparse.SUPPRESS)   # child mode of probe_sharded_graph()
    p.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)   # child mode of the launcher's CPU test
    return p.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_selftest:
        return launch_selftest(world, rank)
    # CHAOREC_FORCE_SHARDED=1: run the N>1 code path (sharded model, RCCL calls, graph capture of them) on one rank
    force_sharded = world == 1 and os.environ.get("CHAOREC_FORCE_SHARDED", "0") == "1"
    if args.model != "LightGCN":
        return main_model(args, world, rank, local_rank, force_sharded)
    if world == 1 and not force_sharded:
        assert torch.cuda.is_available(), "bench.py needs the MI355X"
        torch.cuda.set_device(0)
        return main_single(args, torch.device("cuda", 0))
    return main_sharded(args, world, rank, local_rank, force_sharded)


if __name__ == "__main__":
    main()
