"""Round-4 additions on the GPU: the early-Adam ordering fix for dense readers, the ranking tail, the
self-launching bench (two ranks on this one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def test_early_adam_with_a_dense_reader_equals_the_in_step_update(dev):
    """FusedAdam.early_tables with a table that ops.linear reads WHOLE (VBPR's / MGCN's feature tables: submit(...,
    dense_reader=True)): the node's weight gradient reads every row of the table, so it must be queued before submit()
    starts the table's in-place update on the side stream (it was queued after).  Four steps, eager and
    captured (GraphedTrainStep), bit-identical to the in-step update."""
    from chaorec_amd import ops
    from chaorec_amd.optim import FusedAdam, GraphedTrainStep
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    I, K, R = 6000, 1024, 64
    t0 = torch.randn(I, K, device=dev, generator=g)
    w0 = torch.randn(R, K, device=dev, generator=g) * 0.03
    tgt = torch.randn(I, R, device=dev, generator=g)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(t0.clone())
            self.table._chaorec_projected_only = True
            self.lin = torch.nn.Linear(K, R, bias=True)
            with torch.no_grad():
                self.lin.weight.copy_(w0)
                self.lin.bias.zero_()

        def loss(self):
            y = ops.linear(self.table, self.lin.weight, self.lin.bias)
            return ops.mean_all((y - tgt) ** 2)

    def run(early, captured):
        net = Net().to(dev)
        net.lin.weight.data.copy_(w0)
        opt = FusedAdam(list(net.parameters()), lr=1e-2)
        opt.early_tables = early
        if captured:
            step = GraphedTrainStep(net, opt, batch_fn=lambda: (), loss_fn=net.loss)
            opt.early_tables = early
            for _ in range(4):
                step()
        else:
            for _ in range(4):
                opt.zero_grad()
                net.loss().backward()
                opt.step()
        torch.cuda.synchronize()
        return net.table.detach().clone(), net.lin.weight.detach().clone()

    ref = run(False, False)
    for early, captured in ((True, False), (True, True), (False, True)):
        got = run(early, captured)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), (early, captured)


def _run_bench(extra, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_self_launches_two_ranks_on_this_gpu(dev):
    """`python bench.py --gpus 2` with no launcher around it: two ranks share this one GPU (gloo for the
    collectives -- the line must say it is not an RCCL measurement), the N > 1 line carries the `hbm_regime` and `models`
    sub-records next to the headline, and every record trained (finite loss)."""
    line = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--hbm-steps", "2", "--no-cpu-baseline"], {})
    assert line["n_gpus"] == 2 and line["multi_rank_rccl_measured"] is False and line["scaling"] == "weak"
    assert line["config"]["ranks_share_devices"] is True
    assert np.isfinite(line["loss_mean"]) and line["value"] > 0
    h = line["hbm_regime"]
    assert "error" not in h and h["ms_per_step"] > 0 and np.isfinite(h["loss_mean"]) and "split" in h["launch"]
    for name in ("MMGCN", "FREEDOM"):
        m = line["models"][name]
        assert "error" not in m and m["ms_per_step"] > 0 and m["config"]["exchange_bytes_per_step_per_rank"] > 0, m
