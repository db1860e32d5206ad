"""Stage cuts of the scoring sweep (experiment builds, wrong results) at a steady state of LightGCN/sports.
  python tools/sweep_cuts.py save            product library: 500 training steps, one cold call -> /tmp/sweep_state.pt
  [CHAOREC_EXTRA_HIPCC_FLAGS=-DCHAOREC_SWEEP_EXP=k] python tools/sweep_cuts.py run
                                              20 steady calls on that state (thresholds reset before each), per-kernel
                                              times from hipEvents around the call; run under rocprofv3 --kernel-trace --stats
                                              for the sweep kernel's own duration
CHAOREC_SWEEP_EXP: 1 = hits counted, never stored; 2 = one sign bit instead of sixteen; CHAOREC_SWEEP_STAGE=4: four tiles
per barrier."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import dataload, ops  # noqa: E402
dev = torch.device("cuda:0")
STATE = "/tmp/sweep_state.pt"
if sys.argv[1] == "save":
    from chaorec_amd.Model import LightGCN
    from chaorec_amd.optim import FusedAdam, FusedLightGCNStep
    d = dataload.packed_interactions("sports")
    U, I, edges = d["num_user"], d["num_item"], d["train"]
    torch.manual_seed(42)
    m = LightGCN(U, I, edges, None, 64, 1e-3, 3, "add", dev).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    step = FusedLightGCNStep(m, opt, batch_size=1024, edges=torch.from_numpy(edges.astype(np.int64)).to(dev), seed=42, step_dev=cnt,
                             steps_per_replay=5)
    step.run(500)
    res = m.result.detach().clone()
    hint = torch.empty(U, dtype=torch.float32, device=dev)
    ops.score_topk(res[:U], res[U:U + I], m.hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=False, hint_rank=110)
    torch.cuda.synchronize()
    torch.save({"res": res.cpu(), "hint": hint.cpu(), "hist": [t.cpu() for t in m.hist], "U": U, "I": I}, STATE)
    print("saved", STATE)
    sys.exit(0)
st = torch.load(STATE)
U, I = st["U"], st["I"]
res, old, hist = st["res"].to(dev), st["hint"].to(dev), tuple(t.to(dev) for t in st["hist"])
ue, ie = res[:U].contiguous(), res[U:U + I].contiguous()
hint = old.clone()
counters = torch.zeros(4, dtype=torch.int32, device=dev)
ts = []
for rep in range(25):
    hint.copy_(old)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.score_topk(ue, ie, hist, 1e-6, 50, id_offset=U, hint=hint, hint_valid=True, hint_rank=110, light=True, counters=counters)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3)
print(f"flags {os.environ.get('CHAOREC_EXTRA_HIPCC_FLAGS', '(product)')}: whole call median {np.median(ts[5:]):.1f} us (queues {counters.tolist()})")
