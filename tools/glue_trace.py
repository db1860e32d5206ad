#!/usr/bin/env python3
"""Which torch ops launch the `at::native` kernels of a model's eager train step, with shapes and the Python line that
asked for them: the work list for cutting the glue share.
  python tools/glue_trace.py MMGCN:microlens [steps]      -> table on stdout, JSON under gpurun_out/"""
import collections, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chaorec_amd import graph, dataload
from chaorec_amd.Model import FREEDOM, MMGCN
from chaorec_amd.optim import FusedAdam

spec = sys.argv[1] if len(sys.argv) > 1 else "MMGCN:microlens"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
name, ds = spec.split(":")
dev = torch.device("cuda:0")
packed = dataload.packed_interactions(ds)
U, I, edges = packed["num_user"], packed["num_item"], np.asarray(packed["train"], dtype=np.int64)
uid = graph.user_item_dict_from_edges(edges)
v_feat, t_feat = dataload.synthetic_features(I, ds)
torch.manual_seed(0)
if name == "MMGCN":
    m = MMGCN(U, I, edges.astype(np.int32), uid, v_feat, t_feat, 64, 1e-4, "add", "False", True, dev).to(dev)
else:
    m = FREEDOM(U, I, edges.astype(np.int32), uid, v_feat, t_feat, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev).to(dev)
    m.pre_epoch_processing()
opt = FusedAdam(m.parameters(), lr=1e-3)
sampler = dataload.DeviceBatchSampler(U, I, uid, edges, 1024, dev, name)
it = iter(sampler)
batches = [next(it) for _ in range(steps + 2)]


def step(b):
    opt.zero_grad()
    loss = m.loss(*b)
    loss.backward()
    opt.step()


for b in batches[:2]:
    step(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for b in batches[2:]:
        step(b)
    torch.cuda.synchronize()

# kernel events -> the CPU op that launched them (correlation through the linked cpu parent)
rows = collections.defaultdict(lambda: [0, 0.0])
total = 0.0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CUDA:
        continue
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = ev.self_cuda_time_total
    if dt <= 0:
        continue
    stack = [s for s in (ev.stack or []) if "chaorec_amd" in s or "autograd" in s.lower()]
    where = stack[0].split("/root/repo/")[-1] if stack else ""
    key = (ev.key, str(ev.input_shapes)[:90], where[:80])
    rows[key][0] += ev.count
    rows[key][1] += dt
    total += dt
out = sorted(rows.items(), key=lambda kv: -kv[1][1])
print(f"{name}:{ds}  {steps} eager steps, device time {total / steps / 1e3:.3f} ms per step")
table = []
for (op, shapes, where), (cnt, dt) in out[:70]:
    print(f"{dt / total * 100:5.1f}%  {cnt / steps:6.1f}/step  {dt / cnt:8.1f} us  {op[:44]:44s} {shapes:60s} {where}")
    table.append({"op": op, "shapes": shapes, "where": where, "per_step": cnt / steps, "us": dt / cnt, "share": dt / total})
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"model": spec, "steps": steps, "ms_per_step_device": total / steps / 1e3, "rows": table},
          open(f"gpurun_out/glue_trace_{name}.json", "w"), indent=1)
