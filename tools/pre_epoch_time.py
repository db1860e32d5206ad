import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from chaorec_amd import graph, dataload
from chaorec_amd.Model import FREEDOM, LayerGCN
from chaorec_amd.synthetic import DATASET_SHAPES, synthetic_interactions
dev = torch.device("cuda:0")
for name, ds in (("FREEDOM", "clothing"), ("LayerGCN", "sports")):
    U, I, E = DATASET_SHAPES[ds]
    edges = synthetic_interactions(U, I, E, seed=42)
    uid = graph.user_item_dict_from_edges(edges)
    v, t = dataload.synthetic_features(I, ds)
    torch.manual_seed(0)
    m = (FREEDOM(U, I, edges, uid, v, t, 64, 64, 1e-3, 0.1, 2, 1, 10, 0.8, dev) if name == "FREEDOM"
         else LayerGCN(U, I, edges, uid, 64, 1e-3, 3, 0.1, dev)).to(dev)
    for _ in range(3):
        m.pre_epoch_processing()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        m.pre_epoch_processing()
    torch.cuda.synchronize()
    print(f"{name} {ds}: pre_epoch_processing {(time.time() - t0) / 10 * 1e3:.2f} ms")
