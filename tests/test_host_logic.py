"""CPU-only: host-side logic of chaorec_amd (graph setup, metrics, samplers, CLI) and the C-ABI surface.
No kernel runs here: the product has no CPU compute path (asserted below)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_abi_exports_every_declared_symbol():
    from chaorec_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "chaorec_hip.h")).read()
    declared = set(re.findall(r"\b(chaorec_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/chaorec_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes table and header drifted"
    plain = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, args in re.findall(r"\b(chaorec_\w+)\s*\(([^;{]*?)\)\s*;", plain):
        n_args = 0 if args.strip() in ("", "void") else args.count(",") + 1
        assert len(_lib.SIGNATURES[name][1]) == n_args, f"{name}: ctypes argtypes vs header"
    assert _lib.load().chaorec_abi_version() == _lib.ABI_VERSION == 16
    assert _lib.load().chaorec_spmm_rows_per_wave(64) == 4
    assert _lib.load().chaorec_spmm_rows_per_wave(128) == 2
    assert _lib.load().chaorec_score_topk_workspace_bytes(28940, 15207, 50, 64) > 0


def test_no_cpu_fallback():
    from chaorec_amd import ops, graph
    g = load_golden("lightgcn_tiny.npz")
    csr = graph.lightgcn_csr(g["edges"], int(g["U"]) + int(g["I"]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.spmm_raw(csr, torch.from_numpy(g["x0"]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.score_topk(torch.zeros(4, 64), torch.zeros(100, 64), None, 0.0, 10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.bpr_loss(torch.zeros(4, 64), None, torch.zeros(2, dtype=torch.long), torch.zeros(2, dtype=torch.long),
                     torch.zeros(2, dtype=torch.long), 0, 0.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "chaorec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle_", "").lower() or f == "__init__.py" or \
                    "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("builder", ["lightgcn", "basicgcn"])
def test_csr_builders_match_oracle(oracle, baby, builder):
    from chaorec_amd import graph
    U, I = baby["U"], baby["I"]
    mine = getattr(graph, builder + "_csr")(baby["train"], U + I)
    want = getattr(oracle, builder + "_csr")(baby["train"], U + I)
    assert np.array_equal(mine.rowptr.numpy(), want[0])
    assert np.array_equal(mine.col.numpy(), want[1])
    assert np.array_equal(mine.val.numpy(), want[2])       # torch pow(-0.5) == 1/sqrt in fp32, bit for bit
    assert mine.symmetric and mine.t() is mine


def test_spmm_schedule_descriptors(baby):
    """chaorec_spmm_build_schedule: every row appears exactly once; descriptors carry degree, first entry and the
    first 6 (col,val) pairs; the lead slot of every workgroup walks the heaviest groups in descending order; the
    long-row flag is block-uniform."""
    from chaorec_amd import graph
    csr = graph.lightgcn_csr(baby["train"], baby["U"] + baby["I"])
    sched = csr.schedule(64)
    assert sched is csr.schedule(64) and sched.dtype == torch.int32
    d = sched.numpy().reshape(-1, 16)
    rows = d[:, 0]
    valid = rows >= 0
    assert sorted(rows[valid].tolist()) == list(range(csr.n_rows))
    rp, col, val = csr.rowptr.numpy(), csr.col.numpy(), csr.val.numpy()
    deg = d[:, 1] & 0x7fffffff
    assert np.array_equal(deg[valid], np.diff(rp)[rows[valid]])
    e0 = (d[:, 3].astype(np.int64) << 32) | (d[:, 2].astype(np.int64) & 0xffffffff)
    assert np.array_equal(e0[valid], rp[rows[valid]])
    for k in np.nonzero(valid)[0][::97]:
        n = min(int(deg[k]), 6)
        assert np.array_equal(d[k, 4:4 + n], col[e0[k]:e0[k] + n])
        assert np.array_equal(d[k, 10:10 + n].view(np.float32), val[e0[k]:e0[k] + n])
    # 4 rows per wave slot, 4 slots per workgroup
    heavy = np.where(valid, deg, 0).reshape(-1, 4).max(1).reshape(-1, 4)      # [workgroup, slot]
    assert heavy[0, 0] == np.diff(rp).max() and np.all(np.diff(heavy[:, 0]) <= 0)
    assert np.all(heavy[:, 1:] <= heavy[:, :1])
    flag = (d[:, 1] < 0).reshape(-1, 16)
    assert np.all(flag == flag[:, :1])                                        # block-uniform
    assert np.array_equal(flag[:, 0], heavy.max(1) > 32)


def test_transpose_and_coalesce():
    from chaorec_amd import graph
    rng = np.random.default_rng(0)
    n, m, nnz = 40, 30, 300
    r, c = rng.integers(0, n, nnz), rng.integers(0, m, nnz)
    v = rng.standard_normal(nnz).astype(np.float32)
    dense = np.zeros((n, m), np.float64)
    np.add.at(dense, (r, c), v)
    csr = graph.coo_to_csr_coalesced(r, c, v, n, m)

    def to_dense(x):
        out = np.zeros((x.n_rows, x.n_cols))
        rp, col, val = x.rowptr.numpy(), x.col.numpy(), x.val.numpy()
        for i in range(x.n_rows):
            assert np.all(np.diff(col[rp[i]:rp[i + 1]]) > 0)       # coalesced: strictly ascending columns
            out[i, col[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
        return out
    assert np.allclose(to_dense(csr), dense, atol=1e-6)
    t = csr.t()
    assert np.allclose(to_dense(t), dense.T, atol=1e-6) and t.t() is csr


def test_user_hist_and_dict(oracle, baby):
    from chaorec_amd import graph
    d = graph.user_item_dict_from_edges(baby["train"])
    rp, col = graph.user_hist_csr(d, baby["U"])
    want = oracle.user_hist_csr(baby["train"], baby["U"])
    assert np.array_equal(rp.numpy(), want[0]) and np.array_equal(col.numpy(), want[1])
    rp2, col2 = graph.user_hist_csr_from_edges(baby["train"], baby["U"])
    assert np.array_equal(rp2.numpy(), want[0]) and np.array_equal(col2.numpy(), want[1])


def test_gene_metrics_matches_reference_and_oracle(oracle, baby):
    from chaorec_amd import utils
    g = load_golden("metrics_baby_fixed_rank.npz")
    U, I = baby["U"], baby["I"]
    fixed_rank = np.stack([np.random.default_rng(1000 + u).permutation(I)[:50] + U for u in range(U)])
    k_list = [int(k) for k in g["k_list"]]
    m = utils.gene_metrics(baby["val"], torch.from_numpy(fixed_rank), k_list)
    got = np.array([[m[k][n] for n in g["metric_names"]] for k in k_list])
    assert np.allclose(got, g["val_metrics"], rtol=1e-12, atol=0)          # the reference's own numbers
    # ragged / edge cases against the loop-form oracle: empty positive lists, duplicates, k > hits
    rng = np.random.default_rng(1)
    data = [[u] + rng.integers(U, U + 60, rng.integers(0, 6)).tolist() for u in range(200)]
    data[3] = [3]
    data[4] = [4, U + 1, U + 1, U + 2]
    rank = rng.integers(U, U + 60, (200, 50))
    rank[7] = np.arange(U, U + 50)
    mine = utils.gene_metrics(data, rank, [1, 5, 20, 50])
    ref = oracle.gene_metrics(data, rank, [1, 5, 20, 50])
    for k in ref:
        for name in ref[k]:
            assert mine[k][name] == pytest.approx(ref[k][name], rel=1e-12, abs=1e-15), (k, name)


def test_per_user_metric_functions(oracle):
    from chaorec_amd import metrics
    ranked, test = [5, 3, 9, 1, 7], [9, 7, 2]
    assert metrics.precision_at_k(ranked, test, 5) == 2 / 5
    assert metrics.recall_at_k(ranked, test, 5) == 2 / 3
    assert metrics.recall_at_k(ranked, [], 5) == 0
    assert metrics.hit_rate_at_k(ranked, test, 2) == 0 and metrics.hit_rate_at_k(ranked, test, 3) == 1
    idcg = sum(1 / np.log(i + 2) for i in range(3))
    assert metrics.ndcg_at_k(ranked, test, 5) == pytest.approx((1 / np.log(4) + 1 / np.log(6)) / idcg)
    assert metrics.map_at_k(ranked, test, 5) == pytest.approx((1 / 3 + 2 / 5) / 3)


def test_training_dataset_contract():
    import random
    from chaorec_amd import dataload, graph
    g = load_golden("sampler_tiny.npz")
    U, I, e = int(g["U"]), int(g["I"]), g["edges"]
    d = graph.user_item_dict_from_edges(e)
    ds = dataload.TrainingDataset(U, I, d, e)
    random.seed(0)
    hist = np.zeros((U, I), np.int64)
    for _ in range(300):
        for idx in range(len(ds)):
            u, p, n = ds[idx]
            assert (u, p) == tuple(e[idx]) and n not in d[u] and U <= n < U + I
            hist[u, n - U] += 1
    assert np.array_equal(hist > 0, g["neg_hist"] > 0)      # same support as the reference's sampler
    mm = dataload.TrainingDataset(U, I, d, e, "MMGCN")[0]
    assert mm[0].tolist() == [0, 0] and mm[1][0].item() == e[0][1] and mm[0].dtype == torch.int64


def test_cli_and_yaml():
    from chaorec_amd.arg_parser import parse_args, load_yaml_config
    a = parse_args(["--Model", "FREEDOM", "--data_path", "clothing", "--topk", "5", "10"])
    assert a.topk == [5, 10] and a.batch_size == 1024 and a.dim_E == 64 and a.seed == 42
    assert load_yaml_config("LightGCN")["n_layers"] == [1, 2, 3]
    f = load_yaml_config("FREEDOM")
    assert f["lambda_coeff"] == [0.8] and f["dropout"] == [0.1] and f["ii_topk"] == [10]
    assert load_yaml_config("MMGCN")["reg_weight"] == [0.0001]


def test_synthetic_graph_shape():
    from chaorec_amd.synthetic import synthetic_interactions, synthetic_eval_lists
    e = synthetic_interactions(2000, 900, 9000, seed=3)
    assert e.shape == (9000, 2) and e.dtype == np.int32
    assert np.all(np.diff(e[:, 0]) >= 0) and e[:, 1].min() >= 2000 and e[:, 1].max() < 2900
    assert len(np.unique(e[:, 0].astype(np.int64) * 10000 + e[:, 1])) == 9000
    assert np.bincount(e[:, 0], minlength=2000).min() >= 3
    ev = synthetic_eval_lists(50, 900, e[e[:, 0] < 50], 2)
    seen = set(map(tuple, e.tolist()))
    assert all((r[0], i) not in seen for r in ev for i in r[1:])


def test_early_stopping():
    from chaorec_amd.utils import EarlyStopping
    es = EarlyStopping(patience=2, verbose=False)
    for s in (0.1, 0.2, 0.15, 0.2, 0.1, 0.05):
        es(s, {"s": s})
    assert es.early_stop and es.best_score == 0.2


def test_fused_adam_adjacent_runs():
    """optim._adjacent_run: parameters that continue each other's storage form one launch run."""
    from chaorec_amd.optim import _adjacent_run
    flat = torch.zeros(10 * 4 + 6 * 4 + 3)
    a, b, c = flat[:40].view(10, 4), flat[40:64].view(6, 4), flat[64:67]
    lone = torch.zeros(5)
    assert [id(x) for x in _adjacent_run(a, [a, b, c, lone])] == [id(a), id(b), id(c)]
    assert [id(x) for x in _adjacent_run(b, [a, b, lone, c])] == [id(b)]          # lone breaks the run
    assert [id(x) for x in _adjacent_run(lone, [a, b, c, lone])] == [id(lone)]


def test_eval_lists_csr_layout():
    from chaorec_amd.utils import EvalLists
    data = [[3, 10, 11], [0], [7, 12, 12, 13]]
    ev = EvalLists(data, torch.device("cpu"))
    assert ev.n == 3
    assert ev.row_user.tolist() == [3, 0, 7]
    assert ev.rowptr.tolist() == [0, 2, 2, 5]
    assert ev.items.tolist() == [10, 11, 12, 12, 13]          # duplicates kept: len(test_list) counts them


def test_lightgcn_tables_share_one_buffer_and_keep_their_names(baby):
    """Model/LightGCN.py keeps two nn.Embedding parameters; here they are views of one [N, D] buffer (no per-step
    concatenation).  Names, shapes, values and load_state_dict behave as before; .to()/.float() re-join."""
    from chaorec_amd.Model import LightGCN
    from chaorec_amd import graph
    edges = baby["train"][:2000]
    U, I = baby["U"], baby["I"]
    torch.manual_seed(3)
    m = LightGCN(U, I, edges, graph.user_item_dict_from_edges(edges), 8, 1e-3, 2, "add", torch.device("cpu"))
    assert [n for n, _ in m.named_parameters()] == ["user_embedding.weight", "item_embedding.weight"]
    uw, iw = m.user_embedding.weight, m.item_embedding.weight
    assert uw.shape == (U, 8) and iw.shape == (I, 8)
    assert uw.data_ptr() == m._flat.data_ptr() and iw.data_ptr() == m._flat[U:].data_ptr()
    torch.manual_seed(3)
    eu, ei = torch.nn.Embedding(U, 8), torch.nn.Embedding(I, 8)                 # same init order as the reference
    ref_u, ref_i = torch.nn.init.xavier_uniform_(eu.weight), torch.nn.init.xavier_uniform_(ei.weight)
    assert torch.equal(uw.detach(), ref_u.detach()) and torch.equal(iw.detach(), ref_i.detach())
    sd = {k: v.clone() + 1.0 for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert torch.equal(m._flat[:U], sd["user_embedding.weight"]) and torch.equal(m._flat[U:], sd["item_embedding.weight"])
    m = m.double()
    assert m._flat.dtype == torch.float64 and m.user_embedding.weight.data_ptr() == m._flat.data_ptr()


def test_ngcf_structure_maps_and_parameter_surface():
    """graph.ngcf_structure: entry_row / transpose_entry are consistent (the reversed edge, a bijection, self loops
    fixed) also with a repeated interaction; the NGCF module registers the reference's parameters in its order."""
    from chaorec_amd import graph
    from chaorec_amd.Model import NGCF
    g = load_golden("ngcf_small_drop.npz")
    U, I = int(g["U"]), int(g["I"])
    edges = np.concatenate([g["edges"], g["edges"][:3]])          # three duplicated interactions
    s = graph.ngcf_structure(edges, U + I)
    er, col, te = s.entry_row.numpy(), s.col.numpy(), s.transpose_entry.numpy()
    assert s.nnz == 2 * len(edges) + U + I
    assert np.array_equal(np.sort(te), np.arange(s.nnz))
    assert np.array_equal(er[te], col) and np.array_equal(col[te], er)
    loops = er == col
    assert loops.sum() == U + I and np.array_equal(te[loops], np.nonzero(loops)[0])
    assert np.array_equal(np.repeat(np.arange(U + I), np.diff(s.rowptr.numpy())), er)
    torch.manual_seed(0)
    m = NGCF(U, I, g["edges"], graph.user_item_dict_from_edges(g["edges"]), int(g["D"]), 1e-3, 0.3, int(g["L"]), "add",
             torch.device("cpu"))
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():          # same seed, same creation order -> the reference's initial weights
        assert np.array_equal(p.detach().numpy(), g["p_" + n]), n
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward()


def test_csr_to_keeps_the_transpose_link():
    from chaorec_amd import graph
    a = graph.coo_to_csr_coalesced(torch.tensor([0, 0, 2]), torch.tensor([1, 3, 0]), torch.tensor([1., 2., 3.]), 3, 4)
    t = a.t()
    assert t.t() is a and (t.n_rows, t.n_cols) == (4, 3)
    b = a.to("cpu")                       # a matrix and its transpose reference each other: to() must not recurse
    assert b.t().t() is b and torch.equal(b.t().col, t.col) and torch.equal(b.t().val, t.val)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: with libchaorec_hip.so absent every op raises before doing anything."""
    from chaorec_amd import _lib, ops
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libchaorec_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_raw(torch.zeros(4, 4), torch.zeros(4, 4))


def test_csr_update_from_rewrites_in_place_and_refreshes_the_schedule():
    """FREEDOM / LayerGCN re-prune their graph every epoch into the SAME arrays (a captured step holds their
    addresses): data pointers stay, contents and the cached SpMM schedule become those of the new graph."""
    from chaorec_amd import graph
    rng = np.random.default_rng(0)

    def rand_graph(seed):
        r = np.random.default_rng(seed)
        rows = torch.from_numpy(r.integers(0, 50, 400))
        cols = torch.from_numpy(r.integers(0, 50, 400))
        key = torch.unique(rows * 50 + cols)[:300]
        return graph.coo_to_csr_coalesced(key // 50, key % 50, torch.from_numpy(r.random(300).astype(np.float32)), 50, 50)

    a, b = rand_graph(1), rand_graph(2)
    assert a.nnz == b.nnz == 300
    sched_a = a.schedule(64)
    ptrs = (a.rowptr.data_ptr(), a.col.data_ptr(), a.val.data_ptr(), sched_a.data_ptr())
    before = sched_a.clone()
    assert a.update_from(b)
    assert ptrs == (a.rowptr.data_ptr(), a.col.data_ptr(), a.val.data_ptr(), a.schedule(64).data_ptr())
    assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col) and torch.equal(a.val, b.val)
    assert torch.equal(a.schedule(64), b.schedule(64)) and not torch.equal(a.schedule(64), before)
    c = graph.coo_to_csr_coalesced(torch.tensor([0, 1]), torch.tensor([1, 0]), torch.tensor([1., 1.]), 50, 50)
    assert not a.update_from(c)          # different entry count: the caller must not keep a captured step on it


def test_rank_state_backs_off_from_stale_thresholds():
    """ranking.RankState: a hinted call that queued most users for the retry pass is followed by 1, 2, 4 ... calls
    without hints; a successful hinted call resets the back-off; light mode only after a short queue."""
    import torch
    from chaorec_amd import ranking

    class _Done:
        def query(self):
            return True

    st = ranking.RankState()
    st.hint, st.valid = torch.zeros(4), True
    st.counters_host = torch.zeros(4, dtype=torch.int32)
    U = 1000

    def call(queued):
        hinted = st.use_hints(U)
        light = hinted and st.light()
        st.counters_host[0] = queued if hinted else 0          # what the call's counters would report
        st.copied, st.last_hinted = _Done(), hinted
        return hinted, light

    assert call(900) == (True, False)             # first hinted call (no previous counters on the host): fails massively
    assert call(0)[0] is False                    # back-off 1
    assert call(900)[0] is True                   # tries again, fails again
    assert [call(0)[0] for _ in range(2)] == [False, False]          # back-off 2
    assert call(5) == (True, False)               # succeeds (queue of 5): back-off reset; light not yet (previous call was cold)
    assert call(3) == (True, True)                # short queue last time: no retry pass
    assert call(400)[0] is True                   # 40 % queued: stale again
    assert call(0)[0] is False


def test_torch_generator_and_tensor_graph_builders_on_cpu():
    """Round 3 (config 5 whole): the torch edge generator's contract and the tensor forms of the graph builders, run here
    on CPU tensors (on the GPU box the same functions take CUDA tensors: tests/test_gpu_round3.py)."""
    import numpy as np
    import torch
    from chaorec_amd import graph
    from chaorec_amd.synthetic import synthetic_interactions, synthetic_interactions_torch
    U, I, E = 20000, 5000, 300000
    e = synthetic_interactions_torch(U, I, E, seed=3, chunk_users=7000)
    assert e.dtype == torch.int32 and abs(len(e) - E) <= 0.003 * E
    u, i = e[:, 0].long(), e[:, 1].long() - U
    assert bool((u[1:] >= u[:-1]).all()) and int(i.min()) >= 0 and int(i.max()) < I
    assert (u * I + i).unique().numel() == len(e)                 # unique interactions
    deg = torch.bincount(u, minlength=U)
    assert int(deg.min()) >= 3 and int(deg.max()) <= 256
    assert torch.equal(e, synthetic_interactions_torch(U, I, E, seed=3, chunk_users=7000))   # deterministic per seed / device
    # tensor-input builders == numpy-input builders, bit for bit
    en = synthetic_interactions(3000, 900, 20000, seed=3)
    a, b = graph.lightgcn_csr(en, 3900), graph._lightgcn_csr_device(torch.from_numpy(en), 3900)
    for k in ("rowptr", "col", "val"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    h0, h1 = graph.user_hist_csr_from_edges(en, 3000), graph.user_hist_csr_from_edges(torch.from_numpy(en), 3000)
    assert torch.equal(h0[0], h1[0]) and torch.equal(h0[1], h1[1])


def test_freedom_edge_weights_follow_the_reference_formulas():
    """graph.inv_sqrt_degree_edge_weights / out_degree_normalised_weights / symmetric_bipartite_csr against the sparse-tensor
    expressions of Model/FREEDOM.py:85-99, 128-138, 154-162 (restated with torch.sparse here), bit for bit."""
    import torch
    from chaorec_amd import graph
    g = torch.Generator().manual_seed(0)
    nu, ni, E = 40, 23, 300
    idx = torch.stack((torch.randint(0, nu, (E,), generator=g), torch.randint(0, ni, (E,), generator=g)))
    adj = torch.sparse_coo_tensor(idx, torch.ones_like(idx[0]), (nu, ni))
    r = torch.pow(1e-7 + torch.sparse.sum(adj, -1).to_dense(), -0.5)
    c = torch.pow(1e-7 + torch.sparse.sum(adj.t(), -1).to_dense(), -0.5)
    want = r[idx[0]] * c[idx[1]]
    got = graph.inv_sqrt_degree_edge_weights(idx[0], idx[1], nu, ni)
    assert torch.equal(got, want)
    kn = torch.stack((torch.arange(ni).repeat_interleave(4), torch.randint(0, ni, (4 * ni,), generator=g)))
    a2 = torch.sparse_coo_tensor(kn, torch.ones_like(kn[0]), (ni, ni))
    rs = torch.pow(1e-7 + torch.sparse.sum(a2, -1).to_dense(), -0.5)
    assert torch.equal(graph.out_degree_normalised_weights(kn[0], kn[1], ni), rs[kn[0]] * rs[kn[1]])
    csr = graph.symmetric_bipartite_csr(idx[0], idx[1], got, nu, ni)
    full = torch.sparse_coo_tensor(torch.cat((torch.stack((idx[0], idx[1] + nu)), torch.stack((idx[1] + nu, idx[0]))), 1),
                                   torch.cat((got, got)), (nu + ni, nu + ni)).coalesce()
    rows = torch.repeat_interleave(torch.arange(nu + ni), csr.rowptr[1:] - csr.rowptr[:-1])
    assert torch.equal(torch.stack((rows, csr.col.long())), full.indices()) and torch.allclose(csr.val, full.values(), rtol=1e-6, atol=0)


def test_bench_launches_its_own_ranks_and_propagates_failures():
    """`python bench.py --gpus N` without torchrun: the launcher starts N children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* before any GPU call, relays rank 0's JSON line and exits with a failing rank's code.  The
    children here run the launcher's self-test mode (a gloo all-reduce): no GPU in this container."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "3", "--launch-selftest"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"selftest": True, "n_gpus": 3, "sum": 6.0, "local_rank": 0, "self_launched": True}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-selftest"],
                       env=dict(env, CHAOREC_BENCH_SELFTEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 7 and "a rank ended with 7" in r.stderr
    # without a GPU the real bench must fail loudly through the launcher too (no CPU path), not hang
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=dict(env, CHAOREC_DIST_GRAPH="0"), capture_output=True, text=True, timeout=300)
    if not __import__("torch").cuda.is_available():
        assert r.returncode != 0 and "needs the MI355X" in r.stderr


@pytest.mark.parametrize("name", ["simgcl_small.npz", "xsimgcl_small.npz", "ncl_small.npz", "selfcf_small.npz", "slmrec_small.npz",
                                  "mcln_small.npz", "vgcl_small.npz", "dccf_small.npz", "hccf_small.npz", "sgl_small.npz"])
def test_sparse_family_models_start_from_the_reference_state(name):
    """SimGCL / NCL / SelfCF (through the adapter alone): what needs no GPU -- the same seed gives the
    reference class's parameter names and initial weights, and graph.binary_sym_norm_csr gives its scipy-built
    D^-1/2 A D^-1/2 bit for bit (goldens of tests/golden/gen_sparse_family.py: the reference classes' own output)."""
    from chaorec_amd import graph
    from chaorec_amd.Model import DCCF, HCCF, MCLN, NCL, SelfCF, SGL, SimGCL, SLMRec, VGCL, XSimGCL
    g = load_golden(name)
    U, I = int(g["U"]), int(g["I"])
    uid = graph.user_item_dict_from_edges(g["edges"])
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    if name.startswith("simgcl"):
        m = SimGCL(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_reg"]), cpu)
        adj = m.sparse_norm_adj
    elif name.startswith("xsimgcl"):
        m = XSimGCL(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_reg"]), cpu)
        adj = m.sparse_norm_adj
    elif name.startswith("slmrec"):
        m = SLMRec(U, I, g["edges"], uid, torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]), int(g["D"]), int(g["L"]),
                   float(g["ssl_temp"]), float(g["ssl_alpha"]), cpu)
        adj = m.norm_adj
    elif name.startswith("mcln"):
        m = MCLN(U, I, g["edges"], uid, torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"]), int(g["D"]), float(g["reg"]),
                 int(g["L"]), int(g["n_mca"]), cpu)
        adj = m.norm_adj_mat
    elif name.startswith("sgl"):
        m = SGL(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), "add", float(g["ssl_temp"]), float(g["ssl_reg"]), cpu)
        adj = m.norm_adj
    elif name.startswith("hccf"):
        m = HCCF(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), "add", float(g["ssl_alpha"]), float(g["ssl_temp"]),
                 1.0, 0.5, float(g["mult"]), cpu)
        adj = m.adj
    elif name.startswith("dccf"):
        m = DCCF(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_alpha"]),
                 int(g["K"]), float(g["cen_reg"]), cpu)
        adj = m.norm_adj_mat
    elif name.startswith("vgcl"):
        m = VGCL(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), float(g["ssl_temp"]), float(g["ssl_alpha"]), cpu)
        adj = m.adj_matrix
    elif name.startswith("ncl"):
        m = NCL(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), "add", float(g["ssl_temp"]), float(g["ssl_reg"]), cpu)
        adj = m.norm_adj_mat
    else:
        m = SelfCF(U, I, g["edges"], uid, int(g["D"]), float(g["reg"]), int(g["L"]), float(g["dropout"]), cpu)
        adj = m.online_encoder.sparse_norm_adj
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().numpy(), g["p_" + n]), n
    dense = np.zeros((U + I, U + I), np.float32)
    rp, col, val = adj.rowptr.numpy(), adj.col.numpy(), adj.val.numpy()
    for r in range(U + I):
        dense[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
    ref = np.zeros((U + I, U + I), np.float32)
    ref[g["norm_idx"][0], g["norm_idx"][1]] = g["norm_val"]
    assert np.array_equal(dense, ref)
    with pytest.raises(RuntimeError, match="MI355X only"):          # no CPU compute path: the propagate raises
        if name.startswith("vgcl"):
            m.forward()                                                 # (VGCL's loss() runs no forward itself)
        m.loss(*(torch.from_numpy(g[k]) for k in (("users", "pos", "neg", "ints") if name.startswith("mcln") else ("users", "pos", "neg"))))


@pytest.mark.parametrize("name", ["dhcf", "lgmrec", "powerec", "smore", "mmgcl", "fkan_gcf", "lightgt", "gume", "ddrec", "micro", "mentor", "lightgcl", "bm3", "mgcl", "lattice", "mmssl", "grcn", "mgat"])
def test_round5_family_members_start_from_the_reference_state(name):
    """The six members added in round 5, what needs no GPU: the same seed gives the reference class's parameter names and
    initial weights, the graphs built vectorised here are the reference's scipy / torch ones (SMORE: the weighted user-item
    graph, its R block, both kNN item graphs and their max-pooled union), and the loss raises without the MI355X (no CPU path)."""
    from chaorec_amd import graph
    from chaorec_amd import Model as M
    g = load_golden(name + "_small.npz")
    U, I, D = int(g["U"]), int(g["I"]), int(g["D"])
    uid = graph.user_item_dict_from_edges(g["edges"])
    cpu = torch.device("cpu")
    feats = (torch.from_numpy(g["v_feat"]), torch.from_numpy(g["t_feat"])) if "v_feat" in g else ()
    torch.manual_seed(0)
    adjs = {}
    if name == "dhcf":
        m = M.DHCF(U, I, g["edges"], uid, D, float(g["reg"]), int(g["L"]), 0.0, cpu)
        for k, layer in enumerate(m.layers):
            assert np.array_equal(layer.weight.detach().numpy(), g[f"layer{k}_weight"])
    elif name == "lgmrec":
        m = M.LGMRec(U, I, g["edges"], uid, *feats, D, float(g["reg"]), int(g["L"]), float(g["ssl_alpha"]), cpu)
        adjs["norm"] = (m.norm_adj, (U + I, U + I), 0.0)
    elif name == "powerec":
        m = M.POWERec(U, I, g["edges"], uid, *feats, D, float(g["reg"]), 2, int(g["prompt_num"]), float(g["neg_weight"]), 0.0, cpu)
        adjs["norm"] = (m.norm_adj_matrix, (U + I, U + I), 0.0)
    elif name == "smore":
        m = M.SMORE(U, I, g["edges"], uid, *feats, D, float(g["reg"]), int(g["L"]), int(g["K"]), 0.0, "none", cpu)
        adjs = {"norm": (m.norm_adj, (U + I, U + I), 2e-7), "R": (m.R, (U, I), 2e-7), "image": (m.image_original_adj, (I, I), 2e-6),
                "text": (m.text_original_adj, (I, I), 2e-6), "fusion": (m.fusion_adj, (I, I), 2e-6)}
    elif name == "ddrec":
        m = M.DDRec(U, I, g["edges"], uid, *feats, D, D, float(g["reg"]), int(g["L"]), 0.2, 0.01, 0.0, "add", cpu)
        adjs = {"mm": (m.mm_adj, (I, I), 1e-7), "image": (m.image_adj, (I, I), 1e-7), "text": (m.text_adj, (I, I), 1e-7)}
    elif name == "mgat":
        m = M.MGAT(U, I, g["edges"], uid, *feats, D, float(g["reg"]), cpu)
    elif name == "grcn":
        m = M.GRCN(U, I, g["edges"], uid, *feats, D, int(g["C"]), float(g["reg"]), 0.2, 2, "add", cpu)
    elif name == "mmssl":
        m = M.MMSSL(U, I, g["edges"], uid, *feats, D, float(g["reg"]), 0.1, 0.5, 1e-4, 2, cpu)
        adjs = {"ui": (m.ui_graph, (U, I), 6e-8), "iu": (m.iu_graph, (I, U), 6e-8)}
    elif name == "lattice":
        m = M.LATTICE(U, I, g["edges"], uid, *feats, D, D, float(g["reg"]), int(g["L"]), 2, int(g["K"]), "add", 0.1, cpu)
    elif name == "bm3":
        m = M.BM3(U, I, g["edges"], uid, *feats, D, D, float(g["reg"]), 0.3, int(g["L"]), 2.0, "add", cpu)
    elif name == "mgcl":
        m = M.MGCL(U, I, g["edges"], uid, *feats, D, float(g["reg"]), int(g["L"]), "add", 0.2, 0.01, cpu)
    elif name == "lightgcl":
        m = M.LightGCL(U, I, g["edges"], uid, D, float(g["reg"]), int(g["L"]), "add", 0.01, 0.1, cpu)
        adjs = {"adj": (m.adj_norm, (U, I), 6e-8)}
    elif name == "mentor":
        np.random.seed(0)
        m = M.MENTOR(U, I, g["edges"], uid, *feats, D, 1, float(g["reg"]), 0.2, 0.1, 0.1, 1e-3, 1.5, cpu)
        adjs = {"mm": (m.mm_adj, (I, I), 1e-7)}
    elif name == "micro":
        m = M.MICRO(U, I, g["edges"], uid, *feats, D, int(g["L"]), float(g["reg"]), int(g["K"]), 1, 0.5, 0.1, 0.1, "add", cpu)
    elif name == "gume":
        m = M.GUME(U, I, g["edges"], uid, *feats, D, int(g["L"]), int(g["L_ui"]), 0.1, 0.1, "none", cpu)
        assert np.array_equal(np.array(sorted(map(tuple, m.inter.t().tolist())), dtype=np.int64), g["inter"])
        adjs = {"norm": (m.norm_adj, (U + I, U + I), 2e-7), "R": (m.R, (U, I), 2e-7), "image": (m.image_original_adj, (I, I), 2e-6),
                "text": (m.text_original_adj, (I, I), 2e-6)}
    elif name == "mmgcl":
        m = M.MMGCL(U, I, g["edges"], uid, *feats, D, float(g["reg"]), int(g["L"]), float(g["ssl_alpha"]), float(g["ssl_temp"]),
                    float(g["dropout"]), cpu)
        adjs["norm"] = (m.norm_adj, (U + I, U + I), 2e-7)
    elif name == "lightgt":
        m = M.LightGT(U, I, g["edges"], uid, *feats, D, float(g["reg"]), int(g["L"]), cpu)
        adjs["norm"] = (m.norm_adj_mat, (U + I, U + I), 0.0)
    else:
        m = M.FKAN_GCF(U, I, g["edges"], uid, D, float(g["reg"]), int(g["L"]), 0.0, 0.0, int(g["G"]), cpu)
        adjs["norm"] = (m.norm_adj_matrix, (U + I, U + I), 0.0)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in g["param_names"]]
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().numpy(), g["p_" + n]), n
    for tag, (adj, shape, tol) in adjs.items():
        dense = np.zeros(shape, np.float32)
        rp, col, val = adj.rowptr.numpy(), adj.col.numpy(), adj.val.numpy()
        for r in range(shape[0]):
            dense[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
        ref = np.zeros(shape, np.float32)
        ref[g[tag + "_idx"][0], g[tag + "_idx"][1]] = g[tag + "_val"]
        assert np.array_equal(dense != 0, ref != 0) and np.abs(dense - ref).max() <= tol, tag
    if hasattr(m, "pre_epoch_processing"):
        m.pre_epoch_processing()
    with pytest.raises(RuntimeError, match="MI355X only"):
        if name == "lightgt":
            u2 = torch.stack((torch.from_numpy(g["users"]), torch.from_numpy(g["users"])), 1)
            m.loss(u2, torch.stack((torch.from_numpy(g["pos"]), torch.from_numpy(g["neg"])), 1), torch.from_numpy(g["mask"]),
                   torch.from_numpy(g["user_item"]))
        elif name == "grcn":
            m.loss(torch.from_numpy(np.stack((g["users"], g["users"]), 1)), torch.from_numpy(np.stack((g["pos"], g["neg"]), 1)))
        elif name == "mmssl":
            m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")), 0)
        elif name in ("micro", "lattice"):
            m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")), True)
        else:
            m.loss(*(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))


def test_lightgt_batches_host_and_device_form():
    """dataload.py:89-101 / :109-143 (LightGT's sample format): the host datasets and the device sampler (run on the CPU
    here) give [user, user], [pos, neg], the padding mask and the history sequence behind a -1; short histories are whole,
    long ones are cut to src_len distinct members."""
    from chaorec_amd import dataload, graph
    U, I = 6, 80
    uid = {0: [U + 1, U + 5, U + 7], 1: [U + 2], 2: list(range(U, U + 70)), 3: [U + 3, U + 4], 4: [U + 9], 5: [U + 11, U + 12]}
    edges = np.array([(u, i) for u, items in uid.items() for i in items], dtype=np.int64)
    ds = dataload.TrainingDataset(U, I, uid, edges, "LightGT")
    users2, items2, mask, user_item = ds[0]
    assert users2.tolist() == [0, 0] and items2[0].item() == U + 1 and items2[1].item() not in uid[0]
    assert user_item.shape == (51,) and user_item[0].item() == -1 and sorted(user_item[1:4].tolist()) == [1, 5, 7]
    assert mask.tolist() == [False] * 4 + [True] * 47 and bool((user_item[4:] == 0).all())
    users2, items2, mask, user_item = ds[4 + 10]                      # a sample of user 2: 70 items, 50 kept
    assert users2.tolist() == [2, 2] and not bool(mask.any()) and len(set(user_item[1:].tolist())) == 50
    ev = dataload.EvalDataset(U, I, uid)
    u, user_item, mask = ev[2]
    assert u.tolist() == [2] and user_item.shape == (21,) and not bool(mask.any()) and len(set(user_item[1:].tolist())) == 20
    u, user_item, mask = ev[1]
    assert user_item[:2].tolist() == [-1, 2] and mask.tolist() == [False, False] + [True] * 19
    rowptr, col = graph.user_hist_csr(uid, U)
    batches = list(dataload.device_eval_batches((torch.as_tensor(rowptr), torch.as_tensor(col)), U, 20, 4, torch.device("cpu")))
    assert [b[0].tolist() for b in batches] == [[0, 1, 2, 3], [4, 5]]
    assert batches[0][1][1].tolist() == [-1, 2] + [0] * 19 and batches[0][2][1].tolist() == [False, False] + [True] * 19
    assert len(set(batches[0][1][2, 1:].tolist())) == 20 and not bool(batches[0][2][2].any())


def test_capture_retry_takes_a_capture_lost_to_the_watchdog_race_again():
    """dist.capture_with_retry: a capture that fails with a captured-event error is reset and taken again, any
    other error -- and the last attempt's -- propagates."""
    from chaorec_amd import dist as cdist
    calls = {"capture": 0, "reset": 0}

    def flaky():
        calls["capture"] += 1
        if calls["capture"] < 3:
            raise RuntimeError("HIP error: operation failed due to a previous error during capture")

    n = cdist.capture_with_retry(flaky, lambda: calls.__setitem__("reset", calls["reset"] + 1), what="test", attempts=3)
    assert n == 3 and calls == {"capture": 3, "reset": 2} and cdist.CAPTURE_LOG[-1] == ("test", 3)
    calls.update(capture=0, reset=0)
    with pytest.raises(RuntimeError, match="during capture"):
        cdist.capture_with_retry(flaky, lambda: None, what="test", attempts=2)

    def broken():
        raise RuntimeError("out of memory")

    with pytest.raises(RuntimeError, match="out of memory"):
        cdist.capture_with_retry(broken, lambda: None, what="test")


def test_bench_counts_gpus_from_the_kfd_topology(tmp_path, monkeypatch):
    """bench.visible_gpu_count: the launcher counts devices from /sys/class/kfd/kfd/topology/nodes (a GPU is
    a node with simd_count > 0) without touching the HIP runtime; visibility lists cap it; no topology = 0."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU sockets, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    (tmp_path / "9").mkdir()                                      # a node without a readable properties file
    assert bench.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(tmp_path)) == 1
    assert bench.visible_gpu_count(str(tmp_path / "missing")) == 0


def test_frontier_modes_by_size_and_the_replay_length_of_a_light_epoch(monkeypatch):
    """optim.FusedLightGCNStep.frontier_modes: the row-sparse backward / light forward switch on by graph size (and by the
    environment); the training loop sizes its k-step replays for E // B - 1 steps when an epoch ends in a full step."""
    from chaorec_amd.optim import FusedLightGCNStep
    for k in ("CHAOREC_SPARSE_BACKWARD", "CHAOREC_LIGHT_FORWARD", "CHAOREC_SPARSE_BACKWARD_MIN_ROWS"):
        monkeypatch.delenv(k, raising=False)
    assert FusedLightGCNStep.frontier_modes(44147, 3, 64) == (True, False, False)            # sports: dense
    assert FusedLightGCNStep.frontier_modes(3_250_000, 3, 128) == (True, True, True)         # the config-5 shard
    assert FusedLightGCNStep.frontier_modes(3_250_000, 5, 128) == (True, True, False)        # light forward: L <= 4
    assert FusedLightGCNStep.frontier_modes(3_250_000, 3, 96)[0] is False                    # D / 4 not a lane-group size
    monkeypatch.setenv("CHAOREC_LIGHT_FORWARD", "0")
    assert FusedLightGCNStep.frontier_modes(3_250_000, 3, 128) == (True, True, False)
    monkeypatch.setenv("CHAOREC_SPARSE_BACKWARD", "1")
    assert FusedLightGCNStep.frontier_modes(1000, 2, 64)[1] is True


def test_pair_structure_layout():
    """sparse.PairStructure (the skeleton of the six graph-reweighting models): the structure's first n entries are the distinct
    (user, item) pairs in row-major order, `lower` maps the other half back to pairs, both() lays per-pair values out as the
    [N, N] matrix says, kept_copies() counts the surviving listed copies of a pair."""
    from chaorec_amd import sparse
    U, I = 7, 5
    edges = np.array([(0, U + 1), (0, U + 3), (2, U + 0), (2, U + 3), (2, U + 3), (5, U + 4), (6, U + 1), (6, U + 0)], dtype=np.int64)
    ps = sparse.PairStructure(edges, U, I, torch.device("cpu"))
    assert ps.n_listed == 8 and ps.n == 7 and ps.ew.tolist() == [1, 1, 1, 2, 1, 1, 1]
    st = ps.structure
    rows, cols = st.entry_row.long(), st.col.long()
    assert torch.equal(rows[:ps.n], ps.eu) and torch.equal(cols[:ps.n], U + ps.ei)
    assert torch.equal(rows[ps.n:], U + ps.ei[ps.lower]) and torch.equal(cols[ps.n:], ps.eu[ps.lower])
    up, low = torch.arange(1., ps.n + 1), -torch.arange(1., ps.n + 1)
    dense = torch.zeros(U + I, U + I)
    dense[rows, cols] = ps.both(up, low)
    for k in range(ps.n):
        assert dense[ps.eu[k], U + ps.ei[k]] == up[k] and dense[U + ps.ei[k], ps.eu[k]] == low[k]
    assert torch.equal(dense[rows, cols][st.transpose_entry.long()], dense.T[rows, cols])
    keep = torch.tensor([1, 1, 1, 1, 0, 0, 1, 1], dtype=torch.bool)
    assert ps.kept_copies(keep).tolist() == [1, 1, 1, 1, 0, 1, 1]          # (pairs in key order: (6, 0) before (6, 1))


def test_segment_softmax_is_pygs_softmax_with_multiplicities():
    """Model/GRCN.py's / MGAT.py's per-pair segment softmax against torch_geometric.utils.softmax (its restatement in
    oracle/pyg_standin.py) over the LISTED edges: a pair listed w times counts w times in its group's sum; a dropped pair (w = 0)
    leaves the group."""
    from chaorec_amd.Model.GRCN import _segment_softmax
    from oracle import pyg_standin
    g = torch.Generator().manual_seed(5)
    n_seg, n = 6, 40
    seg = torch.randint(0, n_seg, (n,), generator=g)
    logit = torch.randn(n, generator=g) * 3
    w = torch.randint(0, 3, (n,), generator=g).float()
    got = _segment_softmax(logit, seg, n_seg, w)
    listed = torch.repeat_interleave(torch.arange(n), w.long())              # every kept copy as its own edge
    want = pyg_standin.softmax(logit[listed], seg[listed], num_nodes=n_seg)
    assert torch.allclose(got[listed], want, rtol=1e-6, atol=1e-7)
    assert bool((got[w == 0] == 0).all())
    sums = torch.zeros(n_seg).index_add_(0, seg, w * got)
    live = torch.zeros(n_seg).index_add_(0, seg, w) > 0
    assert torch.allclose(sums[live], torch.ones(int(live.sum())), atol=1e-6)


def test_learned_adj_transposed_layout():
    """sparse.LearnedAdj.transposed(): entry k of A^T's row-major layout is entry perm[k] of A's (what the backward SpMM of a
    learned graph multiplies with)."""
    from chaorec_amd import sparse
    g = torch.Generator().manual_seed(8)
    n, m = 9, 13
    key = torch.unique(torch.randint(0, n * m, (40,), generator=g))
    rows, cols = torch.div(key, m, rounding_mode="floor"), key % m
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
    val = torch.rand(key.numel(), generator=g)
    adj = sparse.LearnedAdj(rowptr, cols, val, n, m)
    assert torch.equal(adj.entry_rows(), rows)
    rowptr_t, col_t, perm = adj.transposed()
    dense = torch.zeros(n, m)
    dense[rows, cols] = val
    rows_t = torch.repeat_interleave(torch.arange(m), rowptr_t[1:] - rowptr_t[:-1])
    dense_t = torch.zeros(m, n)
    dense_t[rows_t, col_t.long()] = val[perm]
    assert torch.equal(dense_t, dense.T)
    d = adj.detach()
    assert d.n_rows == n and d.n_cols == m and torch.equal(d.val, val)


def test_mmssl_row_mean_graph_is_csr_norm():
    """Model/MMSSL.py:176-190 (csr_norm, mean_flag=True) on a count matrix with a repeated pair and an empty row, restated with
    scipy here, against chaorec_amd.Model.MMSSL._row_mean_graph; an empty list is the all-zero operand (None)."""
    import scipy.sparse as sp
    from chaorec_amd.Model.MMSSL import _row_mean_graph
    rows = torch.tensor([0, 0, 2, 2, 2, 4])
    cols = torch.tensor([1, 3, 0, 3, 3, 2])
    got = _row_mean_graph(rows, cols, 5, 4, torch.device("cpu"))
    m = sp.csr_matrix((np.ones(6, np.float32), (rows.numpy(), cols.numpy())), shape=(5, 4))
    rowsum = np.power(np.array(m.sum(1)) + 1e-8, -0.5).flatten()
    want = (sp.diags(rowsum) * m).toarray().astype(np.float32)
    dense = np.zeros((5, 4), np.float32)
    rp, col, val = got.rowptr.numpy(), got.col.numpy(), got.val.numpy()
    for r in range(5):
        dense[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
    assert np.abs(dense - want).max() <= 1e-7
    assert _row_mean_graph(rows[:0], cols[:0], 5, 4, torch.device("cpu")) is None


@pytest.mark.parametrize("canned", ["r05_zzz_default_bench_line.json", "r05_zzz_bench_line_gpus8_one_gpu.json",
                                    "r05_v_config5_full_bench_line.json"])
def test_bench_record_is_a_compact_projection_of_the_detail(canned, tmp_path):
    """the LAST stdout line of bench.py is <= 4 KB of JSON with the contract's keys, whatever the detail
    weighs (round 5's own 24 KB line is the canned detail here); the detail goes to a file and stderr, not to stdout."""
    import json
    from benchlib import record
    detail = json.load(open(os.path.join(ROOT, "profiles", canned)))
    detail.setdefault("cpu_baseline", {"value": None, "reason": "canned"})
    s = record.render(detail, "bench_detail.json")
    assert len(s.encode()) <= record.LINE_LIMIT and "\n" not in s
    line = json.loads(s)
    for k in record.REQUIRED:
        assert k in line, k
    assert line["value"] == pytest.approx(detail["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-5)
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    if "hbm_regime" in detail and "error" not in detail["hbm_regime"]:
        assert line["hbm_regime"]["ms_per_step"] == pytest.approx(detail["hbm_regime"]["ms_per_step"], rel=1e-5)
        assert "spmm_frac" in line["hbm_regime"]
    # a pathological detail (kilobytes of prose in every string) still fits: optional parts are dropped, required ones kept
    fat = json.loads(json.dumps(detail))
    fat["config"]["workload"] = "w" * 5000
    fat["config"]["launch"] = "l" * 5000
    fat["models"] = {f"M{i}": {"ms_per_step": 1.0, "value": 2.0, "roofline": {"dominant_kernel": "k" * 500, "frac": 0.1}} for i in range(40)}
    s2 = record.render(fat, "bench_detail.json")
    assert len(s2.encode()) <= record.LINE_LIMIT
    assert all(k in json.loads(s2) for k in record.REQUIRED)


def test_committed_round6_line_is_the_projection_of_its_committed_detail():
    """profiles/r06_v_bench_line.json (the driver's command at the round's HEAD, as bench.py printed it) is exactly what
    record.compact makes of profiles/r06_v_bench_detail.json: the line holds nothing the detail does not."""
    import json
    from benchlib import record
    detail = json.load(open(os.path.join(ROOT, "profiles", "r06_v_bench_detail.json")))
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_v_bench_line.json")).read())
    want = json.loads(record.render(detail, line.get("detail")))
    assert line == want
    assert len(json.dumps(line, separators=(",", ":")).encode()) <= record.LINE_LIMIT
    for k in record.REQUIRED:
        assert k in line, k
    assert line["models"]["MMGCN"]["roofline"]["bound"] == "mfma" and "stale" not in line["models"]["MMGCN"]["roofline"]
    assert line["cpu_baseline"]["kind"] == "port" and line["roofline"]["traffic"] is not None


def test_bench_record_through_the_launcher_over_gloo(tmp_path):
    """... and end to end: `bench.py --gpus 2` (own launcher, gloo ranks) emitting a canned detail through
    benchlib.record.emit -- the last stdout line the launcher relays is the compact record, the detail is in the file."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    canned = os.path.join(ROOT, "profiles", "r05_zzz_bench_line_gpus8_one_gpu.json")
    detail_path = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       env=dict(env, CHAOREC_BENCH_SELFTEST_EMIT=canned, CHAOREC_BENCH_DETAIL=detail_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last.encode()) <= 4096
    line = json.loads(last)
    want = json.load(open(canned))
    assert line["metric"] == want["metric"] and line["n_gpus"] == want["n_gpus"] and "roofline" in line and "cpu_baseline" in line
    assert json.load(open(detail_path))["config"] == want["config"]
    assert "[bench detail]" in r.stderr and "[bench detail]" not in r.stdout


@pytest.mark.parametrize("n_rows,D,skew", [(1, 64, 0), (17, 128, 3), (1000, 64, 50), (4099, 128, 50), (4099, 32, 3), (700, 256, 0),
                                           (200, 64, 1)])
def test_schedule_tensors_builds_the_host_builders_words(n_rows, D, skew):
    """graph.schedule_tensors (the SpMM row descriptors from tensor operations on the device the CSR lives on,
    instead of a host walk over a host copy of it) against chaorec_spmm_build_schedule: the same int32 words -- degree-sorted
    groups, long rows leading groups / workgroups of their own with the lightest company, inline (col, val) pairs, flags."""
    import ctypes
    from chaorec_amd import _lib, graph
    lib = _lib.load()
    rng = np.random.default_rng(n_rows * 7 + D + skew)
    deg = rng.integers(0, 12, n_rows)
    if skew:
        idx = rng.choice(n_rows, max(1, n_rows // skew), replace=False)
        deg[idx] = rng.integers(33, 400, len(idx))
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int64))
    nnz = int(rowptr[-1])
    col = torch.from_numpy(rng.integers(0, n_rows, nnz).astype(np.int32))
    val = torch.from_numpy(rng.standard_normal(nnz).astype(np.float32))
    n = lib.chaorec_spmm_schedule_len(n_rows, D)
    want = torch.empty(n, dtype=torch.int32)
    rc = lib.chaorec_spmm_build_schedule(ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                         ctypes.c_void_p(val.data_ptr()), n_rows, D, ctypes.c_void_p(want.data_ptr()), n)
    assert rc == 0
    got = graph.schedule_tensors(rowptr, col, val, n_rows, lib.chaorec_spmm_rows_per_wave(D), lib.chaorec_spmm_long_threshold())
    assert got.dtype == torch.int32 and torch.equal(got, want)
