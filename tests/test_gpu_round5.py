"""Round-5 GPU parity tests: the scoring call's raised-threshold pass for overflowing users (pass C), the call as two
phases (CHAOREC_SCORE_FRONT / _BACK) and the user ranges of a large call.  Everything through the C-ABI
(chaorec_amd._lib ctypes), compared with oracle/ (the checker) bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from chaorec_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _hist_random(U, I, max_deg, seed):
    rng = np.random.default_rng(seed)
    rowptr = np.zeros(U + 1, np.int64)
    cols = []
    for u in range(U):
        c = np.sort(rng.choice(I, int(rng.integers(0, max_deg + 1)), replace=False)).astype(np.int32)
        cols.append(c)
        rowptr[u + 1] = rowptr[u] + len(c)
    return rowptr, np.concatenate(cols).astype(np.int32)


@pytest.mark.parametrize("D", [64, 128])
def test_score_topk_overflowing_users_get_a_raised_threshold(dev, oracle, D):
    """Long item range (>= 131072 items: the exact route would stream the table per user).  Two thirds of the users see
    6000 high-scoring items that all lie where the sampler never looks (tiles = 0 mod 4 / items = 4 mod 8), so their sampled
    threshold is far too low and every sweep list overflows.  Pass C re-scores what the lists kept, raises the threshold
    and sweeps those users once more: they must come out certified (no exact-route user) with the oracle's bits -- Model/
    LightGCN.py:147-155's top-K of the masked score row, ties to the lowest index."""
    from chaorec_amd import ops
    rng = np.random.default_rng(5 + D)
    U, I, K = 96, 140000, 50
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.05).astype(np.float32)
    tiles = np.arange(I) // 32
    # (tables in their own order: the sampler takes every 4th TILE; norm-sorted tables -- the default at this length, and
    #  this table keeps its order there, see below --: every 8th ITEM, position = 4 mod 8)
    pool = np.flatnonzero((tiles % 4 != 0) & (np.arange(I) % 4 != 0))
    hot = rng.choice(pool, 6000, replace=False)
    # the hot items have the cold items' NORM (the sweep's error band is c ||u|| max ||i||: one bound for the table) but
    # share a direction v; two thirds of the users have a large component along v and score them far above everything else
    v = rng.standard_normal(D).astype(np.float32)
    v /= np.linalg.norm(v)
    cold_norm = float(np.linalg.norm(ie, axis=1).mean())
    hd = v[None, :] * rng.uniform(0.6, 1.0, (6000, 1)).astype(np.float32) + rng.standard_normal((6000, D)).astype(np.float32) * 0.05
    # (every item at exactly that norm: one norm class, so the norm-sorted layout of long ranges -- a stable sort -- keeps the
    #  table's order and the hot items stay in the tiles the sampler skips)
    ie = (ie / np.linalg.norm(ie, axis=1, keepdims=True) * cold_norm).astype(np.float32)
    ie[hot] = (hd / np.linalg.norm(hd, axis=1, keepdims=True) * cold_norm).astype(np.float32)
    ue -= (ue @ v)[:, None] * v[None, :]                  # nobody sees the hot items ...
    ue[1::3] += 2.0 * v[None, :]                          # ... except these
    ue[2::3] += 1.0 * v[None, :]
    rp, cl = _hist_random(U, I, 30, seed=D)
    rows = [cl[rp[u]:rp[u + 1]] for u in range(U)]
    for u in range(1, U, 6):                              # some of the hot items are in the history of users that see them
        rows[u] = np.unique(np.r_[rows[u], hot[u:u + 40]]).astype(np.int32)
    hist = (np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.int64), np.concatenate(rows).astype(np.int32))
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    st = {}
    hint = torch.full((U,), float("nan"), device=dev)
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K, id_offset=U,
                                  stats=st, hint=hint, hint_valid=False)
    assert np.array_equal(got_v.cpu().numpy(), want_v)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    assert st["rethreshold_users"] >= U // 2, st          # the users that see the hot items went through pass C ...
    assert st["fallback_users"] == 0, st                  # ... and none of them needed the exact routes
    assert bool(torch.isfinite(hint).all())               # every user left a threshold for the next call
    # the thresholds pass C's selection leaves behind serve the next call (pass A) like anybody's
    st2 = {}
    again_i, again_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), dh, 1e-6, K, id_offset=U,
                                      stats=st2, hint=hint, hint_valid=True)
    assert torch.equal(again_i, got_i) and torch.equal(again_v, got_v)
    assert st2["fallback_users"] == 0, st2


def test_score_topk_pass_c_hands_hopeless_users_to_the_exact_routes(dev, oracle):
    """All items equal: no threshold separates anything, pass C's second sweep overflows again and the users end on the
    grouped f32 sweep / the per-user exact kernel -- with the right answer (lowest indices first)."""
    from chaorec_amd import ops
    rng = np.random.default_rng(77)
    U, I, D, K = 40, 131072 + 64, 64, 50
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = np.repeat((rng.standard_normal((1, D)) * 0.2).astype(np.float32), I, 0)
    want_i, want_v = oracle.score_topk(ue, ie, None, 1e-6, K, 0)
    st = {}
    hint = torch.full((U,), float("nan"), device=dev)
    got_i, got_v = ops.score_topk(torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev), None, 1e-6, K, stats=st, hint=hint,
                                  hint_valid=False)
    assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i)
    assert st["fallback_users"] == U, st
    # the grouped f32 sweep leaves a threshold for the next call too (it used to leave the one that had just failed)
    assert bool(torch.isfinite(hint).all()) and bool((hint < torch.from_numpy(want_v[:, K - 1].copy()).to(dev)).all())


def _call(ops, lib, ue, ie, hist, K, U0, hint, hint_valid, phase, ws, idx, val, counters=None):
    nb = lib.chaorec_score_topk_workspace_bytes(ue.shape[0], ie.shape[0], K, ue.shape[1])
    ops._score_call(lib, ue, ie, hist, 1e-6, K, U0, 0, hint, hint_valid, 80, False, counters, idx, val, ws, nb, phase=phase)


@pytest.mark.parametrize("U,I,D", [(300, 9000, 64), (150, 140000, 128), (70, 1000, 64)])
@pytest.mark.parametrize("hinted", [False, True])
def test_score_topk_front_then_back_is_the_whole_call(dev, U, I, D, hinted):
    """chaorec_score_topk_hinted_f32 with CHAOREC_SCORE_FRONT and then CHAOREC_SCORE_BACK (same arguments, same workspace;
    here also on two streams with an event between them) = the call without flags: same indices, values, thresholds,
    counters.  (I = 1000 takes the route without a prefilter: the back call does everything.)"""
    from chaorec_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(U + I)
    K = 50
    ue = torch.randn(U, D, generator=g, device=dev) * 0.2
    ie = torch.randn(I, D, generator=g, device=dev) * 0.2
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 5
    col = (torch.arange(U * 5, device=dev) % 5 * 37 + torch.arange(U * 5, device=dev) // 5 % 11).to(torch.int32)
    hist = (rowptr, col)
    nb = lib.chaorec_score_topk_workspace_bytes(U, I, K, D)
    hint0 = torch.zeros(U, device=dev)
    if hinted:
        ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint0, hint_valid=False)
        ue = ue + 0.01 * torch.randn(U, D, generator=g, device=dev)      # the tables moved a little since

    def run(split):
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        idx = torch.empty((U, K), dtype=torch.int64, device=dev)
        val = torch.empty((U, K), device=dev)
        hint = hint0.clone()
        cnt = torch.full((4,), -1, dtype=torch.int32, device=dev)
        if not split:
            _call(ops, lib, ue, ie, hist, K, U, hint, hinted, 0, ws, idx, val, cnt)
        else:
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            _call(ops, lib, ue, ie, hist, K, U, hint, hinted, ops.SCORE_FRONT, ws, idx, val, cnt)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                _call(ops, lib, ue, ie, hist, K, U, hint, hinted, ops.SCORE_BACK, ws, idx, val, cnt)
            side.synchronize()
        torch.cuda.synchronize()
        return idx, val, hint, cnt

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    if I >= 4096:
        assert bool((a[3] >= 0).all())                    # (the prefilter route reports its queue lengths)


@pytest.mark.parametrize("fill", [0xFF, 0x5A])
def test_ranking_does_not_depend_on_what_its_workspace_held(dev, fill):
    """The ranking call in one piece and as FRONT + BACK phases, unhinted and hinted, over a workspace pre-filled with junk bytes
    (what torch.empty hands out in a long-running process): the same [U, K] lists as over a fresh one."""
    from chaorec_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(3)
    U, I, D, K = 8192, 6000, 64, 50
    ue = torch.randn(U, D, generator=g, device=dev) * 0.2
    ie = torch.randn(I, D, generator=g, device=dev) * 0.2
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 3
    col = ((torch.arange(U * 3, device=dev) % 3) * 1000 + torch.arange(U * 3, device=dev) // 3 % 997).to(torch.int32)
    hist = (rowptr, col)
    nb = lib.chaorec_score_topk_workspace_bytes(U, I, K, D)
    ref_i, ref_v = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U)
    h0 = torch.empty(U, device=dev)
    ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=h0, hint_valid=False)
    for hinted in (False, True):
        for phased in (False, True):
            ws = torch.full((nb,), fill, dtype=torch.uint8, device=dev)
            idx = torch.empty((U, K), dtype=torch.int64, device=dev)
            val = torch.empty((U, K), dtype=torch.float32, device=dev)
            a = (lib, ue, ie, hist, 1e-6, K, U, 0, h0.clone() if hinted else None, hinted, 0, False,
                 torch.zeros(4, dtype=torch.int32, device=dev), idx, val, ws, nb)
            if phased:
                ops._score_call(*a, phase=ops.SCORE_FRONT)
                ops._score_call(*a, phase=ops.SCORE_BACK)
            else:
                ops._score_call(*a)
            assert torch.equal(idx, ref_i) and torch.equal(val, ref_v), (hinted, phased)


@pytest.mark.parametrize("hinted", [False, True])
def test_score_topk_user_ranges_equal_the_call_in_one_piece(dev, oracle, hinted):
    """ops.score_topk over user ranges (the workspace budget forces >= 5 of them, one after the other on the caller's stream) ==
    the same call in one piece == the oracle (a sample of the rows); thresholds and queue counters come out the same."""
    from chaorec_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(3)
    U, I, D, K = 30_000, 6_000, 64, 50
    ue = torch.randn(U, D, generator=g, device=dev) * 0.2
    ie = torch.randn(I, D, generator=g, device=dev) * 0.2
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 3
    col = ((torch.arange(U * 3, device=dev) % 3) * 1000 + torch.arange(U * 3, device=dev) // 3 % 997).to(torch.int32)
    hist = (rowptr, col)
    one = lib.chaorec_score_topk_workspace_bytes(U, I, K, D)
    base = {}
    if hinted:
        h0 = torch.empty(U, device=dev)
        ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=h0, hint_valid=False)
        base = dict(hint_valid=True)
    saved = os.environ.get("CHAOREC_SCORE_WS_LIMIT")
    out = {}
    try:
        for mode, limit in (("ranges", str(one // 5)), ("whole", None)):
            if limit is None:
                os.environ.pop("CHAOREC_SCORE_WS_LIMIT", None)
            else:
                os.environ["CHAOREC_SCORE_WS_LIMIT"] = limit
            st = {}
            kw = dict(base)
            if hinted:
                kw["hint"] = h0.clone()
                kw["counters"] = torch.zeros(4, dtype=torch.int32, device=dev)
            i_, v_ = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, stats=st, **kw)
            torch.cuda.synchronize()
            out[mode] = (i_, v_, st, kw.get("hint"), kw.get("counters"))
    finally:
        if saved is None:
            os.environ.pop("CHAOREC_SCORE_WS_LIMIT", None)
        else:
            os.environ["CHAOREC_SCORE_WS_LIMIT"] = saved
    p, s = out["ranges"], out["whole"]
    assert p[2]["user_chunks"] >= 5 and "user_chunks" not in s[2]
    assert torch.equal(p[0], s[0]) and torch.equal(p[1], s[1])
    if hinted:
        assert torch.equal(p[3], s[3]) and torch.equal(p[4], s[4])
    rows = np.r_[0:40, U // 2:U // 2 + 40, U - 40:U]
    hr = rowptr.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    sub_col = []
    for n, r in enumerate(rows):
        sub_col.append(col[hr[r]:hr[r + 1]].cpu().numpy())
        sub_ptr[n + 1] = sub_ptr[n] + len(sub_col[-1])
    want_i, want_v = oracle.score_topk(ue[rows].cpu().numpy(), ie.cpu().numpy(), (sub_ptr, np.concatenate(sub_col)), 1e-6, K, U)
    assert np.array_equal(p[0][rows].cpu().numpy(), want_i) and np.array_equal(p[1][rows].cpu().numpy(), want_v)


@pytest.mark.parametrize("D", [64, 128])
def test_score_topk_scaled_thresholds_edge_cases(dev, oracle, D):
    """The sweep scales a user's fragments by 1 / |theta_u| (theta_u = T_u - c ||u|| max ||i||) and compares against the
    MFMA's inline constant: negative thresholds (every score of a user negative: the flipped-sign lanes), zero rows, rows
    and thresholds at the ends of the float range, thresholds that are garbage (0, +-tiny, +-huge, NaN, +-inf) must all
    end in the oracle's top-K -- a threshold only ever changes the work."""
    from chaorec_amd import ops
    rng = np.random.default_rng(D)
    U, I, K = 160, 9000, 50
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    ie[:, 0] = np.abs(ie[:, 0]) + 1.0
    ue[0:32, 0] = -3.0                  # all scores negative: T_u < 0
    ue[32:40] = 0.0                     # zero rows: every score 0, ties to the lowest index
    ue[40:48] *= 1e-18                  # tiny rows
    ue[48:56] *= 1e12                   # huge rows
    ue[56:64, 0] = 3.0                  # all scores positive and far from 0
    hist = _hist_random(U, I, 20, seed=D)
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    due, die = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    hint = torch.empty(U, device=dev)
    got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=False)
    assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i)
    good = hint.clone()
    st = {}
    got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=True, stats=st)
    assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i)
    assert st["fallback_users"] <= 16, st               # carried thresholds certify (nearly) everybody, negative ones too
    junk = torch.tensor([0.0, -0.0, 1e-38, -1e-38, 1e-30, -1e-30, 1e30, -1e30, 3e38, -3e38, float("nan"), float("inf"),
                         float("-inf"), 1.0, -1.0, 1e-3], device=dev)
    for shift in range(3):
        hint = good.clone()
        hint[shift::3] = junk[(torch.arange(len(hint[shift::3]), device=dev) + shift) % len(junk)]
        got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=True)
        assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i), shift


# ---- norm-sorted packed tables (score_prefilter.hpp "norm classes"): the bound without its MFMA ----------------------------
class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("D", [64, 128])
def test_score_topk_norm_sorted_table_equals_the_oracle(dev, oracle, D):
    """The long-range layout forced onto a table the oracle ranks in seconds (CHAOREC_PF_CLS_MIN_ITEMS): items sorted by norm
    class, users' fragments re-scaled per run of classes, candidates mapped back through the permutation.  Item norms span
    five octaves (so the walk re-scales many times and several runs are merged), some items are zero rows, some users have
    all-negative scores, zero / tiny / huge rows; history members among the best items.  Cold call, carried thresholds, and
    carried thresholds that are garbage must all give Model/LightGCN.py:147-155's top-K bit for bit."""
    from chaorec_amd import ops
    rng = np.random.default_rng(100 + D)
    U, I, K = 200, 20000, 50
    ue = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    ie *= np.exp2(rng.uniform(-3.0, 2.0, (I, 1))).astype(np.float32)
    ie[rng.choice(I, 50, replace=False)] = 0.0
    ie[:, 0] = np.abs(ie[:, 0])
    ue[0:32, 0] = -3.0                  # (nearly) all scores negative: T_u < 0
    ue[32:40] = 0.0
    ue[40:48] *= 1e-18
    ue[48:56] *= 1e12
    rp, cl = _hist_random(U, I, 20, seed=D)
    raw = ue.astype(np.float64) @ ie.astype(np.float64).T
    rows = [cl[rp[u]:rp[u + 1]] for u in range(U)]
    for u in range(0, U, 5):            # the best items of some users are in their history
        rows[u] = np.unique(np.r_[rows[u], np.argsort(-raw[u])[:30]]).astype(np.int32)
    hist = (np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.int64), np.concatenate(rows).astype(np.int32))
    want_i, want_v = oracle.score_topk(ue, ie, hist, 1e-6, K, U)
    dh = (torch.from_numpy(hist[0]).to(dev), torch.from_numpy(hist[1]).to(dev))
    due, die = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    with _env(CHAOREC_PF_CLS_MIN_ITEMS=1):
        hint = torch.empty(U, device=dev)
        st = {}
        got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=False, stats=st)
        assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i)
        assert st["prefilter_users"] == U and st["fallback_users"] <= 24, st
        good = hint.clone()
        st = {}
        got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=True, stats=st)
        assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i)
        assert st["fallback_users"] <= 16, st
        junk = torch.tensor([0.0, -0.0, 1e-38, -1e-38, 1e-30, -1e-30, 1e30, -1e30, 3e38, -3e38, float("nan"), float("inf"),
                             float("-inf"), 1.0, -1.0, 1e-3], device=dev)
        for shift in range(3):
            hint = good.clone()
            hint[shift::3] = junk[(torch.arange(len(hint[shift::3]), device=dev) + shift) % len(junk)]
            got_i, got_v = ops.score_topk(due, die, dh, 1e-6, K, id_offset=U, hint=hint, hint_valid=True)
            assert np.array_equal(got_v.cpu().numpy(), want_v) and np.array_equal(got_i.cpu().numpy(), want_i), shift


@pytest.mark.parametrize("D,law", [(128, "lognormal"), (64, "level"), (128, "outliers")])
def test_score_topk_norm_sorted_table_equals_the_table_in_its_own_order(dev, D, law):
    """Long item range (the default there): the sorted layout against the per-item bound on its own MFMA k-step
    (CHAOREC_PF_CLS_MIN_ITEMS=0), every row of a few thousand users, cold and with carried thresholds.  Same indices and
    values; the candidate sets may differ (the bound of a run of classes is its largest norm), but not by much."""
    from chaorec_amd import ops
    g = torch.Generator(device=dev).manual_seed(7 + D)
    U, I, K = 4096, 200_000, 50
    ue = torch.randn(U, D, generator=g, device=dev) * 0.1
    ie = torch.randn(I, D, generator=g, device=dev) * 0.1
    if law == "lognormal":
        ie *= torch.exp2(torch.randn(I, 1, generator=g, device=dev) * 0.7)
    elif law == "outliers":
        ie[torch.randint(0, I, (300,), generator=g, device=dev)] *= 30.0
        ie[torch.randint(0, I, (5,), generator=g, device=dev)] *= 1e6
    rowptr = torch.arange(U + 1, dtype=torch.int64, device=dev) * 6
    col = ((torch.arange(U * 6, device=dev) % 6) * 30011 + torch.arange(U * 6, device=dev) // 6 * 7 % 30011).to(torch.int32)
    hist = (rowptr, col)
    res = {}
    for name, v in (("own", 0), ("sorted", 131072)):
        with _env(CHAOREC_PF_CLS_MIN_ITEMS=v):
            hint = torch.empty(U, device=dev)
            st, st2 = {}, {}
            i0, v0 = ops.score_topk(ue, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=False, stats=st)
            ue2 = ue + 0.002 * torch.randn(U, D, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
            i1, v1 = ops.score_topk(ue2, ie, hist, 1e-6, K, id_offset=U, hint=hint, hint_valid=True, stats=st2)
            torch.cuda.synchronize()
            res[name] = (i0, v0, i1, v1, st, st2)
    a, b = res["own"], res["sorted"]
    for k in range(4):
        assert torch.equal(a[k], b[k]), k
    for k in (4, 5):
        assert b[k]["fallback_users"] <= a[k]["fallback_users"] + 8, (a[k], b[k])
        assert b[k]["candidates"] <= 1.5 * a[k]["candidates"] + 128 * U, (a[k], b[k])


@pytest.mark.parametrize("seed", range(8))
def test_score_topk_sorted_layout_fuzz_against_own_order_and_oracle(dev, oracle, seed):
    """Random shapes and tables through the norm-sorted layout (forced by CHAOREC_PF_CLS_MIN_ITEMS on tables of 17 k - 40 k
    items): K from 1 to 64, item counts that are no multiple of the tile, few / odd user counts, heavy-tailed norms, INTEGER
    tables (exact ties by the thousand: the lowest index must win although the sweep walks a permutation), histories of 0 to
    several hundred items, both mask values of the reference -- against the table in its own order for every row, and against
    the oracle for a sample of rows."""
    from chaorec_amd import ops
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([64, 128]))
    U = int(rng.choice([1, 33, 97, 300, 1025]))
    I = int(rng.integers(16_400, 40_000))
    K = int(rng.choice([1, 7, 20, 50, 64]))
    mask_value = float(rng.choice([1e-6, 1e-5]))
    kind = seed % 4
    if kind == 0:                                   # integer tables: massive ties
        ue = rng.integers(-2, 3, (U, D)).astype(np.float32)
        ie = rng.integers(-1, 2, (I, D)).astype(np.float32)
    else:
        ue = (rng.standard_normal((U, D)) * 0.3).astype(np.float32)
        ie = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
        if kind == 1:
            ie *= np.exp2(rng.standard_normal((I, 1)) * 1.2).astype(np.float32)
        elif kind == 2:
            ie[rng.choice(I, 40, replace=False)] *= 1e4
            ie[rng.choice(I, 400, replace=False)] = 0.0
            ue[::7] = -np.abs(ue[::7])
            ie[:, :] = np.abs(ie)                   # users with all-negative scores
    rowptr = np.zeros(U + 1, np.int64)
    cols = []
    for u in range(U):
        deg = int(rng.choice([0, 3, 40, 600])) if u % 5 else 0
        c = np.sort(rng.choice(I, min(deg, I), replace=False)).astype(np.int32)
        cols.append(c)
        rowptr[u + 1] = rowptr[u] + len(c)
    col = np.concatenate(cols).astype(np.int32) if rowptr[-1] else np.zeros(0, np.int32)
    dh = (torch.from_numpy(rowptr).to(dev), torch.from_numpy(col if len(col) else np.zeros(1, np.int32)).to(dev))
    due, die = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    out = {}
    for name, v in (("own", 0), ("sorted", 1)):
        with _env(CHAOREC_PF_CLS_MIN_ITEMS=v):
            hint = torch.empty(U, device=dev)
            a = ops.score_topk(due, die, dh, mask_value, K, id_offset=U, hint=hint, hint_valid=False)
            b = ops.score_topk(due, die, dh, mask_value, K, id_offset=U, hint=hint, hint_valid=True)
            torch.cuda.synchronize()
            out[name] = a + b
    for x, y in zip(out["own"], out["sorted"]):
        assert torch.equal(x, y), (seed, D, U, I, K, kind)
    rows = np.unique(np.r_[0, U // 2, U - 1, rng.integers(0, U, 6)])
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    sub_col = []
    for n, r in enumerate(rows):
        sub_col.append(col[rowptr[r]:rowptr[r + 1]])
        sub_ptr[n + 1] = sub_ptr[n] + len(sub_col[-1])
    sc = np.concatenate(sub_col).astype(np.int32) if sub_ptr[-1] else np.zeros(0, np.int32)
    want_i, want_v = oracle.score_topk(ue[rows], ie, (sub_ptr, sc), mask_value, K, U)
    assert np.array_equal(out["sorted"][0][rows].cpu().numpy(), want_i) and np.array_equal(out["sorted"][1][rows].cpu().numpy(), want_v)
