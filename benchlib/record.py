"""The bench line as a record: the LAST stdout line of bench.py is a compact JSON object of at most LINE_LIMIT bytes.

Everything measured (notes, per-launch tables, prefilter statistics, timed calls, thread sweeps) is the DETAIL: it is
written to bench_detail.json (CHAOREC_BENCH_DETAIL names another path) and to stderr, never to stdout.  The compact line
is a projection of the detail -- `compact()` computes nothing, it only selects and rounds -- so the two cannot disagree.
(Round 5's line had grown to 24 KB and the driver's kept stdout tail no longer held its head: BENCH_r05.parsed was null.)
"""
import json
import os
import sys

from .common import ROOT, flush_c_stdout

LINE_LIMIT = 4096

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _r(x, sig=6):
    """Numbers to `sig` significant digits (ints and everything else untouched)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _roofline(rf):
    if not isinstance(rf, dict):
        return None
    out = {k: _r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                      "algorithmic_bytes_per_launch", "avg_launch_us")}
    sc = rf.get("scoring")
    if isinstance(sc, dict):
        out["scoring"] = {"bound": sc.get("bound"), "kernel": _short(sc.get("kernel", ""), 64), "achieved": _r(sc.get("achieved")),
                          "peak": sc.get("peak"), "unit": sc.get("unit"), "frac": _r(sc.get("frac")),
                          "sweep_only_frac": _r(sc.get("sweep_only_frac"))}
    return out


def _lowest(rf):
    low = ((rf or {}).get("light_step_launches") or {}).get("lowest_frac")
    if not low:
        return None
    return {"name": _short(low.get("launch", ""), 72), "frac": _r(low.get("frac"), 4)}


def _sub(s):
    """A LightGCN sub-record (hbm_regime / config5_whole_on_one_gpu): what the step PERFORMED is its `value`."""
    if not isinstance(s, dict):
        return None
    if "error" in s:
        return {"error": _short(s["error"], 160)}
    rf = s.get("roofline") or {}
    sc = s.get("roofline_scoring") or {}
    out = {"ms_per_step": _r(s.get("ms_per_step")), "value": _r(s.get("value")),
           "value_reference_equivalent": _r(s.get("value_reference_equivalent")),
           "spmm_frac": _r(rf.get("frac"), 4), "lowest_launch": _lowest(rf), "scoring_frac": _r(sc.get("frac"), 4),
           "sweep_only_frac": _r(sc.get("sweep_only_frac"), 4),
           "gene_ranklist_ms": _r(s.get("gene_ranklist_ms", s.get("gene_ranklist_ms_cold")))}
    return {k: v for k, v in out.items() if v is not None}


def _model(m):
    if not isinstance(m, dict):
        return None
    if "error" in m:
        return {"error": _short(m["error"], 160)}
    out = {"ms_per_step": _r(m.get("ms_per_step")), "value": _r(m.get("value"))}
    rf = m.get("roofline")
    if isinstance(rf, dict):
        out["roofline"] = {k: (_short(rf[k], 72) if isinstance(rf.get(k), str) else _r(rf.get(k), 4))
                           for k in ("bound", "dominant_kernel", "share_of_step", "achieved", "peak", "unit", "frac", "stale", "error")
                           if k in rf}
    return out


def _cpu(c):
    if not isinstance(c, dict):
        return c
    out = {k: _r(c.get(k)) for k in ("value", "unit", "cores", "kind", "ms_per_step", "users_scored_per_s") if k in c}
    if "sample" in c:
        out["sample"] = _short(c["sample"], 200)
    if "reason" in c:
        out["reason"] = _short(c["reason"], 160)
    return out


def compact(detail, detail_path=None):
    """-> the dict of the compact line.  Pure projection of `detail` (the full result)."""
    cfg = detail.get("config") or {}
    line = {k: _r(detail.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                           "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    for k in ("users_scored_per_s", "users_scored_per_s_cold", "loss_mean", "multi_rank_rccl_measured"):
        if k in detail:
            line[k] = _r(detail[k])
    line["config"] = {"workload": _short(cfg.get("workload", ""), 400)}
    for k, n in (("gene_ranklist_ms", 0), ("gene_ranklist_ms_cold", 0), ("launch", 200), ("parallelism", 200)):
        if k in cfg:
            line["config"][k] = _short(cfg[k], n) if n else _r(cfg[k])
    line["roofline"] = _roofline(detail.get("roofline"))
    line["cpu_baseline"] = _cpu(detail.get("cpu_baseline"))
    for k in ("hbm_regime", "config5_whole_on_one_gpu"):
        if k in detail:
            line[k] = _sub(detail[k])
    if isinstance(detail.get("models"), dict):
        line["models"] = {name: _model(m) for name, m in detail["models"].items()}
    if detail.get("subrecords_timed_out"):
        line["subrecords_timed_out"] = detail["subrecords_timed_out"]
    if detail_path:
        line["detail"] = detail_path
    return line


def render(detail, detail_path=None):
    """-> the compact line as a string of at most LINE_LIMIT bytes.  Optional parts are dropped, in a fixed order, if a
    run's strings ever push the line past the limit (the required keys never are)."""
    line = compact(detail, detail_path)
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("models", "config5_whole_on_one_gpu", "hbm_regime", "loss_mean", "detail"):
        if len(s.encode()) <= LINE_LIMIT:
            break
        if drop in line:
            line[drop] = "see bench_detail.json"
            s = json.dumps(line, separators=(",", ":"))
    if len(s.encode()) > LINE_LIMIT:
        line["config"] = {"workload": _short(line["config"].get("workload", ""), 160)}
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"].pop("sample", None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s.encode()) <= LINE_LIMIT, len(s)
    return s


def emit(detail):
    """Rank 0's last act: detail -> file + stderr, the compact line -> stdout (the process's LAST stdout line)."""
    path = os.environ.get("CHAOREC_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    rel = None
    try:
        with open(path, "w") as f:
            json.dump(detail, f, indent=1, default=str)
        rel = os.path.relpath(path, ROOT)
    except OSError as exc:
        print(f"[bench] could not write {path}: {exc}", file=sys.stderr, flush=True)
    print("[bench detail] " + json.dumps(detail, default=str), file=sys.stderr, flush=True)
    flush_c_stdout()
    sys.stdout.flush()
    print(render(detail, rel), flush=True)
