"""What would column (source-range) blocking buy the SpMM in the HBM regime?  The blocked form makes every launch gather
from a source block that stays in the 256 MiB Infinity Cache; its ceiling is therefore the rate at which THIS kernel
gathers rows from a cache-resident table.  Measured directly: the same number of output rows and entries (3 M rows x 16
entries, D = 128: 24.6 GB of gathered rows per launch), sources drawn uniformly from a table of 32 MB ... 4 GB.

    python3 tools/spmm_locality.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from chaorec_amd import _lib, graph, ops
    _lib.ensure_built()
    dev = torch.device("cuda:0")
    D, n_rows, deg = 128, 3_000_000, 16
    g = torch.Generator(device=dev).manual_seed(1)
    rowptr = torch.arange(0, (n_rows + 1) * deg, deg, dtype=torch.int64, device=dev)
    val = torch.rand(n_rows * deg, generator=g, device=dev) * 0.1
    out = {}
    for table_mb in (32, 128, 224, 512, 2048, 4096):
        n_cols = table_mb * (1 << 20) // (4 * D)
        col = torch.randint(0, n_cols, (n_rows * deg,), generator=g, device=dev, dtype=torch.int32)
        csr = graph.CSR(rowptr, col, val, n_rows, n_cols)
        x = torch.randn(n_cols, D, generator=g, device=dev)
        y = torch.empty(n_rows, D, device=dev)
        csr.schedule(D)
        for _ in range(2):
            ops.spmm_raw(csr, x, y=y)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.spmm_raw(csr, x, y=y)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        model = n_rows * deg * (4 * D + 8) + n_rows * (4 * D + 8)
        out[f"{table_mb}MB"] = dict(ms=ms, model_TBps=model / ms / 1e9)
        print(f"source table {table_mb:5d} MB ({n_cols:8d} rows): {ms:7.3f} ms per launch = {model / ms / 1e9:5.2f} TB/s of model bytes",
              flush=True)
        del csr, x, y, col
        torch.cuda.empty_cache()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "spmm_locality.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
