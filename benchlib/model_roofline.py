"""`models.<X>.roofline`: the dominant kernel of a model's captured step against the roof that bounds it.

The kernel's duration comes from a committed rocprofv3 kernel trace of `bench.py --model X` on one stream
(profiles/model_kernel_times.json, written by tools/collect_model_profiles.py) and is only quoted when the kernel sources it
was measured on are the ones this run was built from; the ALGORITHMIC flops / bytes are computed here from the live model's
shapes.  Nothing of this is in the timed region."""
import hashlib
import json
import os

from .common import BF16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT


def _sha(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()


def _pick(kernels, substr):
    """The (kernel, grid) group with the most time among those whose name holds `substr`."""
    hit = [k for k in kernels if substr in k["name"]]
    return max(hit, key=lambda k: k["total_us"]) if hit else None


def model_roofline(name, model, batch_rows=None):
    path = os.path.join(ROOT, "profiles", "model_kernel_times.json")
    if not os.path.exists(path):
        return {"error": "profiles/model_kernel_times.json not collected"}
    prof = json.load(open(path))
    stale = [f for f, h in prof.get("sources", {}).items() if _sha(f) != h]
    m = prof.get("models", {}).get(name)
    if m is None:
        return {"error": f"no profile of {name}"}
    step_us = sum(k["total_us"] for k in m["kernels"]) or 1.0
    if name == "MMGCN":
        # the textual branch's first convolution (Model/MMGCN.py:102-105, BasicGCN.py:40-53 on the 768-wide features): its
        # weight gradient  gs^T [A x | A 1]  -- [n, 768]^T [n, 772] -- is the step's largest launch; its forward twin
        # [A x | A 1] [W | b]^T has the same flops
        k = _pick(m["kernels"], "gemm_bf16x3_kernel<true, true, 128, false>")
        const = getattr(model.t_gcn, "_const", None)
        if k is None or const is None:
            return {"error": "dominant kernel not in the profile"}
        n, kdim = const[1].shape
        mdim = model.t_gcn.conv_embed_1.lin.weight.shape[0]
        flops = 2.0 * n * kdim * mdim
        ach = flops / (k["avg_us"] * 1e-6) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0
        out = {"bound": "mfma", "dominant_kernel": k["name"].split("::")[-1] + f" grid {k['grid']}",
               "what": f"weight gradient of the textual branch's first convolution: [{n}, {mdim}]^T [{n}, {kdim}], fp32 operands as three "
                       "bf16 planes, six v_mfma_f32_32x32x16_bf16 plane products per fp32 product",
               "algorithmic_flops": flops, "avg_launch_us": k["avg_us"], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
               "frac": ach / peak, "peak_is": "bf16 dense MFMA peak / 6 plane products (fp32-grade accuracy on the bf16 pipe)",
               "frac_of_bf16_peak_counting_2MNK_only": ach / BF16_MFMA_PEAK_TFLOPS,
               "multiple_of_f32_mfma_peak": ach / F32_MFMA_PEAK_TFLOPS}
    else:
        # FREEDOM: the trainable image table's lazy Adam rows (Model/FREEDOM.py:59, main.py:397 as exact deferred updates): the
        # catch-up launch before the forward replays the steps the batch's rows sat out -- parameter and both moments read and
        # written per touched row
        k = _pick(m["kernels"], "adam_lowrank_rows_kernel<3")
        if k is None:
            return {"error": "dominant kernel not in the profile"}
        kdim = model.image_embedding.weight.shape[1]
        rows = int(batch_rows) if batch_rows else 0
        by = rows * kdim * 4.0 * 6.0
        ach = by / (k["avg_us"] * 1e-6) / 1e9
        out = {"bound": "hbm", "dominant_kernel": k["name"].split("::")[-1] + f" grid {k['grid']}",
               "what": f"catch-up of the {rows} distinct batch rows of the [{model.image_embedding.weight.shape[0]}, {kdim}] image table "
                       "(parameter + two moments, read and written)",
               "algorithmic_bytes": by, "avg_launch_us": k["avg_us"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS,
               "note": "bound by the replayed updates' arithmetic (up to dozens of deferred steps per row), not by bytes"}
    out["share_of_step"] = k["total_us"] / step_us
    out["kernel"], out["traffic"] = out["dominant_kernel"], None       # (the keys of the headline's roofline object)
    out["source"] = m.get("csv")
    out["kernel_time_is"] = "rocprofv3 kernel trace of `bench.py --model %s` on one stream (not measured in this run)" % name
    if stale:
        out["stale"] = "measured on other sources than this build: " + ", ".join(os.path.basename(f) for f in stale)
    return out
