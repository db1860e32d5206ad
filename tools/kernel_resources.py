#!/usr/bin/env python3
"""VGPRs / scratch / LDS of every kernel of the product build (same flags as chaorec_amd/_lib.py), from the compiler's
own metadata:  python3 tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]   (no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chaorec_amd import _lib  # noqa: E402

csrc = os.path.join(ROOT, "chaorec_amd", "csrc")
flags = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]


def one(src):
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["hipcc"] + flags + ["-c", os.path.join(csrc, src), "-o", os.path.join(tmp, "x.o"), "-save-temps=obj"],
                              stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(tmp, asm)).read()
    out = []
    for block in text.split("  - .agpr_count:")[1:]:
        get = lambda key: re.search(r"\.%s:\s*(\S+)" % key, block)
        name = get("name").group(1)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        demangled = re.sub(r"\(.*", "", demangled)
        out.append((src, demangled, int(get("vgpr_count").group(1)), int(get("sgpr_count").group(1)),
                    int(get("private_segment_fixed_size").group(1)), int(get("group_segment_fixed_size").group(1))))
    return out


with ThreadPoolExecutor(max_workers=4) as ex:
    rows = [r for rs in ex.map(one, _lib.SOURCES) for r in rs]
print(f"{'file':18s} {'VGPR':>5s} {'SGPR':>5s} {'scratch B':>9s} {'LDS B':>7s}  kernel")
for src, name, v, s_, priv, lds in sorted(rows):
    print(f"{src:18s} {v:5d} {s_:5d} {priv:9d} {lds:7d}  {name[:110]}")
bad = [r for r in rows if r[4] > 0]
print(f"\n{len(rows)} kernels, {len(bad)} with scratch" + (": " + ", ".join(r[1] for r in bad) if bad else ""))
