import os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
for v in ["-DCHAOREC_PF_RANK=10", "-DCHAOREC_PF_RANK=8", "-DCHAOREC_PF_RANK=7", "-DCHAOREC_PF_RANK=6"]:
    env = dict(os.environ, CHAOREC_EXTRA_HIPCC_FLAGS=v, EPOCHS="0")
    subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT, env=env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "score_profile.py"), "3000"], cwd=ROOT, env=env, capture_output=True, text=True).stdout
    print(v, [l for l in out.splitlines() if l.startswith("cold")], flush=True)
subprocess.check_call([sys.executable, "-c", "from chaorec_amd import _lib; _lib.build(force=True)"], cwd=ROOT)
