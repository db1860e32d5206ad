cd $GRAFT_REPO_ROOT
echo "== gemm shapes (WS v3: ring of 4 tiles)"; timeout 300 python tools/gemm_wide_bench.py mmgcn 2>&1 | tail -7
echo "== WS test"; timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu -k wave_specialised 2>&1 | tail -3
echo "== epoch-1 probe (captured)"; timeout 900 python tools/_epoch1_probe.py 2>&1 | grep -v Warning | tail -24
echo "== epoch-1 probe (eager)"; GRAPH=0 SEEDS=1,2,3,7,42,99 timeout 900 python tools/_epoch1_probe.py 2>&1 | grep -v Warning | tail -9
