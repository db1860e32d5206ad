timeout 1500 python -m pytest tests/test_gpu_round5.py -q -m gpu 2>&1 | tail -5
timeout 600 python3 tools/score_sorted_probe.py 4096 200000 2>&1 | grep 'D=128' 
