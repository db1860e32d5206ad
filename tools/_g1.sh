timeout 1200 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "fuzz" 2>&1 | tail -30
