timeout 900 python -m pytest tests/test_gpu_sparse_family.py -q -m gpu -k "lightgt or LightGT or history_sequences" 2>&1 | grep -v ' INFO \|Warning' | tail -40
