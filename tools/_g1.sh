timeout 900 python -m pytest tests/test_gpu_real_data.py -q -m gpu -x -k "mmgcn_microlens" 2>&1 | grep -v Warning | grep -B2 -A22 'def test_mmgcn_microlens\|Error\|assert' | tail -70
