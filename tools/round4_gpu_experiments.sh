set -u
O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -q -x -k "lagrange or plonk or batched or window_tables" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
