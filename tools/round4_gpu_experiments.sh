set -u
O=gpurun_out/r05l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_goffi.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
