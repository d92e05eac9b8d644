set -u
O=gpurun_out/r04t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batched_multi_exp or registered_bases or prepared_scalars" > $O/pytest_batch.txt 2>&1; tail -15 $O/pytest_batch.txt
timeout 1200 python -m pytest tests/test_gpu_plonk.py -m gpu -q -x > $O/pytest_plonk.txt 2>&1; tail -4 $O/pytest_plonk.txt
