set -u
O=gpurun_out/r05w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "table or registered or prepared or batched or keyio or groth16_prove_with_table or 2p22" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python - <<'PY' | tee $O/table_build.txt
import time, ctypes as C, os, sys
from noir_backend_using_gnark_amd import _lib as lib, bn254 as zb
for which in ("product", "exp_chain29_0"):
    pass
L = lib.lib()
for n in (1000000, 1 << 20, 1 << 22):
    d = lib.DeviceBuffer(n * 64)
    lib.check(L.zk_bn254_g1_generate_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(7), None))
    lib.check(L.zk_dev_sync())
    for rep in range(3):
        t = time.perf_counter(); rb = zb.ResidentBases(d, n=n, table_window_bits=0); lib.check(L.zk_dev_sync()); dt = (time.perf_counter() - t) * 1e3
        print("register %d G1 bases with window tables: %.2f ms" % (n, dt)); rb.free()
PY
