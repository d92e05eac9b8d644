set -u
O=gpurun_out/r05x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_keyio.py -m gpu -q -x -k "table or registered or prepared or batched or keyio or g2 or random_pk" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python - <<'PY' | tee $O/table_build_g2.txt
import time, ctypes as C
from noir_backend_using_gnark_amd import _lib as lib, bn254 as zb
L = lib.lib()
n = 1 << 20
d = lib.DeviceBuffer(n * 128)
lib.check(L.zk_bn254_g2_generate_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(7), None))
lib.check(L.zk_dev_sync())
for rep in range(3):
    t = time.perf_counter(); rb = zb.ResidentBases(d, n=n, is_g2=True, table_window_bits=0); lib.check(L.zk_dev_sync()); dt = (time.perf_counter() - t) * 1e3
    print("register %d G2 bases with window tables: %.2f ms" % (n, dt)); rb.free()
PY
timeout 600 python bench.py --steps 20 --no-2p24 --no-plonk --no-micro --no-export --no-cpu-baseline --no-host-inputs > $O/b.json 2> $O/b.err; python -c "
import json;d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]);print('2^20',d['ms_per_step'],'setup_s',d['setup_s'],d['proof_sha'])"
