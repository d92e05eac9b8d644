#!/bin/bash
# unit twiddles of the stage on bit 1 inside the group that holds bit 0 (experiments library = new ntt.hip, product library = the commit before), one box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3af; mkdir -p $O
cd $R
ZKMI_USE_EXPERIMENTS_LIB=1 timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or compute_h or golden or sharded" > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
cat > /tmp/ntt26.py <<'PY'
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
from noir_backend_using_gnark_amd import _lib
import noir_backend_using_gnark_amd as zk
L = _lib.lib()
out = []
for dec in (zk.DIF, zk.DIT):
  for log_n in (20, 24, 26):
    n = 1 << log_n
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(5), C.c_int(1), C.c_int(0), None))
    f = lambda: _lib.check(L.zk_bn254_ntt_dev(C.c_void_p(d.ptr), C.c_uint32(log_n), C.c_int(0), C.c_int(dec), C.c_int(0), None))
    for _ in range(3): f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort(); out.append(("DIF" if dec == zk.DIF else "DIT", log_n, round(ts[0], 3), round(ts[7], 3)))
    d.free()
print(sys.argv[1], out)
PY
for rep in 1 2 3; do
python /tmp/ntt26.py head 2>&1 | tail -1
ZKMI_USE_EXPERIMENTS_LIB=1 python /tmp/ntt26.py new 2>&1 | tail -1
done | tee $O/ntt.txt
