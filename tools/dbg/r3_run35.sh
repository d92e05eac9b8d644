#!/bin/bash
# which part of the day's restructuring of the pass kernels costs the 2^26 transform 3-6 %: variants of ntt.hip linked into the experiments library, one box.
# The variants are built by hand beforehand (not kept in the tree): `git show <commit>:.../ntt.hip` or the current file with -D switches -> hipcc -c with the flags of
# `make EXPERIMENTS=1` -> hipcc -shared with build_exp/*.o and that ntt.o -> noir_backend_using_gnark_amd/variants/libzkmi_exp_<name>.so; the loop below copies each over
# libzkmi_exp.so in turn.  Results of the runs: DESIGN.md 8 (round 3, the unit-stage item), profiles/r03_j_ntt_kernel_variants_last_run.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3z; mkdir -p $O
cd $R
export ZKMI_USE_EXPERIMENTS_LIB=1
cat > /tmp/ntt26.py <<'PY'
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
from noir_backend_using_gnark_amd import _lib
import noir_backend_using_gnark_amd as zk
L = _lib.lib()
out = []
for log_n in (22, 24, 26):
    n = 1 << log_n
    d = _lib.DeviceBuffer(n * 32)
    _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_uint64(5), C.c_int(1), C.c_int(0), None))
    f = lambda: _lib.check(L.zk_bn254_ntt_dev(C.c_void_p(d.ptr), C.c_uint32(log_n), C.c_int(0), C.c_int(zk.DIF), C.c_int(0), None))
    for _ in range(3): f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort(); out.append((log_n, round(ts[0], 3), round(ts[7], 3)))
    d.free()
print(sys.argv[1], out)
PY
for rep in 1 2 3; do
for v in head mid; do
cp noir_backend_using_gnark_amd/variants/libzkmi_exp_$v.so noir_backend_using_gnark_amd/libzkmi_exp.so
python /tmp/ntt26.py $v 2>&1 | tail -1
done
done | tee $O/ntt_variants.txt
