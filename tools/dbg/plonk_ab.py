import sys, json, ctypes as C
sys.path.insert(0, ".")
import bench
from noir_backend_using_gnark_amd import _lib
L = _lib.lib()
_lib.check(L.zk_init(C.c_int(0)))
d = bench.plonk_block(L, _lib, 22, reps=5)
k = d["kernel_ms_per_proof"]
print(sys.argv[1], d["prove_ms"], d["proof_verifies"], {x: k.get(x) for x in ("msm_reduce_wave", "msm_reduce_l1", "msm_reduce_l2", "msm_fold_multi")})
