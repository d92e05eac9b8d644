import sys, ctypes as C
sys.path.insert(0, ".")
import bench
from noir_backend_using_gnark_amd import _lib
L = _lib.lib()
_lib.check(L.zk_init(C.c_int(0)))
d = bench.plonk_block(L, _lib, 22, reps=2)
print(d["prove_ms"])
