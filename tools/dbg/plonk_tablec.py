"""A/B: PLONK 2^22 prove time against the window width of the SRS's window tables (argv[1]: 0 = planner's choice)."""
import sys, json, ctypes as C, functools
sys.path.insert(0, ".")
import bench
from noir_backend_using_gnark_amd import _lib, bn254 as zb
c = int(sys.argv[1])
orig = zb.ResidentBases
class RB(orig):
    def __init__(self, points, is_g2=False, n=None, table_window_bits=0):
        super().__init__(points, is_g2, n, table_window_bits=c)
zb.ResidentBases = RB
L = _lib.lib()
_lib.check(L.zk_init(C.c_int(0)))
d = bench.plonk_block(L, _lib, 22, reps=5)
print("table c =", c, d["prove_ms"], d["proof_verifies"])
