"""Prints the host-side sections of a Groth16 proof at 2^20 (library event / wall-clock profile) next to the per-step time."""
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
import bench
from noir_backend_using_gnark_amd import _lib, groth16 as zk
L = _lib.lib()
_lib.check(L.zk_init(C.c_int(0)))
inst = bench.Instance(L, _lib, zk, 20, 0, 8, 0, True)
N = 1 << 20
prove = lambda: zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=N, on_device=True)
for _ in range(5): prove()
_lib.profile(True); _lib.profile_reset()
reps = 100
t0 = time.perf_counter()
for _ in range(reps): prove()
dt = (time.perf_counter() - t0) / reps * 1e3
_lib.profile(False)
p = _lib.profile_read()
print(json.dumps({"ms_per_proof": round(dt, 3), "host_sections_ms": {k: round(v[1] / reps, 4) for k, v in p.items() if k.startswith("host")}}))
# python + ctypes overhead of an empty call for scale
t0 = time.perf_counter()
for _ in range(1000): inst.pk.info()
print("pk.info() call us:", round((time.perf_counter() - t0) * 1e3, 2))
