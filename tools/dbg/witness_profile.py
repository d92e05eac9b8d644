"""Kernel-time profile (library event pairs) of a 2^20 Groth16 proof with uniform vs witness-like wire values."""
import ctypes as C, json, sys, time
sys.path.insert(0, ".")
import bench
from noir_backend_using_gnark_amd import _lib, groth16 as zk
L = _lib.lib()
_lib.check(L.zk_init(C.c_int(0)))
N = 1 << 20
for wit in (0, 1):
    inst = bench.Instance(L, _lib, zk, 20, 0, 8, wit, True)
    prove = lambda: zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=N, on_device=True)
    for _ in range(5): prove()
    _lib.profile(True); _lib.profile_reset()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): prove()
    dt = (time.perf_counter() - t0) / reps * 1e3
    _lib.profile(False)
    p = _lib.profile_read()
    top = sorted(((v[1] / reps, k, v[0] // reps) for k, v in p.items()), reverse=True)[:14]
    print(json.dumps({"witness_like": wit, "ms_per_proof": round(dt, 3), "kernels_ms(per proof, launches)": {k: (round(ms, 3), n) for ms, k, n in top}}))
    inst.free()
