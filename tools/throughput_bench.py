#!/usr/bin/env python3
"""Proofs per second with ONE proof in flight (the bench line's latency loop) against TWO (two host threads calling zk_bn254_groth16_prove on the same resident
key: the second proof's transforms and scalar preparation run under the first one's reduction tails).  A proof session takes five of an entry's stream slots, so
two at once need a build with more than the shipped eight (make -C noir_backend_using_gnark_amd/csrc EXPERIMENTS=1 EXTRA=-DZKMI_NSLOTS=12; with eight the
second thread simply waits for the first: the library serialises, it does not fail).

    python tools/throughput_bench.py [log_n] [proofs] [--lib exp]

Prints one JSON object."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
from noir_backend_using_gnark_amd import _lib  # noqa: E402
if "--lib" in sys.argv:
    which = sys.argv[sys.argv.index("--lib") + 1]
    _lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "build_exp", "libzkmi_exp.so") if which == "exp" else which)
import noir_backend_using_gnark_amd as zk  # noqa: E402
from bench_blocks.common import Instance, N_PUBLIC  # noqa: E402

log_n = int(args[0]) if args else 20
count = int(args[1]) if len(args) > 1 else 100
L = _lib.lib()
inst = Instance(L, _lib, zk, log_n, 0, N_PUBLIC, 0, True)
prove = lambda: zk.prove(inst.pk, inst.d_a, inst.d_b, inst.d_c, inst.d_w, inst.r, inst.s, n_constraints=inst.N, on_device=True)
want = prove()
for _ in range(5):
    assert prove() == want


def run(threads):
    bad = []

    def work(k):
        for _ in range(count // threads):
            if prove() != want:
                bad.append(k)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    _lib.check(L.zk_dev_sync())
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    _lib.check(L.zk_dev_sync())
    dt = time.perf_counter() - t0
    return {"threads": threads, "proofs": count // threads * threads, "ms_per_proof": round(1e3 * dt / (count // threads * threads), 3), "bytes_identical": not bad}


out = {"log_n": log_n, "library": os.path.basename(_lib.LIB_PATH), "runs": [run(1), run(2), run(1), run(2), run(3)]}
print(json.dumps(out))
