#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
cat > /tmp/t.py <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import noir_backend_using_gnark_amd as zk
from oracle import bn254_ref as ref
import hashlib, numpy as np
g = json.load(open("tests/golden/bn254_golden.json")) if os.path.exists("tests/golden/bn254_golden.json") else None
import glob
if g is None:
    for f in glob.glob("tests/golden/*.json"):
        d = json.load(open(f))
        if isinstance(d, dict) and "ntt" in d: g = d; print("golden", f); break
from tests.helpers import mont_limbs, sha_image
bad = 0
for rep in range(3):
  for e in g["ntt"]:
    if e["kind"] != "modes": continue
    n = 1 << e["log_n"]
    x = mont_limbs(ref.rand_felts(e["seed"], n))
    d = zk.Domain(n)
    (d.fft_inverse if e["inverse"] else d.fft)(x, e["decimation"], bool(e["coset"]))
    ok = sha_image(x) == e["sha256"]
    if not ok: bad += 1; print("BAD rep", rep, {k: e[k] for k in ("log_n", "inverse", "decimation", "coset")})
print("bad", bad)
PY
echo "== product"; python /tmp/t.py 2>&1 | tail -8
echo "== exp"; ZKMI_USE_EXPERIMENTS_LIB=1 python /tmp/t.py 2>&1 | tail -8

