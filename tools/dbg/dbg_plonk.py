import sys, json
sys.path.insert(0, '.')
import numpy as np
from noir_backend_using_gnark_amd import _lib, bn254 as zb, plonk as zp
from oracle import plonk_ref as pl, bn254_ref as ref
sys.path.insert(0, 'tests')
import tests.test_gpu_plonk as T
R = ref.R
g = json.load(open('tests/golden/plonk_golden.json'))
h2i = lambda h: int(h, 16)
for e in g:
    spr, sol = pl.sparse_r1cs_from_acir(e["acir"], [h2i(v) for v in e["values"]])
    rb, _, _ = T._device_srs(e["srs_size"], h2i(e["srs_alpha"]))
    pk = zp.setup(T._circuit(spr), rb)
    bl = T.M([h2i(v) for v in e["blinders"]])
    for idx in range(len(sol)):
        bad = list(sol); bad[idx] = (bad[idx] + 1) % R
        try:
            zp.prove(pk, T.M(bad), bl)
            r = "no error"
        except Exception as ex:
            r = str(ex)[:60]
        print(e["name"], idx, spr.is_satisfied(bad), r)
