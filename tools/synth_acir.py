"""Synthetic ACIR circuits in the JSON shape the reference's Go side unmarshals (gnark_backend_ffi/acir/acir.go:17-75; fixtures at main.go:233-246),
with a satisfying witness -- the input of bench.py's `export_path` block and of the export-path tests.

Circuit: witnesses 1..n_public are public; opcode i (0-based) constrains witness c = i + 3 from a = i + 1 and b = i + 2, alternating between the two
gate shapes BuildSparseR1CS distinguishes (backend/plonk/sparse_r1cs.go:44-107):
    even i:  qM * a * b            - c + k = 0      (one mul term, ONE linear term  -> qO)
    odd i:   qM * a * b + qL * a + qR * b - c + k = 0      (one mul term, THREE linear terms -> qL qR qO)
Every 1024th opcode is followed by a Directive, which the Go side skips (sparse_r1cs.go:33-37).  Coefficients are small, -1, or full-width values from a
SplitMix64 stream, so the hex literals exercise all three decoding paths.  n_opcodes + n_public <= 2^19 keeps the PLONK domain under the reference's
1,000,000-point SRS (backend/common.go:137)."""
import json

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MASK = (1 << 64) - 1


def _splitmix(seed):
    s = seed & MASK
    while True:
        s = (s + 0x9e3779b97f4a7c15) & MASK
        z = s
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & MASK
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & MASK
        yield z ^ (z >> 31)


def synth(n_opcodes: int, n_public: int = 8, seed: int = 1, compact: bool = True):
    """-> (acir_json: str, values: list[int]) ; values[w - 1] is witness w, n_opcodes + 2 witnesses in all."""
    g = _splitmix(seed)
    hx = lambda v: "%064x" % (v % R)
    one, minus_one = hx(1), hx(-1)
    w = [next(g) % R, next(g) % R]  # witnesses 1 and 2
    ops = []
    sep = (",", ":") if compact else (", ", ": ")
    for i in range(n_opcodes):
        a, b, c = i + 1, i + 2, i + 3
        va, vb = w[a - 1], w[b - 1]
        r = next(g)
        qm = 1 if r & 3 else ((next(g) << 190) | next(g)) % R
        k = (r >> 8) & 0xffff if r & 4 else ((next(g) << 128) | next(g)) % R
        if i & 1:
            ql, qr = (r >> 24) & 0xff, R - 1 if r & 8 else (r >> 32) & 0xffffffff
            vc = (qm * va * vb + ql * va + qr * vb + k) % R
            lin = [[hx(ql), a], [hx(qr), b], [minus_one, c]]
        else:
            vc = (qm * va * vb + k) % R
            lin = [[minus_one, c]]
        w.append(vc)
        # serde's field order (alphabetical), as in the reference's fixtures
        ops.append({"Arithmetic": {"linear_combinations": lin, "mul_terms": [[one if qm == 1 else hx(qm), a, b]], "q_c": hx(k)}})
        if i % 1024 == 1023:
            ops.append({"Directive": {"Invert": {"result": c, "x": a}}})
    acir = {"current_witness_index": n_opcodes + 2, "opcodes": ops, "public_inputs": list(range(1, n_public + 1))}
    return json.dumps(acir, separators=sep), w


def felts_wire_hex(values) -> str:
    """hex( u32 BE count | count x 32 B BE ) -- src/gnark_backend_wrapper/serialize.rs:33-47"""
    return "%08x" % len(values) + "".join("%064x" % v for v in values)
