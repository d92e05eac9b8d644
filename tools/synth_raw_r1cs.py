"""Synthetic RawR1CS payloads in the JSON shape the reference's Rust side serialises for its intended Groth16 FFI
(src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60: struct field order gates, public_inputs, values, num_variables, num_constraints; a Witness is a
bare u32; felts are 64 hex characters, the values vector is hex(u32 BE count | count x 32 B BE)) with a satisfying witness -- the input of bench.py's
`export_path_groth16` block and of the Groth16 export tests.

Circuit: witnesses 1..n_public are public; gate i (0-based) constrains witness c = i + 3 from a = i + 1 and b = i + 2, alternating between a gate with one
linear term and one with three, every gate with ONE mul term:
    even i:  qM * a * b            - c + k = 0
    odd i:   qM * a * b + qL * a + qR * b - c + k = 0
buildR1CS (gnark_backend_ffi/backend/groth16/r1cs.go:9-72) turns every mul term into a product constraint and every gate into a sum constraint:
n_gates gates -> 2 * n_gates constraints, n_gates + 2 witnesses, 1 + (n_gates + 2) + n_gates wires.
`bits` (0..256): that share (in 1/256) of the gates has its constant chosen so that c comes out 0 or 1 -- the 0/1-heavy wire vectors real circuits have
(booleans, range-check bits); 0 = every witness a full-width field element."""
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


def hx(v):
    return "%064x" % (v % R)


def synth(n_gates: int, n_public: int = 8, seed: int = 1, bits: int = 0, first=None):
    """-> (raw_r1cs_json: str, values: list[int]); values[w - 1] is witness w.  first = (v1, v2): other values for the first two witnesses -- with bits = 0
    the same gates (their constants do not depend on the values) and another satisfying assignment."""
    g = _splitmix(seed)
    one, minus_one = hx(1), hx(-1)
    w = [next(g) % R, next(g) % R]
    if first:
        w = [first[0] % R, first[1] % R]
    parts = []
    for i in range(n_gates):
        a, b, c = i + 1, i + 2, i + 3
        va, vb = w[a - 1], w[b - 1]
        r = next(g)
        qm = 1 if r & 3 else ((next(g) << 190) | next(g)) % R
        k = (r >> 8) & 0xffff if r & 4 else ((next(g) << 128) | next(g)) % R
        if i & 1:
            ql, qr = (r >> 24) & 0xff, R - 1 if r & 8 else (r >> 32) & 0xffffffff
            rest = (qm * va * vb + ql * va + qr * vb) % R
            adds = '{"coefficient":"%s","sum":%d},{"coefficient":"%s","sum":%d},{"coefficient":"%s","sum":%d}' % (hx(ql), a, hx(qr), b, minus_one, c)
        else:
            rest = (qm * va * vb) % R
            adds = '{"coefficient":"%s","sum":%d}' % (minus_one, c)
        if ((r >> 48) & 255) < bits:
            k = ((r >> 56) & 1) - rest  # the gate's constant makes c a bit
        w.append((rest + k) % R)
        parts.append('{"mul_terms":[{"coefficient":"%s","multiplicand":%d,"multiplier":%d}],"add_terms":[%s],"constant_term":"%s"}'
                     % (one if qm == 1 else hx(qm), a, b, adds, hx(k)))
    raw = ('{"gates":[' + ",".join(parts) + '],"public_inputs":' + json.dumps(list(range(1, n_public + 1)), separators=(",", ":")) +
           ',"values":"' + felts_wire_hex(w) + '","num_variables":%d,"num_constraints":%d}' % (len(w) + 1, 2 * n_gates))
    return raw, w


def felts_wire_hex(values) -> str:
    """hex( u32 BE count | count x 32 B BE ): the felt-vector format of src/gnark_backend_wrapper/serialize.rs:33-47"""
    return "%08x" % len(values) + "".join("%064x" % v for v in values)


def with_values(raw: str, values) -> str:
    """the same circuit text with another values vector"""
    a = raw.index('"values":"') + len('"values":"')
    b = raw.index('"', a)
    return raw[:a] + felts_wire_hex(values) + raw[b:]
