"""Pure-Python big-int restatement of the BN254 proving hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *generator of golden vectors* and the slow cross-check for `oracle/bn254_oracle.c`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import anything under
`oracle/`; the product path (`noir_backend_using_gnark_amd/`) never does.

PARITY UNPINNED by the reference's own tests: the reference holds no golden MSM / FFT / proof vector
(SURVEY.md §8c).  The arithmetic lives in the un-vendored Go modules pinned at
  /root/reference/gnark_backend_ffi/go.mod:5   github.com/consensys/gnark-crypto v0.9.1
  /root/reference/gnark_backend_ffi/go.mod:23  github.com/consensys/gnark v0.8.0
and is reached from the reference only through
  /root/reference/gnark_backend_ffi/main.go:121,131,141      (groth16.Setup / Prove / Verify)
  /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:21,67  (plonk.Setup / Prove)
  /root/reference/gnark_backend_ffi/backend/common.go:137    (kzg.NewSRS)
What is restated here is the published algorithm of those modules (mathematical definition +
gnark-crypto's memory/ordering conventions); anchors that ARE pinned by the reference:
  - "-1 mod r" literal 30644e72...f0000000               main.go:233
  - felt / felt-vector wire codec (u32 BE count, 32 B BE) src/gnark_backend_wrapper/serialize.rs:10-47,
                                                          gnark_backend_ffi/internal/backend/helpers.go:13-33
  - Groth16 toy instance X=3, Y=2, Z=6                    main.go:81-82,90-107
The pairing at the bottom exists so that the restated Groth16 prover is checked by an independent
*verifier* equation (a proof that fails e(A,B)=e(α,β)·e(IC,γ)·e(C,δ) is wrong whatever its bytes).
"""
from __future__ import annotations

# ------------------------------------------------------------------------------------------------
# constants (all re-derived below in _selfcheck)
X_BN = 4965661367192848881
Q = 36 * X_BN**4 + 36 * X_BN**3 + 24 * X_BN**2 + 6 * X_BN + 1  # base field Fp
R = 36 * X_BN**4 + 36 * X_BN**3 + 18 * X_BN**2 + 6 * X_BN + 1  # scalar field Fr
MONT_R = 1 << 256
FR_GEN = 5  # gnark-crypto fft.Domain.FrMultiplicativeGen for bn254
FR_TWO_ADICITY = 28
FR_ROOT_2_28 = pow(FR_GEN, (R - 1) >> FR_TWO_ADICITY, R)
G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def inv(a: int, m: int) -> int:
    return pow(a, -1, m)


# ------------------------------------------------------------------------------------------------
# memory image helpers: gnark-crypto fr.Element / fp.Element = [4]uint64 little-endian limbs, Montgomery
def to_mont(x: int, m: int) -> int:
    return (x * MONT_R) % m


def from_mont(x: int, m: int) -> int:
    return (x * inv(MONT_R, m)) % m


def limbs_le(x: int) -> bytes:
    return x.to_bytes(32, "little")


def felt_wire(x: int) -> bytes:
    """32-byte big-endian canonical (serialize.rs:10-17; helpers.go:13-22)."""
    return (x % R).to_bytes(32, "big")


def felts_wire(xs) -> bytes:
    """u32 BE count || count * 32B BE (serialize.rs:33-47; helpers.go:24-33 via fr.Vector.UnmarshalBinary)."""
    return len(xs).to_bytes(4, "big") + b"".join(felt_wire(x) for x in xs)


def felts_unwire(b: bytes):
    n = int.from_bytes(b[:4], "big")
    assert len(b) == 4 + 32 * n
    return [int.from_bytes(b[4 + 32 * i: 36 + 32 * i], "big") for i in range(n)]


# ------------------------------------------------------------------------------------------------
# deterministic PRNG shared by oracle, tests and bench: SplitMix64 -> 4 limbs -> mod r
MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def felt(self, m: int = R) -> int:
        l = [self.next() for _ in range(4)]
        return (l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)) % m


def rand_felts(seed: int, n: int, m: int = R):
    g = SplitMix64(seed)
    return [g.felt(m) for _ in range(n)]


# ------------------------------------------------------------------------------------------------
# Fp2 = Fp[u]/(u^2+1)   (gnark-crypto E2{A0,A1})
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_sqr(a): return f2_mul(a, a)
def f2_scalar(a, k): return ((a[0] * k) % Q, (a[1] * k) % Q)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % Q, Q)
    return ((a[0] * d) % Q, (-a[1] * d) % Q)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
B_G1 = 3
B_G2 = f2_mul((3, 0), f2_inv((9, 1)))  # 3/(9+u)


# ------------------------------------------------------------------------------------------------
# generic affine short-Weierstrass arithmetic; None is the point at infinity
class _Field:
    def __init__(self, add, sub, mul, inv_, neg, zero, is_zero):
        self.add, self.sub, self.mul, self.inv, self.neg, self.zero, self.is_zero = add, sub, mul, inv_, neg, zero, is_zero


FP = _Field(lambda a, b: (a + b) % Q, lambda a, b: (a - b) % Q, lambda a, b: (a * b) % Q,
            lambda a: inv(a, Q), lambda a: (-a) % Q, 0, lambda a: a % Q == 0)
FP2 = _Field(f2_add, f2_sub, f2_mul, f2_inv, f2_neg, F2_ZERO, lambda a: a[0] % Q == 0 and a[1] % Q == 0)


def ec_add(F, P, S):
    if P is None: return S
    if S is None: return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if F.is_zero(F.add(y1, y2)):
            return None
        lam = F.mul(F.mul(F.add(F.add(x1, x1), x1), x1), F.inv(F.add(y1, y1)))  # 3x^2/(2y)
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, P):
    return None if P is None else (P[0], F.neg(P[1]))


def ec_mul(F, P, k: int):
    if k < 0:
        return ec_mul(F, ec_neg(F, P), -k)
    acc = None
    while k:
        if k & 1:
            acc = ec_add(F, acc, P)
        P = ec_add(F, P, P)
        k >>= 1
    return acc


def g1_add(P, S): return ec_add(FP, P, S)
def g1_mul(P, k): return ec_mul(FP, P, k % R)
def g1_neg(P): return ec_neg(FP, P)
def g2_add(P, S): return ec_add(FP2, P, S)
def g2_mul(P, k): return ec_mul(FP2, P, k % R)
def g2_neg(P): return ec_neg(FP2, P)


def g1_on_curve(P):
    return P is None or (P[1] * P[1] - P[0] ** 3 - B_G1) % Q == 0


def g2_on_curve(P):
    if P is None: return True
    x, y = P
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B_G2)) == F2_ZERO


def msm_naive(F, points, scalars):
    """Definition of MultiExp: sum_i s_i * P_i.  This is what (*G1Jac).MultiExp / (*G2Jac).MultiExp of
    gnark-crypto v0.9.1 (go.mod:5) computes; the affine result is canonical (algorithm-independent)."""
    assert len(points) == len(scalars)
    acc = None
    for P, s in zip(points, scalars):
        acc = ec_add(F, acc, ec_mul(F, P, s % R))
    return acc


def msm_pippenger(F, points, scalars, c: int = 4):
    """Bucket method in the shape gnark-crypto uses (signed c-bit digits, 2^(c-1) buckets per window,
    running-sum reduce, Horner combine).  Affine arithmetic; used to pin the C oracle's digit recoding."""
    assert len(points) == len(scalars)
    nwin = (255 + c - 1) // c + 0
    half = 1 << (c - 1)
    digits = []
    for s in scalars:
        s %= R
        ds, carry = [], 0
        for w in range(nwin + 1):
            d = ((s >> (w * c)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > half:  # gnark: "if digit > max { digit -= 1<<c; carry = 1 }"
                d -= 1 << c
                carry = 1
            ds.append(d)
        assert carry == 0
        digits.append(ds)
    total = None
    for w in range(nwin, -1, -1):
        for _ in range(c):
            total = ec_add(F, total, total)
        buckets = [None] * (half + 1)
        for P, ds in zip(points, digits):
            d = ds[w]
            if d > 0: buckets[d] = ec_add(F, buckets[d], P)
            elif d < 0: buckets[-d] = ec_add(F, buckets[-d], ec_neg(F, P))
        run, acc = None, None
        for k in range(half, 0, -1):
            run = ec_add(F, run, buckets[k])
            acc = ec_add(F, acc, run)
        total = ec_add(F, total, acc)
    return total


# ------------------------------------------------------------------------------------------------
# point encodings
def g1_affine_mont_bytes(P) -> bytes:
    """gnark-crypto G1Affine memory image: X,Y each [4]uint64 LE Montgomery; infinity = (0,0)."""
    if P is None: return bytes(64)
    return limbs_le(to_mont(P[0], Q)) + limbs_le(to_mont(P[1], Q))


def g2_affine_mont_bytes(P) -> bytes:
    """G2Affine memory image: X.A0, X.A1, Y.A0, Y.A1."""
    if P is None: return bytes(128)
    (x0, x1), (y0, y1) = P
    return b"".join(limbs_le(to_mont(v, Q)) for v in (x0, x1, y0, y1))


def _lex_largest_fp(y: int) -> bool:
    return y > (Q - 1) // 2


def g1_compress(P) -> bytes:
    """gnark-crypto G1Affine.Bytes(): 32 B BE X; top two bits of byte 0:
    0b10 = compressed, y smallest; 0b11 = compressed, y largest; 0b01 = infinity.  [UPSTREAM-RECALL]"""
    if P is None:
        return bytes([0x40]) + bytes(31)
    b = bytearray(P[0].to_bytes(32, "big"))
    b[0] |= 0xC0 if _lex_largest_fp(P[1]) else 0x80
    return bytes(b)


def g2_compress(P) -> bytes:
    """G2Affine.Bytes(): X.A1 || X.A0 big-endian, flags as G1; "largest" compares A1 first, then A0."""
    if P is None:
        return bytes([0x40]) + bytes(63)
    (x0, x1), (y0, y1) = P
    largest = _lex_largest_fp(y1) if y1 != 0 else _lex_largest_fp(y0)
    b = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    b[0] |= 0xC0 if largest else 0x80
    return bytes(b)


# ------------------------------------------------------------------------------------------------
# fft.Domain restatement (gnark-crypto v0.9.1 ecc/bn254/fr/fft)  [UPSTREAM-RECALL for ordering conventions]
DIT, DIF = 0, 1  # same numeric values as gnark-crypto's `Decimation` iota (DIT first)


def bitrev(i: int, logn: int) -> int:
    return int(format(i, "0%db" % logn)[::-1], 2) if logn else 0


def bit_reverse(a):
    """fft.BitReverse: in-place permutation a[i] <-> a[bitrev(i)]."""
    n = len(a)
    logn = n.bit_length() - 1
    out = list(a)
    for i in range(n):
        out[bitrev(i, logn)] = a[i]
    return out


class Domain:
    def __init__(self, m: int):
        n = 1
        while n < m: n <<= 1
        self.n = n
        self.logn = n.bit_length() - 1
        assert self.logn <= FR_TWO_ADICITY
        self.gen = pow(FR_ROOT_2_28, 1 << (FR_TWO_ADICITY - self.logn), R)
        self.gen_inv = inv(self.gen, R)
        self.card_inv = inv(n, R)
        self.coset = FR_GEN
        self.coset_inv = inv(FR_GEN, R)

    def _ntt_natural(self, x, w):
        """X[k] = sum_j x[j] w^{jk}, natural in / natural out (iterative, O(n log n))."""
        n, logn = self.n, self.logn
        a = [x[bitrev(i, logn)] for i in range(n)]
        length = 2
        while length <= n:
            wl = pow(w, n // length, R)
            for s in range(0, n, length):
                t = 1
                for j in range(length // 2):
                    u, v = a[s + j], a[s + j + length // 2] * t % R
                    a[s + j], a[s + j + length // 2] = (u + v) % R, (u - v) % R
                    t = t * wl % R
            length <<= 1
        return a

    def fft(self, a, decimation, coset=False):
        """(*Domain).FFT(a, decimation, coset...):  DIF: natural in -> bit-reversed out;
        DIT: bit-reversed in -> natural out.  With coset the *logical* input x_j is scaled by g^j
        (DIF scales a[i] by CosetTable[i]; DIT, whose memory order is bit-reversed, by CosetTableReversed[i])."""
        n, logn = self.n, self.logn
        assert len(a) == n
        x = list(a) if decimation == DIF else bit_reverse(a)
        if coset:
            x = [v * pow(self.coset, j, R) % R for j, v in enumerate(x)]
        X = self._ntt_natural(x, self.gen)
        return bit_reverse(X) if decimation == DIF else X

    def fft_inverse(self, a, decimation, coset=False):
        """(*Domain).FFTInverse: same data movement with TwiddlesInv, then scale by CardinalityInv and,
        if coset, the *logical* output y_j by g^-j (DIT -> CosetTableInv, DIF -> CosetTableInvReversed)."""
        n = self.n
        assert len(a) == n
        x = list(a) if decimation == DIF else bit_reverse(a)
        X = self._ntt_natural(x, self.gen_inv)
        X = [v * self.card_inv % R for v in X]
        if coset:
            X = [v * pow(self.coset_inv, j, R) % R for j, v in enumerate(X)]
        return bit_reverse(X) if decimation == DIF else X


def dft_naive(x, w):
    n = len(x)
    return [sum(x[j] * pow(w, j * k, R) for j in range(n)) % R for k in range(n)]


def compute_h(a, b, c, dom: Domain):
    """gnark v0.8.0 internal/backend/bn254/groth16 prove.go computeH  [UPSTREAM-RECALL]:
       pad a,b,c to N; FFTInverse(DIF) x3; FFT(DIT, coset) x3; a = (a*b - c) * 1/(g^N - 1);
       FFTInverse(a, DIF, coset).  Result is left in BIT-REVERSED order (pk.G1.Z is stored bit-reversed
       by Setup to match); the caller uses h[:N-1]."""
    n = dom.n
    a = list(a) + [0] * (n - len(a))
    b = list(b) + [0] * (n - len(b))
    c = list(c) + [0] * (n - len(c))
    a, b, c = (dom.fft_inverse(v, DIF) for v in (a, b, c))
    a, b, c = (dom.fft(v, DIT, True) for v in (a, b, c))
    den = inv((pow(dom.coset, n, R) - 1) % R, R)
    a = [((x * y - z) * den) % R for x, y, z in zip(a, b, c)]
    return dom.fft_inverse(a, DIF, True)


# ------------------------------------------------------------------------------------------------
# Groth16 (prove with pinned randomness; setup from explicit toxic waste for tests)
class R1CS:
    """constraints: list of (L, R, O) each a dict wire->coeff.  Wire order = gnark's: [ONE, public..., secret...]."""
    def __init__(self, n_public: int, n_secret: int, constraints):
        self.n_public = n_public  # including the ONE wire
        self.n_secret = n_secret
        self.constraints = constraints

    @property
    def n_wires(self): return self.n_public + self.n_secret

    def eval_abc(self, w):
        dot = lambda lin: sum(cf * w[i] for i, cf in lin.items()) % R
        a = [dot(L) for L, _, _ in self.constraints]
        b = [dot(Rr) for _, Rr, _ in self.constraints]
        c = [dot(O) for _, _, O in self.constraints]
        return a, b, c


def groth16_setup(r1cs: R1CS, tau, alpha, beta, gamma, delta):
    """Textbook Groth16 CRS in gnark's pk layout (G1.{Alpha,Beta,Delta,A[],B[],K[],Z[]}, G2.{Beta,Delta,B[]}),
    Lagrange basis over the size-N domain.  Z is stored bit-reversed (see compute_h)."""
    dom = Domain(len(r1cs.constraints))
    n, m = dom.n, r1cs.n_wires
    # Lagrange basis at tau: L_j(tau) = (tau^n - 1)/n * w^j/(tau - w^j)
    zt = (pow(tau, n, R) - 1) % R
    lag = [zt * dom.card_inv % R * pow(dom.gen, j, R) % R * inv((tau - pow(dom.gen, j, R)) % R, R) % R for j in range(n)]
    A = [0] * m; B = [0] * m; C = [0] * m
    for j, (L, Rr, O) in enumerate(r1cs.constraints):
        for i, cf in L.items(): A[i] = (A[i] + cf * lag[j]) % R
        for i, cf in Rr.items(): B[i] = (B[i] + cf * lag[j]) % R
        for i, cf in O.items(): C[i] = (C[i] + cf * lag[j]) % R
    gi, di = inv(gamma, R), inv(delta, R)
    kk = [(beta * A[i] + alpha * B[i] + C[i]) % R for i in range(m)]
    G, H = G1_GEN, G2_GEN
    zs = [pow(tau, i, R) * zt % R * di % R for i in range(n)]
    pk = dict(
        domain=dom,
        g1_alpha=g1_mul(G, alpha), g1_beta=g1_mul(G, beta), g1_delta=g1_mul(G, delta),
        g1_a=[g1_mul(G, x) for x in A], g1_b=[g1_mul(G, x) for x in B],
        g1_k=[g1_mul(G, kk[i] * di % R) for i in range(r1cs.n_public, m)],
        g1_z=bit_reverse([g1_mul(G, z) for z in zs]),
        g2_beta=g2_mul(H, beta), g2_delta=g2_mul(H, delta),
        g2_b=[g2_mul(H, x) for x in B],
    )
    vk = dict(
        g1_alpha=pk["g1_alpha"], g2_beta=pk["g2_beta"], g2_gamma=g2_mul(H, gamma), g2_delta=pk["g2_delta"],
        g1_ic=[g1_mul(G, kk[i] * gi % R) for i in range(r1cs.n_public)],
    )
    return pk, vk


def groth16_prove(pk, n_public, a, b, c, w, r, s):
    """gnark v0.8.0 groth16.Prove body (main.go:131 reaches it) with r,s as INPUTS  [UPSTREAM-RECALL]:
         Ar  = MSM(G1.A, w) + alpha + r*delta
         Bs1 = MSM(G1.B, w) + beta  + s*delta
         Bs  = MSM(G2.B, w) + beta2 + s*delta2
         Krs = MSM(G1.K, w[nPub:]) + MSM(G1.Z, h[:N-1]) + s*Ar + r*Bs1 - r*s*delta
       (points at infinity in A/B are kept as (0,0) entries here instead of gnark's InfinityA/B filtering:
        the sums are identical.)"""
    dom = pk["domain"]
    h = compute_h(a, b, c, dom)
    ar = g1_add(g1_add(msm_naive(FP, pk["g1_a"], w), pk["g1_alpha"]), g1_mul(pk["g1_delta"], r))
    bs1 = g1_add(g1_add(msm_naive(FP, pk["g1_b"], w), pk["g1_beta"]), g1_mul(pk["g1_delta"], s))
    bs = g2_add(g2_add(msm_naive(FP2, pk["g2_b"], w), pk["g2_beta"]), g2_mul(pk["g2_delta"], s))
    krs = g1_add(msm_naive(FP, pk["g1_k"], w[n_public:]), msm_naive(FP, pk["g1_z"][:dom.n - 1], h[:dom.n - 1]))
    krs = g1_add(krs, g1_add(g1_mul(ar, s), g1_mul(bs1, r)))
    krs = g1_add(krs, g1_neg(g1_mul(pk["g1_delta"], r * s % R)))
    return ar, bs, krs


def groth16_proof_bytes(ar, bs, krs) -> bytes:
    """Proof.WriteTo (raw=false): Ar | Bs | Krs compressed = 32 + 64 + 32 = 128 bytes."""
    return g1_compress(ar) + g2_compress(bs) + g1_compress(krs)


# ------------------------------------------------------------------------------------------------
# optimal-ate-free, textbook Tate-style pairing check via Fp12 (slow, tests only).
# Fp12 is represented as polynomials over Fp modulo w^12 - 18 w^6 + 82 (the standard BN254 tower flattened).
_FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82


def _p12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for i in range(22, 11, -1):
        top = t[i]
        if top:
            t[i] = 0
            t[i - 6] += 18 * top
            t[i - 12] -= 82 * top
    return [v % Q for v in t[:12]]


def _p12_inv(a):
    # extended Euclid on polynomials over Fp
    def deg(p):
        d = len(p) - 1
        while d and p[d] % Q == 0: d -= 1
        return d
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % Q, 0, 0, 0, 0, 0, 1]
    while deg(low):
        dl, dh = deg(low), deg(high)
        r = [0] * 13
        # poly division high / low
        temp = list(high)
        o = [0] * 13
        for i in range(dh - dl, -1, -1):
            o[i] = temp[dl + i] * inv(low[dl], Q) % Q
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - o[i] * low[c]) % Q
        r = o
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    c = inv(low[0], Q)
    return [v * c % Q for v in lm[:12]]


def _p12_pow(a, e):
    out = [1] + [0] * 11
    while e:
        if e & 1: out = _p12_mul(out, a)
        a = _p12_mul(a, a)
        e >>= 1
    return out


def _p12_scalar(k): return [k % Q] + [0] * 11


class _F12:
    add = staticmethod(lambda a, b: [(x + y) % Q for x, y in zip(a, b)])
    sub = staticmethod(lambda a, b: [(x - y) % Q for x, y in zip(a, b)])
    mul = staticmethod(_p12_mul)
    inv = staticmethod(_p12_inv)
    neg = staticmethod(lambda a: [(-x) % Q for x in a])
    zero = [0] * 12
    is_zero = staticmethod(lambda a: all(x % Q == 0 for x in a))


def _twist(P):
    """G2 (over Fp2, u^2=-1) -> curve over Fp12: u = w^6 - 9; (x,y) -> (x*w^2, y*w^3)."""
    (x0, x1), (y0, y1) = P
    xc = [(x0 - 9 * x1) % Q, x1]
    yc = [(y0 - 9 * y1) % Q, y1]
    nx = [0] * 12; ny = [0] * 12
    nx[2], nx[8] = xc[0], xc[1]
    ny[3], ny[9] = yc[0], yc[1]
    return (nx, ny)


def _linefunc(P1, P2, T):
    F = _F12
    x1, y1 = P1; x2, y2 = P2; xt, yt = T
    if x1 != x2:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    elif y1 == y2:
        m = F.mul(F.mul(_p12_scalar(3), F.mul(x1, x1)), F.inv(F.mul(_p12_scalar(2), y1)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    else:
        return F.sub(xt, x1)


_ATE_LOOP = 29793968203157093288  # 6x+2
_LOG_ATE = 63


def _ec12_add(P, S):
    return ec_add(_F12, P, S)


def miller_loop(Q2, P1):
    if Q2 is None or P1 is None:
        return [1] + [0] * 11
    Qt = _twist(Q2)
    P = (_p12_scalar(P1[0]), _p12_scalar(P1[1]))
    Rr = Qt
    f = [1] + [0] * 11
    for i in range(_LOG_ATE, -1, -1):
        f = _p12_mul(_p12_mul(f, f), _linefunc(Rr, Rr, P))
        Rr = _ec12_add(Rr, Rr)
        if _ATE_LOOP & (1 << i):
            f = _p12_mul(f, _linefunc(Rr, Qt, P))
            Rr = _ec12_add(Rr, Qt)
    Q1 = (_p12_pow(Qt[0], Q), _p12_pow(Qt[1], Q))
    nQ2 = (_p12_pow(Q1[0], Q), _F12.neg(_p12_pow(Q1[1], Q)))
    f = _p12_mul(f, _linefunc(Rr, Q1, P))
    Rr = _ec12_add(Rr, Q1)
    f = _p12_mul(f, _linefunc(Rr, nQ2, P))
    return f


def final_exp(f):
    return _p12_pow(f, (Q ** 12 - 1) // R)


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 for pairs (P in G1, Q in G2)."""
    f = [1] + [0] * 11
    for P1, Q2 in pairs:
        f = _p12_mul(f, miller_loop(Q2, P1))
    return final_exp(f) == [1] + [0] * 11


def groth16_verify(vk, proof, public_w) -> bool:
    """e(A,B) = e(alpha,beta) e(IC,gamma) e(C,delta)  (groth16.Verify, main.go:141)."""
    ar, bs, krs = proof
    ic = msm_naive(FP, vk["g1_ic"], public_w)
    return pairing_product_is_one([
        (g1_neg(ar), bs), (vk["g1_alpha"], vk["g2_beta"]), (ic, vk["g2_gamma"]), (krs, vk["g2_delta"])])


# ------------------------------------------------------------------------------------------------
def _selfcheck():
    assert Q == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    assert R == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    assert pow(FR_ROOT_2_28, 1 << 27, R) == R - 1
    assert FR_ROOT_2_28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert g1_on_curve(G1_GEN) and g2_on_curve(G2_GEN)
    assert g1_mul(G1_GEN, R - 1) == (1, Q - 2) and ec_mul(FP, G1_GEN, R) is None
    assert ec_mul(FP2, G2_GEN, R) is None


if __name__ == "__main__":
    _selfcheck()
    print("bn254_ref selfcheck ok")
