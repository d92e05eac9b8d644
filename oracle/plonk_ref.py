"""CPU restatement of gnark v0.8.0's PLONK backend for BN254 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

What is restated (all [UPSTREAM-RECALL]: gnark / gnark-crypto are un-vendored Go modules pinned at
/root/reference/gnark_backend_ffi/go.mod:23 and :5; no Go toolchain exists here, so nothing below could be checked against
the upstream binary -- PARITY UNPINNED, exactly like the Groth16 oracle):
    plonk.Setup   internal/backend/bn254/plonk/setup.go    reached at /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:21
    plonk.Prove   internal/backend/bn254/plonk/prove.go    reached at backend/plonk/plonk.go:67 (the reference's ONLY live prove path:
                                                           PlonkProveWithPK, gnark_backend_ffi/main.go:24-37)
    plonk.Verify  internal/backend/bn254/plonk/verify.go   reached at backend/plonk/plonk.go:47
    kzg.Commit / Open / BatchOpenSinglePoint / FoldProof / BatchVerifyMultiPoints   gnark-crypto ecc/bn254/fr/kzg/kzg.go
    kzg.NewSRS                                                                       reached at backend/common.go:137
    fiatshamir.Transcript (SHA-256; challenges "gamma", "beta", "alpha", "zeta")     gnark-crypto fiat-shamir/transcript.go
    iop.BuildRatioCopyConstraint, iop.DivideByXMinusOne, (*Polynomial).Blind         gnark-crypto ecc/bn254/fr/iop
The constraint system is the reference's own: one gate qL*xa + qR*xb + qO*xc + qM*xa*xb + qC = 0 per ACIR arithmetic opcode
(/root/reference/gnark_backend_ffi/backend/plonk/sparse_r1cs.go:44-107).

The prover's randomness (the 9 blinding scalars; upstream: fr.SetRandom) is an INPUT, which makes the proof bytes a function of
the inputs; the Fiat-Shamir challenges are derived as upstream does unless they are pinned by the caller.

What pins this file, in the absence of upstream: the verifier below is written from the verification EQUATIONS (pairing check of the
two KZG openings + the quotient identity at zeta), so prove -> verify closing on satisfiable circuits and failing on tampered ones
checks the prover's algebra end to end; the byte layout (Proof.WriteTo) is recall only.

Python integers throughout; NTTs and MSMs may be delegated to the C oracle (oracle/bn254_oracle.c, another independent restatement)
through `fast=True` so that 2^12-2^14-gate instances finish in seconds."""
from __future__ import annotations

import hashlib

from . import bn254_ref as ref
from .bn254_ref import DIF, DIT, FP, R, Domain, bit_reverse, g1_add, g1_mul, g1_neg, inv

# ------------------------------------------------------------------------------------------------ constraint system


class SparseR1CS:
    """cs.SparseR1CS as the reference builds it (sparse_r1cs.go:18-107): variables = [public..., secret...] (no internal variables, no
    ONE wire); one constraint per gate: (qL, qR, qO, qM, qC, xa, xb, xc) with qL*xa + qR*xb + qO*xc + qM*xa*xb + qC == 0."""

    def __init__(self, n_public: int, n_secret: int, constraints):
        self.n_public, self.n_secret = n_public, n_secret
        self.constraints = [tuple(c) for c in constraints]

    @property
    def n_vars(self):
        return self.n_public + self.n_secret

    def is_satisfied(self, solution) -> bool:
        return all((ql * solution[xa] + qr * solution[xb] + qo * solution[xc] + qm * solution[xa] * solution[xb] + qc) % R == 0
                   for ql, qr, qo, qm, qc, xa, xb, xc in self.constraints)


def handle_values(public_inputs, n_values: int, layout: str = "reference"):
    """HandleValues (backend/common.go:45-76) on witnesses 1 .. n_values: -> (order, index_map, n_public) where variable k holds witness order[k]
    (public variables first: cs.AddPublicVariable numbers them 0 .., cs.AddSecretVariable continues after them [UPSTREAM-RECALL gnark v0.8.0
    constraint/system.go]) and index_map[w] is the variable the gates use for witness w.

    layout="reference" follows the two loops literally: loop 1 (common.go:48-58) adds a public variable for every (witness, equal public input)
    pair, in witness order; loop 2 (common.go:59-74) adds, when there ARE public inputs, a SECRET variable for every (witness, DIFFERENT public
    input) pair -- |P| copies of a private witness, |P| - 1 copies of a public one -- else one per witness; indexMap[i] is overwritten each time,
    so the last variable added for a witness is the one the gates name.  For |P| <= 1 this is one variable per witness.
    layout="one_var" is that one-variable-per-witness layout for any |P| (what HandleValues presumably meant)."""
    pub = list(public_inputs)
    order, index = [], {}
    if layout == "reference":
        for i in range(1, n_values + 1):
            for p in pub:
                if i == p:
                    index[i] = len(order)
                    order.append(i)
        n_public = len(order)
        for i in range(1, n_values + 1):
            if pub:
                for p in pub:
                    if i != p:
                        index[i] = len(order)
                        order.append(i)
            else:
                index[i] = len(order)
                order.append(i)
        return order, index, n_public
    assert layout == "one_var", layout
    order = [i for i in range(1, n_values + 1) if i in pub] + [i for i in range(1, n_values + 1) if i not in pub]
    return order, {w: k for k, w in enumerate(order)}, len([i for i in order if i in pub])


def sparse_r1cs_from_acir(acir: dict, values, layout: str = "reference"):
    """The reference's lowering of an ACIR circuit (backend/plonk/sparse_r1cs.go:18-107 + backend/common.go:45-76): arithmetic opcodes
    become gates (only MulTerms[0]; SimpleTerms of length 1 -> qO, 2 -> qL, qR, 3 -> qL, qR, qO, any other length -> none); directives and
    black-box calls emit nothing.  `values` = the witness values handed over (witness i = values[i - 1]).  Returns (spr, solution) with the
    public variables first; the variable layout is handle_values(..., layout).  In the reference layout a term naming a witness that has
    no variable gets variable 0 -- Go's map lookup yields the zero value (sparse_r1cs.go:53-54) -- in "one_var" that is a KeyError."""
    order, index, n_public = handle_values(acir.get("public_inputs", []), len(values), layout)
    var = (lambda w: index.get(w, 0)) if layout == "reference" else (lambda w: index[w])
    solution = [values[w - 1] % R for w in order]
    h2i = lambda h: int(h, 16) % R
    gates = []
    for op in acir["opcodes"]:
        if "Arithmetic" not in op:
            continue
        a = op["Arithmetic"]
        ql = qr = qo = qm = 0
        xa = xb = xc = 0
        if a["mul_terms"]:
            c, wl, wr = a["mul_terms"][0]
            qm, xa, xb = h2i(c), var(wl), var(wr)
        lin = a["linear_combinations"]
        if len(lin) == 1:
            qo, xc = h2i(lin[0][0]), var(lin[0][1])
        elif len(lin) in (2, 3):
            ql, xa = h2i(lin[0][0]), var(lin[0][1])
            qr, xb = h2i(lin[1][0]), var(lin[1][1])
            if len(lin) == 3:
                qo, xc = h2i(lin[2][0]), var(lin[2][1])
        gates.append((ql, qr, qo, qm, h2i(a["q_c"]), xa, xb, xc))
    return SparseR1CS(n_public, len(order) - n_public, gates), solution


# ------------------------------------------------------------------------------------------------ plumbing (python <-> C oracle)
def _np():
    import numpy as np
    return np


def ints_to_mont_np(xs, m=R):
    np = _np()
    return np.frombuffer(b"".join(((x % m) * ref.MONT_R % m).to_bytes(32, "little") for x in xs), dtype=np.uint64).reshape(-1, 4).copy()


def mont_np_to_ints(a, m=R):
    np = _np()
    raw = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    rinv = inv(ref.MONT_R % m, m)
    return [int.from_bytes(raw[i:i + 32], "little") * rinv % m for i in range(0, len(raw), 32)]


def g1_to_np(P):
    np = _np()
    return np.frombuffer(ref.g1_affine_mont_bytes(P), dtype=np.uint64).copy()


def g1_from_np(a):
    x, y = mont_np_to_ints(_np().asarray(a, dtype=_np().uint64).reshape(2, 4), ref.Q)
    return None if (x == 0 and y == 0) else (x, y)


class _Backend:
    """NTT / MSM providers: pure Python (definition-level) or the C oracle (fast=True)."""

    def __init__(self, fast: bool):
        self.fast = fast
        if fast:
            from . import oracle as orc
            self.orc = orc

    def ntt(self, dom: Domain, v, inverse: bool, decimation: int, coset: bool = False):
        if not self.fast or dom.n < 64:
            return (dom.fft_inverse if inverse else dom.fft)(v, decimation, coset)
        return mont_np_to_ints(self.orc.fr_ntt(ints_to_mont_np(v), inverse, decimation, coset))

    def msm_g1(self, srs_g1, scalars):
        """kzg.Commit: MultiExp(srs.G1[:len(p)], p)."""
        n = len(scalars)
        if not self.fast or n < 64:
            pts = srs_g1[:n] if isinstance(srs_g1, list) else [g1_from_np(srs_g1[i]) for i in range(n)]
            return ref.msm_pippenger(FP, pts, scalars, 4) if n > 8 else ref.msm_naive(FP, pts, scalars)
        np = _np()
        pts = srs_g1[:n] if not isinstance(srs_g1, list) else np.stack([g1_to_np(p) for p in srs_g1[:n]])
        return g1_from_np(self.orc.g1_msm(np.ascontiguousarray(pts), ints_to_mont_np(scalars)))


# ------------------------------------------------------------------------------------------------ KZG
def kzg_new_srs(size: int, alpha: int, fast: bool = False):
    """kzg.NewSRS(size, alpha): G1[i] = alpha^i * G1, G2 = [G2, alpha * G2]  (backend/common.go:137 calls it with size 1_000_000)."""
    if fast:
        from . import oracle as orc
        np = _np()
        g = g1_to_np(ref.G1_GEN)
        pw, out = 1, []
        for _ in range(size):
            out.append(orc.g1_mul(g, ints_to_mont_np([pw])[0]))
            pw = pw * alpha % R
        g1 = np.stack(out)
    else:
        g1, pw = [], 1
        for _ in range(size):
            g1.append(g1_mul(ref.G1_GEN, pw))
            pw = pw * alpha % R
    return dict(g1=g1, g2=[ref.G2_GEN, ref.g2_mul(ref.G2_GEN, alpha)])


def poly_eval(p, x):
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % R
    return acc


def divide_by_x_minus_a(f, fa, a):
    """kzg.dividePolyByXminusA: (f - f(a)) / (X - a) by synthetic division; len(result) = len(f) - 1."""
    f = list(f)
    f[0] = (f[0] - fa) % R
    for i in range(len(f) - 2, -1, -1):
        f[i] = (f[i] + f[i + 1] * a) % R
    return f[1:]


# ------------------------------------------------------------------------------------------------ Fiat-Shamir
def fr_bytes(x: int) -> bytes:
    """fr.Element.Marshal(): 32 bytes big-endian, canonical."""
    return (x % R).to_bytes(32, "big")


def g1_raw_bytes(P) -> bytes:
    """G1Affine.RawBytes() / Marshal(): uncompressed X || Y big-endian (64 B); infinity = 0x40 flag + zeros."""
    if P is None:
        return bytes([0x40]) + bytes(63)
    return P[0].to_bytes(32, "big") + P[1].to_bytes(32, "big")


class Transcript:
    """fiatshamir.NewTranscript(sha256.New(), ids...): challenge_i = H(id_i || challenge_{i-1} || bindings_i)."""

    def __init__(self, *ids):
        self.ids = list(ids)
        self.bindings = {i: [] for i in ids}
        self.values = {}

    def bind(self, cid, b: bytes):
        assert cid not in self.values
        self.bindings[cid].append(bytes(b))

    def compute(self, cid) -> bytes:
        if cid in self.values:
            return self.values[cid]
        h = hashlib.sha256()
        h.update(cid.encode())
        pos = self.ids.index(cid)
        if pos:
            h.update(self.values[self.ids[pos - 1]])  # errPreviousChallengeNotComputed otherwise
        for b in self.bindings[cid]:
            h.update(b)
        self.values[cid] = h.digest()
        return self.values[cid]

    def challenge_fr(self, cid, *points) -> int:
        """plonk deriveRandomness: bind the points' RawBytes, compute, fr.SetBytes (big-endian, reduced mod r)."""
        for P in points:
            self.bind(cid, g1_raw_bytes(P))
        return int.from_bytes(self.compute(cid), "big") % R


def kzg_derive_gamma(point, digests, claimed_values) -> int:
    """kzg.deriveGamma (v0.9.1): one-challenge transcript bound to the point, the digests and the claimed values."""
    t = Transcript("gamma")
    t.bind("gamma", fr_bytes(point))
    for d in digests:
        t.bind("gamma", g1_raw_bytes(d))
    for v in claimed_values:
        t.bind("gamma", fr_bytes(v))
    return int.from_bytes(t.compute("gamma"), "big") % R


# ------------------------------------------------------------------------------------------------ Setup
def _canonical(be, dom, lagrange):
    """Lagrange (regular) -> canonical (regular): FFTInverse(DIF) then BitReverse, as setup.go does."""
    return bit_reverse(be.ntt(dom, lagrange, True, DIF))


def build_permutation(spr: SparseR1CS, size: int):
    """setup.go buildPermutation: position -> permuted position over the 3*size wire slots (L | R | O), placeholders included."""
    npub, nc = spr.n_public, len(spr.constraints)
    lro = [0] * (3 * size)
    for i in range(npub):
        lro[i] = i
    for i, (_, _, _, _, _, xa, xb, xc) in enumerate(spr.constraints):
        lro[npub + i] = xa
        lro[size + npub + i] = xb
        lro[2 * size + npub + i] = xc
    perm = [-1] * (3 * size)
    cycle = [-1] * max(1, spr.n_vars)
    for i in range(3 * size):
        if cycle[lro[i]] != -1:
            perm[i] = cycle[lro[i]]
        cycle[lro[i]] = i
    for i in range(3 * size):
        if perm[i] == -1:
            perm[i] = cycle[lro[i]]
    return perm


def plonk_setup(spr: SparseR1CS, srs, fast: bool = False):
    """plonk.Setup(spr, srs) -> (pk, vk).  pk keeps what gnark's ProvingKey keeps (canonical Ql..Qo, CQk, LQk, S1..S3, Permutation,
    the two domains) -- the Lagrange-coset copies gnark caches are derived by whoever needs them."""
    be = _Backend(fast)
    npub, nc = spr.n_public, len(spr.constraints)
    size_system = nc + npub
    d0 = Domain(size_system)
    d1 = Domain((8 if size_system < 6 else 4) * size_system)
    n = d0.n
    ql, qr, qm, qo, qk = ([0] * n for _ in range(5))
    for i in range(npub):
        ql[i] = R - 1  # placeholder gates  -PUB_i + qk_i = 0 ; qk_i is completed by the prover
    for i, (cl, cr, co, cm, cc, _, _, _) in enumerate(spr.constraints):
        ql[npub + i], qr[npub + i], qm[npub + i], qo[npub + i], qk[npub + i] = cl % R, cr % R, cm % R, co % R, cc % R
    lqk = list(qk)
    perm = build_permutation(spr, n)
    u = d0.coset
    ident = [pow(d0.gen, i, R) for i in range(n)]
    ident = ident + [u * x % R for x in ident] + [u * u % R * x % R for x in ident]  # getIDSmallDomain
    s_lag = [[ident[perm[j * n + i]] for i in range(n)] for j in range(3)]
    pk = dict(spr=spr, d0=d0, d1=d1, n=n, perm=perm,
              ql=_canonical(be, d0, ql), qr=_canonical(be, d0, qr), qm=_canonical(be, d0, qm), qo=_canonical(be, d0, qo),
              cqk=_canonical(be, d0, qk), lqk=lqk,
              s1=_canonical(be, d0, s_lag[0]), s2=_canonical(be, d0, s_lag[1]), s3=_canonical(be, d0, s_lag[2]), srs=srs)
    commit = lambda p: be.msm_g1(srs["g1"], p)
    vk = dict(size=n, size_inv=d0.card_inv, generator=d0.gen, n_public=npub, coset_shift=u, srs_g2=srs["g2"],
              s=[commit(pk["s1"]), commit(pk["s2"]), commit(pk["s3"])],
              ql=commit(pk["ql"]), qr=commit(pk["qr"]), qm=commit(pk["qm"]), qo=commit(pk["qo"]), qk=commit(pk["cqk"]))
    pk["vk"] = vk
    return pk, vk


# ------------------------------------------------------------------------------------------------ Prove
def evaluate_lro(spr: SparseR1CS, n: int, solution):
    """prove.go evaluateLROSmallDomain: l, r, o in Lagrange form on the small domain (placeholders first, padding = solution[0])."""
    s0 = solution[0] if solution else 0
    l, r, o = [s0] * n, [s0] * n, [s0] * n
    for i in range(spr.n_public):
        l[i] = solution[i]
    for i, (_, _, _, _, _, xa, xb, xc) in enumerate(spr.constraints):
        k = spr.n_public + i
        l[k], r[k], o[k] = solution[xa], solution[xb], solution[xc]
    return l, r, o


def _blind(canon, n, rnd):
    """(*Polynomial).Blind(len(rnd) - 1): p += Q(X) * (X^n - 1) with Q's coefficients = rnd."""
    p = list(canon) + [0] * len(rnd)
    for i, b in enumerate(rnd):
        p[i] = (p[i] - b) % R
        p[n + i] = (p[n + i] + b) % R
    return p


def _bind_public_data(fs: Transcript, vk, public_inputs):
    """prove.go bindPublicData: S1..S3, Ql, Qr, Qm, Qo, Qk digests (Marshal = uncompressed), then the public inputs."""
    for d in (vk["s"][0], vk["s"][1], vk["s"][2], vk["ql"], vk["qr"], vk["qm"], vk["qo"], vk["qk"]):
        fs.bind("gamma", g1_raw_bytes(d))
    for w in public_inputs:
        fs.bind("gamma", fr_bytes(w))


def plonk_prove(pk, solution, blinders, challenges=None, fast: bool = False, trace=None):
    """plonk.Prove(spr, pk, witness) after the solver: `solution` = values of all variables (public first).
    blinders: 9 scalars -- (l: 2, r: 2, o: 2, z: 3), the fr.SetRandom draws of Blind(1) x3 and Blind(2) in call order.
    challenges: None (SHA-256 Fiat-Shamir, as upstream) or dict(gamma, beta, alpha, zeta, kzg_gamma) to pin them.
    Returns the proof dict (see plonk_proof_bytes)."""
    be = _Backend(fast)
    spr, d0, d1, n, vk, srs = pk["spr"], pk["d0"], pk["d1"], pk["n"], pk["vk"], pk["srs"]
    N4, rho = d1.n, d1.n // d0.n
    npub = spr.n_public
    commit = lambda p: be.msm_g1(srs["g1"], p)
    fs = Transcript("gamma", "beta", "alpha", "zeta")
    pin = challenges or {}

    # l, r, o: Lagrange -> canonical, blinded with degree-1 masks, committed
    l_lag, r_lag, o_lag = evaluate_lro(spr, n, solution)
    cl, cr, co = (_canonical(be, d0, v) for v in (l_lag, r_lag, o_lag))
    bl, br, bo = _blind(cl, n, blinders[0:2]), _blind(cr, n, blinders[2:4]), _blind(co, n, blinders[4:6])
    lro = [commit(bl), commit(br), commit(bo)]
    _bind_public_data(fs, vk, solution[:npub])
    gamma = fs.challenge_fr("gamma", *lro)
    beta = fs.challenge_fr("beta")
    gamma, beta = pin.get("gamma", gamma), pin.get("beta", beta)

    # Z: the copy-constraint ratio (iop.BuildRatioCopyConstraint) on the UNblinded l, r, o; canonical, blinded with a degree-2 mask
    u = d0.coset
    ident = [pow(d0.gen, i, R) for i in range(n)]
    ident = ident + [u * x % R for x in ident] + [u * u % R * x % R for x in ident]
    perm = pk["perm"]
    num, den = [1] * n, [1] * n
    for i in range(n - 1):
        a = b = 1
        for j, w in enumerate((l_lag, r_lag, o_lag)):
            a = a * ((w[i] + beta * ident[i + j * n] + gamma) % R) % R
            b = b * ((w[i] + beta * ident[perm[i + j * n]] + gamma) % R) % R
        num[i + 1], den[i + 1] = num[i] * a % R, den[i] * b % R
    z_lag = [num[i] * inv(den[i], R) % R for i in range(n)]
    bz = _blind(_canonical(be, d0, z_lag), n, blinders[6:9])
    z_digest = commit(bz)
    alpha = pin.get("alpha", fs.challenge_fr("alpha", z_digest))

    # qk completed with the public inputs (Lagrange), canonical
    qk_lag = list(solution[:npub]) + list(pk["lqk"][npub:])
    qk_full = _canonical(be, d0, qk_lag)

    # quotient on the coset of the big domain: every polynomial evaluated at g * W^i (natural index order here; gnark keeps the
    # bit-reversed layout -- the polynomial h is the same)
    def coset_eval(p):
        return bit_reverse(be.ntt(d1, list(p) + [0] * (N4 - len(p)), False, DIF, True))

    e_l, e_r, e_o, e_z, e_qk = (coset_eval(p) for p in (bl, br, bo, bz, qk_full))
    e_ql, e_qr, e_qm, e_qo = (coset_eval(pk[k]) for k in ("ql", "qr", "qm", "qo"))
    e_s1, e_s2, e_s3 = (coset_eval(pk[k]) for k in ("s1", "s2", "s3"))
    l1_canon = [d0.card_inv] * n  # L_1 = (X^n - 1) / (n (X - 1)) = (1/n) sum X^i
    e_l1 = coset_eval(l1_canon)
    W = d1.gen
    xs = [d1.coset * pow(W, i, R) % R for i in range(N4)]
    xn_minus_one_inv = [inv((pow(xs[i], n, R) - 1) % R, R) for i in range(rho)]  # x^n takes rho values on the coset
    uu = u * u % R
    t = [0] * N4
    for i in range(N4):
        lv, rv, ov, zv, zs = e_l[i], e_r[i], e_o[i], e_z[i], e_z[(i + rho) % N4]  # z(omega * x): rho steps of W
        ic = (e_ql[i] * lv + e_qr[i] * rv + e_qm[i] * lv % R * rv + e_qo[i] * ov + e_qk[i]) % R
        a = (lv + beta * xs[i] + gamma) * (rv + beta * u % R * xs[i] + gamma) % R * (ov + beta * uu % R * xs[i] + gamma) % R * zv % R
        b = (lv + beta * e_s1[i] + gamma) * (rv + beta * e_s2[i] + gamma) % R * (ov + beta * e_s3[i] + gamma) % R * zs % R
        one = (zv - 1) * e_l1[i] % R
        t[i] = ((one * alpha + (b - a)) % R * alpha + ic) % R * xn_minus_one_inv[i % rho] % R
    h = be.ntt(d1, bit_reverse(t), True, DIT, True)  # coset interpolation: bit-reversed in -> natural canonical out
    assert all(c == 0 for c in h[3 * (n + 2):]), "the constraint system is not satisfied (the quotient is not a polynomial)"
    h1, h2, h3 = h[:n + 2], h[n + 2:2 * (n + 2)], h[2 * (n + 2):3 * (n + 2)]
    hd = [commit(h1), commit(h2), commit(h3)]
    zeta = pin.get("zeta", fs.challenge_fr("zeta", *hd))

    # openings
    lz, rz, oz = poly_eval(bl, zeta), poly_eval(br, zeta), poly_eval(bo, zeta)
    zeta_sh = zeta * d0.gen % R
    zu = poly_eval(bz, zeta_sh)
    z_open_h = commit(divide_by_x_minus_a(bz, zu, zeta_sh))

    # linearised polynomial (prove.go computeLinearizedPolynomial)
    s1z, s2z = poly_eval(pk["s1"], zeta), poly_eval(pk["s2"], zeta)
    c_s3 = (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) % R * zu % R * beta % R
    c_z = (-(lz + beta * zeta + gamma) * (rz + beta * u % R * zeta + gamma) % R * (oz + beta * uu % R * zeta + gamma)) % R
    lag1 = (pow(zeta, n, R) - 1) * inv((zeta - 1) % R, R) % R * alpha % R * alpha % R * d0.card_inv % R
    rl = lz * rz % R
    lin = []
    for i in range(len(bz)):
        v = bz[i] * c_z % R
        if i < n:
            v = (v + pk["s3"][i] * c_s3) % R
        v = v * alpha % R
        if i < n:
            v = (v + pk["qm"][i] * rl + pk["ql"][i] * lz + pk["qr"][i] * rz + pk["qo"][i] * oz + pk["cqk"][i]) % R
        lin.append((v + bz[i] * lag1) % R)
    lin_digest = commit(lin)

    # folded quotient: h1 + zeta^(n+2) h2 + zeta^(2(n+2)) h3, and its digest
    zp = pow(zeta, n + 2, R)
    folded_h = [((h3[i] * zp + h2[i]) % R * zp + h1[i]) % R for i in range(n + 2)]
    folded_h_digest = g1_add(g1_mul(g1_add(g1_mul(hd[2], zp), hd[1]), zp), hd[0])

    # kzg.BatchOpenSinglePoint of (foldedH, linPol, l, r, o, s1, s2) at zeta
    polys = [folded_h, lin, bl, br, bo, pk["s1"], pk["s2"]]
    digests = [folded_h_digest, lin_digest, lro[0], lro[1], lro[2], vk["s"][0], vk["s"][1]]
    claimed = [poly_eval(p, zeta) for p in polys]
    kg = pin.get("kzg_gamma", kzg_derive_gamma(zeta, digests, claimed))
    folded = [0] * max(len(p) for p in polys)
    acc = 1
    for p in polys:
        for j, c in enumerate(p):
            folded[j] = (folded[j] + c * acc) % R
        acc = acc * kg % R
    folded_eval = 0
    for v in reversed(claimed):
        folded_eval = (folded_eval * kg + v) % R
    batch_h = commit(divide_by_x_minus_a(folded, folded_eval, zeta))
    proof = dict(lro=lro, z=z_digest, h=hd, batch_h=batch_h, claimed=claimed, z_open_h=z_open_h, zu=zu)
    if trace is not None:
        trace.update(gamma=gamma, beta=beta, alpha=alpha, zeta=zeta, kzg_gamma=kg, bl=bl, br=br, bo=bo, bz=bz, h=h[:3 * (n + 2)], lin=lin,
                     folded_h_digest=folded_h_digest, lin_digest=lin_digest)
    return proof


def plonk_proof_bytes(proof) -> bytes:
    """Proof.WriteTo (marshal.go): LRO[0..2], Z, H[0..2] compressed (7 x 32 B); BatchedProof = H (32 B) | u32 BE count | claimed
    values (7 x 32 B BE); ZShiftedOpening = H (32 B) | claimed value (32 B).  548 bytes."""
    out = b"".join(ref.g1_compress(p) for p in (*proof["lro"], proof["z"], *proof["h"]))
    out += ref.g1_compress(proof["batch_h"]) + len(proof["claimed"]).to_bytes(4, "big") + b"".join(fr_bytes(v) for v in proof["claimed"])
    out += ref.g1_compress(proof["z_open_h"]) + fr_bytes(proof["zu"])
    return out


# ------------------------------------------------------------------------------------------------ Verify
def _kzg_check(digest, h, value, point, g2) -> bool:
    """e(C - v*G1 + z*H, G2) == e(H, alpha*G2)"""
    lhs = g1_add(g1_add(digest, g1_neg(g1_mul(ref.G1_GEN, value))), g1_mul(h, point))
    return ref.pairing_product_is_one([(lhs, g2[0]), (g1_neg(h), g2[1])])


def plonk_verify(vk, proof, public_inputs, challenges=None) -> bool:
    """plonk.Verify(proof, vk, publicWitness): re-derive the challenges, check the quotient identity at zeta, rebuild the linearised
    digest from the verifying key, fold the batched opening and check the two KZG openings (at zeta and omega*zeta) by pairings."""
    n, u = vk["size"], vk["coset_shift"]
    fs = Transcript("gamma", "beta", "alpha", "zeta")
    _bind_public_data(fs, vk, public_inputs)
    gamma = fs.challenge_fr("gamma", *proof["lro"])
    beta = fs.challenge_fr("beta")
    alpha = fs.challenge_fr("alpha", proof["z"])
    zeta = fs.challenge_fr("zeta", *proof["h"])
    if challenges:
        gamma, beta, alpha, zeta = (challenges.get(k, v) for k, v in (("gamma", gamma), ("beta", beta), ("alpha", alpha), ("zeta", zeta)))
    zn = pow(zeta, n, R)
    zz = (zn - 1) % R
    # PI(zeta) = sum_i L_i(zeta) w_i ; L_i(zeta) = w^i/n * (zeta^n - 1)/(zeta - w^i)
    pi = 0
    for i, w in enumerate(public_inputs):
        wi = pow(vk["generator"], i, R)
        pi = (pi + wi * vk["size_inv"] % R * zz % R * inv((zeta - wi) % R, R) % R * w) % R
    l1 = zz * vk["size_inv"] % R * inv((zeta - 1) % R, R) % R
    quot, lin_z, lz, rz, oz, s1z, s2z = proof["claimed"]
    zu = proof["zu"]
    t = (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) % R * (oz + gamma) % R * alpha % R * zu % R
    lhs = (lin_z + pi + t - alpha * alpha % R * l1) % R
    if lhs != quot * zz % R:
        return False
    # digests: folded quotient and linearised polynomial, from public data only
    zp = pow(zeta, n + 2, R)
    folded_h = g1_add(g1_mul(g1_add(g1_mul(proof["h"][2], zp), proof["h"][1]), zp), proof["h"][0])
    uu = u * u % R
    c_s3 = (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) % R * zu % R * beta % R * alpha % R
    c_z = ((-(lz + beta * zeta + gamma) * (rz + beta * u % R * zeta + gamma) % R * (oz + beta * uu % R * zeta + gamma)) % R * alpha + alpha * alpha % R * l1) % R
    lin_digest = ref.msm_naive(FP, [vk["ql"], vk["qr"], vk["qm"], vk["qo"], vk["qk"], vk["s"][2], proof["z"]], [lz, rz, lz * rz % R, oz, 1, c_s3, c_z])
    digests = [folded_h, lin_digest, proof["lro"][0], proof["lro"][1], proof["lro"][2], vk["s"][0], vk["s"][1]]
    kg = (challenges or {}).get("kzg_gamma", kzg_derive_gamma(zeta, digests, proof["claimed"]))
    fd, fe, acc = None, 0, 1
    for d, v in zip(digests, proof["claimed"]):
        fd = g1_add(fd, g1_mul(d, acc))
        fe = (fe + v * acc) % R
        acc = acc * kg % R
    return _kzg_check(fd, proof["batch_h"], fe, zeta, vk["srs_g2"]) and _kzg_check(proof["z"], proof["z_open_h"], zu, zeta * vk["generator"] % R, vk["srs_g2"])


# ------------------------------------------------------------------------------------------------ wire formats
# What the reference moves between Rust and Go as hex strings / keeps in srs.hex  [REF gnark_backend_ffi/internal/backend/helpers.go:49-94
# (Serialize/DeserializeProvingKey, ...VerifyingKey, ...Proof), backend/common.go:86-125 (LoadSRS / SaveSRS)]: the bytes of gnark's
# WriteTo methods.  [UPSTREAM-RECALL] gnark-crypto v0.9.1 ecc/bn254/marshal.go Encoder: integers big-endian, fr / fp elements 32 B big-endian
# canonical, points COMPRESSED (G1 32 B, G2 64 B), slices of elements / points prefixed with a u32 big-endian length; []int64 goes through
# encoding/binary (raw big-endian words, NO length prefix).
def g1_decompress(b: bytes):
    """inverse of G1Affine.Bytes(): mask 0b10 / 0b11 = compressed with the smallest / largest y, 0b01 = infinity; anything else is an error"""
    flag = b[0] >> 6
    if flag == 1:
        if any(b[1:]) or b[0] & 0x3F:
            raise ValueError("invalid infinity encoding")
        return None
    if flag == 0:
        raise ValueError("uncompressed point where a compressed one is expected")
    x = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:32], "big")
    if x >= ref.Q:
        raise ValueError("invalid fp.Element encoding")
    rhs = (x * x * x + 3) % ref.Q
    y = pow(rhs, (ref.Q + 1) // 4, ref.Q)
    if y * y % ref.Q != rhs:
        raise ValueError("invalid compressed coordinate: square root doesn't exist")
    if (y > (ref.Q - 1) // 2) != (flag == 3):
        y = ref.Q - y
    return (x, y)


def _f2_pow(a, e):
    r = ref.F2_ONE
    while e:
        if e & 1:
            r = ref.f2_mul(r, a)
        a = ref.f2_sqr(a)
        e >>= 1
    return r


def f2_sqrt(a):
    """square root in Fp2 = Fp[u]/(u^2+1), q = 3 mod 4 (Adj & Rodriguez-Henriquez, Alg. 9); None if a is not a square"""
    q = ref.Q
    if a == ref.F2_ZERO:
        return ref.F2_ZERO
    a1 = _f2_pow(a, (q - 3) // 4)
    alpha = ref.f2_mul(a1, ref.f2_mul(a1, a))
    a0 = ref.f2_mul((alpha[0], (-alpha[1]) % q), alpha)  # alpha^q * alpha (Frobenius = conjugation)
    if a0 == (q - 1, 0):
        return None
    x0 = ref.f2_mul(a1, a)
    if alpha == (q - 1, 0):
        return ref.f2_mul((0, 1), x0)
    b = _f2_pow(ref.f2_add(ref.F2_ONE, alpha), (q - 1) // 2)
    return ref.f2_mul(b, x0)


def g2_decompress(b: bytes, subgroup_check: bool = True):
    """inverse of G2Affine.Bytes(): X.A1 || X.A0 big-endian, flags as G1; "largest" compares Y.A1 first, then Y.A0.  The gnark-crypto Decoder
    checks r-torsion membership by default (the twist has a cofactor): restated as r * P = infinity."""
    flag = b[0] >> 6
    if flag == 1:
        return None
    if flag == 0:
        raise ValueError("uncompressed point where a compressed one is expected")
    x1 = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:32], "big")
    x0 = int.from_bytes(b[32:64], "big")
    if x0 >= ref.Q or x1 >= ref.Q:
        raise ValueError("invalid fp.Element encoding")
    x = (x0, x1)
    rhs = ref.f2_add(ref.f2_mul(ref.f2_sqr(x), x), ref.B_G2)
    y = f2_sqrt(rhs)
    if y is None:
        raise ValueError("invalid compressed coordinate: square root doesn't exist")
    largest = ref._lex_largest_fp(y[1]) if y[1] != 0 else ref._lex_largest_fp(y[0])
    if largest != (flag == 3):
        y = ref.f2_neg(y)
    if subgroup_check and ref.ec_mul(ref.FP2, (x, y), R) is not None:
        raise ValueError("invalid point: subgroup check failed")
    return (x, y)


def kzg_srs_bytes(srs) -> bytes:
    """kzg.SRS.WriteTo: G2[0] | G2[1] (64 B each) | u32 BE len(G1) | G1 points (32 B each)."""
    g1 = srs["g1"] if isinstance(srs["g1"], list) else [g1_from_np(p) for p in srs["g1"]]
    return ref.g2_compress(srs["g2"][0]) + ref.g2_compress(srs["g2"][1]) + len(g1).to_bytes(4, "big") + b"".join(ref.g1_compress(p) for p in g1)


def kzg_srs_from_bytes(b: bytes):
    """kzg.SRS.ReadFrom"""
    g2 = [g2_decompress(b[0:64]), g2_decompress(b[64:128])]
    n = int.from_bytes(b[128:132], "big")
    if len(b) != 132 + 32 * n:
        raise ValueError("SRS: %d bytes, the count says %d points" % (len(b), n))
    return dict(g1=[g1_decompress(b[132 + 32 * i:164 + 32 * i]) for i in range(n)], g2=g2)


def _fr_vec_bytes(v) -> bytes:
    return len(v).to_bytes(4, "big") + b"".join(fr_bytes(x) for x in v)


def _domain_bytes(d: Domain) -> bytes:
    """fft.Domain.WriteTo: Cardinality u64 | CardinalityInv | Generator | GeneratorInv | FrMultiplicativeGen | FrMultiplicativeGenInv"""
    return d.n.to_bytes(8, "big") + b"".join(fr_bytes(x) for x in (d.card_inv, d.gen, d.gen_inv, d.coset, d.coset_inv))


def plonk_vk_bytes(vk) -> bytes:
    """plonk.VerifyingKey.WriteTo: Size u64 | SizeInv | Generator | NbPublicVariables u64 | CosetShift | S[0..2] | Ql Qr Qm Qo Qk (compressed)"""
    out = vk["size"].to_bytes(8, "big") + fr_bytes(vk["size_inv"]) + fr_bytes(vk["generator"]) + vk["n_public"].to_bytes(8, "big") + fr_bytes(vk["coset_shift"])
    return out + b"".join(ref.g1_compress(p) for p in (*vk["s"], vk["ql"], vk["qr"], vk["qm"], vk["qo"], vk["qk"]))


def plonk_pk_bytes(pk) -> bytes:
    """plonk.ProvingKey.WriteTo: the verifying key, Domain[0], Domain[1], then Ql, Qr, Qm, Qo, CQk, LQk, S1, S2, S3 as length-prefixed
    []fr.Element and Permutation as 3n raw big-endian int64."""
    out = plonk_vk_bytes(pk["vk"]) + _domain_bytes(pk["d0"]) + _domain_bytes(pk["d1"])
    for k in ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3"):
        out += _fr_vec_bytes(pk[k])
    return out + b"".join(int(p).to_bytes(8, "big", signed=True) for p in pk["perm"])


def plonk_pk_from_bytes(b: bytes):
    """plonk.ProvingKey.ReadFrom (the spr and the SRS are attached by the caller, as the reference does: plonk.go:53-63)"""
    o = 0

    def u64():
        nonlocal o
        o += 8
        return int.from_bytes(b[o - 8:o], "big")

    def fr():
        nonlocal o
        o += 32
        v = int.from_bytes(b[o - 32:o], "big")
        if v >= R:
            raise ValueError("invalid fr.Element encoding")
        return v

    def g1():
        nonlocal o
        o += 32
        return g1_decompress(b[o - 32:o])

    vk = dict(size=u64(), size_inv=fr(), generator=fr(), n_public=u64(), coset_shift=fr())
    vk["s"] = [g1(), g1(), g1()]
    for k in ("ql", "qr", "qm", "qo", "qk"):
        vk[k] = g1()
    doms = []
    for _ in range(2):
        card = u64()
        doms.append((card, [fr() for _ in range(5)]))
    n = doms[0][0]
    pk = dict(vk=vk, n=n, d0=Domain(doms[0][0]), d1=Domain(doms[1][0]))
    for k in ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3"):
        cnt = int.from_bytes(b[o:o + 4], "big")
        o += 4
        if cnt != n:
            raise ValueError("proving key: vector of %d elements, domain of %d" % (cnt, n))
        pk[k] = [fr() for _ in range(cnt)]
    if len(b) - o != 3 * n * 8:
        raise ValueError("proving key: %d bytes left for the permutation, %d expected" % (len(b) - o, 3 * n * 8))
    pk["perm"] = [int.from_bytes(b[o + 8 * i:o + 8 * i + 8], "big", signed=True) for i in range(3 * n)]
    return pk


# ---- Groth16 keys: what ProveWithPK / Preprocess of the reference's intended Groth16 FFI move as hex
# [REF gnark_backend_ffi/backend/groth16/r1cs.go:107-143 (hex of ProvingKey.WriteTo -> provingKey.ReadFrom), :214-266 (Preprocess: both keys out)]
def groth16_pk_compact(pk):
    """oracle key (wire-indexed A / B / G2.B with None at infinity) -> gnark's stored form: the slices without their points at infinity plus
    InfinityA / InfinityB  [UPSTREAM-RECALL gnark v0.8.0 internal/backend/bn254/groth16/setup.go: `if a.IsZero() { InfinityA[i] = true; n++ }`]"""
    inf_a = [P is None for P in pk["g1_a"]]
    inf_b = [P is None for P in pk["g1_b"]]
    assert inf_b == [P is None for P in pk["g2_b"]]
    return dict(pk, g1_a=[P for P in pk["g1_a"] if P is not None], g1_b=[P for P in pk["g1_b"] if P is not None],
                g2_b=[P for P in pk["g2_b"] if P is not None], infinity_a=inf_a, infinity_b=inf_b)


def groth16_pk_bytes(pk) -> bytes:
    """groth16.ProvingKey.WriteTo (raw = false)  [UPSTREAM-RECALL gnark v0.8.0 internal/backend/bn254/groth16/marshal.go]: Domain.WriteTo, then
    through the gnark-crypto Encoder: G1.Alpha, G1.Beta, G1.Delta, G1.A, G1.B, G1.Z, G1.K, G2.Beta, G2.Delta, G2.B, nbWires u64, NbInfinityA u64,
    NbInfinityB u64, InfinityA, InfinityB -- the two []bool go through encoding/binary: one byte (0 / 1) per wire and NO length prefix (which is
    why nbWires is written in front of them).  Takes the wire-indexed oracle key or its compact form."""
    if "infinity_a" not in pk:
        pk = groth16_pk_compact(pk)
    g1s = lambda v: len(v).to_bytes(4, "big") + b"".join(ref.g1_compress(P) for P in v)
    out = _domain_bytes(pk["domain"]) + b"".join(ref.g1_compress(pk[k]) for k in ("g1_alpha", "g1_beta", "g1_delta"))
    out += g1s(pk["g1_a"]) + g1s(pk["g1_b"]) + g1s(pk["g1_z"]) + g1s(pk["g1_k"])
    out += ref.g2_compress(pk["g2_beta"]) + ref.g2_compress(pk["g2_delta"])
    out += len(pk["g2_b"]).to_bytes(4, "big") + b"".join(ref.g2_compress(P) for P in pk["g2_b"])
    nw = len(pk["infinity_a"])
    out += nw.to_bytes(8, "big") + sum(pk["infinity_a"]).to_bytes(8, "big") + sum(pk["infinity_b"]).to_bytes(8, "big")
    return out + bytes(int(x) for x in pk["infinity_a"]) + bytes(int(x) for x in pk["infinity_b"])


def groth16_pk_from_bytes(b: bytes):
    """groth16.ProvingKey.ReadFrom -> the compact form (g1_a / g1_b / g2_b without their points at infinity + the bitmaps)"""
    o = 0

    def take(k):
        nonlocal o
        if o + k > len(b):
            raise ValueError("proving key: truncated")
        o += k
        return b[o - k:o]

    card = int.from_bytes(take(8), "big")
    dom_rest = take(160)
    if card == 0 or card & (card - 1) or _domain_bytes(Domain(card))[8:] != dom_rest:
        raise ValueError("proving key: not a radix-2 domain of gnark-crypto's")
    pk = dict(domain=Domain(card))
    for k in ("g1_alpha", "g1_beta", "g1_delta"):
        pk[k] = g1_decompress(take(32))

    def g1s():
        n = int.from_bytes(take(4), "big")
        return [g1_decompress(take(32)) for _ in range(n)]

    pk["g1_a"], pk["g1_b"], pk["g1_z"], pk["g1_k"] = g1s(), g1s(), g1s(), g1s()
    pk["g2_beta"], pk["g2_delta"] = g2_decompress(take(64)), g2_decompress(take(64))
    n = int.from_bytes(take(4), "big")
    pk["g2_b"] = [g2_decompress(take(64)) for _ in range(n)]
    nw, na, nb = (int.from_bytes(take(8), "big") for _ in range(3))
    ia, ib = take(nw), take(nw)
    if o != len(b):
        raise ValueError("proving key: %d trailing bytes" % (len(b) - o))
    if any(x > 1 for x in ia + ib):
        raise ValueError("proving key: a bool that is neither 0 nor 1")
    pk["infinity_a"], pk["infinity_b"] = [bool(x) for x in ia], [bool(x) for x in ib]
    if sum(ia) != na or sum(ib) != nb or len(pk["g1_a"]) != nw - na or len(pk["g1_b"]) != nw - nb or len(pk["g2_b"]) != nw - nb:
        raise ValueError("proving key: the point counts do not match InfinityA / InfinityB")
    if len(pk["g1_z"]) != card or len(pk["g1_k"]) > nw:
        raise ValueError("proving key: bad Z / K length")
    return pk


def groth16_pk_expand(pk):
    """compact form -> wire-indexed (None at infinity), the layout bn254_ref.groth16_prove takes"""
    out = dict(pk)
    for key, inf in (("g1_a", "infinity_a"), ("g1_b", "infinity_b"), ("g2_b", "infinity_b")):
        it = iter(pk[key])
        out[key] = [None if f else next(it) for f in pk[inf]]
    return out


def groth16_vk_bytes(vk) -> bytes:
    """groth16.VerifyingKey.WriteTo (raw = false)  [UPSTREAM-RECALL gnark v0.8.0 marshal.go: "[alpha]1,[beta]1,[beta]2,[gamma]2,[delta]1,[delta]2,
    uint32(len(Kvk)),[Kvk]1"]; the oracle's vk carries g1_beta / g1_delta only through the proving key, so they are passed in it."""
    out = ref.g1_compress(vk["g1_alpha"]) + ref.g1_compress(vk["g1_beta"]) + ref.g2_compress(vk["g2_beta"]) + ref.g2_compress(vk["g2_gamma"])
    out += ref.g1_compress(vk["g1_delta"]) + ref.g2_compress(vk["g2_delta"])
    return out + len(vk["g1_ic"]).to_bytes(4, "big") + b"".join(ref.g1_compress(P) for P in vk["g1_ic"])


# ------------------------------------------------------------------------------------------------ RawR1CS (the reference's Groth16 payload)
def r1cs_from_raw(raw: dict):
    """buildR1CS of the reference's intended Groth16 FFI  [REF gnark_backend_ffi/backend/groth16/r1cs.go:9-72 (commented out), payload
    src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60]: every mul term gets an internal product variable p with
    (1 * multiplicand) * (1 * multiplier) = 1 * p, every gate ends in (1 * ONE) * (sum coefficient * p + sum coefficient * x + constant * ONE) = 0.
    Made well-defined where the sketch is not: wires = [ONE, public witnesses in witness order, the other witnesses, product variables];
    values[w - 1] is witness w; the product variable is the plain product (the sketch puts the coefficient on the product constraint's output AND
    on the term, which cancels it); the gate's constant term IS part of the sum (the sketch drops it); a mul term with coefficient 0 emits nothing.
    Returns (bn254_ref.R1CS, full wire values)."""
    values = ref.felts_unwire(bytes.fromhex(raw["values"])) if isinstance(raw["values"], str) else list(raw["values"])
    n = len(values)
    pub = [w for w in raw.get("public_inputs", []) if 1 <= w <= n]
    order = [w for w in range(1, n + 1) if w in pub] + [w for w in range(1, n + 1) if w not in pub]
    wire = {w: 1 + k for k, w in enumerate(order)}
    wvals = [1] + [values[w - 1] % R for w in order]
    h2i = lambda h: int(h, 16) % R
    cons = []
    for g in raw["gates"]:
        terms = {}
        for t in g["mul_terms"]:
            c = h2i(t["coefficient"])
            if c == 0:
                continue
            a, b = wire[t["multiplicand"]], wire[t["multiplier"]]
            p = len(wvals)
            wvals.append(wvals[a] * wvals[b] % R)
            cons.append(({a: 1}, {b: 1}, {p: 1}))
            terms[p] = (terms.get(p, 0) + c) % R
        for t in g["add_terms"]:
            x = wire[t["sum"]]
            terms[x] = (terms.get(x, 0) + h2i(t["coefficient"])) % R
        k = h2i(g["constant_term"])
        if k:
            terms[0] = (terms.get(0, 0) + k) % R
        cons.append(({0: 1}, terms, {}))
    return ref.R1CS(1 + len(pub), len(wvals) - 1 - len(pub), cons), wvals
