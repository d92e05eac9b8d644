"""Test infrastructure: big-integer restatement of the four compute phases of the block-sharded computeH
(include/zkmi.h, zk_bn254_groth16_h_shard_dev) and a lock-step driver that plays all G ranks in one process.

The decomposition: a radix-2 transform over D = G*M points = g = log2(G) "cross" stages on the top index bits (first
stages of DIF, last stages of DIT) + a size-M transform of every block.  Cross stages run on transposed data
T[s][j] = element s*M + rank*C + j (C = M/G).  Values are plain integers mod r here (no Montgomery form)."""
from oracle import bn254_ref as ref

R = ref.R


def dif_inplace(a, w):
    """natural in -> bit-reversed out; w = primitive len(a)-th root (gnark FFT(DIF) data movement)."""
    n = len(a)
    span, step = n // 2, 1
    while span >= 1:
        for start in range(0, n, 2 * span):
            for i in range(span):
                x, y = a[start + i], a[start + i + span]
                a[start + i] = (x + y) % R
                a[start + i + span] = (x - y) * pow(w, i * step, R) % R
        span //= 2
        step *= 2


def dit_inplace(a, w):
    """bit-reversed in -> natural out."""
    n = len(a)
    h = 1
    while h < n:
        for start in range(0, n, 2 * h):
            for i in range(h):
                x, y = a[start + i], a[start + i + h] * pow(w, i * (n // (2 * h)), R) % R
                a[start + i] = (x + y) % R
                a[start + i + h] = (x - y) % R
        h *= 2


def cross_dif(T, w_d, log_g, log_m, rank):
    G, M = 1 << log_g, 1 << log_m
    C = M // G
    for j in range(C):
        low = rank * C + j
        for t in range(log_g):
            h = G >> (t + 1)
            for s in range(G):
                if s & h:
                    continue
                e = ((s & (h - 1)) * M + low) << t
                x, y = T[s * C + j], T[(s + h) * C + j]
                T[s * C + j] = (x + y) % R
                T[(s + h) * C + j] = (x - y) * pow(w_d, e, R) % R


def cross_dit(T, w_d, log_g, log_m, rank):
    G, M = 1 << log_g, 1 << log_m
    C = M // G
    for j in range(C):
        low = rank * C + j
        for u in range(log_g):
            h = 1 << u
            for s in range(G):
                if s & h:
                    continue
                e = ((s & (h - 1)) * M + low) * (G >> (u + 1))
                x, y = T[s * C + j], T[(s + h) * C + j] * pow(w_d, e, R) % R
                T[s * C + j] = (x + y) % R
                T[(s + h) * C + j] = (x - y) % R


def phase_int(phase, a, b, c, log_d, log_g, rank):
    """In place on python lists of integers mod r (this rank's arrays)."""
    dom = ref.Domain(1 << log_d)
    log_m = log_d - log_g
    M, G = 1 << log_m, 1 << log_g
    w_m, w_m_inv = pow(dom.gen, G, R), pow(dom.gen_inv, G, R)
    if phase == 0:
        for v in (a, b, c):
            if v is not None:
                cross_dif(v, dom.gen_inv, log_g, log_m, rank)
    elif phase == 1:
        for v in (a, b, c):
            if v is None:
                continue
            dif_inplace(v, w_m_inv)
            for j in range(M):
                v[j] = v[j] * dom.card_inv % R * pow(dom.coset, ref.bitrev(rank * M + j, log_d), R) % R
            dit_inplace(v, w_m)
    elif phase == 4:   # first third of phase 2, per array
        for v in (a, b, c):
            if v is not None:
                cross_dit(v, dom.gen, log_g, log_m, rank)
    elif phase in (2, 5):
        if phase == 2:
            for v in (a, b, c):
                cross_dit(v, dom.gen, log_g, log_m, rank)
        den = ref.inv((pow(dom.coset, dom.n, R) - 1) % R, R)
        for j in range(M):
            a[j] = (a[j] * b[j] - c[j]) * den % R
        cross_dif(a, dom.gen_inv, log_g, log_m, rank)
    elif phase == 3:
        dif_inplace(a, w_m_inv)
        for j in range(M):
            a[j] = a[j] * dom.card_inv % R * pow(dom.coset_inv, ref.bitrev(rank * M + j, log_d), R) % R
    elif phase == 6:   # the six-transform schedule: block of the coefficients of any array given (c)
        for v in (a, b, c):
            if v is None:
                continue
            dif_inplace(v, w_m_inv)
            for j in range(M):
                v[j] = v[j] * dom.card_inv % R
    elif phase == 7:   # transposed a, b after phase 4: product, then the cross stages of the final inverse
        for j in range(M):
            a[j] = a[j] * b[j] % R
        cross_dif(a, dom.gen_inv, log_g, log_m, rank)
    elif phase == 8:   # block of a, block of c's coefficients: rest of FFTInverse(DIF, coset), h = (a - c) / (g^D - 1)
        den = ref.inv((pow(dom.coset, dom.n, R) - 1) % R, R)
        dif_inplace(a, w_m_inv)
        for j in range(M):
            a[j] = (a[j] * dom.card_inv % R * pow(dom.coset_inv, ref.bitrev(rank * M + j, log_d), R) - c[j]) * den % R
    else:
        raise ValueError(phase)


def exchange_all(blocks):
    """All-to-all among virtual ranks: blocks[r] is rank r's array (list or numpy rows); returns the transposed arrays."""
    G = len(blocks)
    C = len(blocks[0]) // G
    out = []
    for r in range(G):
        rows = [blocks[s][r * C:(r + 1) * C] for s in range(G)]
        if hasattr(rows[0], "shape"):
            import numpy as np
            out.append(np.concatenate(rows))
        else:
            out.append([x for row in rows for x in row])
    return out


def run_virtual(phase, A, B, Cc, log_d, per_array=False):
    """Lock-step schedule of parallel.compute_h_sharded over G = len(A) virtual ranks.  phase(p, a, b, c, log_d, log_g, rank)
    works in place on whatever array type the blocks are.  Returns the list of h blocks.
    per_array: the pipelined schedule's calls -- phases 0, 1, 4 one array at a time (b = c = None), then phase 5."""
    G = len(A)
    log_g = G.bit_length() - 1
    for p in range(3):
        A, B, Cc = exchange_all(A), exchange_all(B), exchange_all(Cc)
        for r in range(G):
            if not per_array:
                phase(p, A[r], B[r], Cc[r], log_d, log_g, r)
                continue
            for v in (A[r], B[r], Cc[r]):
                phase(4 if p == 2 else p, v, None, None, log_d, log_g, r)
            if p == 2:
                phase(5, A[r], B[r], Cc[r], log_d, log_g, r)
    A = exchange_all(A)
    for r in range(G):
        phase(3, A[r], None, None, log_d, log_g, r)
    return A


def ntt_step_int(step, a, log_d, log_g, rank, inverse, decimation, coset):
    """Big-integer restatement of zk_bn254_ntt_shard_dev's three steps (include/zkmi.h), in place on this rank's list."""
    dom = ref.Domain(1 << log_d)
    log_m = log_d - log_g
    M, G = 1 << log_m, 1 << log_g
    w_d = dom.gen_inv if inverse else dom.gen
    w_m = pow(w_d, G, R)
    if step == 0:
        (cross_dif if decimation == ref.DIF else cross_dit)(a, w_d, log_g, log_m, rank)
    elif step == 1:
        if not inverse and coset and decimation == ref.DIT:
            for j in range(M):
                a[j] = a[j] * pow(dom.coset, ref.bitrev(rank * M + j, log_d), R) % R
        (dif_inplace if decimation == ref.DIF else dit_inplace)(a, w_m)
        if inverse:
            for j in range(M):
                f = dom.card_inv
                if coset and decimation == ref.DIF:
                    f = f * pow(dom.coset_inv, ref.bitrev(rank * M + j, log_d), R) % R
                a[j] = a[j] * f % R
    elif step == 2:
        if coset and ((not inverse and decimation == ref.DIF) or (inverse and decimation == ref.DIT)):
            base = dom.coset_inv if inverse else dom.coset
            for j in range(M):
                a[j] = a[j] * pow(base, rank * M + j, R) % R
    else:
        raise ValueError(step)


def run_virtual_ntt(step, X, log_d, inverse, decimation, coset):
    """Lock-step schedule of parallel.ntt_sharded over G = len(X) virtual ranks (step(step, a, log_d, log_g, rank, inverse, decimation, coset) works in
    place on whatever the blocks are).  Returns the list of output blocks."""
    G = len(X)
    log_g = G.bit_length() - 1
    run = lambda st, B: [step(st, B[r], log_d, log_g, r, inverse, decimation, coset) for r in range(G)]
    if decimation == 1:
        if coset and not inverse:
            run(2, X)
        X = exchange_all(X)
        run(0, X)
        X = exchange_all(X)
        run(1, X)
        return X
    run(1, X)
    X = exchange_all(X)
    run(0, X)
    X = exchange_all(X)
    if coset and inverse:
        run(2, X)
    return X


def run_virtual_six(phase, A, B, Cc, log_d):
    """Lock-step schedule of parallel.compute_h_sharded(six_transforms=True) over G = len(A) virtual ranks: c only through phases 0 and 6."""
    G = len(A)
    log_g = G.bit_length() - 1
    A, B, Cc = exchange_all(A), exchange_all(B), exchange_all(Cc)
    for r in range(G):
        for v in (A[r], B[r], Cc[r]):
            phase(0, v, None, None, log_d, log_g, r)
    A, B, Cc = exchange_all(A), exchange_all(B), exchange_all(Cc)
    for r in range(G):
        phase(1, A[r], None, None, log_d, log_g, r)
        phase(1, B[r], None, None, log_d, log_g, r)
        phase(6, Cc[r], None, None, log_d, log_g, r)
    A, B = exchange_all(A), exchange_all(B)
    for r in range(G):
        phase(4, A[r], None, None, log_d, log_g, r)
        phase(4, B[r], None, None, log_d, log_g, r)
        phase(7, A[r], B[r], None, log_d, log_g, r)
    A = exchange_all(A)
    for r in range(G):
        phase(8, A[r], None, Cc[r], log_d, log_g, r)
    return A
