/* Curve "template" for the CPU oracle (TEST INFRASTRUCTURE) -- included twice by bn254_oracle.c, once per
 * group, with:  FE (coordinate type), F(x) (field-op prefix), G(x) (group-op prefix), CURVE_LIMBS.
 * Formulas: EFD short-Weierstrass a=0 "xyzz" (madd-2008-s, mdbl-2008-s-1, add-2008-s, dbl-2008-s-1); these are
 * the extended-Jacobian bucket coordinates gnark-crypto's MultiExp uses (g1JacExtended / g2JacExtended,
 * gnark-crypto v0.9.1, /root/reference/gnark_backend_ffi/go.mod:5)  [UPSTREAM-RECALL]. */

typedef struct { FE x, y; } G(aff);
typedef struct { FE x, y, zz, zzz; } G(xyzz);

static inline int G(aff_is_inf)(const G(aff) *p) { return F(is_zero)(&p->x) && F(is_zero)(&p->y); }
static inline int G(is_inf)(const G(xyzz) *p) { return F(is_zero)(&p->zz); }
static inline void G(set_inf)(G(xyzz) *p) { memset(p, 0, sizeof *p); }

static void G(from_aff)(G(xyzz) *r, const G(aff) *p) {
    if (G(aff_is_inf)(p)) { G(set_inf)(r); return; }
    r->x = p->x; r->y = p->y; F(set_one)(&r->zz); F(set_one)(&r->zzz);
}

/* r = 2*p for affine p */
static void G(dbl_aff)(G(xyzz) *r, const G(aff) *p) {
    if (G(aff_is_inf)(p)) { G(set_inf)(r); return; }
    FE u, v, w, s, m, t;
    F(add)(&u, &p->y, &p->y);
    F(sqr)(&v, &u);
    F(mul)(&w, &u, &v);
    F(mul)(&s, &p->x, &v);
    F(sqr)(&m, &p->x); F(add)(&t, &m, &m); F(add)(&m, &m, &t);
    F(sqr)(&r->x, &m); F(sub)(&r->x, &r->x, &s); F(sub)(&r->x, &r->x, &s);
    F(sub)(&t, &s, &r->x); F(mul)(&t, &m, &t);
    F(mul)(&u, &w, &p->y);
    F(sub)(&r->y, &t, &u);
    r->zz = v; r->zzz = w;
}

/* r = 2*r */
static void G(dbl)(G(xyzz) *r) {
    if (G(is_inf)(r)) return;
    FE u, v, w, s, m, t;
    F(add)(&u, &r->y, &r->y);
    F(sqr)(&v, &u);
    F(mul)(&w, &u, &v);
    F(mul)(&s, &r->x, &v);
    F(sqr)(&m, &r->x); F(add)(&t, &m, &m); F(add)(&m, &m, &t);
    FE x3;
    F(sqr)(&x3, &m); F(sub)(&x3, &x3, &s); F(sub)(&x3, &x3, &s);
    F(sub)(&t, &s, &x3); F(mul)(&t, &m, &t);
    F(mul)(&u, &w, &r->y);
    F(sub)(&r->y, &t, &u);
    r->x = x3;
    F(mul)(&r->zz, &v, &r->zz);
    F(mul)(&r->zzz, &w, &r->zzz);
}

/* r += p (affine) ; neg != 0 subtracts */
static void G(madd)(G(xyzz) *r, const G(aff) *p, int neg) {
    if (G(aff_is_inf)(p)) return;
    FE py = p->y;
    if (neg) F(neg)(&py, &py);
    if (G(is_inf)(r)) { r->x = p->x; r->y = py; F(set_one)(&r->zz); F(set_one)(&r->zzz); return; }
    FE u2, s2, pp, rr, ppp, q, t;
    F(mul)(&u2, &p->x, &r->zz);
    F(mul)(&s2, &py, &r->zzz);
    F(sub)(&pp, &u2, &r->x);   /* P */
    F(sub)(&rr, &s2, &r->y);   /* R */
    if (F(is_zero)(&pp)) {
        if (F(is_zero)(&rr)) { G(aff) pa; pa.x = p->x; pa.y = py; G(dbl_aff)(r, &pa); }
        else G(set_inf)(r);
        return;
    }
    FE p2;
    F(sqr)(&p2, &pp);
    F(mul)(&ppp, &pp, &p2);
    F(mul)(&q, &r->x, &p2);
    FE x3;
    F(sqr)(&x3, &rr); F(sub)(&x3, &x3, &ppp); F(sub)(&x3, &x3, &q); F(sub)(&x3, &x3, &q);
    F(sub)(&t, &q, &x3); F(mul)(&t, &rr, &t);
    F(mul)(&q, &r->y, &ppp);
    F(sub)(&r->y, &t, &q);
    r->x = x3;
    F(mul)(&r->zz, &r->zz, &p2);
    F(mul)(&r->zzz, &r->zzz, &ppp);
}

/* r += p (xyzz) */
static void G(add)(G(xyzz) *r, const G(xyzz) *p) {
    if (G(is_inf)(p)) return;
    if (G(is_inf)(r)) { *r = *p; return; }
    FE u1, u2, s1, s2, pp, rr, p2, ppp, q, t, x3;
    F(mul)(&u1, &r->x, &p->zz);
    F(mul)(&u2, &p->x, &r->zz);
    F(mul)(&s1, &r->y, &p->zzz);
    F(mul)(&s2, &p->y, &r->zzz);
    F(sub)(&pp, &u2, &u1);
    F(sub)(&rr, &s2, &s1);
    if (F(is_zero)(&pp)) {
        if (F(is_zero)(&rr)) G(dbl)(r);
        else G(set_inf)(r);
        return;
    }
    F(sqr)(&p2, &pp);
    F(mul)(&ppp, &pp, &p2);
    F(mul)(&q, &u1, &p2);
    F(sqr)(&x3, &rr); F(sub)(&x3, &x3, &ppp); F(sub)(&x3, &x3, &q); F(sub)(&x3, &x3, &q);
    F(sub)(&t, &q, &x3); F(mul)(&t, &rr, &t);
    F(mul)(&q, &s1, &ppp);
    F(sub)(&r->y, &t, &q);
    r->x = x3;
    F(mul)(&r->zz, &r->zz, &p->zz); F(mul)(&r->zz, &r->zz, &p2);
    F(mul)(&r->zzz, &r->zzz, &p->zzz); F(mul)(&r->zzz, &r->zzz, &ppp);
}

static void G(to_aff)(G(aff) *r, const G(xyzz) *p) {
    if (G(is_inf)(p)) { memset(r, 0, sizeof *r); return; }
    FE zi, zi2, zi3;
    /* x = X/ZZ, y = Y/ZZZ ; 1/ZZZ, then 1/ZZ = ZZZ^-1 ... keep it simple: two inversions folded into one */
    FE prod; F(mul)(&prod, &p->zz, &p->zzz);
    F(inv)(&zi, &prod);               /* 1/(zz*zzz) */
    F(mul)(&zi2, &zi, &p->zzz);       /* 1/zz  */
    F(mul)(&zi3, &zi, &p->zz);        /* 1/zzz */
    F(mul)(&r->x, &p->x, &zi2);
    F(mul)(&r->y, &p->y, &zi3);
}

/* r = k * p, k canonical little-endian limbs (double-and-add, MSB first) */
static void G(scalar_mul)(G(xyzz) *r, const G(aff) *p, const uint64_t k[4]) {
    G(set_inf)(r);
    for (int i = 255; i >= 0; i--) {
        G(dbl)(r);
        if ((k[i >> 6] >> (i & 63)) & 1) G(madd)(r, p, 0);
    }
}

static int G(on_curve)(const G(aff) *p) {
    if (G(aff_is_inf)(p)) return 1;
    FE l, r;
    F(sqr)(&l, &p->y);
    F(sqr)(&r, &p->x); F(mul)(&r, &r, &p->x); F(add)(&r, &r, &G(curve_b));
    return F(eq)(&l, &r);
}

/* ---- bucket-method MSM in gnark-crypto's shape.
 * digits: signed c-bit, d in [-2^(c-1), 2^(c-1)]  (partitionScalars);  per window: 2^(c-1) xyzz buckets,
 * add or subtract by sign, zero digits skipped;  running-sum reduce;  Horner combine with c doublings. */
static void G(msm_window)(G(xyzz) *out, const G(aff) *pts, const int32_t *digits /* n, this window */, size_t n, int c) {
    size_t nb = (size_t)1 << (c - 1);
    G(xyzz) *buckets = (G(xyzz) *)calloc(nb, sizeof(G(xyzz)));
    for (size_t i = 0; i < n; i++) {
        int32_t d = digits[i];
        if (d == 0) continue;
        if (d > 0) G(madd)(&buckets[d - 1], &pts[i], 0);
        else G(madd)(&buckets[-d - 1], &pts[i], 1);
    }
    G(xyzz) run, acc;
    G(set_inf)(&run); G(set_inf)(&acc);
    for (size_t k = nb; k-- > 0;) {
        if (!G(is_inf)(&buckets[k])) G(add)(&run, &buckets[k]);
        G(add)(&acc, &run);
    }
    *out = acc;
    free(buckets);
}

static int G(msm)(G(aff) *out, const G(aff) *pts, const uint64_t *scalars /* canonical, n*4 */, size_t n, int c, int nthreads) {
    if (n == 0) { memset(out, 0, sizeof *out); return 0; }
    if (c <= 0) c = msm_best_c(n);
    int nwin = (255 + c - 1) / c;  /* fr.Bits+1 = 255 bits so the top digit never carries out */
    int32_t *digits = (int32_t *)malloc(sizeof(int32_t) * n * (size_t)nwin);
    msm_partition_scalars(digits, scalars, n, c, nwin);
    G(xyzz) *wsum = (G(xyzz) *)malloc(sizeof(G(xyzz)) * (size_t)nwin);
    /* one task per window, like gnark-crypto's one goroutine per chunk; if there are more threads than
     * windows, split the points of each window in `split` ranges (gnark: "if nbTasks > nbChunks split the msm") */
    int split = 1;
    if (nthreads > nwin) { split = (nthreads + nwin - 1) / nwin; if ((size_t)split > n / 1024 + 1) split = (int)(n / 1024 + 1); }
    G(xyzz) *part = (G(xyzz) *)malloc(sizeof(G(xyzz)) * (size_t)nwin * (size_t)split);
    int ntask = nwin * split;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int t = 0; t < ntask; t++) {
        int w = t / split, sidx = t % split;
        size_t lo = n * (size_t)sidx / (size_t)split, hi = n * (size_t)(sidx + 1) / (size_t)split;
        G(msm_window)(&part[t], pts + lo, digits + (size_t)w * n + lo, hi - lo, c);
    }
    for (int w = 0; w < nwin; w++) {
        wsum[w] = part[w * split];
        for (int sidx = 1; sidx < split; sidx++) G(add)(&wsum[w], &part[w * split + sidx]);
    }
    G(xyzz) total;
    G(set_inf)(&total);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) G(dbl)(&total);
        G(add)(&total, &wsum[w]);
    }
    G(to_aff)(out, &total);
    free(part); free(wsum); free(digits);
    return 0;
}

static void G(msm_naive)(G(aff) *out, const G(aff) *pts, const uint64_t *scalars, size_t n) {
    G(xyzz) total, t;
    G(set_inf)(&total);
    for (size_t i = 0; i < n; i++) {
        G(scalar_mul)(&t, &pts[i], scalars + 4 * i);
        G(add)(&total, &t);
    }
    G(to_aff)(out, &total);
}
