/* CPU oracle, PLONK part -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Included at the end of bn254_oracle.c (it uses that file's static field, curve,
 * MSM and fft.Domain restatements); the API is declared in bn254_oracle.h.
 *
 * A C / OpenMP restatement of oracle/plonk_ref.py's plonk_setup and plonk_prove -- which restate gnark v0.8.0 internal/backend/bn254/plonk
 * {setup.go, prove.go} and gnark-crypto v0.9.1 {kzg, fiat-shamir, iop} [UPSTREAM-RECALL], reached from the reference at
 * /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:21 (plonk.Setup) and :67 (plonk.Prove, the reference's only live prove path:
 * PlonkProveWithPK, gnark_backend_ffi/main.go:24-37) -- so that 2^20 .. 2^22-gate instances have proof BYTES to compare and a CPU time to report.
 * Same inputs (the 9 blinding scalars are an input), same Fiat-Shamir transcript, same Proof.WriteTo layout; pinned against plonk_ref.py on the
 * committed fixtures (tests/test_plonk_oracle.py).  PARITY UNPINNED against upstream, like everything else under oracle/. */
#include <stdio.h>
#include <time.h>
/* ORC_PLONK_TRACE=1: wall-clock of the prover's phases on stderr (where a CPU proof spends its time; tooling only) */
static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void orc_trace(const char *what, double *t0) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("ORC_PLONK_TRACE"); on = e && *e == '1'; }
    double t1 = orc_now();
    if (on) fprintf(stderr, "[orc_plonk] %-28s %8.3f s\n", what, t1 - *t0);
    *t0 = t1;
}

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4), for fiatshamir.NewTranscript(sha256.New(), ...) */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill; } sha256_t;
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror32(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
static void sha256_block(sha256_t *s, const uint8_t *p) {
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    memcpy(a, s->h, sizeof a);
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ror32(a[4], 6) ^ ror32(a[4], 11) ^ ror32(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
        uint32_t t1 = a[7] + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = ror32(a[0], 2) ^ ror32(a[0], 13) ^ ror32(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
        uint32_t t2 = S0 + mj;
        a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) s->h[i] += a[i];
}
static void sha256_init(sha256_t *s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv); s->len = 0; s->fill = 0;
}
static void sha256_update(sha256_t *s, const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    s->len += n;
    while (n) {
        size_t k = 64 - s->fill; if (k > n) k = n;
        memcpy(s->buf + s->fill, p, k); s->fill += k; p += k; n -= k;
        if (s->fill == 64) { sha256_block(s, s->buf); s->fill = 0; }
    }
}
static void sha256_final(sha256_t *s, uint8_t out[32]) {
    uint64_t bits = s->len * 8;
    uint8_t pad = 0x80; sha256_update(s, &pad, 1);
    pad = 0; while (s->fill != 56) sha256_update(s, &pad, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha256_update(s, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(s->h[i] >> 24); out[4 * i + 1] = (uint8_t)(s->h[i] >> 16); out[4 * i + 2] = (uint8_t)(s->h[i] >> 8); out[4 * i + 3] = (uint8_t)s->h[i]; }
}
void orc_sha256(const uint8_t *data, size_t n, uint8_t out[32]) { sha256_t s; sha256_init(&s); sha256_update(&s, data, n); sha256_final(&s, out); }

/* ------------------------------------------------------------------ encodings for the transcript and Proof.WriteTo */
/* fr.Element.Marshal(): 32 bytes big-endian, canonical */
static void fr_be_bytes(uint8_t out[32], const fe *mont) { fe c; fld_from_mont(&FR, &c, mont); fe_to_be(out, &c); }
/* G1Affine.RawBytes() / Marshal(): uncompressed X || Y big-endian (64 B); infinity = 0x40 flag + zeros */
static void g1_raw_bytes(uint8_t out[64], const g1_aff *p) {
    if (g1_aff_is_inf(p)) { memset(out, 0, 64); out[0] = 0x40; return; }
    fe x, y; fld_from_mont(&FP, &x, &p->x); fld_from_mont(&FP, &y, &p->y);
    fe_to_be(out, &x); fe_to_be(out + 32, &y);
}
/* fr.SetBytes of a 32-byte digest: big-endian integer reduced mod r -> Montgomery */
static void fr_from_digest(fe *out, const uint8_t d[32]) {
    fe x;
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int b = 0; b < 8; b++) v = (v << 8) | d[8 * (3 - i) + b]; x.l[i] = v; }
    fr_reduce256(&x);
    fld_to_mont(&FR, out, &x);
}

/* ------------------------------------------------------------------ small Fr helpers */
static inline void fr_add(fe *r, const fe *a, const fe *b) { fld_add(&FR, r, a, b); }
static inline void fr_sub(fe *r, const fe *a, const fe *b) { fld_sub(&FR, r, a, b); }
static fe *fr_zeros(size_t n) { return (fe *)calloc(n ? n : 1, sizeof(fe)); }
static int thr(int nthreads) { return nthreads > 0 ? nthreads : 1; }

/* p(x), Horner per chunk and a combination of the chunks (the value of a polynomial does not depend on the association) */
static void poly_eval_c(fe *out, const fe *p, size_t len, const fe *x, int nthreads) {
    const size_t chunk = 1 << 14, nch = (len + chunk - 1) / chunk;
    if (!len) { memset(out, 0, sizeof *out); return; }
    fe *part = fr_zeros(nch);
#pragma omp parallel for schedule(static) num_threads(thr(nthreads)) if (len >= 4096)
    for (size_t c = 0; c < nch; c++) {
        size_t lo = c * chunk, hi = lo + chunk < len ? lo + chunk : len;
        fe acc; memset(&acc, 0, sizeof acc);
        for (size_t i = hi; i-- > lo;) { fr_mul(&acc, &acc, x); fr_add(&acc, &acc, &p[i]); }
        part[c] = acc;
    }
    fe xc, acc; fr_pow_u64(&xc, x, (uint64_t)chunk); memset(&acc, 0, sizeof acc);
    for (size_t c = nch; c-- > 0;) { fr_mul(&acc, &acc, &xc); fr_add(&acc, &acc, &part[c]); }
    free(part);
    *out = acc;
}
/* v[i] <- v[0] * ... * v[i]: per-chunk running products, a scan over the chunks' totals, one more multiplication per element */
static void prefix_products(fe *v, size_t n, int nthreads) {
    const size_t chunk = 1 << 14, nch = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) num_threads(thr(nthreads)) if (n >= 4096)
    for (size_t c = 0; c < nch; c++) {
        const size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        for (size_t i = lo + 1; i < hi; i++) fr_mul(&v[i], &v[i], &v[i - 1]);
    }
    fe *carry = (fe *)malloc(sizeof(fe) * nch);
    carry[0] = FR.one;
    for (size_t c = 1; c < nch; c++) fr_mul(&carry[c], &carry[c - 1], &v[c * chunk - 1]);
#pragma omp parallel for schedule(static) num_threads(thr(nthreads)) if (n >= 4096)
    for (size_t c = 1; c < nch; c++) {
        const size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        for (size_t i = lo; i < hi; i++) fr_mul(&v[i], &v[i], &carry[c]);
    }
    free(carry);
}
/* kzg.dividePolyByXminusA: (f - f(a)) / (X - a) by synthetic division, in place; the quotient is f[1 .. len) afterwards */
static void divide_by_x_minus_a_c(fe *f, size_t len, const fe *fa, const fe *a) {
    fr_sub(&f[0], &f[0], fa);
    for (size_t i = len - 1; i-- > 0;) { fe t; fr_mul(&t, &f[i + 1], a); fr_add(&f[i], &f[i], &t); }
}
/* Lagrange (regular) -> canonical (regular): FFTInverse(DIF) then BitReverse, as setup.go / prove.go do */
static void to_canonical(const domain_t *d, fe *v, int nthreads) { domain_fft(d, v, 1, ORC_DIF, 0, nthreads); orc_fr_bit_reverse((uint64_t *)v, d->logn); }
/* kzg.Commit: MultiExp(srs.G1[:len(p)], p) */
static void kzg_commit(g1_aff *out, const g1_aff *srs, const fe *p, size_t len, int nthreads) {
    fe *k = (fe *)malloc(sizeof(fe) * (len ? len : 1));
    scalars_canonical(k, (const uint64_t *)p, len, 1);
    g1_msm(out, srs, (const uint64_t *)k, len, 0, nthreads);
    free(k);
}

/* ------------------------------------------------------------------ the key (what gnark's ProvingKey keeps; plonk_ref.plonk_setup) */
struct orc_plonk_pk {
    unsigned log_n, log_n4;
    size_t n, n4, n_public, n_constraints, n_vars;
    fe *poly[9];       /* canonical ql, qr, qm, qo, cqk | Lagrange lqk | canonical s1, s2, s3 : n each */
    uint32_t *perm;    /* 3n */
    uint32_t *x[3];    /* xa, xb, xc: n_constraints each */
    g1_aff vk[8];      /* [S1], [S2], [S3], [Ql], [Qr], [Qm], [Qo], [Qk] */
    const g1_aff *srs; /* borrowed: >= n + 3 points, alive as long as the key */
    domain_t d0, d1;
};
enum { PQL = 0, PQR, PQM, PQO, PCQK, PLQK, PS1, PS2, PS3 };

static unsigned ceil_log2(size_t m) { unsigned l = 0; while (((size_t)1 << l) < m) l++; return l; }

void orc_plonk_pk_free(orc_plonk_pk *pk) {
    if (!pk) return;
    for (int i = 0; i < 9; i++) free(pk->poly[i]);
    free(pk->perm);
    for (int i = 0; i < 3; i++) free(pk->x[i]);
    if (pk->d0.tw) domain_free(&pk->d0);
    if (pk->d1.tw) domain_free(&pk->d1);
    free(pk);
}

/* setup.go buildPermutation: position -> permuted position over the 3 * size wire slots (L | R | O), placeholders included */
static void build_permutation_c(orc_plonk_pk *pk) {
    const size_t n = pk->n, npub = pk->n_public, nc = pk->n_constraints;
    uint32_t *lro = (uint32_t *)calloc(3 * n, sizeof(uint32_t));
    for (size_t i = 0; i < npub; i++) lro[i] = (uint32_t)i;
    for (size_t i = 0; i < nc; i++) { lro[npub + i] = pk->x[0][i]; lro[n + npub + i] = pk->x[1][i]; lro[2 * n + npub + i] = pk->x[2][i]; }
    const size_t nv = pk->n_vars ? pk->n_vars : 1;
    int64_t *cycle = (int64_t *)malloc(sizeof(int64_t) * nv);
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * 3 * n);
    for (size_t i = 0; i < nv; i++) cycle[i] = -1;
    for (size_t i = 0; i < 3 * n; i++) perm[i] = -1;
    for (size_t i = 0; i < 3 * n; i++) {
        if (cycle[lro[i]] != -1) perm[i] = cycle[lro[i]];
        cycle[lro[i]] = (int64_t)i;
    }
    for (size_t i = 0; i < 3 * n; i++)
        if (perm[i] == -1) perm[i] = cycle[lro[i]];
    for (size_t i = 0; i < 3 * n; i++) pk->perm[i] = (uint32_t)perm[i];
    free(lro); free(cycle); free(perm);
}

/* getIDSmallDomain: [gen^i | u gen^i | u^2 gen^i], 3n entries */
static fe *id_small_domain(const orc_plonk_pk *pk, int nthreads) {
    const size_t n = pk->n;
    fe u, uu; fld_set_u64(&FR, &u, 5); fr_mul(&uu, &u, &u);
    fe *id = (fe *)malloc(sizeof(fe) * 3 * n);
    fr_powers(id, n, &pk->d0.gen, NULL, nthreads);
    fr_powers(id + n, n, &pk->d0.gen, &u, nthreads);
    fr_powers(id + 2 * n, n, &pk->d0.gen, &uu, nthreads);
    return id;
}

orc_plonk_pk *orc_plonk_setup(const orc_plonk_circuit *c, int nthreads) {
    orc_init();
    const size_t npub = c->n_public, nc = c->n_constraints, size_system = nc + npub;
    if (!size_system || !c->srs_g1) return NULL;
    orc_plonk_pk *pk = (orc_plonk_pk *)calloc(1, sizeof *pk);
    pk->log_n = ceil_log2(size_system);
    pk->log_n4 = ceil_log2((size_system < 6 ? 8 : 4) * size_system);  /* fft.NewDomain(4 * sizeSystem), 8 * when sizeSystem < 6 */
    pk->n = (size_t)1 << pk->log_n; pk->n4 = (size_t)1 << pk->log_n4;
    pk->n_public = npub; pk->n_constraints = nc; pk->n_vars = c->n_vars;
    if (c->srs_len < pk->n + 3) { free(pk); return NULL; }
    for (size_t i = 0; i < nc; i++)
        if (c->xa[i] >= c->n_vars || c->xb[i] >= c->n_vars || c->xc[i] >= c->n_vars) { free(pk); return NULL; }
    pk->srs = (const g1_aff *)c->srs_g1;
    domain_init(&pk->d0, pk->log_n); domain_init(&pk->d1, pk->log_n4);
    const size_t n = pk->n;
    for (int i = 0; i < 9; i++) pk->poly[i] = fr_zeros(n);
    pk->perm = (uint32_t *)calloc(3 * n, sizeof(uint32_t));
    const uint32_t *xs[3] = {c->xa, c->xb, c->xc};
    for (int k = 0; k < 3; k++) { pk->x[k] = (uint32_t *)malloc(sizeof(uint32_t) * (nc ? nc : 1)); memcpy(pk->x[k], xs[k], sizeof(uint32_t) * nc); }
    fe minus_one; memset(&minus_one, 0, sizeof minus_one); fr_sub(&minus_one, &minus_one, &FR.one);
    for (size_t i = 0; i < npub; i++) pk->poly[PQL][i] = minus_one;  /* placeholder gates -PUB_i + qk_i = 0; qk_i is completed by the prover */
    const uint64_t *src[5] = {c->ql, c->qr, c->qm, c->qo, c->qk};
    const int dst[5] = {PQL, PQR, PQM, PQO, PCQK};
    for (int k = 0; k < 5; k++) memcpy(pk->poly[dst[k]] + npub, src[k], nc * sizeof(fe));
    memcpy(pk->poly[PLQK], pk->poly[PCQK], n * sizeof(fe));
    build_permutation_c(pk);
    fe *id = id_small_domain(pk, nthreads);
    for (int j = 0; j < 3; j++) {
        fe *s = pk->poly[PS1 + j];
#pragma omp parallel for schedule(static) num_threads(thr(nthreads)) if (n >= 4096)
        for (size_t i = 0; i < n; i++) s[i] = id[pk->perm[(size_t)j * n + i]];
    }
    free(id);
    for (int k = 0; k < 9; k++)
        if (k != PLQK) to_canonical(&pk->d0, pk->poly[k], nthreads);
    const int order[8] = {PS1, PS2, PS3, PQL, PQR, PQM, PQO, PCQK};
    for (int k = 0; k < 8; k++) kzg_commit(&pk->vk[k], pk->srs, pk->poly[order[k]], n, nthreads);
    return pk;
}

void orc_plonk_pk_sizes(const orc_plonk_pk *pk, size_t *n, size_t *n4) { if (n) *n = pk->n; if (n4) *n4 = pk->n4; }
/* which 0..8: canonical ql, qr, qm, qo, cqk, Lagrange lqk, canonical s1, s2, s3 (n elements); 9: the 8 verifying-key digests (64 limbs); 10: permutation (3n u32) */
void orc_plonk_pk_get(const orc_plonk_pk *pk, int which, void *out) {
    if (which >= 0 && which < 9) memcpy(out, pk->poly[which], pk->n * sizeof(fe));
    else if (which == 9) memcpy(out, pk->vk, sizeof pk->vk);
    else if (which == 10) memcpy(out, pk->perm, 3 * pk->n * sizeof(uint32_t));
}

/* (*Polynomial).Blind(k - 1): p += Q(X) (X^n - 1) with Q's coefficients = rnd[0 .. k); p has n + k entries, the top k zero on entry */
static void blind_c(fe *p, size_t n, const fe *rnd, int k) {
    for (int i = 0; i < k; i++) { fr_sub(&p[i], &p[i], &rnd[i]); fr_add(&p[n + i], &p[n + i], &rnd[i]); }
}
/* natural-order evaluations on the coset g * <W> of the big domain: bit_reverse(FFT(p padded, DIF, coset)) */
static fe *coset_eval_c(const orc_plonk_pk *pk, const fe *p, size_t len, int nthreads) {
    fe *e = fr_zeros(pk->n4);
    memcpy(e, p, len * sizeof(fe));
    domain_fft(&pk->d1, e, 0, ORC_DIF, 1, nthreads);
    orc_fr_bit_reverse((uint64_t *)e, pk->log_n4);
    return e;
}
typedef struct { sha256_t s; } fs_t;
static void fs_begin(fs_t *t, const char *id, const uint8_t *prev) { sha256_init(&t->s); sha256_update(&t->s, id, strlen(id)); if (prev) sha256_update(&t->s, prev, 32); }
static void fs_point(fs_t *t, const g1_aff *p) { uint8_t b[64]; g1_raw_bytes(b, p); sha256_update(&t->s, b, 64); }
static void fs_fr(fs_t *t, const fe *x) { uint8_t b[32]; fr_be_bytes(b, x); sha256_update(&t->s, b, 32); }
static void fs_end(fs_t *t, uint8_t digest[32], fe *challenge) { sha256_final(&t->s, digest); fr_from_digest(challenge, digest); }

/* plonk.Prove after the solver (plonk_ref.plonk_prove).  solution: the values of all variables, public first (Montgomery).  blinders: 9 scalars -- (l: 2, r: 2,
 * o: 2, z: 3), the fr.SetRandom draws of Blind(1) x 3 and Blind(2) in call order.  challenges_out (optional): gamma, beta, alpha, zeta, kzg gamma (5 x 4 limbs).
 * Returns 0; -2 when the quotient is not a polynomial (the constraint system is not satisfied); -1 on bad arguments. */
int orc_plonk_prove(const orc_plonk_pk *pk, const uint64_t *solution_, const uint64_t *blinders_, int nthreads, uint8_t proof_out[548], uint64_t *challenges_out) {
    orc_init();
    if (!pk || !solution_ || !blinders_ || !proof_out) return -1;
    const size_t n = pk->n, N4 = pk->n4, rho = N4 / n, npub = pk->n_public, nc = pk->n_constraints;
    const fe *sol = (const fe *)solution_, *bl_ = (const fe *)blinders_;
    const int T = thr(nthreads);
    fe u, uu; fld_set_u64(&FR, &u, 5); fr_mul(&uu, &u, &u);
    double tr = orc_now();

    /* l, r, o: Lagrange (evaluateLROSmallDomain: placeholders first, padding = solution[0]) -> canonical, blinded with degree-1 masks, committed */
    fe *lag[3], *b[3];  /* lag: unblinded Lagrange l, r, o (the ratio below uses them); b: blinded canonical, n + 2 entries */
    fe s0; memset(&s0, 0, sizeof s0); if (pk->n_vars) s0 = sol[0];
    for (int k = 0; k < 3; k++) {
        lag[k] = (fe *)malloc(sizeof(fe) * n);
        for (size_t i = 0; i < n; i++) lag[k][i] = s0;
    }
    for (size_t i = 0; i < npub; i++) lag[0][i] = sol[i];
#pragma omp parallel for schedule(static) num_threads(T) if (nc >= 4096)
    for (size_t i = 0; i < nc; i++) { lag[0][npub + i] = sol[pk->x[0][i]]; lag[1][npub + i] = sol[pk->x[1][i]]; lag[2][npub + i] = sol[pk->x[2][i]]; }
    g1_aff lro[3];
    for (int k = 0; k < 3; k++) {
        b[k] = fr_zeros(n + 2);
        memcpy(b[k], lag[k], n * sizeof(fe));
        to_canonical(&pk->d0, b[k], nthreads);
        blind_c(b[k], n, bl_ + 2 * k, 2);
        kzg_commit(&lro[k], pk->srs, b[k], n + 2, nthreads);
    }
    orc_trace("lro: canonical + 3 commits", &tr);
    /* gamma <- bindPublicData (S1..S3, Ql, Qr, Qm, Qo, Qk digests, the public inputs) then [L], [R], [O]; beta <- nothing but gamma */
    fs_t fs; uint8_t dg[32], db[32], da[32], dz[32]; fe gamma, beta, alpha, zeta;
    fs_begin(&fs, "gamma", NULL);
    for (int k = 0; k < 8; k++) fs_point(&fs, &pk->vk[k]);
    for (size_t i = 0; i < npub; i++) fs_fr(&fs, &sol[i]);
    for (int k = 0; k < 3; k++) fs_point(&fs, &lro[k]);
    fs_end(&fs, dg, &gamma);
    fs_begin(&fs, "beta", dg); fs_end(&fs, db, &beta);

    /* Z: the copy-constraint ratio (iop.BuildRatioCopyConstraint) on the UNblinded l, r, o; canonical, blinded with a degree-2 mask */
    fe *id = id_small_domain(pk, nthreads);
    fe *num = (fe *)malloc(sizeof(fe) * n), *den = (fe *)malloc(sizeof(fe) * n);
    num[0] = FR.one; den[0] = FR.one;
#pragma omp parallel for schedule(static) num_threads(T) if (n >= 4096)
    for (size_t i = 0; i < n - 1; i++) {
        fe a = FR.one, bb = FR.one, t;
        for (int j = 0; j < 3; j++) {
            fr_mul(&t, &beta, &id[i + (size_t)j * n]); fr_add(&t, &t, &lag[j][i]); fr_add(&t, &t, &gamma); fr_mul(&a, &a, &t);
            fr_mul(&t, &beta, &id[pk->perm[i + (size_t)j * n]]); fr_add(&t, &t, &lag[j][i]); fr_add(&t, &t, &gamma); fr_mul(&bb, &bb, &t);
        }
        num[i + 1] = a; den[i + 1] = bb;
    }
    prefix_products(num, n, nthreads); prefix_products(den, n, nthreads);
    fe *bz = fr_zeros(n + 3);
    {   /* z = num / den: one inversion per chunk (Montgomery's trick) */
        const size_t chunk = 1 << 12, nch = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) num_threads(T) if (n >= 4096)
        for (size_t c = 0; c < nch; c++) {
            const size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
            fe pre[1 << 12];
            fe acc = FR.one;
            for (size_t i = lo; i < hi; i++) { pre[i - lo] = acc; fr_mul(&acc, &acc, &den[i]); }
            fe inv; fld_inv(&FR, &inv, &acc);
            for (size_t i = hi; i-- > lo;) { fe di; fr_mul(&di, &inv, &pre[i - lo]); fr_mul(&inv, &inv, &den[i]); fr_mul(&bz[i], &num[i], &di); }
        }
    }
    free(num); free(den); free(id);
    for (int k = 0; k < 3; k++) free(lag[k]);
    to_canonical(&pk->d0, bz, nthreads);
    blind_c(bz, n, bl_ + 6, 3);
    g1_aff z_digest; kzg_commit(&z_digest, pk->srs, bz, n + 3, nthreads);
    orc_trace("z: ratio + canonical + commit", &tr);
    fs_begin(&fs, "alpha", db); fs_point(&fs, &z_digest); fs_end(&fs, da, &alpha);

    /* qk completed with the public inputs (Lagrange), canonical */
    fe *qk_full = (fe *)malloc(sizeof(fe) * n);
    memcpy(qk_full, pk->poly[PLQK], n * sizeof(fe));
    for (size_t i = 0; i < npub; i++) qk_full[i] = sol[i];
    to_canonical(&pk->d0, qk_full, nthreads);

    /* quotient on the coset of the big domain, natural index order: x_i = g W^i */
    fe *e_l = coset_eval_c(pk, b[0], n + 2, nthreads), *e_r = coset_eval_c(pk, b[1], n + 2, nthreads), *e_o = coset_eval_c(pk, b[2], n + 2, nthreads);
    fe *e_z = coset_eval_c(pk, bz, n + 3, nthreads), *e_qk = coset_eval_c(pk, qk_full, n, nthreads);
    free(qk_full);
    fe *e_ql = coset_eval_c(pk, pk->poly[PQL], n, nthreads), *e_qr = coset_eval_c(pk, pk->poly[PQR], n, nthreads), *e_qm = coset_eval_c(pk, pk->poly[PQM], n, nthreads);
    fe *e_qo = coset_eval_c(pk, pk->poly[PQO], n, nthreads);
    fe *e_s1 = coset_eval_c(pk, pk->poly[PS1], n, nthreads), *e_s2 = coset_eval_c(pk, pk->poly[PS2], n, nthreads), *e_s3 = coset_eval_c(pk, pk->poly[PS3], n, nthreads);
    fe *l1 = (fe *)malloc(sizeof(fe) * n);  /* L_1 = (X^n - 1) / (n (X - 1)) = (1 / n) sum X^i */
    for (size_t i = 0; i < n; i++) l1[i] = pk->d0.card_inv;
    fe *e_l1 = coset_eval_c(pk, l1, n, nthreads);
    free(l1);
    orc_trace("13 coset evaluations", &tr);
    fe *xs = (fe *)malloc(sizeof(fe) * N4);
    fr_powers(xs, N4, &pk->d1.gen, &u, nthreads);
    fe *xn_inv = (fe *)malloc(sizeof(fe) * rho);  /* x^n takes rho values on the coset */
    for (size_t i = 0; i < rho; i++) { fe t; fr_pow_u64(&t, &xs[i], (uint64_t)n); fr_sub(&t, &t, &FR.one); fld_inv(&FR, &xn_inv[i], &t); }
    fe bu, buu; fr_mul(&bu, &beta, &u); fr_mul(&buu, &beta, &uu);
    fe *t_ = (fe *)malloc(sizeof(fe) * N4);
#pragma omp parallel for schedule(static) num_threads(T) if (N4 >= 4096)
    for (size_t i = 0; i < N4; i++) {
        const fe lv = e_l[i], rv = e_r[i], ov = e_o[i], zv = e_z[i], zs = e_z[(i + rho) % N4];  /* z(omega x): rho steps of W */
        fe ic, t, a, bb, f, one;
        fr_mul(&ic, &e_ql[i], &lv);
        fr_mul(&t, &e_qr[i], &rv); fr_add(&ic, &ic, &t);
        fr_mul(&t, &e_qm[i], &lv); fr_mul(&t, &t, &rv); fr_add(&ic, &ic, &t);
        fr_mul(&t, &e_qo[i], &ov); fr_add(&ic, &ic, &t);
        fr_add(&ic, &ic, &e_qk[i]);
        fr_mul(&a, &beta, &xs[i]); fr_add(&a, &a, &lv); fr_add(&a, &a, &gamma);
        fr_mul(&f, &bu, &xs[i]); fr_add(&f, &f, &rv); fr_add(&f, &f, &gamma); fr_mul(&a, &a, &f);
        fr_mul(&f, &buu, &xs[i]); fr_add(&f, &f, &ov); fr_add(&f, &f, &gamma); fr_mul(&a, &a, &f);
        fr_mul(&a, &a, &zv);
        fr_mul(&bb, &beta, &e_s1[i]); fr_add(&bb, &bb, &lv); fr_add(&bb, &bb, &gamma);
        fr_mul(&f, &beta, &e_s2[i]); fr_add(&f, &f, &rv); fr_add(&f, &f, &gamma); fr_mul(&bb, &bb, &f);
        fr_mul(&f, &beta, &e_s3[i]); fr_add(&f, &f, &ov); fr_add(&f, &f, &gamma); fr_mul(&bb, &bb, &f);
        fr_mul(&bb, &bb, &zs);
        fr_sub(&one, &zv, &FR.one); fr_mul(&one, &one, &e_l1[i]);
        fr_mul(&t, &one, &alpha); fr_sub(&f, &bb, &a); fr_add(&t, &t, &f); fr_mul(&t, &t, &alpha); fr_add(&t, &t, &ic);
        fr_mul(&t_[i], &t, &xn_inv[i % rho]);
    }
    orc_trace("quotient loop", &tr);
    fe *evs[] = {e_l, e_r, e_o, e_z, e_qk, e_ql, e_qr, e_qm, e_qo, e_s1, e_s2, e_s3, e_l1, xs, xn_inv};
    for (size_t k = 0; k < sizeof evs / sizeof *evs; k++) free(evs[k]);
    orc_fr_bit_reverse((uint64_t *)t_, pk->log_n4);
    domain_fft(&pk->d1, t_, 1, ORC_DIT, 1, nthreads);  /* coset interpolation: bit-reversed in -> natural canonical out */
    fe *h = t_;
    int bad = 0;
    for (size_t i = 3 * (n + 2); i < N4; i++) bad |= !fe_is_zero(&h[i]);
    if (bad) { free(h); for (int k = 0; k < 3; k++) free(b[k]); free(bz); return -2; }
    orc_trace("coset interpolation of t", &tr);
    fe *h1 = h, *h2 = h + (n + 2), *h3 = h + 2 * (n + 2);
    g1_aff hd[3];
    kzg_commit(&hd[0], pk->srs, h1, n + 2, nthreads); kzg_commit(&hd[1], pk->srs, h2, n + 2, nthreads); kzg_commit(&hd[2], pk->srs, h3, n + 2, nthreads);
    orc_trace("3 commits of h", &tr);
    fs_begin(&fs, "zeta", da);
    for (int k = 0; k < 3; k++) fs_point(&fs, &hd[k]);
    fs_end(&fs, dz, &zeta);

    /* openings */
    fe lz, rz, oz, zu, zeta_sh;
    poly_eval_c(&lz, b[0], n + 2, &zeta, nthreads); poly_eval_c(&rz, b[1], n + 2, &zeta, nthreads); poly_eval_c(&oz, b[2], n + 2, &zeta, nthreads);
    fr_mul(&zeta_sh, &zeta, &pk->d0.gen);
    poly_eval_c(&zu, bz, n + 3, &zeta_sh, nthreads);
    g1_aff z_open_h;
    {
        fe *q = (fe *)malloc(sizeof(fe) * (n + 3));
        memcpy(q, bz, (n + 3) * sizeof(fe));
        divide_by_x_minus_a_c(q, n + 3, &zu, &zeta_sh);
        kzg_commit(&z_open_h, pk->srs, q + 1, n + 2, nthreads);
        free(q);
    }
    orc_trace("openings at zeta + [z/(X-wz)]", &tr);
    /* linearised polynomial (prove.go computeLinearizedPolynomial) */
    fe s1z, s2z, c_s3, c_z, lag1, rl, t, f;
    poly_eval_c(&s1z, pk->poly[PS1], n, &zeta, nthreads); poly_eval_c(&s2z, pk->poly[PS2], n, &zeta, nthreads);
    fr_mul(&t, &beta, &s1z); fr_add(&t, &t, &lz); fr_add(&t, &t, &gamma);
    fr_mul(&f, &beta, &s2z); fr_add(&f, &f, &rz); fr_add(&f, &f, &gamma);
    fr_mul(&c_s3, &t, &f); fr_mul(&c_s3, &c_s3, &zu); fr_mul(&c_s3, &c_s3, &beta);
    fr_mul(&t, &beta, &zeta); fr_add(&t, &t, &lz); fr_add(&t, &t, &gamma);
    fr_mul(&f, &bu, &zeta); fr_add(&f, &f, &rz); fr_add(&f, &f, &gamma); fr_mul(&t, &t, &f);
    fr_mul(&f, &buu, &zeta); fr_add(&f, &f, &oz); fr_add(&f, &f, &gamma); fr_mul(&t, &t, &f);
    memset(&c_z, 0, sizeof c_z); fr_sub(&c_z, &c_z, &t);
    fr_pow_u64(&t, &zeta, (uint64_t)n); fr_sub(&t, &t, &FR.one);
    fr_sub(&f, &zeta, &FR.one); fld_inv(&FR, &f, &f);
    fr_mul(&lag1, &t, &f); fr_mul(&lag1, &lag1, &alpha); fr_mul(&lag1, &lag1, &alpha); fr_mul(&lag1, &lag1, &pk->d0.card_inv);
    fr_mul(&rl, &lz, &rz);
    fe *lin = (fe *)malloc(sizeof(fe) * (n + 3));
#pragma omp parallel for schedule(static) num_threads(T) if (n >= 4096)
    for (size_t i = 0; i < n + 3; i++) {
        fe v, w;
        fr_mul(&v, &bz[i], &c_z);
        if (i < n) { fr_mul(&w, &pk->poly[PS3][i], &c_s3); fr_add(&v, &v, &w); }
        fr_mul(&v, &v, &alpha);
        if (i < n) {
            fr_mul(&w, &pk->poly[PQM][i], &rl); fr_add(&v, &v, &w);
            fr_mul(&w, &pk->poly[PQL][i], &lz); fr_add(&v, &v, &w);
            fr_mul(&w, &pk->poly[PQR][i], &rz); fr_add(&v, &v, &w);
            fr_mul(&w, &pk->poly[PQO][i], &oz); fr_add(&v, &v, &w);
            fr_add(&v, &v, &pk->poly[PCQK][i]);
        }
        fr_mul(&w, &bz[i], &lag1);
        fr_add(&lin[i], &v, &w);
    }
    g1_aff lin_digest; kzg_commit(&lin_digest, pk->srs, lin, n + 3, nthreads);

    orc_trace("linearised polynomial + commit", &tr);
    /* folded quotient h1 + zeta^(n+2) h2 + zeta^(2(n+2)) h3 and its digest */
    fe zp; fr_pow_u64(&zp, &zeta, (uint64_t)(n + 2));
    fe *folded_h = (fe *)malloc(sizeof(fe) * (n + 2));
#pragma omp parallel for schedule(static) num_threads(T) if (n >= 4096)
    for (size_t i = 0; i < n + 2; i++) { fe v; fr_mul(&v, &h3[i], &zp); fr_add(&v, &v, &h2[i]); fr_mul(&v, &v, &zp); fr_add(&folded_h[i], &v, &h1[i]); }
    g1_aff folded_h_digest;
    {
        fe zpc; fld_from_mont(&FR, &zpc, &zp);
        g1_xyzz acc, t2; g1_aff ta;
        g1_scalar_mul(&acc, &hd[2], zpc.l); g1_madd(&acc, &hd[1], 0);
        g1_to_aff(&ta, &acc);
        g1_scalar_mul(&t2, &ta, zpc.l); g1_madd(&t2, &hd[0], 0);
        g1_to_aff(&folded_h_digest, &t2);
    }
    free(h);

    /* kzg.BatchOpenSinglePoint of (foldedH, linPol, l, r, o, s1, s2) at zeta */
    const fe *polys[7] = {folded_h, lin, b[0], b[1], b[2], pk->poly[PS1], pk->poly[PS2]};
    const size_t plen[7] = {n + 2, n + 3, n + 2, n + 2, n + 2, n, n};
    const g1_aff *digests[7] = {&folded_h_digest, &lin_digest, &lro[0], &lro[1], &lro[2], &pk->vk[0], &pk->vk[1]};
    fe claimed[7];
    for (int k = 0; k < 7; k++) poly_eval_c(&claimed[k], polys[k], plen[k], &zeta, nthreads);
    uint8_t dk[32]; fe kg;  /* kzg.deriveGamma: a one-challenge transcript bound to the point, the digests and the claimed values */
    fs_begin(&fs, "gamma", NULL); fs_fr(&fs, &zeta);
    for (int k = 0; k < 7; k++) fs_point(&fs, digests[k]);
    for (int k = 0; k < 7; k++) fs_fr(&fs, &claimed[k]);
    fs_end(&fs, dk, &kg);
    fe *folded = fr_zeros(n + 3);
    fe acc = FR.one;
    for (int k = 0; k < 7; k++) {
        const fe *p = polys[k]; const fe ac = acc;
#pragma omp parallel for schedule(static) num_threads(T) if (n >= 4096)
        for (size_t j = 0; j < plen[k]; j++) { fe w; fr_mul(&w, &p[j], &ac); fr_add(&folded[j], &folded[j], &w); }
        fr_mul(&acc, &acc, &kg);
    }
    fe folded_eval; memset(&folded_eval, 0, sizeof folded_eval);
    for (int k = 6; k >= 0; k--) { fr_mul(&folded_eval, &folded_eval, &kg); fr_add(&folded_eval, &folded_eval, &claimed[k]); }
    divide_by_x_minus_a_c(folded, n + 3, &folded_eval, &zeta);
    g1_aff batch_h; kzg_commit(&batch_h, pk->srs, folded + 1, n + 2, nthreads);
    free(folded); free(folded_h); free(lin); free(bz);
    for (int k = 0; k < 3; k++) free(b[k]);

    orc_trace("batch opening + commit", &tr);
    /* Proof.WriteTo (marshal.go): LRO[0..2], Z, H[0..2] compressed (7 x 32 B); BatchedProof = H (32 B) | u32 BE count | claimed values (7 x 32 B BE);
     * ZShiftedOpening = H (32 B) | claimed value (32 B).  548 bytes. */
    uint8_t *o = proof_out;
    const g1_aff *seven[7] = {&lro[0], &lro[1], &lro[2], &z_digest, &hd[0], &hd[1], &hd[2]};
    for (int k = 0; k < 7; k++, o += 32) orc_g1_compress((const uint64_t *)seven[k], o);
    orc_g1_compress((const uint64_t *)&batch_h, o); o += 32;
    o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 7; o += 4;
    for (int k = 0; k < 7; k++, o += 32) fr_be_bytes(o, &claimed[k]);
    orc_g1_compress((const uint64_t *)&z_open_h, o); o += 32;
    fr_be_bytes(o, &zu);
    if (challenges_out) {
        memcpy(challenges_out, &gamma, 32); memcpy(challenges_out + 4, &beta, 32); memcpy(challenges_out + 8, &alpha, 32);
        memcpy(challenges_out + 12, &zeta, 32); memcpy(challenges_out + 16, &kg, 32);
    }
    return 0;
}
