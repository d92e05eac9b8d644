/* CPU oracle for the BN254 proving hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See bn254_oracle.h.
 *
 * Restates (plain C, 4x64-bit Montgomery limbs, unsigned __int128) the algorithms of
 *   gnark-crypto v0.9.1  ecc/bn254 {fp, fr, G1/G2 MultiExp}, ecc/bn254/fr/fft {Domain.FFT, FFTInverse, BitReverse}
 *   gnark v0.8.0         internal/backend/bn254/groth16 {computeH, Prove}
 * pinned at /root/reference/gnark_backend_ffi/go.mod:5,23 and reached from the reference at
 * /root/reference/gnark_backend_ffi/main.go:121,131 and backend/plonk/plonk.go:21,67.  Their source is not
 * vendored under /root/reference, so each function below cites the upstream symbol it follows [UPSTREAM-RECALL]
 * and is pinned against oracle/bn254_ref.py (pure-Python big ints) + tests/golden/.   PARITY UNPINNED by the
 * reference's own tests.
 */
#include "bn254_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

/* ------------------------------------------------------------------ moduli */
typedef struct {
    fe p;        /* modulus */
    fe one;      /* R mod p */
    fe r2;       /* R^2 mod p */
    uint64_t ninv; /* -p^-1 mod 2^64 (gnark "qInvNeg") */
} field_t;

static field_t FR = {{{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}}, {{0}}, {{0}}, 0};
static field_t FP = {{{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}}, {{0}}, {{0}}, 0};
static int g_init_done = 0;

static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
    return 1;
}
static inline uint64_t fe_add_raw(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t fe_sub_raw(fe *r, const fe *a, const fe *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline void fld_add(const field_t *F, fe *r, const fe *a, const fe *b) {
    fe t; fe_add_raw(&t, a, b);           /* p < 2^254 so no carry out */
    if (fe_geq(&t, &F->p)) fe_sub_raw(&t, &t, &F->p);
    *r = t;
}
static inline void fld_sub(const field_t *F, fe *r, const fe *a, const fe *b) {
    fe t;
    if (fe_sub_raw(&t, a, b)) fe_add_raw(&t, &t, &F->p);
    *r = t;
}
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof *a) == 0; }
static inline void fld_neg(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    fe t; fe_sub_raw(&t, &F->p, a); *r = t;
}

/* Montgomery multiplication, CIOS (gnark-crypto fp.Element.Mul / fr.Element.Mul: "no-carry" CIOS with qInvNeg) */
static inline void fld_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->ninv;
        c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &F->p)) fe_sub_raw(&o, &o, &F->p);
    *r = o;
}
static void fld_pow(const field_t *F, fe *r, const fe *a, const fe *e) {
    fe acc = F->one, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) fld_mul(F, &acc, &acc, &base);
        fld_mul(F, &base, &base, &base);
    }
    *r = acc;
}
static void fld_inv(const field_t *F, fe *r, const fe *a) { /* Fermat: a^(p-2); inv(0) = 0 like gnark */
    fe e = F->p, two = {{2, 0, 0, 0}};
    fe_sub_raw(&e, &e, &two);
    fld_pow(F, r, a, &e);
}
static inline void fld_to_mont(const field_t *F, fe *r, const fe *a) { fld_mul(F, r, a, &F->r2); }
static inline void fld_from_mont(const field_t *F, fe *r, const fe *a) { fe one = {{1, 0, 0, 0}}; fld_mul(F, r, a, &one); }
static void fld_set_u64(const field_t *F, fe *r, uint64_t v) { fe t = {{v, 0, 0, 0}}; fld_to_mont(F, r, &t); }

static void field_init(field_t *F) {
    /* ninv by Newton iteration; R mod p and R^2 mod p by repeated doubling */
    uint64_t x = 1;
    for (int i = 0; i < 7; i++) x *= 2 - F->p.l[0] * x;
    F->ninv = (uint64_t)0 - x;
    fe t = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        fld_add(F, &t, &t, &t);
        if (i == 255) F->one = t;
    }
    F->r2 = t;
}

/* ------------------------------------------------------------------ Fp / Fp2 op names for curve_tmpl.h */
static fe g1_curve_b;                 /* 3 */
typedef struct { fe a0, a1; } fe2;    /* gnark-crypto E2{A0,A1}, u^2 = -1 */
static fe2 g2_curve_b;                /* 3/(9+u) */

static inline void fp_add(fe *r, const fe *a, const fe *b) { fld_add(&FP, r, a, b); }
static inline void fp_sub(fe *r, const fe *a, const fe *b) { fld_sub(&FP, r, a, b); }
static inline void fp_mul(fe *r, const fe *a, const fe *b) { fld_mul(&FP, r, a, b); }
static inline void fp_sqr(fe *r, const fe *a) { fld_mul(&FP, r, a, a); }
static inline void fp_neg(fe *r, const fe *a) { fld_neg(&FP, r, a); }
static inline void fp_inv(fe *r, const fe *a) { fld_inv(&FP, r, a); }
static inline int fp_is_zero(const fe *a) { return fe_is_zero(a); }
static inline int fp_eq(const fe *a, const fe *b) { return fe_eq(a, b); }
static inline void fp_set_one(fe *r) { *r = FP.one; }

static inline void fp2_add(fe2 *r, const fe2 *a, const fe2 *b) { fp_add(&r->a0, &a->a0, &b->a0); fp_add(&r->a1, &a->a1, &b->a1); }
static inline void fp2_sub(fe2 *r, const fe2 *a, const fe2 *b) { fp_sub(&r->a0, &a->a0, &b->a0); fp_sub(&r->a1, &a->a1, &b->a1); }
static inline void fp2_neg(fe2 *r, const fe2 *a) { fp_neg(&r->a0, &a->a0); fp_neg(&r->a1, &a->a1); }
static inline void fp2_mul(fe2 *r, const fe2 *a, const fe2 *b) {
    fe t0, t1, t2, t3;
    fp_mul(&t0, &a->a0, &b->a0);
    fp_mul(&t1, &a->a1, &b->a1);
    fp_mul(&t2, &a->a0, &b->a1);
    fp_mul(&t3, &a->a1, &b->a0);
    fp_sub(&r->a0, &t0, &t1);
    fp_add(&r->a1, &t2, &t3);
}
static inline void fp2_sqr(fe2 *r, const fe2 *a) { fp2_mul(r, a, a); }
static inline void fp2_inv(fe2 *r, const fe2 *a) {
    fe n, t, d;
    fp_sqr(&n, &a->a0); fp_sqr(&t, &a->a1); fp_add(&n, &n, &t);
    fp_inv(&d, &n);
    fp_mul(&r->a0, &a->a0, &d);
    fp_mul(&t, &a->a1, &d); fp_neg(&r->a1, &t);
}
static inline int fp2_is_zero(const fe2 *a) { return fe_is_zero(&a->a0) && fe_is_zero(&a->a1); }
static inline int fp2_eq(const fe2 *a, const fe2 *b) { return fe_eq(&a->a0, &b->a0) && fe_eq(&a->a1, &b->a1); }
static inline void fp2_set_one(fe2 *r) { r->a0 = FP.one; memset(&r->a1, 0, sizeof(fe)); }

/* ------------------------------------------------------------------ MSM helpers shared by G1/G2 */
/* gnark-crypto MultiExp "bestC": argmin over c of (fr.Bits+1)*(n + 2^c)/c   [UPSTREAM-RECALL; c range 2..16] */
static int msm_best_c(size_t n) {
    int best = 2; double bc = 1e300;
    for (int c = 2; c <= 16; c++) {
        double cost = 255.0 * ((double)n + (double)((size_t)1 << c)) / (double)c;
        if (cost < bc) { bc = cost; best = c; }
    }
    return best;
}
/* gnark-crypto partitionScalars: signed digits, "if digit > 2^(c-1) { digit -= 2^c; carry = 1 }" */
static void msm_partition_scalars(int32_t *digits, const uint64_t *scalars, size_t n, int c, int nwin) {
    const int64_t half = (int64_t)1 << (c - 1);
    const uint64_t mask = ((uint64_t)1 << c) - 1;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const uint64_t *s = scalars + 4 * i;
        int64_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            int bit = w * c, limb = bit >> 6, off = bit & 63;
            uint64_t v = 0;
            if (limb < 4) {
                v = s[limb] >> off;
                if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            }
            int64_t d = (int64_t)(v & mask) + carry;
            carry = 0;
            if (d > half) { d -= (int64_t)1 << c; carry = 1; }
            digits[(size_t)w * n + i] = (int32_t)d;
        }
    }
}

#define FE fe
#define F(x) fp_##x
#define G(x) g1_##x
#include "curve_tmpl.h"
#undef FE
#undef F
#undef G
#define FE fe2
#define F(x) fp2_##x
#define G(x) g2_##x
#include "curve_tmpl.h"
#undef FE
#undef F
#undef G

static g1_aff G1_GEN;
static g2_aff G2_GEN;

static void set_dec_limbs(fe *r, uint64_t l0, uint64_t l1, uint64_t l2, uint64_t l3) {
    fe t = {{l0, l1, l2, l3}};
    fld_to_mont(&FP, r, &t);
}

static int effective_threads(void);
static void orc_init(void) {
    if (g_init_done) return;
    (void)effective_threads();
#pragma omp critical(orc_init_lock)
    {
        if (!g_init_done) {
            field_init(&FR); field_init(&FP);
            fld_set_u64(&FP, &g1_curve_b, 3);
            /* b' = 3/(9+u) */
            fe2 nine_u, three;
            fld_set_u64(&FP, &nine_u.a0, 9); nine_u.a1 = FP.one;
            fld_set_u64(&FP, &three.a0, 3); memset(&three.a1, 0, sizeof(fe));
            fe2 iv; fp2_inv(&iv, &nine_u); fp2_mul(&g2_curve_b, &three, &iv);
            fld_set_u64(&FP, &G1_GEN.x, 1); fld_set_u64(&FP, &G1_GEN.y, 2);
            /* G2 generator, canonical limbs LE */
            set_dec_limbs(&G2_GEN.x.a0, 0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL);
            set_dec_limbs(&G2_GEN.x.a1, 0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL);
            set_dec_limbs(&G2_GEN.y.a0, 0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL);
            set_dec_limbs(&G2_GEN.y.a1, 0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL);
            g_init_done = 1;
        }
    }
}

/* The threads worth starting: the CPUs this process may actually USE -- min(OpenMP's default, the cgroup's CPU quota).  The GPU boxes show 256 logical CPUs
 * under a quota of 16 (cpu.max "1600000 100000"): 128 OpenMP threads time-sliced onto 16 CPUs spend their life in barriers (a 2^12-gate PLONK proof took 42 s
 * there against 0.4 s on 8 unthrottled cores).  Read once; OMP_NUM_THREADS still wins when it asks for fewer. */
static int cgroup_cpu_limit(void) {
    long quota = -1, period = 100000;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");                 /* cgroup v2: "<quota|max> <period>" */
    if (f) {
        char q[32] = {0};
        if (fscanf(f, "%31s %ld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {  /* cgroup v1 */
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) { if (fscanf(f, "%ld", &period) != 1) period = 100000; fclose(f); }
    }
    if (quota <= 0 || period <= 0) return 0;
    long n = (quota + period - 1) / period;
    return n < 1 ? 1 : (int)n;
}
static int g_threads = 0;
static int effective_threads(void) {
    if (g_threads) return g_threads;
    int n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads();
#endif
    const int lim = cgroup_cpu_limit();
    if (lim > 0 && lim < n) n = lim;
    g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);  /* the parallel regions without a num_threads clause */
#endif
    return g_threads;
}
int orc_max_threads(void) { return effective_threads(); }
/* logical CPUs the box shows (what a naive count would have used); for the bench line's record beside `cores` */
int orc_host_cpus(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ field API */
static const field_t *which_field(int which) { orc_init(); return which ? &FP : &FR; }
void orc_fe_mul(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fld_mul(which_field(which), (fe *)out, (const fe *)a, (const fe *)b); }
void orc_fe_add(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fld_add(which_field(which), (fe *)out, (const fe *)a, (const fe *)b); }
void orc_fe_sub(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { fld_sub(which_field(which), (fe *)out, (const fe *)a, (const fe *)b); }
void orc_fe_inv(int which, const uint64_t a[4], uint64_t out[4]) { fld_inv(which_field(which), (fe *)out, (const fe *)a); }
void orc_fe_to_mont(int which, const uint64_t a[4], uint64_t out[4]) { fld_to_mont(which_field(which), (fe *)out, (const fe *)a); }
void orc_fe_from_mont(int which, const uint64_t a[4], uint64_t out[4]) { fld_from_mont(which_field(which), (fe *)out, (const fe *)a); }

/* ------------------------------------------------------------------ PRNG */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* 256-bit value mod r: 2^256 < 6r, so at most 5 conditional subtractions */
static void fr_reduce256(fe *x) { orc_init(); while (fe_geq(x, &FR.p)) fe_sub_raw(x, x, &FR.p); }

void orc_rand_fr(uint64_t seed, size_t n, uint64_t *out, int mont) {
    orc_init();
    uint64_t s = seed;
    for (size_t i = 0; i < n; i++) {
        fe x; for (int k = 0; k < 4; k++) x.l[k] = splitmix64(&s);
        fr_reduce256(&x);
        if (mont) fld_to_mont(&FR, &x, &x);
        memcpy(out + 4 * i, &x, 32);
    }
}
void orc_rand_fr_witness(uint64_t seed, size_t n, uint64_t *out, int mont) {
    orc_init();
    uint64_t s = seed;
    for (size_t i = 0; i < n; i++) {
        fe x; for (int k = 0; k < 4; k++) x.l[k] = splitmix64(&s);
        uint64_t sel = splitmix64(&s) & 3;
        if (sel < 2) { x.l[0] &= 1; x.l[1] = x.l[2] = x.l[3] = 0; }
        else if (sel == 2) { x.l[0] &= 0xffffffffULL; x.l[1] = x.l[2] = x.l[3] = 0; }
        fr_reduce256(&x);
        if (mont) fld_to_mont(&FR, &x, &x);
        memcpy(out + 4 * i, &x, 32);
    }
}

/* ------------------------------------------------------------------ points */
static void scalars_canonical(fe *dst, const uint64_t *src, size_t n, int mont) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fe x; memcpy(&x, src + 4 * i, 32);
        if (mont) fld_from_mont(&FR, &x, &x);
        dst[i] = x;
    }
}

void orc_g1_gen_points(uint64_t seed, size_t n, uint64_t *out, int nthreads) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    orc_rand_fr(seed, n, (uint64_t *)k, 0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (size_t i = 0; i < n; i++) {
        g1_xyzz t; g1_aff a;
        g1_scalar_mul(&t, &G1_GEN, k[i].l);
        g1_to_aff(&a, &t);
        memcpy(out + 8 * i, &a, 64);
    }
    free(k);
}
void orc_g2_gen_points(uint64_t seed, size_t n, uint64_t *out, int nthreads) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    orc_rand_fr(seed, n, (uint64_t *)k, 0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (size_t i = 0; i < n; i++) {
        g2_xyzz t; g2_aff a;
        g2_scalar_mul(&t, &G2_GEN, k[i].l);
        g2_to_aff(&a, &t);
        memcpy(out + 16 * i, &a, 128);
    }
    free(k);
}
int orc_g1_on_curve(const uint64_t p[8]) { orc_init(); return g1_on_curve((const g1_aff *)p); }
int orc_g2_on_curve(const uint64_t p[16]) { orc_init(); return g2_on_curve((const g2_aff *)p); }

int orc_g1_msm(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, int c, int nthreads, uint64_t out[8]) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    scalars_canonical(k, scalars, n, scalars_mont);
    int rc = g1_msm((g1_aff *)out, (const g1_aff *)points, (const uint64_t *)k, n, c, nthreads);
    free(k);
    return rc;
}
int orc_g2_msm(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, int c, int nthreads, uint64_t out[16]) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    scalars_canonical(k, scalars, n, scalars_mont);
    int rc = g2_msm((g2_aff *)out, (const g2_aff *)points, (const uint64_t *)k, n, c, nthreads);
    free(k);
    return rc;
}
int orc_g1_msm_naive(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out[8]) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    scalars_canonical(k, scalars, n, scalars_mont);
    g1_msm_naive((g1_aff *)out, (const g1_aff *)points, (const uint64_t *)k, n);
    free(k);
    return 0;
}
int orc_g2_msm_naive(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out[16]) {
    orc_init();
    fe *k = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    scalars_canonical(k, scalars, n, scalars_mont);
    g2_msm_naive((g2_aff *)out, (const g2_aff *)points, (const uint64_t *)k, n);
    free(k);
    return 0;
}
void orc_g1_add(const uint64_t a[8], const uint64_t b[8], uint64_t out[8]) {
    orc_init();
    g1_xyzz t; g1_from_aff(&t, (const g1_aff *)a); g1_madd(&t, (const g1_aff *)b, 0);
    g1_aff o; g1_to_aff(&o, &t); memcpy(out, &o, 64);
}
void orc_g2_add(const uint64_t a[16], const uint64_t b[16], uint64_t out[16]) {
    orc_init();
    g2_xyzz t; g2_from_aff(&t, (const g2_aff *)a); g2_madd(&t, (const g2_aff *)b, 0);
    g2_aff o; g2_to_aff(&o, &t); memcpy(out, &o, 128);
}
void orc_g1_mul(const uint64_t a[8], const uint64_t k[4], int k_mont, uint64_t out[8]) {
    orc_init();
    fe kk; memcpy(&kk, k, 32); if (k_mont) fld_from_mont(&FR, &kk, &kk);
    g1_xyzz t; g1_scalar_mul(&t, (const g1_aff *)a, kk.l);
    g1_aff o; g1_to_aff(&o, &t); memcpy(out, &o, 64);
}
void orc_g2_mul(const uint64_t a[16], const uint64_t k[4], int k_mont, uint64_t out[16]) {
    orc_init();
    fe kk; memcpy(&kk, k, 32); if (k_mont) fld_from_mont(&FR, &kk, &kk);
    g2_xyzz t; g2_scalar_mul(&t, (const g2_aff *)a, kk.l);
    g2_aff o; g2_to_aff(&o, &t); memcpy(out, &o, 128);
}

/* G1Affine.Bytes(): 32 B big-endian X, flags in the two top bits of byte 0:
 * mCompressedSmallest 0b10<<6, mCompressedLargest 0b11<<6, mCompressedInfinity 0b01<<6  [UPSTREAM-RECALL] */
static void fe_to_be(uint8_t out[32], const fe *canon) {
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(canon->l[i] >> (8 * b));
}
static int fp_lex_largest(const fe *canon) { /* canon > (q-1)/2 */
    fe h = FP.p; /* (q-1)/2 : q odd -> shift right */
    for (int i = 0; i < 4; i++) h.l[i] = (h.l[i] >> 1) | (i < 3 ? h.l[i + 1] << 63 : 0);
    return !fe_geq(&h, canon);
}
void orc_g1_compress(const uint64_t a[8], uint8_t out[32]) {
    orc_init();
    const g1_aff *p = (const g1_aff *)a;
    if (g1_aff_is_inf(p)) { memset(out, 0, 32); out[0] = 0x40; return; }
    fe x, y; fld_from_mont(&FP, &x, &p->x); fld_from_mont(&FP, &y, &p->y);
    fe_to_be(out, &x);
    out[0] |= fp_lex_largest(&y) ? 0xC0 : 0x80;
}
void orc_g2_compress(const uint64_t a[16], uint8_t out[64]) {
    orc_init();
    const g2_aff *p = (const g2_aff *)a;
    if (g2_aff_is_inf(p)) { memset(out, 0, 64); out[0] = 0x40; return; }
    fe x0, x1, y0, y1;
    fld_from_mont(&FP, &x0, &p->x.a0); fld_from_mont(&FP, &x1, &p->x.a1);
    fld_from_mont(&FP, &y0, &p->y.a0); fld_from_mont(&FP, &y1, &p->y.a1);
    fe_to_be(out, &x1); fe_to_be(out + 32, &x0);
    int largest = fe_is_zero(&y1) ? fp_lex_largest(&y0) : fp_lex_largest(&y1);
    out[0] |= largest ? 0xC0 : 0x80;
}

/* ------------------------------------------------------------------ fft.Domain */
static const uint64_t ROOT_2_28[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};

typedef struct {
    unsigned logn; size_t n;
    fe gen, gen_inv, card_inv;
    fe *tw;      /* gen^i, i < n/2 */
    fe *tw_inv;  /* gen_inv^i */
} domain_t;

static void fr_mul(fe *r, const fe *a, const fe *b) { fld_mul(&FR, r, a, b); }
static void fr_pow_u64(fe *r, const fe *a, uint64_t e) { fe ee = {{e, 0, 0, 0}}; fld_pow(&FR, r, a, &ee); }
/* out[i] = first * base^i, i < n -- chunked: one exponentiation per chunk, then a running product (the same values as the sequential product; only the
 * association differs, and field multiplication is exact) */
static void fr_powers(fe *out, size_t n, const fe *base, const fe *first, int nthreads) {
    const size_t chunk = 1 << 14, nch = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (n >= 4096)
    for (size_t c = 0; c < nch; c++) {
        size_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        fe p; fr_pow_u64(&p, base, (uint64_t)lo);
        if (first) fr_mul(&p, &p, first);
        for (size_t i = lo; i < hi; i++) { out[i] = p; fr_mul(&p, &p, base); }
    }
}

static void domain_init(domain_t *d, unsigned logn) {
    orc_init();
    d->logn = logn; d->n = (size_t)1 << logn;
    fe w; memcpy(&w, ROOT_2_28, 32); fld_to_mont(&FR, &w, &w);
    for (unsigned i = logn; i < 28; i++) fr_mul(&w, &w, &w);   /* gen = root^(2^(28-logn)) */
    d->gen = w; fld_inv(&FR, &d->gen_inv, &w);
    fe nn; fld_set_u64(&FR, &nn, (uint64_t)d->n); fld_inv(&FR, &d->card_inv, &nn);
    size_t h = d->n > 1 ? d->n / 2 : 1;
    d->tw = (fe *)malloc(sizeof(fe) * h); d->tw_inv = (fe *)malloc(sizeof(fe) * h);
    fr_powers(d->tw, h, &d->gen, NULL, orc_max_threads());
    fr_powers(d->tw_inv, h, &d->gen_inv, NULL, orc_max_threads());
}
static void domain_free(domain_t *d) { free(d->tw); free(d->tw_inv); }

static size_t bitrev_sz(size_t i, unsigned logn) {
    size_t r = 0;
    for (unsigned b = 0; b < logn; b++) r |= ((i >> b) & 1) << (logn - 1 - b);
    return r;
}
void orc_fr_bit_reverse(uint64_t *a_, unsigned logn) {
    fe *a = (fe *)a_; size_t n = (size_t)1 << logn;
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; i++) { size_t j = bitrev_sz(i, logn); if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
}

/* difFFT: natural in -> bit-reversed out; butterfly (x, y) -> (x + y, (x - y) * w)   (gnark-crypto fft.go difFFT)
 * ditFFT: bit-reversed in -> natural out; butterfly (x, y) -> (x + y*w, x - y*w)     (gnark-crypto fft.go ditFFT)
 * Every butterfly is the textbook one with the textbook twiddle; what is organised for the machine is only the ORDER in which they run (field arithmetic is
 * exact, so the order cannot change a word): below 2^17 points one parallel loop per stage; above, the stages whose span exceeds a 1 MB block run five at a time
 * on tiles of 32 rows x 64 consecutive elements held in a thread's cache, and the stages inside a block run block by block -- 2 + ceil((log n - 15) / 5) passes
 * over memory instead of log n (at 2^26 on 128 cores the one-pass-per-stage form was memory-bound at 4.3 s per transform). */
#define NTT_LB 15u   /* stages with span <= 2^NTT_LB elements (1 MB) run inside contiguous blocks */
#define NTT_G 5u     /* stages per tiled group */
#define NTT_C 64u    /* consecutive elements per tile row (2 KB) */
static inline void bfly_dif(fe *x, fe *y, const fe *w) { fe t, u; fld_add(&FR, &u, x, y); fld_sub(&FR, &t, x, y); fr_mul(y, &t, w); *x = u; }
static inline void bfly_dit(fe *x, fe *y, const fe *w) { fe t, u; fr_mul(&t, y, w); fld_add(&FR, &u, x, &t); fld_sub(&FR, y, x, &t); *x = u; }
/* one tiled group: `g` stages over rows `mrow` apart; dif: first in-buffer stage pairs rows g-1 bits apart and halves, dit: starts at 1 and doubles.
 * tw_shift0: twiddle index of in-buffer stage u = j << (dif ? tw_shift0 + u : tw_shift0 - u), j = k * mrow + column */
static void ntt_group(fe *a, const fe *tw, size_t n, size_t mrow, unsigned g, unsigned tw_shift0, int dif, int nthreads) {
    const size_t rows = (size_t)1 << g, big = rows * mrow, ntile_row = mrow / NTT_C, ntiles = (n / big) * ntile_row;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (size_t tile = 0; tile < ntiles; tile++) {
        fe buf[(1u << NTT_G) * NTT_C];
        const size_t c0 = (tile % ntile_row) * NTT_C;
        fe *base = a + (tile / ntile_row) * big + c0;
        for (size_t r = 0; r < rows; r++) memcpy(buf + r * NTT_C, base + r * mrow, NTT_C * sizeof(fe));
        for (unsigned u = 0; u < g; u++) {
            const size_t h = dif ? (size_t)1 << (g - 1 - u) : (size_t)1 << u;
            const unsigned sh = dif ? tw_shift0 + u : tw_shift0 - u;
            for (size_t q = 0; q < rows; q += 2 * h)
                for (size_t k = 0; k < h; k++) {
                    fe *x = buf + (q + k) * NTT_C, *y = x + h * NTT_C;
                    const size_t j0 = k * mrow + c0;
                    if (dif) for (size_t c = 0; c < NTT_C; c++) bfly_dif(&x[c], &y[c], &tw[(j0 + c) << sh]);
                    else for (size_t c = 0; c < NTT_C; c++) bfly_dit(&x[c], &y[c], &tw[(j0 + c) << sh]);
                }
        }
        for (size_t r = 0; r < rows; r++) memcpy(base + r * mrow, buf + r * NTT_C, NTT_C * sizeof(fe));
    }
}
static void dif_inplace(fe *a, const fe *tw, unsigned logn, int nthreads) {
    const size_t n = (size_t)1 << logn;
    unsigned s = 0;  /* next stage; its half-span is n >> (s + 1), its twiddle index j << s */
    if (logn > NTT_LB + 1) {
        while (logn - s > NTT_LB) {
            unsigned g = logn - s - NTT_LB; if (g > NTT_G) g = NTT_G;
            ntt_group(a, tw, n, n >> (s + g), g, s, 1, nthreads);
            s += g;
        }
        const size_t blk = n >> s;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
        for (size_t b = 0; b < n / blk; b++) {
            fe *p = a + b * blk;
            for (unsigned t = s; t < logn; t++) {
                const size_t m = n >> (t + 1);
                for (size_t q = 0; q < blk; q += 2 * m)
                    for (size_t j = 0; j < m; j++) bfly_dif(&p[q + j], &p[q + j + m], &tw[j << t]);
            }
        }
        return;
    }
    for (size_t m = n / 2, stride = 1; m >= 1; m >>= 1, stride <<= 1) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (n >= 4096)
        for (size_t idx = 0; idx < n / 2; idx++) {
            size_t blk = idx / m, j = idx % m, i0 = blk * 2 * m + j;
            bfly_dif(&a[i0], &a[i0 + m], &tw[j * stride]);
        }
    }
}
static void dit_inplace(fe *a, const fe *tw, unsigned logn, int nthreads) {
    const size_t n = (size_t)1 << logn;
    if (logn > NTT_LB + 1) {
        const size_t blk = (size_t)1 << NTT_LB;  /* stages t = 0 .. NTT_LB - 1 (half-span 1 << t, twiddle index j << (logn - 1 - t)) inside contiguous blocks */
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
        for (size_t b = 0; b < n / blk; b++) {
            fe *p = a + b * blk;
            for (unsigned t = 0; t < NTT_LB; t++) {
                const size_t m = (size_t)1 << t;
                for (size_t q = 0; q < blk; q += 2 * m)
                    for (size_t j = 0; j < m; j++) bfly_dit(&p[q + j], &p[q + j + m], &tw[j << (logn - 1 - t)]);
            }
        }
        for (unsigned t = NTT_LB; t < logn;) {
            unsigned g = logn - t; if (g > NTT_G) g = NTT_G;
            ntt_group(a, tw, n, (size_t)1 << t, g, logn - 1 - t, 0, nthreads);
            t += g;
        }
        return;
    }
    for (size_t m = 1, stride = n / 2; m < n; m <<= 1, stride >>= 1) {
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (n >= 4096)
        for (size_t idx = 0; idx < n / 2; idx++) {
            size_t blk = idx / m, j = idx % m, i0 = blk * 2 * m + j;
            bfly_dit(&a[i0], &a[i0 + m], &tw[j * stride]);
        }
    }
}

/* a[i] *= base^(e(i)), e(i) = i or bitrev(i); and optional extra constant factor */
static void scale_powers(fe *a, unsigned logn, const fe *base, int reversed, const fe *extra, int nthreads) {
    size_t n = (size_t)1 << logn;
    fe *pw = (fe *)malloc(sizeof(fe) * n);
    fr_powers(pw, n, base, extra, nthreads);
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (n >= 4096)
    for (size_t i = 0; i < n; i++) { size_t e = reversed ? bitrev_sz(i, logn) : i; fr_mul(&a[i], &a[i], &pw[e]); }
    free(pw);
}

static void domain_fft(const domain_t *d, fe *a, int inverse, int decimation, int coset, int nthreads) {
    fe g, ginv; fld_set_u64(&FR, &g, 5); fld_inv(&FR, &ginv, &g);    /* FrMultiplicativeGen = 5 */
    if (!inverse) {
        /* (*Domain).FFT: coset scaling first -- DIF by CosetTable[i], DIT by CosetTableReversed[i] */
        if (coset) scale_powers(a, d->logn, &g, decimation == ORC_DIT, NULL, nthreads);
        if (decimation == ORC_DIF) dif_inplace(a, d->tw, d->logn, nthreads); else dit_inplace(a, d->tw, d->logn, nthreads);
    } else {
        /* (*Domain).FFTInverse: TwiddlesInv, then * CardinalityInv (and CosetTableInv[i] for DIT / CosetTableInvReversed[i] for DIF) */
        if (decimation == ORC_DIF) dif_inplace(a, d->tw_inv, d->logn, nthreads); else dit_inplace(a, d->tw_inv, d->logn, nthreads);
        if (coset) scale_powers(a, d->logn, &ginv, decimation == ORC_DIF, &d->card_inv, nthreads);
        else {
            size_t n = d->n;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (n >= 4096)
            for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &d->card_inv);
        }
    }
}

void orc_fr_ntt(uint64_t *a, unsigned logn, int inverse, int decimation, int coset, int nthreads) {
    domain_t d; domain_init(&d, logn);
    domain_fft(&d, (fe *)a, inverse, decimation, coset, nthreads);
    domain_free(&d);
}

/* gnark v0.8.0 groth16 computeH  [UPSTREAM-RECALL]:  3x FFTInverse(DIF); 3x FFT(DIT, coset);
 * a = (a*b - c) * (g^N - 1)^-1; FFTInverse(a, DIF, coset).  Output stays in bit-reversed order. */
void orc_groth16_compute_h(const uint64_t *a_, const uint64_t *b_, const uint64_t *c_, size_t n, unsigned logN,
                           uint64_t *h_out, int nthreads) {
    domain_t d; domain_init(&d, logN);
    size_t N = d.n;
    fe *a = (fe *)h_out, *b = (fe *)calloc(N, sizeof(fe)), *c = (fe *)calloc(N, sizeof(fe));
    memset(a, 0, N * sizeof(fe));
    memcpy(a, a_, n * sizeof(fe)); memcpy(b, b_, n * sizeof(fe)); memcpy(c, c_, n * sizeof(fe));
    domain_fft(&d, a, 1, ORC_DIF, 0, nthreads); domain_fft(&d, b, 1, ORC_DIF, 0, nthreads); domain_fft(&d, c, 1, ORC_DIF, 0, nthreads);
    domain_fft(&d, a, 0, ORC_DIT, 1, nthreads); domain_fft(&d, b, 0, ORC_DIT, 1, nthreads); domain_fft(&d, c, 0, ORC_DIT, 1, nthreads);
    fe den, g, e = {{(uint64_t)N, 0, 0, 0}};
    fld_set_u64(&FR, &g, 5); fld_pow(&FR, &den, &g, &e); fld_sub(&FR, &den, &den, &FR.one); fld_inv(&FR, &den, &den);
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (N >= 4096)
    for (size_t i = 0; i < N; i++) { fe t; fr_mul(&t, &a[i], &b[i]); fld_sub(&FR, &t, &t, &c[i]); fr_mul(&a[i], &t, &den); }
    domain_fft(&d, a, 1, ORC_DIF, 1, nthreads);
    free(b); free(c); domain_free(&d);
}

/* gnark v0.8.0 groth16.Prove body with r, s as inputs (see bn254_ref.groth16_prove)  [UPSTREAM-RECALL] */
int orc_groth16_prove(const orc_groth16_pk *pk, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                      size_t n_constraints, const uint64_t *w, const uint64_t r_[4], const uint64_t s_[4],
                      int nthreads, uint8_t proof_out[128], uint64_t *points_out) {
    orc_init();
    size_t N = (size_t)1 << pk->log_domain;
    if (n_constraints > N || pk->n_public > pk->n_wires) return -1;
    fe *h = (fe *)malloc(sizeof(fe) * N);
    orc_groth16_compute_h(a, b, c, n_constraints, pk->log_domain, (uint64_t *)h, nthreads);
    fe r, s, rs;
    memcpy(&r, r_, 32); memcpy(&s, s_, 32); fr_mul(&rs, &r, &s);
    g1_aff t1, t2, ar, bs1, krs; g2_aff u1, bs;
    /* Ar = MSM(A, w) + alpha + r*delta */
    orc_g1_msm(pk->g1_a, w, pk->n_wires, 1, 0, nthreads, (uint64_t *)&t1);
    orc_g1_add((uint64_t *)&t1, pk->g1_alpha, (uint64_t *)&t1);
    orc_g1_mul(pk->g1_delta, r.l, 1, (uint64_t *)&t2);
    orc_g1_add((uint64_t *)&t1, (uint64_t *)&t2, (uint64_t *)&ar);
    /* Bs1 = MSM(B, w) + beta + s*delta */
    orc_g1_msm(pk->g1_b, w, pk->n_wires, 1, 0, nthreads, (uint64_t *)&t1);
    orc_g1_add((uint64_t *)&t1, pk->g1_beta, (uint64_t *)&t1);
    orc_g1_mul(pk->g1_delta, s.l, 1, (uint64_t *)&t2);
    orc_g1_add((uint64_t *)&t1, (uint64_t *)&t2, (uint64_t *)&bs1);
    /* Bs = MSM(G2.B, w) + beta2 + s*delta2 */
    orc_g2_msm(pk->g2_b, w, pk->n_wires, 1, 0, nthreads, (uint64_t *)&u1);
    orc_g2_add((uint64_t *)&u1, pk->g2_beta, (uint64_t *)&u1);
    g2_aff u2; orc_g2_mul(pk->g2_delta, s.l, 1, (uint64_t *)&u2);
    orc_g2_add((uint64_t *)&u1, (uint64_t *)&u2, (uint64_t *)&bs);
    /* Krs = MSM(K, w[nPub:]) + MSM(Z, h[:N-1]) + s*Ar + r*Bs1 - rs*delta */
    orc_g1_msm(pk->g1_k, w + 4 * pk->n_public, pk->n_wires - pk->n_public, 1, 0, nthreads, (uint64_t *)&t1);
    orc_g1_msm(pk->g1_z, (uint64_t *)h, N - 1, 1, 0, nthreads, (uint64_t *)&t2);
    orc_g1_add((uint64_t *)&t1, (uint64_t *)&t2, (uint64_t *)&krs);
    orc_g1_mul((uint64_t *)&ar, s.l, 1, (uint64_t *)&t1); orc_g1_add((uint64_t *)&krs, (uint64_t *)&t1, (uint64_t *)&krs);
    orc_g1_mul((uint64_t *)&bs1, r.l, 1, (uint64_t *)&t1); orc_g1_add((uint64_t *)&krs, (uint64_t *)&t1, (uint64_t *)&krs);
    orc_g1_mul(pk->g1_delta, rs.l, 1, (uint64_t *)&t1); fp_neg(&t1.y, &t1.y);
    orc_g1_add((uint64_t *)&krs, (uint64_t *)&t1, (uint64_t *)&krs);
    orc_g1_compress((uint64_t *)&ar, proof_out);
    orc_g2_compress((uint64_t *)&bs, proof_out + 32);
    orc_g1_compress((uint64_t *)&krs, proof_out + 96);
    if (points_out) { memcpy(points_out, &ar, 64); memcpy(points_out + 8, &bs, 128); memcpy(points_out + 24, &krs, 64); }
    free(h);
    return 0;
}

/* ------------------------------------------------------------------ PLONK (plonk.Setup / plonk.Prove restated; see the file) */
#include "plonk_oracle_impl.h"
