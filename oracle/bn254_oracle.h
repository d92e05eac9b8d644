/* CPU oracle for the BN254 proving hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (noir_backend_using_gnark_amd/, libzkmi.so) never links, loads or calls it.
 *
 * PARITY UNPINNED by the reference's own tests (no MSM / FFT / proof vector exists in /root/reference;
 * SURVEY.md §8c).  What is restated is the published algorithm of the un-vendored modules the reference
 * pins: gnark-crypto v0.9.1 (/root/reference/gnark_backend_ffi/go.mod:5) and gnark v0.8.0 (go.mod:23),
 * reached through groth16.Prove (/root/reference/gnark_backend_ffi/main.go:131) and plonk.Prove
 * (/root/reference/gnark_backend_ffi/backend/plonk/plonk.go:67).  It is pinned instead against the
 * independent pure-Python big-int generator oracle/bn254_ref.py and the committed vectors in tests/golden/.
 *
 * All field elements cross this API as gnark-crypto's memory image: uint64_t[4], little-endian limbs,
 * Montgomery form (x * 2^256 mod p).  Affine infinity is (0,0).
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DIT = 0, ORC_DIF = 1 }; /* gnark-crypto fft.Decimation */

/* field (which: 0 = Fr, 1 = Fp) */
void orc_fe_mul(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_fe_add(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_fe_sub(int which, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_fe_inv(int which, const uint64_t a[4], uint64_t out[4]);
void orc_fe_to_mont(int which, const uint64_t a[4], uint64_t out[4]);
void orc_fe_from_mont(int which, const uint64_t a[4], uint64_t out[4]);

/* SplitMix64 -> 4 limbs -> mod r; output Montgomery if mont != 0 else canonical */
void orc_rand_fr(uint64_t seed, size_t n, uint64_t *out, int mont);
/* "witness-like": 50% in {0,1}, 25% < 2^32, 25% uniform */
void orc_rand_fr_witness(uint64_t seed, size_t n, uint64_t *out, int mont);

/* P_i = k_i * G, k_i from SplitMix64(seed) (valid subgroup points; G1Affine / G2Affine memory image) */
void orc_g1_gen_points(uint64_t seed, size_t n, uint64_t *out /* n*8 */, int nthreads);
void orc_g2_gen_points(uint64_t seed, size_t n, uint64_t *out /* n*16 */, int nthreads);
int orc_g1_on_curve(const uint64_t p[8]);
int orc_g2_on_curve(const uint64_t p[16]);

/* (*G1Jac).MultiExp restated: bucket method, signed c-bit digits, one task per window; affine out.
 * returns 0, or -1 on n mismatch style errors.  c == 0 -> gnark's cost model picks it. */
int orc_g1_msm(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, int c, int nthreads, uint64_t out[8]);
int orc_g2_msm(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, int c, int nthreads, uint64_t out[16]);
/* definition: sum_i s_i P_i by double-and-add */
int orc_g1_msm_naive(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out[8]);
int orc_g2_msm_naive(const uint64_t *points, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out[16]);
/* out = a + b ; out = k * a (k canonical or Montgomery Fr) */
void orc_g1_add(const uint64_t a[8], const uint64_t b[8], uint64_t out[8]);
void orc_g2_add(const uint64_t a[16], const uint64_t b[16], uint64_t out[16]);
void orc_g1_mul(const uint64_t a[8], const uint64_t k[4], int k_mont, uint64_t out[8]);
void orc_g2_mul(const uint64_t a[16], const uint64_t k[4], int k_mont, uint64_t out[16]);
/* G1Affine.Bytes() / G2Affine.Bytes() compressed encodings */
void orc_g1_compress(const uint64_t a[8], uint8_t out[32]);
void orc_g2_compress(const uint64_t a[16], uint8_t out[64]);

/* (*Domain).FFT / FFTInverse on a[0..2^logn) in place; fft.BitReverse */
void orc_fr_ntt(uint64_t *a, unsigned logn, int inverse, int decimation, int coset, int nthreads);
void orc_fr_bit_reverse(uint64_t *a, unsigned logn);
/* gnark groth16 computeH: a,b,c have n <= 2^logN entries each; h_out has 2^logN entries (bit-reversed order) */
void orc_groth16_compute_h(const uint64_t *a, const uint64_t *b, const uint64_t *c, size_t n, unsigned logN,
                           uint64_t *h_out, int nthreads);

/* Groth16 prove with pinned (r, s).  Layout mirrors gnark's ProvingKey. */
typedef struct {
    unsigned log_domain;          /* N = 2^log_domain */
    size_t n_wires, n_public;     /* n_public includes the ONE wire */
    const uint64_t *g1_alpha, *g1_beta, *g1_delta;     /* 8 limbs each */
    const uint64_t *g1_a, *g1_b;                       /* n_wires points each */
    const uint64_t *g1_k;                              /* n_wires - n_public points */
    const uint64_t *g1_z;                              /* N points, bit-reversed order; N-1 used */
    const uint64_t *g2_beta, *g2_delta;                /* 16 limbs each */
    const uint64_t *g2_b;                              /* n_wires points */
} orc_groth16_pk;
/* a,b,c: n_constraints each (Montgomery Fr); w: n_wires (Montgomery Fr); r,s Montgomery Fr.
 * proof_out = Ar | Bs | Krs compressed (128 bytes); points_out (optional) = Ar(8) Bs(16) Krs(8) limbs. */
int orc_groth16_prove(const orc_groth16_pk *pk, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                      size_t n_constraints, const uint64_t *w, const uint64_t r[4], const uint64_t s[4],
                      int nthreads, uint8_t proof_out[128], uint64_t *points_out);

/* ---- PLONK (oracle/plonk_oracle_impl.h): plonk.Setup and plonk.Prove of gnark v0.8.0 restated, the C twin of oracle/plonk_ref.py.
 * The constraint system is the reference's own: one gate qL*xa + qR*xb + qO*xc + qM*xa*xb + qC = 0 per arithmetic opcode
 * (/root/reference/gnark_backend_ffi/backend/plonk/sparse_r1cs.go:44-107); variables = public first, then secret. */
typedef struct {
    size_t n_public, n_constraints, n_vars;
    const uint64_t *ql, *qr, *qm, *qo, *qk;   /* n_constraints Montgomery Fr each */
    const uint32_t *xa, *xb, *xc;             /* n_constraints variable indices each */
    const uint64_t *srs_g1;                   /* kzg.SRS.G1: srs_len >= domain size + 3 affine points; must outlive the key */
    size_t srs_len;
} orc_plonk_circuit;
typedef struct orc_plonk_pk orc_plonk_pk;
orc_plonk_pk *orc_plonk_setup(const orc_plonk_circuit *c, int nthreads);   /* NULL on bad arguments */
void orc_plonk_pk_free(orc_plonk_pk *pk);
void orc_plonk_pk_sizes(const orc_plonk_pk *pk, size_t *n, size_t *n4);
/* which 0..8: canonical ql, qr, qm, qo, cqk, Lagrange lqk, canonical s1, s2, s3 (n elements each); 9: the verifying key's digests
 * [S1] [S2] [S3] [Ql] [Qr] [Qm] [Qo] [Qk] (8 affine points); 10: the permutation (3n uint32) */
void orc_plonk_pk_get(const orc_plonk_pk *pk, int which, void *out);
/* solution: all variables (Montgomery), blinders: 9 Montgomery scalars (l: 2, r: 2, o: 2, z: 3); proof_out = Proof.WriteTo's 548 bytes;
 * challenges_out (optional): gamma, beta, alpha, zeta, kzg gamma.  0 ok, -2 the constraint system is not satisfied, -1 bad arguments. */
int orc_plonk_prove(const orc_plonk_pk *pk, const uint64_t *solution, const uint64_t *blinders, int nthreads, uint8_t proof_out[548],
                    uint64_t *challenges_out);
void orc_sha256(const uint8_t *data, size_t n, uint8_t out[32]);

int orc_max_threads(void);   /* threads worth starting: min(OpenMP default, the cgroup CPU quota) */
int orc_host_cpus(void);     /* logical CPUs the box shows */
#ifdef __cplusplus
}
#endif
#endif
