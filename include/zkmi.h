/* zkmi.h -- C ABI of libzkmi.so: the MI355X (gfx950) BN254 proving hot path.
 *
 * This is the drop-in boundary for the ONE path this repository replaces inside
 * lambdaclass/noir_backend_using_gnark: the gnark-crypto calls that gnark's provers make
 *     (*G1Jac).MultiExp / (*G2Jac).MultiExp      ecc/bn254/multiexp.go          (Pippenger MSM)
 *     (*fft.Domain).FFT / FFTInverse / BitReverse ecc/bn254/fr/fft/fft.go        (radix-2 NTT over Fr)
 *     computeH + the 5 MSMs of groth16.Prove      gnark internal/backend/bn254/groth16/prove.go
 *     kzg.Commit (= one G1 MSM)                   ecc/bn254/fr/kzg/kzg.go
 * The reference pins those modules at /root/reference/gnark_backend_ffi/go.mod:5 (gnark-crypto v0.9.1) and
 * go.mod:23 (gnark v0.8.0) and reaches them ONLY through
 *     groth16.Prove   /root/reference/gnark_backend_ffi/main.go:131
 *     plonk.Prove     /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:67
 *     plonk.Setup     /root/reference/gnark_backend_ffi/backend/plonk/plonk.go:21
 *     kzg.NewSRS      /root/reference/gnark_backend_ffi/backend/common.go:137
 * (there is no direct MultiExp / fft call site in the reference: SURVEY.md §0 fact 2).  INTEGRATION.md shows the
 * cgo stubs that bind each entry point below at those gnark-crypto call sites; the outer Rust->Go FFI
 * (PlonkProveWithPK & co, /root/reference/gnark_backend_ffi/main.go:24-78) is untouched by construction.
 *
 * Conventions
 *  - plain C types only; every pointer is valid for the duration of the call only (cgo rule: C must not retain Go
 *    pointers) unless the function name ends in _dev, in which case the pointer is a HIP device pointer owned by
 *    the caller.
 *  - field elements are gnark-crypto's memory image: uint64_t[4], little-endian limbs, Montgomery form
 *    (x * 2^256 mod p).  G1Affine = {X, Y}; G2Affine = {X.A0, X.A1, Y.A0, Y.A1}; the point at infinity is (0,0).
 *  - results that are points are AFFINE (canonical); the Go shim calls FromAffine to satisfy *G1Jac.
 *  - return value: 0 = ok, negative = zk_status error; zk_last_error() gives a thread-local message.
 *  - thread-safe and re-entrant: gnark issues its MSMs from several goroutines; each call picks a stream slot.
 *  - there is NO CPU fallback: without a usable HIP device every compute entry point returns ZK_ERR_NO_DEVICE.
 */
#ifndef ZKMI_H
#define ZKMI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } zk_fr;                 /* fr.Element  (go.mod:5 ecc/bn254/fr)  */
typedef struct { uint64_t l[4]; } zk_fp;                 /* fp.Element  (go.mod:5 ecc/bn254/fp)  */
typedef struct { zk_fp x, y; } zk_g1_affine;             /* bn254.G1Affine */
typedef struct { zk_fp x0, x1, y0, y1; } zk_g2_affine;   /* bn254.G2Affine {X E2{A0,A1}, Y E2{A0,A1}} */

typedef enum {
    ZK_OK = 0,
    ZK_ERR_LEN = -1,        /* MultiExp: "len(points) != len(scalars)" */
    ZK_ERR_NB_TASKS = -2,   /* MultiExp: "invalid config: config.NbTasks > 1024" */
    ZK_ERR_NO_DEVICE = -3,  /* no usable gfx950 device / HIP runtime -- never falls back to the CPU */
    ZK_ERR_HIP = -4,        /* a HIP call failed; see zk_last_error() */
    ZK_ERR_ARG = -5,        /* bad argument (null pointer, size not a power of two, log_n > 28, ...) */
    ZK_ERR_HANDLE = -6,     /* unknown or freed handle (or a proving key that live msm5 sessions still use) */
    ZK_ERR_BUSY = -7        /* no stream slot became free within the time limit (ZKMI_SLOT_TIMEOUT_S, default 120 s): a caller
                               that holds an msm5 session (5 of the 8 slots) and asks for more would otherwise wait forever */
} zk_status;

/* mirrors ecc.MultiExpConfig{NbTasks, ScalarsMont} of gnark-crypto v0.9.1, plus device-side knobs */
typedef struct {
    int nb_tasks;      /* accepted for compatibility; > 1024 is rejected like upstream; otherwise ignored      */
    int scalars_mont;  /* 1: scalars are Montgomery fr.Element images; 0: regular (canonical) form -- upstream's zero value
                          (gnark v0.8.0 calls FromMont() on the wire values / h and passes the default config)         */
    int window_bits;   /* 0 = auto (cost model for the GPU), else c in [2, 22]; against registered bases with window
                          tables an explicit value selects the plain (table-less) method                              */
    int device_mask;   /* several GPUs in ONE process (zk_init_devices): bit i = device entry i takes part; 0 = the process default
                          (zk_set_default_devices; every entry once zk_init_devices was called, else the calling thread's entry)  */
} zk_msm_cfg;

enum { ZK_DIT = 0, ZK_DIF = 1 }; /* fft.Decimation (same iota order as gnark-crypto) */

/* ---- lifecycle ------------------------------------------------------------------------------------------------ */
int zk_device_count(void);             /* number of visible HIP devices (0 if none / runtime missing)             */
int zk_init(int device);               /* optional: bind the calling process to `device` (default: 0, lazily)     */
/* Several GPUs in ONE process (the reference is one process: nargo -> Rust -> cgo; gnark_backend_ffi/main.go:24-37).  zk_init_devices gives the process
 * its device list: one ENTRY per listed HIP device, in order (devices == NULL or n == 0: every visible device); a device may be listed several times
 * (each listing is an entry with streams and workspaces of its own: "virtual devices").  From then on the calls that carry a device_mask -- and, through
 * the process default, those that do not (zk_bn254_ntt on host slices, zk_bn254_bases_register*) -- spread over the entries: MSMs by point range, a
 * resident Groth16 key by wire range with computeH block-sharded, transforms by blocks; results are the single-GPU bytes.  Resident objects carry their
 * entry in their handle, so every call on a handle runs on that handle's GPU whatever thread makes it; zk_set_entry picks the entry for the calls of the
 * calling THREAD that take no handle (zk_dev_alloc, zk_bn254_ntt_dev, ...).  Calling zk_init_devices again may only extend the list. */
int zk_init_devices(const int *devices, size_t n);
/* The stream slots a first proof will take, created ahead of time: a HIP stream costs 7-14 ms to create on MI355X / ROCm 7.2 and the first one of a process
 * 40-160 ms (tools/hip_start_bench.hip), so the library creates them when a slot is first used -- or here, for a caller that has host work to do meanwhile (the
 * export shim reads srs.hex).  n slots of the calling thread's device entry (at most all 8). */
int zk_warm_streams(int n);
/* The same for the five slots of a Groth16 proof session INCLUDING their high-priority streams (what zk_bn254_groth16_prove takes at once).  After
 * zk_init_flags(ZK_INIT_LEAN_STREAMS) this is also what ALLOWS those streams: until it is called a lean process runs every chain on the slots' own streams (its
 * first proof creates no stream; the export path calls it when a key's second proof is asked for). */
int zk_warm_session_streams(void);
/* The same from the library's background thread, one slot at a time (each slot is held only while its stream is created): proofs that arrive meanwhile find
 * their five slots and keep running on the slots' own streams until all five high-priority streams exist.  Returns at once. */
int zk_warm_session_streams_background(void);
/* 1 when no background job (window tables of a key or SRS that proves again, zk_warm_session_streams_background) is queued or running; waits up to timeout_ms
 * for that (< 0: as long as it takes), 0 on timeout.  For callers that want the tables in place before they measure or compare; the provers never wait. */
int zk_background_wait(int timeout_ms);
/* Background jobs start when no call is in flight: zk_background_hold(+1) / (-1) brackets a call of several phases (the export shim does it around every
 * export; the library's export entry points do it themselves); a job waits for the count to reach zero and the stream slots to be free, at most
 * zk_background_set_yield_ms (default 250, 0 = start at once; [0, 60000]).  Neither changes a result. */
void zk_background_hold(int delta);
int zk_background_set_yield_ms(int ms);
/* Process-wide start-up choices; call before anything that touches a device.  ZK_INIT_LEAN_STREAMS: a device entry creates only the five streams every caller
 * needs with itself and every other stream on first use (the default also creates the five high-priority streams of a Groth16 proof session up front: 40 ms more
 * start-up, 1 % less per 2^20 proof -- WHICH streams share a hardware queue follows creation order).  For a process that makes one proof and exits. */
#define ZK_INIT_LEAN_STREAMS 1u
int zk_init_flags(uint32_t flags);
int zk_device_entries(int *devices_out, size_t cap); /* number of entries; devices_out[i] = HIP device of entry i */
int zk_set_entry(int entry);
int zk_set_default_devices(uint32_t mask);
uint32_t zk_default_devices(void);
const char *zk_last_error(void);
const char *zk_version(void);

/* ---- MSM: (*G1Jac).MultiExp / (*G2Jac).MultiExp ------------------------------------------------------------------
 * out = sum_i scalars[i] * points[i].  n_points != n_scalars -> ZK_ERR_LEN (upstream error).  Host pointers. */
int zk_bn254_g1_msm(const zk_g1_affine *points, size_t n_points, const zk_fr *scalars, size_t n_scalars,
                    const zk_msm_cfg *cfg, zk_g1_affine *out);
int zk_bn254_g2_msm(const zk_g2_affine *points, size_t n_points, const zk_fr *scalars, size_t n_scalars,
                    const zk_msm_cfg *cfg, zk_g2_affine *out);
/* Same, inputs already resident in HBM (device pointers); `stream` is a hipStream_t or NULL for the library's own. */
int zk_bn254_g1_msm_dev(const void *d_points, const void *d_scalars, size_t n, const zk_msm_cfg *cfg,
                        zk_g1_affine *out_host, void *stream);
int zk_bn254_g2_msm_dev(const void *d_points, const void *d_scalars, size_t n, const zk_msm_cfg *cfg,
                        zk_g2_affine *out_host, void *stream);
/* Partial result for range-sharded multi-GPU MSM: the un-normalised sum as XYZZ (4 coordinates, same limb image);
 * partials from several ranks are combined with zk_bn254_g1_sum_xyzz / zk_bn254_g2_sum_xyzz on any rank. */
int zk_bn254_g1_msm_partial_dev(const void *d_points, const void *d_scalars, size_t n, const zk_msm_cfg *cfg,
                                uint64_t out_xyzz[16], void *stream);
int zk_bn254_g2_msm_partial_dev(const void *d_points, const void *d_scalars, size_t n, const zk_msm_cfg *cfg,
                                uint64_t out_xyzz[32], void *stream);
int zk_bn254_g1_sum_xyzz(const uint64_t *partials, size_t n_partials, zk_g1_affine *out);
int zk_bn254_g2_sum_xyzz(const uint64_t *partials, size_t n_partials, zk_g2_affine *out);

/* Resident bases (Groth16 pk.G1.{A,B,K,Z}, pk.G2.B, kzg SRS.G1): upload once, reuse for every proof. */
int zk_bn254_bases_register(const void *points, size_t n, int is_g2, uint64_t *handle);
int zk_bn254_bases_free(uint64_t handle);
int zk_bn254_msm_bases(uint64_t handle, size_t offset, const zk_fr *scalars, size_t n, const zk_msm_cfg *cfg, void *out);
/* the same with the points (register) / the scalars (msm) already in HBM -- KZG commits of polynomials that the NTTs left on the device.
 * Registration builds precomputed window tables 2^(c*w)*P_i when they fit (>= 4096 bases): every commit then feeds one bucket set. */
int zk_bn254_bases_register_dev(const void *d_points, size_t n, int is_g2, uint64_t *handle);
/* The same with the table geometry chosen by the caller: table_window_bits = 0 auto (tables for >= 4096 bases when they fit),
 * -1 no tables, else c in [8, 24] (tables at any n: how the tests reach the widths the planner picks at 2^22 .. 2^26 points). */
int zk_bn254_bases_register_cfg(const void *points, size_t n, int is_g2, int on_device, int table_window_bits, uint64_t *handle);
/* Window tables for a base array registered without them (table_window_bits as above; 0 = the planner's width, nothing below 4096 bases; a handle that has
 * a table is left alone).  The tables of 1,000,000 G1 points take 17 ms to build and save a 2^19-gate PLONK proof 1.4 ms: a process that makes ONE proof (nargo
 * prove) is better off without them, one that makes many builds them when the second proof is asked for (csrc/goffi.cpp does exactly that). */
int zk_bn254_bases_build_table(uint64_t handle, int table_window_bits);
/* The same on the library's background thread: returns at once; multi-exps against the handle run without the table until it is published.  (What the
 * export shim does when a process's SECOND proving call arrives: that call no longer waits for the build.) */
int zk_bn254_bases_build_table_background(uint64_t handle, int table_window_bits);
int zk_bn254_msm_bases_dev(uint64_t handle, size_t offset, const void *d_scalars, size_t n, const zk_msm_cfg *cfg, void *out);
/* The Lagrange form of a registered G1 base array over the domain of 2^log_n points, as a base array of its own: out[i] = (1/n) sum_j w^(-ij) in[j] for i < n (the
 * inverse transform taken in the exponent: n/2 log2 n + n point-by-scalar multiplications on the device), then in[n] - in[0] and in[n+1] - in[1] (for a KZG SRS:
 * [tau^n - 1], [tau^(n+1) - tau], the points of gnark's blinding).  sum_i e_i out[i] == sum_j c_j in[j] whenever c = FFTInverse(e): a polynomial is committed from its
 * EVALUATIONS -- wire values, small -- instead of its coefficients.  Needs 2^log_n + 2 points on one device entry; free the result with zk_bn254_bases_free. */
int zk_bn254_bases_lagrange(uint64_t handle, uint32_t log_n, uint64_t *out_handle);
/* `count` scalar vectors of n elements each against ONE registered base array: out[k] = MultiExp(bases[offset : offset + n], scalars[k]) -- plonk.Prove's
 * three simultaneous kzg.Commit calls (l, r, o; h1, h2, h3: gnark v0.8.0 backend/plonk/bn254/prove.go, reached from gnark_backend_ffi/backend/plonk/plonk.go:53-73).
 * With a G1 window table and count <= 3 they are ONE multi-scalar multiplication with a bucket set per vector (one recoding, one accumulate launch); otherwise the
 * vectors run one after the other.  out: count affine points of the handle's group; the same points as count calls of zk_bn254_msm_bases. */
int zk_bn254_msm_bases_batch(uint64_t handle, size_t offset, const zk_fr *const *scalars, uint32_t count, size_t n, const zk_msm_cfg *cfg, void *out);
int zk_bn254_msm_bases_batch_dev(uint64_t handle, size_t offset, const void *const *d_scalars, uint32_t count, size_t n, const zk_msm_cfg *cfg, void *out);
/* Scalars kept resident and recoded ONCE for every base array they pair with.  groth16.Prove's five MultiExp calls (gnark v0.8.0 groth16 prove.go, reached
 * from gnark_backend_ffi/main.go:131) pair A, B1, K and G2.B with the SAME wire values: through zk_bn254_msm_bases each call uploads the 32 B x n scalars and
 * recodes them (digits, radix sort, task plan).  zk_bn254_scalars_register uploads once (cfg->scalars_mont says which form they are in);
 * zk_bn254_msm_bases_prepared(bases, bases_offset, scalars, skip, cfg, out) = sum_{i >= skip} scalars[i] * bases[bases_offset + i - skip] shares one recoding
 * among base arrays registered over the same index space (A, B1, G2.B) and makes a second one from the resident copy for K (skip = n_public) -- callable
 * from concurrent threads, like upstream's goroutines.  Same affine result as zk_bn254_msm_bases on the same data. */
int zk_bn254_scalars_register(const zk_fr *scalars, size_t n, const zk_msm_cfg *cfg, uint64_t *handle);
int zk_bn254_scalars_free(uint64_t handle);
int zk_bn254_msm_bases_prepared(uint64_t bases, size_t bases_offset, uint64_t scalars_handle, size_t skip, const zk_msm_cfg *cfg, void *out);

/* ---- NTT: (*fft.Domain).FFT / FFTInverse / fft.BitReverse ------------------------------------------------------
 * In place on a[0 .. 2^log_n).  decimation: ZK_DIF natural in -> bit-reversed out; ZK_DIT bit-reversed in ->
 * natural out.  coset != 0 evaluates on / interpolates from the coset g*H, g = 5 (FrMultiplicativeGen).
 * FFTInverse also scales by 1/N exactly like upstream.
 * Precondition (as for every zk_fr in this header): the elements are fr.Element images, i.e. Montgomery values REDUCED below r -- what gnark's arithmetic
 * always produces.  The butterflies keep lazily reduced 29-bit-limb values whose bounds (tools/u29_ntt_model.py) start from an entry below 2.2 r; an image in
 * [4r, 2^256) handed to the DIT transform without coset meets a subtraction bias of 4 r in the twiddle-free first stage and gives an undefined result. */
int zk_bn254_ntt(zk_fr *a, uint32_t log_n, int inverse, int decimation, int coset);
/* the same over several device entries of this process (bit i of device_mask = entry i; 2, 4 or 8 entries): block k of the array goes to entry k over that
 * GPU's own PCIe link, the transform runs block-sharded with two all-to-all transposes between the GPUs.  zk_bn254_ntt == device_mask 0 (process default). */
int zk_bn254_ntt_devices(zk_fr *a, uint32_t log_n, int inverse, int decimation, int coset, uint32_t device_mask);
int zk_bn254_ntt_dev(void *d_a, uint32_t log_n, int inverse, int decimation, int coset, void *stream);
int zk_bn254_bit_reverse(zk_fr *a, uint32_t log_n);
int zk_bn254_bit_reverse_dev(void *d_a, uint32_t log_n, void *stream);

/* ---- Groth16: computeH and Prove (gnark v0.8.0 internal/backend/bn254/groth16/prove.go) -------------------------
 * computeH: a, b, c hold n <= 2^log_N evaluations each; h_out receives 2^log_N coefficients in the order gnark's
 * computeH leaves them (bit-reversed; the caller uses h[:N-1]). */
int zk_bn254_groth16_compute_h(const zk_fr *a, const zk_fr *b, const zk_fr *c, size_t n, uint32_t log_N, zk_fr *h_out);
int zk_bn254_groth16_compute_h_dev(const void *d_a, const void *d_b, const void *d_c, size_t n, uint32_t log_N,
                                   void *d_h_out, void *stream);
/* computeH sharded over G = 2^log_g GPUs (one process per GPU; SURVEY.md 8e: the transposes are all-to-all over xGMI,
 * issued by the host between the phases -- the library never communicates).  Rank rho owns the block
 * [rho*M, (rho+1)*M), M = 2^log_D / G, of a, b, c (natural order) and receives the same block of h (gnark's
 * bit-reversed order).  "Transposed" = after an all-to-all of the block viewed as G chunks of M/G elements.
 *   phase 0: a, b, c transposed      -> cross stages of FFTInverse(DIF)                         -> transpose back
 *   phase 1: a, b, c blocks          -> block part of FFTInverse(DIF), *1/D*g^bitrev(i), block part of FFT(DIT, coset) -> transpose
 *   phase 2: a, b, c transposed      -> cross stages of FFT(DIT); a = (a*b - c)/(g^D - 1); cross stages of FFTInverse(DIF) -> transpose a back
 *   phase 3: a block                 -> block part of FFTInverse(DIF, coset): a is this rank's block of h.
 * Phases 0 and 1 act on each array independently and skip null pointers; phase 2 = phase 4 (cross stages of FFT(DIT), per array) followed by
 * phase 5 (the pointwise step and the final cross stages, all three arrays) -- so that the host can pipeline the transposes of b, c under the
 * stages of a (parallel.compute_h_sharded(pipelined=True): async collectives, one array ahead).
 * The six-transform schedule (default of parallel.compute_h_sharded; exact by linearity): c stays in coefficient form --
 *   phase 6: blocks given (c)    -> rest of FFTInverse(DIF) with 1/D: block of the coefficients, bit-reversed order (no further transpose of c)
 *   phase 7: a, b transposed, after phase 4 on each -> a = a*b; cross stages of FFTInverse(DIF) on a            -> transpose a back
 *   phase 8: a block + c block   -> block part of FFTInverse(DIF, coset) on a; a = (a - c)/(g^D - 1): this rank's block of h.
 *   Order: transpose a, b, c; 0 each; transpose each; 1 on a, b and 6 on c; transpose a, b; 4 on a, b; 7; transpose a; 8  -- 9 transposes instead of 10.
 * log_g = 0 degenerates to zk_bn254_groth16_compute_h_dev (no exchange).  In place; asynchronous on `stream` if given. */
int zk_bn254_groth16_h_shard_dev(int phase, void *d_a, void *d_b, void *d_c, uint32_t log_D, uint32_t log_g,
                                 uint32_t rank, void *stream);

/* A standalone (*Domain).FFT / FFTInverse sharded by blocks over G = 2^log_g GPUs (BASELINE configs[4] on several GPUs; the decomposition of the sharded
 * computeH above, for any of the eight mode combinations).  Rank rho holds block rho of the STORED order (natural for DIF input / DIT output, bit-reversed for
 * DIF output / DIT input); the library never communicates -- the host runs the steps with its all-to-all transposes (X) in between:
 *     FFT(DIF):        [2 if coset]  X  0  X  1              FFT(DIT):         1  X  0  X
 *     FFTInverse(DIF):               X  0  X  1              FFTInverse(DIT):  1  X  0  X  [2 if coset]
 *   step 0: the log_g cross stages on TRANSPOSED data;  step 1: the size-M transform of the block, with 1/D and the coset factors that can ride on it;
 *   step 2: the coset factor g^(+-i) on the natural-order block where it cannot (a no-op in the other modes).  log_g <= 3; in place; asynchronous on
 *   `stream` if given.  noir_backend_using_gnark_amd/parallel.py: ntt_sharded. */
int zk_bn254_ntt_shard_dev(int step, void *d_a, uint32_t log_D, uint32_t log_g, uint32_t rank, int inverse, int decimation, int coset,
                           void *stream);

/* ProvingKey image (host pointers; copied to the device by zk_bn254_groth16_pk_load).  Same field names as gnark's
 * groth16.ProvingKey{G1{Alpha,Beta,Delta,A,B,K,Z}, G2{Beta,Delta,B}, InfinityA, InfinityB, NbInfinityA, NbInfinityB}
 * (gnark v0.8.0 internal/backend/bn254/groth16/setup.go; the reference reaches it at gnark_backend_ffi/main.go:121,131 and
 * meant to pass it through backend/groth16/r1cs.go:107-143).
 *   - gnark's layout: infinity_a / infinity_b point at the []bool images (one byte per wire, != 0: that wire's point is the point
 *     at infinity and is NOT stored); g1_a then holds n_wires - nb_infinity_a points, g1_b and g2_b n_wires - nb_infinity_b,
 *     exactly the slices gnark keeps.  pk_load expands them on the device (one scatter kernel) to wire-indexed arrays, so the
 *     prover reads the full wire vector and needs no per-MSM filtering of the scalars.
 *   - dense layout: infinity_a == infinity_b == NULL; g1_a / g1_b / g2_b have n_wires entries, (0,0) = infinity contributes nothing.
 * K has n_wires - n_public entries; Z has 2^log_domain entries in gnark's (bit-reversed) order, N-1 are used. */
typedef struct {
    uint32_t log_domain;
    size_t n_wires, n_public;
    const zk_g1_affine *g1_alpha, *g1_beta, *g1_delta;
    const zk_g1_affine *g1_a, *g1_b, *g1_k, *g1_z;
    const zk_g2_affine *g2_beta, *g2_delta;
    const zk_g2_affine *g2_b;
    int bases_on_device; /* 1: g1_a, g1_b, g1_k, g1_z, g2_b are DEVICE pointers that stay owned by the caller */
    int flags;           /* bit 0: do NOT build the precomputed window tables 2^(c*w)*P_i (they cost ~13x the bases in HBM and
                            are what makes the resident-key MSMs ~20% cheaper; skipped automatically when HBM is short)
                            bit 1: all 2^log_domain entries of Z are used (a non-final slice of a range-sharded key)
                            bit 2: WINDOW-sharded key (BASELINE north_star / configs[2] wording): the key holds ALL wires, but its window
                                   tables only the rows w = shard_rank + k * shard_count of the ceil(255/c) digit windows -- 1/shard_count
                                   of the table memory; zk_bn254_groth16_msm5_pk then returns the partial sums over those windows of
                                   the FULL wire / h vectors, and the records of all ranks finalize as in the range-sharded mode.
                                   Needs the tables (an error when they do not fit). */
    const uint8_t *infinity_a, *infinity_b; /* gnark's InfinityA / InfinityB ([]bool, HOST pointers, n_wires bytes) or NULL */
    size_t nb_infinity_a, nb_infinity_b;    /* gnark's NbInfinityA / NbInfinityB; must equal the number of non-zero bytes */
    int table_window_bits;                  /* 0: planner's choice; else the window width c in [8, 24] of the tables */
    int device_mask;                        /* several GPUs in ONE process: bit i = device entry i holds a range slice of the key (2, 4 or 8 entries);
                                               0 = the process default.  zk_bn254_groth16_prove on such a key runs computeH block-sharded over the
                                               entries and the five MSMs per slice; same proof bytes. */
    uint32_t shard_rank, shard_count;       /* flags bit 2: this rank and the number of ranks (window sharding) */
} zk_groth16_pk;
int zk_bn254_groth16_pk_load(const zk_groth16_pk *pk, uint64_t *handle);
int zk_bn254_groth16_pk_free(uint64_t handle);   /* ZK_ERR_HANDLE while an msm5 session still uses the key */
/* Geometry of a loaded key (any out pointer may be NULL): lets a caller validate len(w) before handing a bare pointer over. */
int zk_bn254_groth16_pk_info(uint64_t handle, size_t *n_wires, size_t *n_public, uint32_t *log_domain, int *has_tables);
/* Window tables for a resident key that was loaded without them (flags bit 0): what a caller does once a key turns out to be used again -- the export path
 * (zk_groth16_prove_with_pk) reads a key text without tables for its first proof and builds them when the second is asked for.  table_window_bits: 0 = the
 * planner's choice, else [8, 24].  A key that has its tables already, or whose tables do not fit, is left as it is: *built (optional) = 1 / 0. */
int zk_bn254_groth16_pk_build_tables(uint64_t handle, int table_window_bits, int *built);
/* HBM held by a resident Groth16 key (the base arrays it owns + its window tables) */
int zk_bn254_groth16_pk_bytes(uint64_t handle, size_t *bytes);
/* Prove with the prover randomness (r, s) as INPUTS (upstream draws them from crypto/rand; pinning them is what
 * makes "bit-exact proof bytes" well defined).  a, b, c: n_constraints evaluations (solver output); w: n_wires wire
 * values; all Montgomery fr.Element.  proof_out = Ar | Bs | Krs in gnark's compressed encoding (32+64+32 bytes),
 * i.e. Proof.WriteTo.  `on_device` != 0: a, b, c, w are device pointers. */
int zk_bn254_groth16_prove(uint64_t pk_handle, const void *a, const void *b, const void *c, size_t n_constraints,
                           const void *w, size_t n_wires, const zk_fr *r, const zk_fr *s, int on_device, uint8_t proof_out[128]);
/* n_wires = len(w): must equal the key's wire count (ZK_ERR_LEN otherwise -- the library reads exactly that many elements). */

/* ---- R1CS on the device: groth16.Setup and the solver's a, b, c (gnark v0.8.0; reference: groth16.Setup at gnark_backend_ffi/main.go:121, the
 * R1CS the reference meant to build from Noir's RawR1CS at backend/groth16/r1cs.go:9-72) ------------------------------------------------------
 * L, R, O: sparse matrices by constraint (CSR: *_ptr has n_constraints + 1 entries starting at 0, *_idx wire ids, *_val Montgomery
 * coefficients); wires = [ONE, public..., secret..., internal...], n_public counts the ONE wire like gnark's GetNbPublicVariables(). */
typedef struct {
    size_t n_constraints, n_wires, n_public;
    const uint32_t *l_ptr, *l_idx; const zk_fr *l_val;
    const uint32_t *r_ptr, *r_idx; const zk_fr *r_val;
    const uint32_t *o_ptr, *o_idx; const zk_fr *o_val;
} zk_r1cs;
int zk_bn254_r1cs_load(const zk_r1cs *r1cs, uint64_t *handle);
int zk_bn254_r1cs_free(uint64_t handle);
/* a = L w, b = R w, c = O w: what r1cs.Solve leaves for a system without hints (one lane per matrix row); device pointers. */
int zk_bn254_r1cs_eval_abc_dev(uint64_t handle, const void *d_w, size_t n_wires, void *d_a, void *d_b, void *d_c, void *stream);
/* groth16.Setup with the toxic waste (tau, alpha, beta, gamma, delta; Montgomery, non-zero) as INPUT -- upstream draws it; pinning it is what
 * makes a key reproducible.  The key is built in HBM (Lagrange basis at tau, transposed products, fixed-base scalar multiplications) and comes
 * back as a resident proving key; vk_g1 (1 + n_public points: [alpha]G1, then K_i / gamma) and vk_g2 ([beta]G2, [gamma]G2, [delta]G2) on
 * the host.  flags bit 0: no window tables. */
int zk_bn254_groth16_setup(uint64_t r1cs_handle, const zk_fr toxic[5], int flags, uint64_t *pk_handle, zk_g1_affine *vk_g1,
                           zk_g2_affine vk_g2[3]);
/* groth16.Prove from the witness: a, b, c by the device solver step above, then zk_bn254_groth16_prove on resident data. */
int zk_bn254_groth16_prove_r1cs(uint64_t r1cs_handle, uint64_t pk_handle, const void *w, size_t n_wires, const zk_fr *r,
                                const zk_fr *s, int on_device, uint8_t proof_out[128]);

/* ---- Groth16 key wire formats (gnark v0.8.0 groth16 marshal.go): what the reference's intended Groth16 FFI moves as hex --
 * ProveWithPK(rawR1CS, encodedProvingKey) -> provingKey.ReadFrom at gnark_backend_ffi/backend/groth16/r1cs.go:107-143; Preprocess -> both keys at
 * r1cs.go:214-266.  ProvingKey.WriteTo (compressed points): Domain 168 B | G1.Alpha, Beta, Delta | G1.A, G1.B, G1.Z, G1.K (u32 count + 32 B each) |
 * G2.Beta, Delta | G2.B (u32 count + 64 B each) | nbWires, NbInfinityA, NbInfinityB (u64) | InfinityA, InfinityB (nbWires bytes each).
 *   pk_read : every point is decompressed ON THE DEVICE (G1: one square root; G2: an Fp2 square root + the r-torsion check gnark-crypto's Decoder
 *             applies), then the key is loaded like zk_bn254_groth16_pk_load's gnark layout.  flags: bit 0 = no window tables.
 *   pk_write: out == NULL only returns the size in *out_len.  A / B / G2.B leave without their points at infinity.
 *   vk_write: VerifyingKey.WriteTo = [alpha]1 [beta]1 [beta]2 [gamma]2 [delta]1 [delta]2 u32 len(K) K, from zk_bn254_groth16_setup's vk_g1 (n_k =
 *             n_public points after [alpha]1) / vk_g2 and the resident key. */
int zk_bn254_groth16_pk_read(const void *data, size_t len, int is_hex, int flags, int table_window_bits, uint64_t *handle);
int zk_bn254_groth16_pk_write(uint64_t handle, int as_hex, void *out, size_t cap, size_t *out_len);
int zk_bn254_groth16_vk_write(uint64_t pk_handle, const zk_g1_affine *vk_g1, size_t n_k, const zk_g2_affine vk_g2[3], int as_hex, void *out,
                              size_t cap, size_t *out_len);

/* ---- Verification on the HOST (SURVEY 8f keeps it last: O(1) next to a proof, no device work, ~10 ms on one core) -- the counterpart of the reference's
 * PlonkVerifyWithVK (gnark_backend_ffi/main.go:44-56 -> backend/plonk/plonk.go:28-51) and of the intended Groth16 VerifyWithVK (backend/groth16/r1cs.go:176-212).
 * Inputs are gnark's wire images (Proof.WriteTo, VerifyingKey.WriteTo as bytes or hex text) and the public witness as Montgomery fr.Elements (Groth16: without
 * the constant wire).  *accepted = 1 / 0 for well-formed inputs; what gnark's ReadFrom rejects is ZK_ERR_ARG / ZK_ERR_LEN.
 *   zk_bn254_groth16_verify : e(Ar, Bs) == e(alpha, beta) e(sum w_i K_i, gamma) e(Krs, delta)
 *   zk_bn254_plonk_verify   : gnark v0.8.0 plonk.Verify -- challenges from the SHA-256 transcript, quotient identity at zeta, two KZG checks against
 *                             srs_g2 = the SRS's ([1]2, [alpha]2) (zk_bn254_kzg_srs_read / _new_srs_dev return them)
 *   zk_bn254_pairing_check  : prod_i e(p_i, q_i) == 1 (optimal ate over the host field types; building block of the two above) */
int zk_bn254_groth16_verify(const uint8_t proof[128], const void *vk, size_t vk_len, int vk_is_hex, const zk_fr *public_inputs, size_t n_public,
                            int *accepted);
int zk_bn254_plonk_verify(const uint8_t proof[548] /* ZK_PLONK_PROOF_BYTES */, const void *vk, size_t vk_len, int vk_is_hex, const zk_g2_affine srs_g2[2],
                          const zk_fr *public_inputs, size_t n_public, int *accepted);
int zk_bn254_pairing_check(const zk_g1_affine *p, const zk_g2_affine *q, size_t n, int *is_one);

/* The two halves of zk_bn254_groth16_prove, exposed so that one proof can be range-sharded over several GPUs
 * (one process per GPU): every rank runs the five MSMs on ITS slice of the bases / wire values / h, the un-normalised
 * XYZZ sums (4 x G1 = 64 limbs, then G2 = 32 limbs; order A, B1, K, Z, B2) are all-gathered, and any rank finishes.
 *   msm5_dev : d_a, d_b (G1), d_b2 (G2) against d_w[0..nw); d_k against d_wk[0..nk); d_z against d_h[0..nz).
 *   finalize : sums n_partials x 96-limb records, adds alpha/beta/delta terms with (r, s), writes the 128-byte proof. */
int zk_bn254_groth16_msm5_dev(const void *d_a, const void *d_b, const void *d_b2, const void *d_w, size_t nw,
                              const void *d_k, const void *d_wk, size_t nk, const void *d_z, const void *d_h, size_t nz,
                              uint64_t out_xyzz[96], void *stream);
/* msm5 against a loaded key whose base arrays are THIS rank's slices (n_wires / n_public / log_domain describe the
 * slice; flags bit 1 on every rank but the last): uses the key's resident window tables.  d_w: n_wires wire values of
 * the slice, complete before the call (their preparation starts at once); d_h: this rank's block of h, awaited on
 * `stream` (the stream computeH was enqueued on) if given. */
int zk_bn254_groth16_msm5_pk(uint64_t pk_handle, const void *d_w, const void *d_h, uint64_t out_xyzz[96], void *stream);
/* The same in two calls: _begin starts the scalar-side preparation of the wire values at once (it does not need h), so it
 * runs while the host drives the sharded computeH and its exchanges; _end (always call it: it releases the session)
 * enqueues the rest behind `stream` and waits for the record. */
int zk_bn254_groth16_msm5_pk_begin(uint64_t pk_handle, const void *d_w, uint64_t *session);
int zk_bn254_groth16_msm5_pk_end(uint64_t session, const void *d_h, uint64_t out_xyzz[96], void *stream);
/* Gives a session up without finishing it (an exception between _begin and _end): waits for the work already enqueued and
 * releases the five stream slots.  Unknown / already ended sessions return ZK_ERR_HANDLE. */
int zk_bn254_groth16_msm5_pk_abort(uint64_t session);
/* The session's high-priority stream (valid until _end): run the sharded computeH and its exchanges on it -- that is the
 * stream zk_bn254_groth16_prove runs computeH on, and it never shares a hardware queue with the preparation of w. */
int zk_bn254_groth16_msm5_session_stream(uint64_t session, void **stream_out);
int zk_bn254_groth16_finalize(uint64_t pk_handle, const uint64_t *partials, size_t n_partials, const zk_fr *r,
                              const zk_fr *s, uint8_t proof_out[128]);

/* ---- PLONK: plonk.Setup / plonk.Prove (gnark v0.8.0 internal/backend/bn254/plonk/{setup,prove}.go) ------------------------------
 * The reference's only live prove path: PlonkProveWithPK (gnark_backend_ffi/main.go:24-37) -> plonk.Prove at
 * backend/plonk/plonk.go:67; plonk.Setup at backend/plonk/plonk.go:21; the constraint system is one gate
 *     qL*xa + qR*xb + qO*xc + qM*xa*xb + qK == 0
 * per ACIR arithmetic opcode (backend/plonk/sparse_r1cs.go:44-107).  The commitments are kzg.Commit = one G1 MSM over the SRS
 * registered with zk_bn254_bases_register* (>= domain size + 3 points; kzg.NewSRS at backend/common.go:137); the transforms are the
 * fft.Domain calls above on the small domain n and the big domain 4n (8n below 6 gates).  Everything between the solver's output and
 * the 548 proof bytes runs on the device except the Fiat-Shamir hashes and a few G1 scalar multiplications. */
typedef struct {
    size_t n_public, n_constraints, n_vars; /* spr.NbPublicVariables, len(spr.Constraints), number of variables (public first) */
    const void *ql, *qr, *qo, *qm, *qk;     /* n_constraints fr.Elements each (Montgomery); host pointers unless coeffs_on_device */
    const uint32_t *xa, *xb, *xc;           /* wire ids of every gate (HOST pointers) */
    int coeffs_on_device;
    int reserved;
} zk_plonk_circuit;
/* gnark's ProvingKey as plonk.Setup / ProvingKey.ReadFrom leave it: canonical (regular) Ql, Qr, Qm, Qo, CQk, S1..S3Canonical and LQk of
 * 2^log_n entries each, Permutation of 3 * 2^log_n entries, the verifying key's digests -- plus the gates' wire ids (from the spr). */
typedef struct {
    uint32_t log_n;
    size_t n_public, n_constraints, n_vars;
    const zk_fr *ql, *qr, *qm, *qo, *cqk, *lqk, *s1, *s2, *s3;
    const int64_t *permutation;
    const uint32_t *xa, *xb, *xc;
    const zk_g1_affine *vk_s /* [3] */, *vk_ql, *vk_qr, *vk_qm, *vk_qo, *vk_qk;
} zk_plonk_pk;
typedef struct {                 /* plonk.VerifyingKey (without the SRS) */
    uint64_t size, n_public;     /* Domain[0].Cardinality, NbPublicVariables */
    zk_fr size_inv, generator, coset_shift;
    zk_g1_affine s[3], ql, qr, qm, qo, qk;
} zk_plonk_vk;
int zk_bn254_plonk_setup(const zk_plonk_circuit *circuit, uint64_t srs_handle, uint64_t *pk_handle, zk_plonk_vk *vk_out);
int zk_bn254_plonk_pk_load(const zk_plonk_pk *pk, uint64_t srs_handle, uint64_t *pk_handle);
int zk_bn254_plonk_pk_free(uint64_t pk_handle);
/* The SRS in LAGRANGE form over this key's domain, built once ([L_i(tau)] = the inverse transform of the SRS "in the exponent": n/2 log2 n + n point-by-scalar
 * multiplications -- 0.15 s at 2^19 gates, 1.3 s at 2^22).  From then on zk_bn254_plonk_prove commits l, r, o from the WIRE VALUES: the same three digests
 * (sum_i l_i [L_i(tau)] + b0 [tau^n - 1] + b1 [tau^(n+1) - tau] is kzg.Commit of the blinded coefficients), from scalars that are bits and words in any real circuit
 * instead of uniform field elements -- a quarter to a third of the additions -- and without waiting for the three inverse transforms.  For a prover that keeps its
 * key (a process that makes one proof should not call it); needs the SRS on one device entry.  A key that has it: ZK_OK, untouched. */
int zk_bn254_plonk_pk_lagrange_srs(uint64_t pk_handle);
/* plonk.ProvingKey.ReadFrom / WriteTo on gnark's bytes (is_hex / as_hex: the hex text the reference ships, internal/backend/helpers.go:49-60,
 * 82-87): verifying key | Domain[0] | Domain[1] | Ql Qr Qm Qo CQk LQk S1 S2 S3 (u32 BE length + 32 B BE elements) | Permutation (3n raw BE
 * int64).  The nine vectors are decoded / encoded on the device; the wire ids come from the spr the caller rebuilt (plonk.go:54). */
int zk_bn254_plonk_pk_read(const void *data, size_t len, int is_hex, size_t n_vars, size_t n_constraints, const uint32_t *xa,
                           const uint32_t *xb, const uint32_t *xc, uint64_t srs_handle, uint64_t *pk_handle);
int zk_bn254_plonk_pk_write(uint64_t pk_handle, int as_hex, void *out, size_t cap, size_t *out_len);
/* Shape of a resident key (any out pointer may be NULL): Domain[0].Cardinality, NbPublicVariables, number of gates, number of variables. */
int zk_bn254_plonk_pk_info(uint64_t pk_handle, size_t *domain_size, size_t *n_public, size_t *n_constraints, size_t *n_vars);
/* Canonical polynomials of a resident key, for inspection: which = 0..8 -> Ql, Qr, Qm, Qo, CQk, S1, S2, S3 (canonical), LQk; n entries. */
int zk_bn254_plonk_pk_export(uint64_t pk_handle, int which, zk_fr *out_host, size_t n);
enum { ZK_PLONK_PROOF_BYTES = 548 }; /* Proof.WriteTo: 7 x 32 B digests | 32 B + u32 count + 7 x 32 B | 32 B + 32 B */
/* plonk.Prove after the solver.  solution: n_vars values of all variables (public first; Montgomery; device pointer if on_device).
 * blinders: the 9 scalars upstream draws with fr.SetRandom -- Blind(1) of l, r, o (2 each) and Blind(2) of z (3) -- as INPUTS, which is
 * what makes the proof bytes reproducible.  challenges: NULL = SHA-256 Fiat-Shamir exactly as upstream (transcript "gamma", "beta",
 * "alpha", "zeta" + kzg's folding "gamma"); else 5 pinned values (gamma, beta, alpha, zeta, kzg gamma; Montgomery).
 * ZK_ERR_ARG when the solution does not satisfy the circuit (the quotient is not a polynomial; upstream fails in Solve). */
int zk_bn254_plonk_prove(uint64_t pk_handle, const void *solution, size_t n_vars, int on_device, const zk_fr blinders[9],
                         const zk_fr *challenges, uint8_t proof_out[ZK_PLONK_PROOF_BYTES]);
/* synthetic data (bench / tests): qk[i] = -(ql*a + qr*b + qo*c + qm*a*b) for gate i with a, b, c = solution[xa[i]], ... -- makes any
 * assignment of random coefficients and wires satisfiable.  All pointers are device pointers. */
int zk_bn254_plonk_synth_qk_dev(void *d_qk, const void *d_ql, const void *d_qr, const void *d_qo, const void *d_qm, const void *d_xa,
                                const void *d_xb, const void *d_xc, const void *d_solution, size_t n_constraints, void *stream);

/* ---- the callers either side of the PLONK path: the reference's exported entry points, restated over the device path -----------------------
 * PlonkPreprocess (gnark_backend_ffi/main.go:58-78) and PlonkProveWithPK (main.go:24-37) take the ACIR as JSON (acir/acir.go:17-75), the witness
 * values as the hex felt vector (internal/backend/helpers.go:24-33) and the key as hex of ProvingKey.WriteTo (helpers.go:49-60, 82-87); the
 * lowering is BuildSparseR1CS / HandleValues (backend/plonk/sparse_r1cs.go:18-107, backend/common.go:45-76).  Differences: the SRS is a
 * resident handle instead of srs.hex re-read per call; errors are codes instead of log.Fatal; outputs go to caller buffers (no terminator);
 * the blinding scalars can be pinned (NULL = /dev/urandom).
 * layout: how witnesses become constraint-system variables --
 *   ZK_ACIR_LAYOUT_REFERENCE             HandleValues exactly as written (backend/common.go:45-76): public variables in witness order, then one
 *                                        SECRET variable per (witness, non-matching public input) when there are public inputs -- so with |P| >= 2
 *                                        public inputs a private witness has |P| variables, a public one |P| - 1 secret copies besides its public
 *                                        variable -- and the gates name the LAST variable added for a witness (indexMap, last write wins); a term
 *                                        that names a witness without a variable gets variable 0 (Go's map zero value).  Keys and proofs made this
 *                                        way are the reference's own; libgnark_backend.so uses it.
 *   ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS   public witnesses first, then the others, one variable each; unknown witnesses are an error.  Identical to
 *                                        the reference layout for zero or one public input.
 *   zk_plonk_preprocess: pk_hex_out == NULL only returns the sizes in *pk_len / *vk_len; pk_handle (optional) keeps the key resident.
 *   zk_plonk_prove_with_pk: pk_hex == NULL uses the resident key pk_handle (the reference deserialises the key on every call); ZK_ERR_ARG when the
 *                           key's public / variable / gate counts are not the circuit's. */
enum { ZK_ACIR_LAYOUT_REFERENCE = 0, ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS = 1 };
int zk_plonk_preprocess(const char *acir_json, size_t acir_len, const char *values_hex, size_t values_len, int layout,
                        uint64_t srs_handle, char *pk_hex_out, size_t pk_cap, size_t *pk_len, char *vk_hex_out, size_t vk_cap,
                        size_t *vk_len, uint64_t *pk_handle);
int zk_plonk_prove_with_pk(const char *acir_json, size_t acir_len, const char *values_hex, size_t values_len, int layout,
                           const char *pk_hex, size_t pk_len, uint64_t pk_handle, uint64_t srs_handle, const zk_fr *blinders,
                           char proof_hex_out[2 * ZK_PLONK_PROOF_BYTES]);
/* BuildSparseR1CS alone: the gates of an ACIR circuit (Montgomery coefficients, variable ids with the public variables first) and the
 * witness order (variable k holds witness order[k] + 1).  Array pointers may be NULL (first call sizes them through *n_constraints / *n_vars). */
int zk_acir_to_sparse_r1cs(const char *acir_json, size_t acir_len, size_t n_values, int layout, size_t *n_public, size_t *n_vars,
                           size_t *n_constraints, zk_fr *ql, zk_fr *qr, zk_fr *qo, zk_fr *qm, zk_fr *qk, uint32_t *xa, uint32_t *xb,
                           uint32_t *xc, uint32_t *order);
/* What the verifier needs of HandleValues (backend/common.go:45-60; PlonkVerifyWithVK, main.go:44-56): out[k] = 0-based position, in the witness-value
 * vector, of public variable k.  *n_public is set even when cap is too small (ZK_ERR_ARG then). */
int zk_acir_public_witnesses(const char *acir_json, size_t acir_len, size_t n_values, int layout, uint32_t *out, size_t cap, size_t *n_public);
/* What the export path keeps resident between calls.  The reference re-reads everything per call (main.go:24-37 -> backend/plonk/plonk.go:53-73: the ACIR
 * is unmarshalled and lowered, the key hex-decoded and ReadFrom'd); at 2^19 gates that is 0.24 GB of JSON and 0.33 GB of key text in front of a 10 ms
 * prover.  zk_plonk_prove_with_pk / zk_plonk_preprocess / zk_acir_public_witnesses identify both texts by a 128-bit content key and keep the lowered
 * circuit (wiring + variable order in HBM) and the decoded key (with its big-coset forms) resident, least recently used first out, within
 * ZKMI_TABLE_CAP_GB (0 keeps nothing).  A key text is only ever matched together with the circuit text and the SRS it was first seen with. */
int zk_export_cache_info(size_t *n_circuits, size_t *n_keys, size_t *bytes);
int zk_acir_lower_resident(const char *acir_json, size_t acir_len, size_t n_values, int layout, int with_coefficients); /* host only: lower now, keep resident
    (start-up overlap: the export shim runs it beside the SRS load); with_coefficients: the selectors too, kept for the zk_plonk_preprocess that follows */
int zk_export_cache_clear(void);
/* Size (G1 points) of an SRS the export shim creates when srs.hex is missing: 1,000,000 like the reference (backend/common.go:137) unless a TEST set another
 * (4 .. 2^28) before the shim's first call -- a setter instead of an environment variable, so that nothing in a prover's environment can change its SRS. */
int zk_export_set_new_srs_size(size_t n);
size_t zk_export_new_srs_size(void);
/* HBM held by a resident PLONK key */
int zk_bn254_plonk_pk_bytes(uint64_t handle, size_t *bytes);

/* buildR1CS of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:9-72; payload RawR1CS, src/gnark_backend_wrapper/groth16/
 * acir_to_r1cs.rs:18-60) on a JSON RawR1CS: a resident R1CS plus the full wire vector [ONE, public, secret, product variables] in HBM
 * (*d_witness, n_wires Montgomery elements; release with zk_dev_free) -- feed them to zk_bn254_groth16_setup / zk_bn254_groth16_prove_r1cs
 * (on_device = 1).  See frontend.hip for the points where the reference's sketch is made well-defined. */
int zk_groth16_r1cs_from_raw(const char *raw_json, size_t len, uint64_t *r1cs_handle, void **d_witness, size_t *n_wires, size_t *n_public);
/* The exported entry points of that FFI, restated over the device path (errors are codes instead of log.Fatal; outputs go to caller buffers, no
 * terminator; randomness may be pinned, NULL = /dev/urandom as upstream draws it):
 *   zk_groth16_preprocess      Preprocess    (r1cs.go:214-266): groth16.Setup -> hex(ProvingKey.WriteTo), hex(VerifyingKey.WriteTo); toxic = tau, alpha,
 *                              beta, gamma, delta.  pk_hex_out == NULL only returns the sizes; pk_handle (optional) keeps the key resident.
 *   zk_groth16_prove_with_pk   ProveWithPK   (r1cs.go:107-143): hex key (or pk_handle when pk_hex == NULL) -> 256 hex characters of Proof.WriteTo;
 *                              rs = the prover's (r, s).
 *   zk_groth16_prove_with_meta ProveWithMeta (r1cs.go:74-105): Setup + Prove, the key never leaves HBM. */
int zk_groth16_preprocess(const char *raw_json, size_t raw_len, const zk_fr *toxic, char *pk_hex_out, size_t pk_cap, size_t *pk_len,
                          char *vk_hex_out, size_t vk_cap, size_t *vk_len, uint64_t *pk_handle);
int zk_groth16_prove_with_pk(const char *raw_json, size_t raw_len, const char *pk_hex, size_t pk_len, uint64_t pk_handle, const zk_fr *rs,
                             char proof_hex_out[256]);
int zk_groth16_prove_with_meta(const char *raw_json, size_t raw_len, const zk_fr *toxic, const zk_fr *rs, char proof_hex_out[256]);
/* What these three keep resident between calls (the sketch re-reads everything per call, r1cs.go:107-128: json.Unmarshal of the RawR1CS, buildR1CS,
 * hex.DecodeString + ProvingKey.ReadFrom): a RawR1CS text carries the circuit AND this proof's values string, so the circuit is identified by the content
 * keys of the text before and after that string (same offset, same length); the key text by its content key.  Resident: the R1CS in HBM with the witness ->
 * wire order and the operands of every product variable (the wire vector of a later proof is assembled on the device from the values string alone), the
 * decoded key -- without window tables for its first proof, with them from its second on.  Same LRU bound as the PLONK entries (zk_export_cache_info counts both).
 *   zk_groth16_lower_resident  read a RawR1CS text into the cache now; to_device = 0: host only (start-up overlap: the export shim runs it beside the HIP
 *                              runtime's start), 1: also upload the circuit (beside the key's decoding)
 *   zk_groth16_key_resident    decode a key text into the cache now (no window tables)
 *   zk_groth16_public_inputs   buildWitnesses' public part for VerifyWithVK (r1cs.go:176-212): the values of the public wires after ONE, in wire order
 *                              (Montgomery); host only; *n_public is set even when cap is too small (ZK_ERR_ARG then). */
int zk_groth16_lower_resident(const char *raw_json, size_t raw_len, int to_device);
int zk_groth16_key_resident(const char *pk_hex, size_t pk_len);
int zk_groth16_public_inputs(const char *raw_json, size_t raw_len, zk_fr *out, size_t cap, size_t *n_public);

/* What the MSM planner picks for n points (with / without resident window tables): window width c and the number of c-bit
 * digits per scalar, i.e. mixed additions per scalar multiplication -- used by bench.py to turn launches into work. */
int zk_bn254_msm_plan_info(size_t n, int window_tables, uint32_t *window_bits, uint32_t *digits);

/* ---- felt-vector wire codec (the data format in front of the hot path) -------------------------------------------
 * The reference hands witness values to the Go side as hex( u32 BE count || count x 32 B BE canonical felts )
 * [REF src/gnark_backend_wrapper/serialize.rs:33-47,71-106; gnark_backend_ffi/internal/backend/helpers.go:24-33
 * DeserializeFelts = hex.DecodeString + fr.Vector.UnmarshalBinary].  One kernel decodes the hex text, reverses the byte
 * order, rejects non-canonical values like gnark-crypto ("invalid fr.Element encoding": >= r is an error, not reduced) and
 * converts to the Montgomery image, writing the vector straight into HBM -- ready to be the `w` of zk_bn254_groth16_prove
 * (on_device) or the scalars of zk_bn254_msm_bases.  Errors: ZK_ERR_LEN (length does not match the count), ZK_ERR_ARG
 * (bad hex character, non-canonical felt, capacity).
 *   decode_hex      : hex on the host -> d_out (device, capacity `cap` felts); *n_out = count.
 *   decode_hex_dev  : the whole text already in HBM (d_text must sit 8 bytes before a 16-byte boundary; n = its count).
 *   decode_bytes_dev: raw fr.Vector.MarshalBinary bytes in HBM (d_raw 4 bytes before a 16-byte boundary).
 *   encode_hex      : Montgomery vector in HBM -> 8 + 64 n hex characters on the host (encode_felts), no terminator. */
int zk_bn254_felts_decode_hex(const char *hex, size_t hex_len, void *d_out, size_t cap, size_t *n_out);
int zk_bn254_felts_decode_hex_dev(const void *d_text, size_t text_len, void *d_out, size_t cap, size_t n, int to_mont,
                                  void *stream);
int zk_bn254_felts_decode_bytes_dev(const void *d_raw, size_t raw_len, void *d_out, size_t cap, size_t n, int to_mont,
                                    void *stream);
int zk_bn254_felts_encode_hex(const void *d_in, size_t n, char *hex_out, size_t cap);

/* ---- synthetic data on the device (bench / tests; deterministic SplitMix64 streams) -------------- */
int zk_bn254_fr_random_dev(void *d_out, size_t n, uint64_t seed, int mont, int witness_like, void *stream);
int zk_bn254_g1_generate_dev(void *d_out, size_t n, uint64_t seed, void *stream); /* P_i = k_i * G1 */
int zk_bn254_g2_generate_dev(void *d_out, size_t n, uint64_t seed, void *stream); /* P_i = k_i * G2 */
int zk_bn254_fr_mul_dev(void *d_out, const void *d_a, const void *d_b, size_t n, void *stream); /* out = a*b (Montgomery) */

/* ---- kzg.NewSRS(size, alpha) (gnark-crypto ecc/bn254/fr/kzg; the reference: backend/common.go:137 with size 1_000_000, main.go:176):
 * d_g1_out[i] = alpha^i * G1 for i < size, on the device (register it with zk_bn254_bases_register_dev); g2_out = [G2, alpha * G2] on the
 * host.  alpha: Montgomery fr.Element (toxic waste: the caller's business, exactly as upstream's test-only constructor). */
int zk_bn254_kzg_new_srs_dev(void *d_g1_out, size_t size, const zk_fr *alpha, zk_g2_affine g2_out[2], void *stream);
/* kzg.SRS.ReadFrom / WriteTo (what LoadSRS / SaveSRS move through srs.hex: backend/common.go:86-125, re-read on every call at
 * backend/plonk/plonk.go:16,34,58): G2[0] | G2[1] (64 B compressed) | u32 BE count | count x 32 B compressed G1, as bytes or hex text.
 * _read decompresses the G1 points on the device (one square root each) straight into a registered base array (window tables per
 * table_window_bits as in zk_bn254_bases_register_cfg) and returns the two G2 points; _write is the inverse.  Errors: ZK_ERR_LEN (size
 * does not match the count), ZK_ERR_ARG (bad hex, bad flags, x >= q, no square root, G2 point outside the r-torsion). */
int zk_bn254_kzg_srs_read(const void *data, size_t len, int is_hex, int table_window_bits, uint64_t *handle, size_t *n_g1,
                          zk_g2_affine g2_out[2]);
int zk_bn254_kzg_srs_write(uint64_t handle, const zk_g2_affine g2[2], int as_hex, void *out, size_t cap, size_t *out_len);
/* The two G2 points of an SRS image, on the HOST: all that plonk.Verify needs of the SRS (backend/plonk/plonk.go:28-51 re-reads the whole file for it).  No
 * device work: a process that only verifies never starts the HIP runtime.  Header and length are checked as in _read; the G1 points are not looked at. */
int zk_bn254_kzg_srs_g2(const void *data, size_t len, int is_hex, zk_g2_affine g2_out[2]);

/* ---- device memory plumbing for hosts without a HIP binding (ctypes tests, the cgo shim) ------------------------ */
int zk_dev_alloc(void **d_ptr, size_t bytes);
int zk_dev_free(void *d_ptr);
int zk_dev_h2d(void *d_dst, const void *h_src, size_t bytes);
int zk_dev_d2h(void *h_dst, const void *d_src, size_t bytes);
int zk_dev_sync(void);

/* ---- per-kernel timing (hipEvent pairs on the stream the kernels run on; feeds bench.py's roofline object) ------- */
int zk_profile_enable(int on);
/* a host-side section of a caller above the C ABI (libgnark_backend.so's SRS handling), accounted beside the library's own; no-op unless profiling is on */
int zk_profile_host(const char *name, double ms);                 /* 1: record an event pair around every kernel launch */
int zk_profile_reset(void);
int zk_profile_count(void);                    /* number of distinct kernel names seen */
int zk_profile_get(int idx, char *name_out, size_t name_cap, uint64_t *launches, double *total_ms);

/* ---- host-only self test (runs WITHOUT a GPU): the shared 32-bit-limb field / XYZZ code vs the 64-bit host code;
 * returns the number of mismatches (0 = ok). */
int zk_selftest_host(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
