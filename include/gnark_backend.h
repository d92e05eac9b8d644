/* gnark_backend.h -- the OUTER boundary: the reference's ten Go exports, as C sees them.
 *
 * This is the header cgo generates for the reference's Go archive (`go build -buildmode=c-archive` writes libgnark_backend.h beside libgnark_backend.a;
 * /root/reference/build.rs:1-22 runs `make build-go` and links `static=gnark_backend`), written by hand for the library that takes the archive's place: noir_backend_using_gnark_amd/
 * libgnark_backend.so (or .a) built from csrc/goffi.cpp over libzkmi.  Names, argument order, argument meaning and error behaviour are the reference's;
 * the Rust side binds them unchanged at
 *     /root/reference/src/gnark_backend_wrapper/plonk/mod.rs:10-25      (PLONK: the reference's live path)
 *     /root/reference/src/gnark_backend_wrapper/groth16/mod.rs:14-20    (Groth16: the intended FFI; commented out on the Go side, backend/groth16/r1cs.go:74-266)
 * with the two structures of /root/reference/src/gnark_backend_wrapper/c_go_structures.rs:5-26.
 *
 * Go's C ABI for exported functions: a Go `string` argument is a GoString passed BY VALUE (pointer + length, no terminator needed); a `*C.char` result is
 * malloc'ed by C.CString and owned by the caller (the Rust side never frees it -- neither does anything here); a two-value result is a struct returned by
 * value; a Go `bool` is one byte.  Errors: the Go side calls log.Fatal -- the process ends with status 1 and the message on stderr; so does this library.
 *
 * Strings that cross: ACIR as JSON (acir/acir.go:17-75); RawR1CS as JSON (src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60); witness values and
 * public inputs as hex(u32 BE count | count x 32 B BE) (internal/backend/helpers.go:13-33; PlonkPreprocess receives that string JSON-quoted, main.go:66-72);
 * proofs and keys as hex(gnark WriteTo) (helpers.go:35-94).  The inner boundary -- the gnark-crypto call sites (MultiExp, FFT) and the provers on resident
 * data -- is include/zkmi.h. */
#ifndef GNARK_BACKEND_H
#define GNARK_BACKEND_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { const char *p; ptrdiff_t n; } GoString;                 /* cgo's _GoString_; Rust: GoString { ptr, length } (c_go_structures.rs:7-10) */
typedef struct { char *proving_key; char *verifying_key; } KeyPair;      /* cgo's struct for (*C.char, *C.char); Rust: KeyPair (c_go_structures.rs:22-26) */
typedef unsigned char GoUint8;                                           /* Go bool */

/* ---- PLONK (/root/reference/gnark_backend_ffi/main.go) */
char   *PlonkProveWithPK(GoString acirJSON, GoString encodedValues, GoString encodedProvingKey);                            /* main.go:24-37  -> hex(plonk.Proof.WriteTo), 1096 characters */
GoUint8 PlonkVerifyWithMeta(GoString acirJSON, GoString encodedValues, GoString encodedProof);                              /* main.go:39-42  upstream: `return false` */
GoUint8 PlonkVerifyWithVK(GoString acirJSON, GoString encodedProof, GoString encodedPublicInputs, GoString encodedVerifyingKey); /* main.go:44-56 */
KeyPair PlonkPreprocess(GoString acirJSON, GoString encodedRandomValues);                                                   /* main.go:58-78  encodedRandomValues is JSON-quoted */
char   *PlonkProveWithMeta(GoString acirJSON, GoString encodedValues);                                                      /* plonk/mod.rs:12; upstream a commented-out stub returning "Unimplemented" (backend/groth16/r1cs.go:268-271): here Preprocess + Prove */

/* ---- Groth16 (/root/reference/gnark_backend_ffi/backend/groth16/r1cs.go, commented out upstream; declared at groth16/mod.rs:14-20) */
char   *ProveWithMeta(GoString rawR1CS);                                                                                    /* r1cs.go:74-105  Setup + Prove */
char   *ProveWithPK(GoString rawR1CS, GoString encodedProvingKey);                                                          /* r1cs.go:107-143 -> hex(groth16.Proof.WriteTo), 256 characters */
GoUint8 VerifyWithMeta(GoString rawR1CS, GoString encodedProof);                                                            /* r1cs.go:145-187 here: false, like PLONK's */
GoUint8 VerifyWithVK(GoString rawR1CS, GoString encodedProof, GoString encodedVerifyingKey);                                /* r1cs.go:189-236 */
KeyPair Preprocess(GoString rawR1CS);                                                                                       /* r1cs.go:238-266 */

#ifdef __cplusplus
}
#endif
#endif
