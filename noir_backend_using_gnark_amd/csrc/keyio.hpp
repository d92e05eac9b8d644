// Wire formats of keys and the SRS decoded / encoded on the device (keyio.hip); used by the PLONK key reader in plonk.hip.
#pragma once
#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"

namespace zkmi {

// 2*n_bytes hex characters at d_text -> n_bytes bytes at d_out (both device); *d_status |= 1 on a character that is not a hex digit
int hex_decode_dev(Slot* s, hipStream_t st, const void* d_text, size_t n_bytes, void* d_out, int* d_status);
int hex_encode_dev(Slot* s, hipStream_t st, const void* d_bytes, size_t n_bytes, void* d_text);
// n x 32-byte big-endian canonical fr.Elements at d_raw (any 4-byte alignment) <-> Montgomery images; *d_status |= 2 on a value >= r
int fr_from_be_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status);
int fr_to_be_dev(Slot* s, hipStream_t st, const void* d_in, size_t n, void* d_raw);
// n x 32-byte compressed G1 points (G1Affine.Bytes()) <-> affine Montgomery images; *d_status |= 4 on an invalid encoding
int g1_decompress_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status);
int g1_compress_dev(Slot* s, hipStream_t st, const void* d_pts, size_t n, void* d_raw);
// n x 64-byte compressed G2 points <-> affine Montgomery images; *d_status |= 8 on an invalid encoding, |= 16 on a point outside the r-torsion.
// d_idx (compress only, may be NULL): gather -- point i of the output is d_pts[d_idx[i]]
// (takes n * G2_DECOMPRESS_SCRATCH bytes from the slot's arena, 256-byte aligned per call: the caller's reserve() includes them)
static constexpr size_t G2_DECOMPRESS_SCRATCH = 256;
int g2_decompress_dev(Slot* s, hipStream_t st, const void* d_raw, size_t n, void* d_out, int* d_status);
int g2_compress_dev(Slot* s, hipStream_t st, const void* d_pts, const uint32_t* d_idx, size_t n, void* d_raw);
int g1_compress_idx_dev(Slot* s, hipStream_t st, const void* d_pts, const uint32_t* d_idx, size_t n, void* d_raw);
// one byte per point of a wire-indexed array: 1 = the point at infinity (gnark's InfinityA / InfinityB)
int inf_flags_dev(Slot* s, hipStream_t st, int is_g2, const void* d_pts, size_t n, void* d_out);
// host: one compressed G2 point (G2Affine.Bytes()) -> affine; false on an invalid encoding / a point outside the r-torsion
bool g2_decompress_host(const uint8_t in[64], Affine<HFp2>* out);
// a loaded Groth16 proving key as the key writer sees it (groth16.hip); the pointers stay the key's
struct Groth16View {
    uint32_t log_domain;
    size_t n_wires, n_public, nz;
    Affine<HFp> alpha, beta, delta;
    Affine<HFp2> beta2, delta2;
    const void *d_a, *d_b, *d_k, *d_z, *d_b2;
};
int groth16_pk_view(uint64_t handle, Groth16View* v);
int groth16_pk_adopt(uint64_t handle);  // the key takes ownership of its five device base arrays
// registered bases: device pointer and count
int bases_ptr(uint64_t handle, const void** d, size_t* n, int* is_g2);

}  // namespace zkmi
