// Felt-vector wire codec on the device: the format in which the reference hands witness values to the Go side,
//     hex( u32 big-endian count || count x 32-byte big-endian canonical felts )
// [REF src/gnark_backend_wrapper/serialize.rs:33-47 (encode_felts), :71-106 (deserialize_felts);
//  gnark_backend_ffi/internal/backend/helpers.go:24-33 (DeserializeFelts = hex.DecodeString + fr.Vector.UnmarshalBinary)].
// It is the data format immediately in front of the hot path: the decoded vector is the `w` of groth16 prove / the
// scalars of the KZG commits.  One kernel fuses hex decoding, byte-order reversal, the canonical-range check of
// gnark-crypto's BigEndian.Element (a value >= r is an error, it is NOT reduced) and the conversion to gnark's in-memory
// Montgomery image, writing straight into HBM.  Byte / integer work: 64 hex characters in, 32 bytes out per felt
// (algorithmic 96 B per felt, HBM-bound next to one Montgomery product per felt).
#include <string.h>

#include "ctx.hpp"
#include "ff.hpp"

namespace zkmi {

enum : int { WIRE_OK = 0, WIRE_BAD_HEX = 1, WIRE_NOT_CANONICAL = 2 };

// 4 hex characters (one little-endian word of the text) -> 2 bytes in text order (low byte = first pair); *bad |= invalid character
__device__ __forceinline__ uint32_t hex4(uint32_t w, uint32_t* bad) {
    uint32_t nib = (w & 0x0f0f0f0fu) + ((w >> 6) & 0x01010101u) * 9u;  // '0'-'9' -> 0-9, 'a'-'f' / 'A'-'F' -> 10-15
    // validation by re-encoding: the lower-case character of the nibble must equal the input character -- folded to lower case
    // (bit 5 set) ONLY where it is a letter (bit 6 set); digits are compared as they are, so that the control bytes 0x10-0x19
    // (which `| 0x20` would map onto '0'-'9') are rejected like hex.DecodeString rejects them.  Accepted set, checked over all
    // 256 byte values: '0'-'9', 'A'-'F', 'a'-'f'.
    uint32_t gt9 = ((nib + 0x06060606u) >> 4) & 0x01010101u;
    uint32_t enc = nib + 0x30303030u + gt9 * 0x27u;
    *bad |= (enc ^ (w | ((w >> 1) & 0x20202020u))) | (nib & 0xf0f0f0f0u);  // 'g'..'o' would give nibbles 16..24 that re-encode to themselves
    uint32_t b = ((nib << 4) | (nib >> 8)) & 0x00ff00ffu;  // bytes 0 and 2 hold (n0 n1), (n2 n3)
    return (b & 0xffu) | ((b >> 8) & 0xff00u);
}

__device__ __forceinline__ bool geq_mod_r(const uint32_t x[8]) {
    for (int i = 7; i >= 0; i--)
        if (x[i] != FrParams::MOD[i]) return x[i] > FrParams::MOD[i];
    return true;
}

// text: 64 hex characters per felt (the 8-character count header already skipped)
__global__ __launch_bounds__(256) void k_felts_decode_hex(const uint4* __restrict__ text, size_t n, int to_mont, Fr* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t bad = 0;
    Fr x;
#pragma unroll
    for (int q = 0; q < 4; q++) {  // 16 characters = 8 bytes = limbs 7-2q and 6-2q (big-endian text)
        uint4 t = text[i * 4 + q];
        uint32_t h0 = hex4(t.x, &bad), h1 = hex4(t.y, &bad), h2 = hex4(t.z, &bad), h3 = hex4(t.w, &bad);
        // h0 = bytes (B0, B1) with B0 in the low byte; the limb is B0 B1 B2 B3 big-endian
        x.l[7 - 2 * q] = (__builtin_bswap32(h0) >> 16 << 16) | (__builtin_bswap32(h1) >> 16);
        x.l[6 - 2 * q] = (__builtin_bswap32(h2) >> 16 << 16) | (__builtin_bswap32(h3) >> 16);
    }
    if (bad) { atomicMax(status, (int)WIRE_BAD_HEX); return; }
    if (geq_mod_r(x.l)) { atomicMax(status, (int)WIRE_NOT_CANONICAL); return; }
    out[i] = to_mont ? x.to_mont() : x;
}

// raw bytes (already hex-decoded): 32 big-endian bytes per felt
__global__ __launch_bounds__(256) void k_felts_decode_bytes(const uint4* __restrict__ raw, size_t n, int to_mont, Fr* __restrict__ out, int* __restrict__ status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = raw[i * 2], b = raw[i * 2 + 1];
    Fr x;
    x.l[7] = __builtin_bswap32(a.x); x.l[6] = __builtin_bswap32(a.y); x.l[5] = __builtin_bswap32(a.z); x.l[4] = __builtin_bswap32(a.w);
    x.l[3] = __builtin_bswap32(b.x); x.l[2] = __builtin_bswap32(b.y); x.l[1] = __builtin_bswap32(b.z); x.l[0] = __builtin_bswap32(b.w);
    if (geq_mod_r(x.l)) { atomicMax(status, (int)WIRE_NOT_CANONICAL); return; }
    out[i] = to_mont ? x.to_mont() : x;
}

__device__ __forceinline__ uint32_t hexenc2(uint32_t b16) {  // 2 bytes (text order, low byte first) -> 4 lower-case hex characters
    uint32_t nib = ((b16 >> 4) & 0x0fu) | ((b16 & 0x0fu) << 8) | (((b16 >> 12) & 0x0fu) << 16) | (((b16 >> 8) & 0x0fu) << 24);
    uint32_t gt9 = ((nib + 0x06060606u) >> 4) & 0x01010101u;
    return nib + 0x30303030u + gt9 * 0x27u;
}

__global__ __launch_bounds__(256) void k_felts_encode_hex(const Fr* __restrict__ in, size_t n, int from_mont, uint4* __restrict__ text) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = in[i];
    if (from_mont) x = x.from_mont();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t hi = __builtin_bswap32(x.l[7 - 2 * q]), lo = __builtin_bswap32(x.l[6 - 2 * q]);  // bytes in text order, low byte first
        text[i * 4 + q] = make_uint4(hexenc2(hi & 0xffffu), hexenc2(hi >> 16), hexenc2(lo & 0xffffu), hexenc2(lo >> 16));
    }
}

static int parse_count_hex(const char* hex, size_t hex_len, size_t* n) {
    if (hex_len < 8) return set_err(ZK_ERR_ARG, "felt vector: %zu characters cannot hold the 4-byte count", hex_len);
    uint32_t v = 0;
    for (int k = 0; k < 8; k++) {
        char c = hex[k];
        int d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
        if (d < 0) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character in the count");
        v = (v << 4) | (uint32_t)d;
    }
    *n = v;
    return ZK_OK;
}

static int status_to_rc(int st) {
    if (st == WIRE_BAD_HEX) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character");
    if (st == WIRE_NOT_CANONICAL) return set_err(ZK_ERR_ARG, "felt vector: invalid fr.Element encoding (value >= r)");
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// d_text: device copy of the WHOLE hex string (count header included), 16-byte aligned + 8 (i.e. the felts start 16-byte aligned) is
// not required: the kernel reads from d_text + 8, so d_text must be 8 bytes past a 16-byte boundary -- zk_bn254_felts_decode_hex
// stages the text that way.
}  // extern "C"
namespace zkmi { int felts_decode_hex_on_slot(Slot* s, hipStream_t st, const void* d_text, size_t text_len, void* d_out, size_t cap, size_t n, int to_mont, int* d_status); }
extern "C" {
int zk_bn254_felts_decode_hex_dev(const void* d_text, size_t text_len, void* d_out, size_t cap, size_t n, int to_mont, void* stream) {
    if (!d_text || (n && !d_out)) return set_err(ZK_ERR_ARG, "null pointer");
    if (!n && text_len == 8) return ZK_OK;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    ZK_TRY(g.s->reserve(256));
    return felts_decode_hex_on_slot(g.s, stream ? (hipStream_t)stream : g.s->stream, d_text, text_len, d_out, cap, n, to_mont, (int*)g.s->alloc(64));
}
}  // extern "C"
namespace zkmi {
// The same for a caller that already HOLDS a stream slot (the export path's wire assembly): no second slot is taken -- eight concurrent callers that each held
// one slot and waited for another would wait for ever -- and nothing of the slot's arena is touched: the status word is the caller's (4 bytes in HBM).
int felts_decode_hex_on_slot(Slot* s, hipStream_t st, const void* d_text, size_t text_len, void* d_out, size_t cap, size_t n, int to_mont, int* d_status) {
    if (!d_text || (n && !d_out) || !d_status) return set_err(ZK_ERR_ARG, "null pointer");
    if (text_len != 8 + 64 * n) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts (%zu characters)", text_len, n, 8 + 64 * n);
    if (n > cap) return set_err(ZK_ERR_ARG, "felt vector of %zu felts does not fit %zu", n, cap);
    if ((((uintptr_t)d_text) + 8) & 15) return set_err(ZK_ERR_ARG, "hex text must start 8 bytes before a 16-byte boundary");
    if (!n) return ZK_OK;
    ZK_HIP(hipMemsetAsync(d_status, 0, 4, st));
    ZK_LAUNCH(s, st, "felts_decode_hex", k_felts_decode_hex, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
              reinterpret_cast<const uint4*>((const char*)d_text + 8), n, to_mont, (Fr*)d_out, d_status);
    int h_status = 0;
    ZK_HIP(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(s, st));
    return status_to_rc(h_status);
}
}  // namespace zkmi
extern "C" {

// DeserializeFelts: hex string on the host -> Montgomery fr.Element vector in HBM (d_out, capacity cap); *n_out = count.
int zk_bn254_felts_decode_hex(const char* hex, size_t hex_len, void* d_out, size_t cap, size_t* n_out) {
    if (!hex || !n_out) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_TRY(ensure_init());
    size_t n = 0;
    ZK_TRY(parse_count_hex(hex, hex_len, &n));
    *n_out = n;
    if (hex_len != 8 + 64 * n) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts (%zu characters)", hex_len, n, 8 + 64 * n);
    if (n > cap) return set_err(ZK_ERR_ARG, "felt vector of %zu felts does not fit %zu", n, cap);
    if (!n) return ZK_OK;
    if (!d_out) return set_err(ZK_ERR_ARG, "null pointer");
    char* d_text = nullptr;
    ZK_HIP(hipMalloc((void**)&d_text, hex_len + 16));
    int rc = ZK_OK;
    if (hipMemcpy(d_text + 8, hex, hex_len, hipMemcpyHostToDevice) != hipSuccess) rc = set_err(ZK_ERR_HIP, "upload of the hex text failed");
    if (rc == ZK_OK) rc = zk_bn254_felts_decode_hex_dev(d_text + 8, hex_len, d_out, cap, n, 1, nullptr);
    (void)hipFree(d_text);
    return rc;
}

// fr.Vector.UnmarshalBinary on raw bytes resident in HBM: d_raw = u32 BE count || n x 32 B BE (count passed by the caller, who read it)
int zk_bn254_felts_decode_bytes_dev(const void* d_raw, size_t raw_len, void* d_out, size_t cap, size_t n, int to_mont, void* stream) {
    if (!d_raw || (n && !d_out)) return set_err(ZK_ERR_ARG, "null pointer");
    if (raw_len != 4 + 32 * n) return set_err(ZK_ERR_LEN, "felt vector: %zu bytes, the count says %zu felts (%zu bytes)", raw_len, n, 4 + 32 * n);
    if (n > cap) return set_err(ZK_ERR_ARG, "felt vector of %zu felts does not fit %zu", n, cap);
    if ((((uintptr_t)d_raw) + 4) & 15) return set_err(ZK_ERR_ARG, "raw vector must start 4 bytes before a 16-byte boundary");
    if (!n) return ZK_OK;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    ZK_TRY(g.s->reserve(256));
    int* d_status = (int*)g.s->alloc(64);
    ZK_HIP(hipMemsetAsync(d_status, 0, 4, st));
    ZK_LAUNCH(g.s, st, "felts_decode_bytes", k_felts_decode_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
              reinterpret_cast<const uint4*>((const char*)d_raw + 4), n, to_mont, (Fr*)d_out, d_status);
    int h_status = 0;
    ZK_HIP(hipMemcpyAsync(&h_status, d_status, 4, hipMemcpyDeviceToHost, st));
    ZK_TRY(slot_sync(g.s, st));
    return status_to_rc(h_status);
}

// encode_felts / fr.Vector.MarshalBinary + hex: Montgomery vector in HBM -> hex string on the host (8 + 64 n characters, no NUL)
int zk_bn254_felts_encode_hex(const void* d_in, size_t n, char* hex_out, size_t cap) {
    if ((n && !d_in) || !hex_out) return set_err(ZK_ERR_ARG, "null pointer");
    if (n >> 32) return set_err(ZK_ERR_ARG, "felt vector count %zu does not fit 32 bits", n);
    if (cap < 8 + 64 * n) return set_err(ZK_ERR_ARG, "output holds %zu characters, %zu needed", cap, 8 + 64 * n);
    ZK_TRY(ensure_init());
    static const char dig[] = "0123456789abcdef";
    for (int k = 0; k < 8; k++) hex_out[k] = dig[(n >> (28 - 4 * k)) & 15];
    if (!n) return ZK_OK;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    ZK_TRY(g.s->reserve(64 * n + 256));
    uint4* d_text = (uint4*)g.s->alloc(64 * n);
    ZK_LAUNCH(g.s, st, "felts_encode_hex", k_felts_encode_hex, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const Fr*)d_in, n, 1, d_text);
    ZK_HIP(hipMemcpyAsync(hex_out + 8, d_text, 64 * n, hipMemcpyDeviceToHost, st));
    return slot_sync(g.s, st);
}

}  // extern "C"
