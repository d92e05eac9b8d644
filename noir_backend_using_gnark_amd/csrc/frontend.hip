// The callers either side of the hot path (SURVEY §8 rows f3 and the outer boundary of §8b): what the reference's Go shim does between the
// strings Noir hands over and gnark's prover -- restated on top of the device path, so that the reference's exported entry points have
// semantic equivalents here that never leave the GPU between the witness text and the proof bytes:
//     PlonkPreprocess(acirJSON, encodedValues)              -> (hex pk, hex vk)    /root/reference/gnark_backend_ffi/main.go:58-78
//     PlonkProveWithPK(acirJSON, encodedValues, encodedPK)  -> hex proof           /root/reference/gnark_backend_ffi/main.go:24-37
// through
//     acir.ACIR JSON                                  gnark_backend_ffi/acir/acir.go:17-75, opcode/arithmetic_opcode.go:18-83, term/*.go
//     BuildSparseR1CS / handleArithmeticOpcode        backend/plonk/sparse_r1cs.go:18-107 (one gate per arithmetic opcode: MulTerms[0] only;
//                                                     SimpleTerms of length 1 -> qO, 2 -> qL qR, 3 -> qL qR qO; directives / black boxes emit nothing)
//     HandleValues / BuildWitnesses                   backend/common.go:22-76 (public variables first, in witness order, then the secret ones)
//     DeserializeFelts                                internal/backend/helpers.go:24-33 (wire.hip, on the device)
//     Serialize / DeserializeProvingKey, VerifyingKey internal/backend/helpers.go:49-94 (plonk.hip / keyio.hip, on the device)
// Differences, on purpose: the SRS is a handle the caller keeps resident (the reference re-reads srs.hex on every call, plonk.go:16,34,58);
// failures are error codes (the reference log.Fatal()s); the nine blinding scalars can be pinned (NULL: drawn from the OS generator like
// upstream's fr.SetRandom).  Variable layout: ZK_ACIR_LAYOUT_REFERENCE (the default, what libgnark_backend.so uses) reproduces HandleValues
// literally -- with two or more public inputs it appends one secret variable per (witness, non-matching public input) and the gates use the
// LAST copy (common.go:59-68), so keys and proofs are interchangeable with the reference's; ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS is the layout
// without the duplicates (the two coincide for zero or one public input).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "ff.hpp"
#include "host_ff.hpp"

namespace zkmi {

// ------------------------------------------------------------------------------------------------ a small JSON reader
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const char* p;
    const char* end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool str(std::string* out) {
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        out->clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'u': {  // only the ASCII range can occur in this schema
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        out->push_back((char)(v & 0x7f));
                        p += 4;
                        break;
                    }
                    default: out->push_back(*p);
                }
                p++;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool value(JVal* v, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v->kind = JVal::OBJ;
            p++;
            ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!str(&k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                v->obj.emplace_back(k, JVal());
                if (!value(&v->obj.back().second, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v->kind = JVal::ARR;
            p++;
            ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                v->arr.emplace_back();
                if (!value(&v->arr.back(), depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v->kind = JVal::STR; return str(&v->str); }
        // the length test comes first: the text is a GoString payload, not NUL-terminated
        if (end - p >= 4 && !memcmp(p, "true", 4)) { v->kind = JVal::BOOL; v->b = true; p += 4; return true; }
        if (end - p >= 5 && !memcmp(p, "false", 5)) { v->kind = JVal::BOOL; p += 5; return true; }
        if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return true; }
        char* e = nullptr;
        std::string tmp(p, (size_t)(end - p) < 40 ? end : p + 40);
        v->num = strtod(tmp.c_str(), &e);
        if (e == tmp.c_str()) return fail("unexpected character");
        v->kind = JVal::NUM;
        p += e - tmp.c_str();
        return true;
    }
};

// ------------------------------------------------------------------------------------------------ ACIR -> gates
struct Gates {
    size_t n_public = 0, n_vars = 0;
    std::vector<HFr> ql, qr, qo, qm, qk;
    std::vector<uint32_t> xa, xb, xc;
    std::vector<uint32_t> order;  // variable k holds witness order[k] (1-based witness index - 1): the gather that builds the solution
};

// fr.Element.SetString on a hex literal of the ACIR (FieldElement: 64 hex characters big-endian, canonical or not: reduced mod r)
static bool felt_from_hex(const std::string& h, HFr* out) {
    if (h.size() > 64 || h.empty()) return false;
    uint8_t be[32] = {0};
    std::string s(64 - h.size(), '0');
    s += h;
    for (int i = 0; i < 32; i++) {
        auto hv = [](int c) { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; };
        int hi = hv(s[2 * i]), lo = hv(s[2 * i + 1]);
        if (hi < 0 || lo < 0) return false;
        be[i] = (uint8_t)((hi << 4) | lo);
    }
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | be[8 * (3 - i) + b];
        t[i] = v;
    }
    while (HFr::geq_mod(t)) HFr::sub_mod(t);
    *out = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
    return true;
}
static bool as_index(const JVal& v, uint32_t* out) {
    if (v.kind != JVal::NUM || v.num < 0 || v.num > 4294967295.0 || v.num != (double)(uint64_t)v.num) return false;
    *out = (uint32_t)v.num;
    return true;
}

// BuildSparseR1CS (sparse_r1cs.go:18-107) + HandleValues (common.go:45-76).  n_values = number of witness values handed over (witnesses 1..n).
static int lower_acir(const char* json, size_t len, size_t n_values, int layout, Gates* G) {
    if (layout != ZK_ACIR_LAYOUT_REFERENCE && layout != ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS) return set_err(ZK_ERR_ARG, "unknown ACIR variable layout %d", layout);
    JParser P{json, json + len, ""};
    JVal root;
    if (!P.value(&root) || root.kind != JVal::OBJ) return set_err(ZK_ERR_ARG, "ACIR JSON: %s", P.err.empty() ? "not an object" : P.err.c_str());
    const JVal* ops = root.get("opcodes");
    const JVal* pubs = root.get("public_inputs");
    if (!ops || ops->kind != JVal::ARR) return set_err(ZK_ERR_ARG, "ACIR JSON: no opcodes array");
    std::vector<uint32_t> pub;
    if (pubs && pubs->kind == JVal::ARR)
        for (auto& e : pubs->arr) {
            uint32_t w;
            if (!as_index(e, &w)) return set_err(ZK_ERR_ARG, "ACIR JSON: bad public input");
            pub.push_back(w);
        }
    // index[w] = variable of witness w (1-based); -1 = none
    std::vector<int64_t> index(n_values + 1, -1);
    G->order.clear();
    const bool exact = layout == ZK_ACIR_LAYOUT_REFERENCE;
    if (exact) {
        // HandleValues, literally (common.go:45-76).  Loop 1: one public variable per (witness, matching public input), in witness order.  Loop 2: with
        // public inputs, one SECRET variable per (witness, NON-matching public input) -- i.e. |P| copies of a private witness, |P| - 1 copies of a public
        // one -- and indexMap keeps the last index assigned, so that with |P| >= 2 every gate names a secret copy; without public inputs one secret
        // variable per witness.  cs.AddPublicVariable / AddSecretVariable number the variables public first, then secret, in the order of the calls.
        const size_t k = pub.size();
        size_t n_sec = 0;
        for (size_t w = 1; w <= n_values; w++) {
            size_t same = 0;
            for (uint32_t p : pub) same += p == w;
            n_sec += k ? k - same : 1;
        }
        if (n_sec + n_values * k >= ((size_t)1 << 31)) return set_err(ZK_ERR_ARG, "HandleValues: %zu witnesses x %zu public inputs make too many variables", n_values, k);
        for (size_t w = 1; w <= n_values; w++)
            for (uint32_t p : pub)
                if (p == w) { index[w] = (int64_t)G->order.size(); G->order.push_back((uint32_t)(w - 1)); }
        G->n_public = G->order.size();
        for (size_t w = 1; w <= n_values; w++) {
            if (k) {
                for (uint32_t p : pub)
                    if (p != w) { index[w] = (int64_t)G->order.size(); G->order.push_back((uint32_t)(w - 1)); }
            } else {
                index[w] = (int64_t)G->order.size();
                G->order.push_back((uint32_t)(w - 1));
            }
        }
    } else {
        // one variable per witness: public witnesses first (in witness order), then the others
        std::vector<bool> is_pub(n_values + 1, false);
        for (uint32_t w : pub)
            if (w >= 1 && w <= n_values) is_pub[w] = true;
        for (size_t w = 1; w <= n_values; w++)
            if (is_pub[w]) { index[w] = (int64_t)G->order.size(); G->order.push_back((uint32_t)(w - 1)); }
        G->n_public = G->order.size();
        for (size_t w = 1; w <= n_values; w++)
            if (!is_pub[w]) { index[w] = (int64_t)G->order.size(); G->order.push_back((uint32_t)(w - 1)); }
    }
    G->n_vars = G->order.size();
    auto var_of = [&](const JVal& v, uint32_t* out) -> bool {
        uint32_t w;
        if (!as_index(v, &w)) return false;
        if (w < 1 || w > n_values || index[w] < 0) {
            if (!exact) return false;
            *out = 0;  // the reference's map lookup of a witness that has no variable yields the zero value: variable 0 (sparse_r1cs.go:53-54)
            return true;
        }
        *out = (uint32_t)index[w];
        return true;
    };
    for (auto& op : ops->arr) {
        if (op.kind != JVal::OBJ) return set_err(ZK_ERR_ARG, "ACIR JSON: opcode is not an object");
        const JVal* a = op.get("Arithmetic");
        if (!a) {
            if (op.get("Directive") || op.get("BlackBoxFuncCall")) continue;  // no constraints (sparse_r1cs.go:33-37)
            return set_err(ZK_ERR_ARG, "unknown opcode type");
        }
        const JVal *mul = a->get("mul_terms"), *lin = a->get("linear_combinations"), *qc = a->get("q_c");
        if (!mul || !lin || !qc || mul->kind != JVal::ARR || lin->kind != JVal::ARR || qc->kind != JVal::STR) return set_err(ZK_ERR_ARG, "ACIR JSON: malformed arithmetic opcode");
        HFr ql = HFr::zero(), qr = ql, qo = ql, qm = ql, qk;
        uint32_t xa = 0, xb = 0, xc = 0;
        if (!mul->arr.empty()) {  // qM * (xa * xb): only the first mul term
            const JVal& t = mul->arr[0];
            if (t.kind != JVal::ARR || t.arr.size() != 3 || t.arr[0].kind != JVal::STR || !felt_from_hex(t.arr[0].str, &qm) || !var_of(t.arr[1], &xa) || !var_of(t.arr[2], &xb))
                return set_err(ZK_ERR_ARG, "ACIR JSON: malformed mul term");
        }
        auto term = [&](const JVal& t, HFr* c, uint32_t* x) -> bool {
            return t.kind == JVal::ARR && t.arr.size() == 2 && t.arr[0].kind == JVal::STR && felt_from_hex(t.arr[0].str, c) && var_of(t.arr[1], x);
        };
        const size_t nl = lin->arr.size();
        bool ok = true;
        if (nl == 1) ok = term(lin->arr[0], &qo, &xc);
        else if (nl == 2 || nl == 3) {
            ok = term(lin->arr[0], &ql, &xa) && term(lin->arr[1], &qr, &xb);
            if (ok && nl == 3) ok = term(lin->arr[2], &qo, &xc);
        }
        if (!ok || !felt_from_hex(qc->str, &qk)) return set_err(ZK_ERR_ARG, "ACIR JSON: malformed linear combination / q_c");
        G->ql.push_back(ql); G->qr.push_back(qr); G->qo.push_back(qo); G->qm.push_back(qm); G->qk.push_back(qk);
        G->xa.push_back(xa); G->xb.push_back(xb); G->xc.push_back(xc);
    }
    return ZK_OK;
}

__global__ void k_gather_fr(const Fr* __restrict__ vals, const uint32_t* __restrict__ order, size_t n, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = vals[order[i]];
}

static int count_from_hex(const char* hex, size_t len, size_t* n) {
    if (len < 8) return set_err(ZK_ERR_ARG, "felt vector: %zu characters cannot hold the 4-byte count", len);
    size_t v = 0;
    for (int k = 0; k < 8; k++) {
        int c = hex[k], d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
        if (d < 0) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character in the count");
        v = (v << 4) | (size_t)d;
    }
    *n = v;
    return ZK_OK;
}

static int circuit_of(const Gates& G, zk_plonk_circuit* c) {
    memset(c, 0, sizeof *c);
    c->n_public = G.n_public;
    c->n_constraints = G.xa.size();
    c->n_vars = G.n_vars;
    c->ql = G.ql.data(); c->qr = G.qr.data(); c->qo = G.qo.data(); c->qm = G.qm.data(); c->qk = G.qk.data();
    c->xa = G.xa.data(); c->xb = G.xb.data(); c->xc = G.xc.data();
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// PlonkPreprocess (main.go:58-78): ACIR + the witness-value vector (only its length and the public/secret split matter to Setup) -> hex of
// ProvingKey.WriteTo and hex of VerifyingKey.WriteTo (= the first 368 bytes of the former).  pk_hex_out may be NULL with pk_cap = 0 to query
// the sizes.  The key also stays resident: *pk_handle (optional) can be handed to zk_bn254_plonk_prove directly.
int zk_plonk_preprocess(const char* acir_json, size_t acir_len, const char* values_hex, size_t values_len, int layout, uint64_t srs_handle, char* pk_hex_out,
                        size_t pk_cap, size_t* pk_len, char* vk_hex_out, size_t vk_cap, size_t* vk_len, uint64_t* pk_handle) {
    if (!acir_json || !values_hex || !pk_len || !vk_len) return set_err(ZK_ERR_ARG, "null pointer");
    size_t n_values = 0;
    ZK_TRY(count_from_hex(values_hex, values_len, &n_values));
    if (values_len != 8 + 64 * n_values) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts", values_len, n_values);
    Gates G;
    ZK_TRY(lower_acir(acir_json, acir_len, n_values, layout, &G));
    // sizes first (a call with pk_hex_out == NULL only asks for them): domain n = next power of two >= gates + public inputs
    size_t n = 1;
    while (n < G.xa.size() + G.n_public) n <<= 1;
    *pk_len = 2 * (704 + 9 * (4 + 32 * n) + 24 * n);
    *vk_len = 2 * 368;
    if (!pk_hex_out) return ZK_OK;
    if (pk_cap < *pk_len || (vk_hex_out && vk_cap < *vk_len)) return set_err(ZK_ERR_ARG, "outputs hold %zu / %zu characters, %zu / %zu needed", pk_cap, vk_cap, *pk_len, *vk_len);
    zk_plonk_circuit c;
    circuit_of(G, &c);
    uint64_t h = 0;
    ZK_TRY(zk_bn254_plonk_setup(&c, srs_handle, &h, nullptr));
    int rc = zk_bn254_plonk_pk_write(h, 1, pk_hex_out, pk_cap, pk_len);
    if (rc == ZK_OK && vk_hex_out) memcpy(vk_hex_out, pk_hex_out, *vk_len);  // ProvingKey.WriteTo starts with VerifyingKey.WriteTo
    if (pk_handle && rc == ZK_OK) *pk_handle = h;
    else (void)zk_bn254_plonk_pk_free(h);
    return rc;
}

// PlonkProveWithPK (main.go:24-37): ACIR + hex witness values + hex proving key -> hex of Proof.WriteTo (2 * 548 characters, no terminator).
// pk_hex may be NULL when pk_handle names a resident key (zk_plonk_preprocess / zk_bn254_plonk_pk_read) -- the reference deserialises the
// key on every call.  blinders: 9 scalars or NULL (drawn from /dev/urandom, as upstream draws them with fr.SetRandom).
int zk_plonk_prove_with_pk(const char* acir_json, size_t acir_len, const char* values_hex, size_t values_len, int layout, const char* pk_hex, size_t pk_len,
                           uint64_t pk_handle, uint64_t srs_handle, const zk_fr* blinders, char proof_hex_out[2 * ZK_PLONK_PROOF_BYTES]) {
    if (!acir_json || !values_hex || !proof_hex_out || (!pk_hex && !pk_handle)) return set_err(ZK_ERR_ARG, "null pointer");
    size_t n_values = 0;
    ZK_TRY(count_from_hex(values_hex, values_len, &n_values));
    Gates G;
    ZK_TRY(lower_acir(acir_json, acir_len, n_values, layout, &G));
    ZK_TRY(ensure_init());
    // witness values: DeserializeFelts on the device, then the public-first gather (BuildWitnesses)
    void *d_vals = nullptr, *d_sol = nullptr, *d_order = nullptr;
    struct Free { void** p[3]; ~Free() { for (auto q : p) if (*q) (void)hipFree(*q); } } guard{{&d_vals, &d_sol, &d_order}};
    ZK_HIP(hipMalloc(&d_vals, (n_values ? n_values : 1) * 32));
    ZK_HIP(hipMalloc(&d_sol, (G.n_vars ? G.n_vars : 1) * 32));
    ZK_HIP(hipMalloc(&d_order, (G.n_vars ? G.n_vars : 1) * 4));
    size_t n_dec = 0;
    ZK_TRY(zk_bn254_felts_decode_hex(values_hex, values_len, d_vals, n_values ? n_values : 1, &n_dec));
    {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->stream;
        if (G.n_vars) {
            ZK_HIP(hipMemcpyAsync(d_order, G.order.data(), G.n_vars * 4, hipMemcpyHostToDevice, st));
            ZK_LAUNCH(g.s, st, "witness_gather", k_gather_fr, dim3((unsigned)((G.n_vars + 255) / 256)), dim3(256), 0, (const Fr*)d_vals, (const uint32_t*)d_order, G.n_vars, (Fr*)d_sol);
        }
        ZK_TRY(slot_sync(g.s, st));
    }
    uint64_t h = pk_handle;
    if (pk_hex) ZK_TRY(zk_bn254_plonk_pk_read(pk_hex, pk_len, 1, G.n_vars, G.xa.size(), G.xa.data(), G.xb.data(), G.xc.data(), srs_handle, &h));
    {   // the key must be the key of THIS circuit: same public / variable / gate counts (a key of another shape would read the solution out of bounds)
        size_t kp = 0, kv = 0, kc = 0;
        int rc = zk_bn254_plonk_pk_info(h, nullptr, &kp, &kc, &kv);
        if (rc == ZK_OK && (kp != G.n_public || kv != G.n_vars || kc != G.xa.size()))
            rc = set_err(ZK_ERR_ARG, "proving key is for %zu public / %zu variables / %zu gates, the circuit has %zu / %zu / %zu", kp, kv, kc, G.n_public, G.n_vars, G.xa.size());
        if (rc != ZK_OK) {
            if (pk_hex) (void)zk_bn254_plonk_pk_free(h);
            return rc;
        }
    }
    zk_fr rnd[9];
    if (!blinders) {
        FILE* f = fopen("/dev/urandom", "rb");
        uint8_t raw[9 * 32];
        if (!f || fread(raw, 1, sizeof raw, f) != sizeof raw) {
            if (f) fclose(f);
            if (pk_hex) (void)zk_bn254_plonk_pk_free(h);
            return set_err(ZK_ERR_ARG, "no randomness source for the blinding scalars");
        }
        fclose(f);
        for (int i = 0; i < 9; i++) {
            uint64_t t[4];
            memcpy(t, raw + 32 * i, 32);
            t[3] &= 0x3fffffffffffffffULL;  // < 2^254, then one conditional subtraction: uniform enough for blinding (upstream: rejection sampling)
            while (HFr::geq_mod(t)) HFr::sub_mod(t);
            HFr m = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
            memcpy(&rnd[i], &m, 32);
        }
        blinders = rnd;
    }
    uint8_t proof[ZK_PLONK_PROOF_BYTES];
    int rc = zk_bn254_plonk_prove(h, d_sol, G.n_vars, 1, blinders, nullptr, proof);
    if (pk_hex) (void)zk_bn254_plonk_pk_free(h);
    ZK_TRY(rc);
    static const char dig[] = "0123456789abcdef";
    for (size_t i = 0; i < ZK_PLONK_PROOF_BYTES; i++) {
        proof_hex_out[2 * i] = dig[proof[i] >> 4];
        proof_hex_out[2 * i + 1] = dig[proof[i] & 15];
    }
    return ZK_OK;
}

// buildR1CS of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:9-72, commented out there; payload RawR1CS of
// src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60: {"gates":[{"mul_terms":[{"coefficient","multiplicand","multiplier"}],"add_terms":
// [{"coefficient","sum"}],"constant_term"}],"public_inputs","values" (hex felt vector),"num_variables","num_constraints"}): every mul term gets an
// internal product variable p with (1 * multiplicand) * (1 * multiplier) = 1 * p; every gate ends in
// (1 * ONE) * (sum coefficient * p + sum coefficient * x + constant * ONE) = 0.  Made well-defined where the sketch is not: wires = [ONE, public
// witnesses in witness order, the other witnesses, product variables]; values[w - 1] is witness w; the product variable is the plain product (the
// sketch puts the coefficient on the product constraint's output AND on the term, which cancels it); the constant term IS in the sum (the sketch
// drops it); a mul term with coefficient 0 emits nothing.  Out: a resident R1CS (zk_bn254_r1cs_*) and the full wire vector in HBM
// (*d_witness: n_wires Montgomery elements, to be released with zk_dev_free) -- ready for zk_bn254_groth16_setup / _prove_r1cs(on_device = 1).
int zk_groth16_r1cs_from_raw(const char* raw_json, size_t len, uint64_t* r1cs_handle, void** d_witness, size_t* n_wires, size_t* n_public) {
    if (!raw_json || !r1cs_handle || !d_witness) return set_err(ZK_ERR_ARG, "null pointer");
    JParser P{raw_json, raw_json + len, ""};
    JVal root;
    if (!P.value(&root) || root.kind != JVal::OBJ) return set_err(ZK_ERR_ARG, "RawR1CS JSON: %s", P.err.empty() ? "not an object" : P.err.c_str());
    const JVal *gates = root.get("gates"), *pubs = root.get("public_inputs"), *vals = root.get("values");
    if (!gates || gates->kind != JVal::ARR || !vals || vals->kind != JVal::STR) return set_err(ZK_ERR_ARG, "RawR1CS JSON: gates / values missing");
    // witness values: hex felt vector, decoded on the host here (they feed the host-side solver step for the product variables)
    const std::string& vh = vals->str;
    size_t n = 0;
    ZK_TRY(count_from_hex(vh.data(), vh.size(), &n));
    if (vh.size() != 8 + 64 * n) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts", vh.size(), n);
    std::vector<HFr> wv(1, HFr::one());
    std::vector<bool> is_pub(n + 1, false);
    if (pubs && pubs->kind == JVal::ARR)
        for (auto& e : pubs->arr) {
            uint32_t w;
            if (!as_index(e, &w)) return set_err(ZK_ERR_ARG, "RawR1CS JSON: bad public input");
            if (w >= 1 && w <= n) is_pub[w] = true;
        }
    std::vector<uint32_t> wire(n + 1, 0);
    size_t npub = 1;
    auto felt_at = [&](size_t w, HFr* out) -> bool {  // canonical values only, like fr.Vector.UnmarshalBinary
        uint64_t t[4];
        for (int i = 0; i < 4; i++) {
            uint64_t v = 0;
            for (int b = 0; b < 16; b++) {
                int c = vh[8 + 64 * (w - 1) + 16 * (3 - i) + b], d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
                if (d < 0) return false;
                v = (v << 4) | (uint64_t)d;
            }
            t[i] = v;
        }
        if (HFr::geq_mod(t)) return false;
        *out = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
        return true;
    };
    for (int pass = 0; pass < 2; pass++)
        for (size_t w = 1; w <= n; w++)
            if (is_pub[w] == (pass == 0)) {
                HFr v;
                if (!felt_at(w, &v)) return set_err(ZK_ERR_ARG, "felt vector: invalid hex character or fr.Element encoding");
                wire[w] = (uint32_t)wv.size();
                wv.push_back(v);
                if (pass == 0) npub++;
            }
    auto wire_of = [&](const JVal* v, uint32_t* out) -> bool {
        uint32_t w;
        if (!v || !as_index(*v, &w) || w < 1 || w > n) return false;
        *out = wire[w];
        return true;
    };
    std::vector<uint32_t> ptr[3], idx[3];
    std::vector<HFr> val[3];
    for (int m = 0; m < 3; m++) ptr[m].push_back(0);
    auto end_row = [&]() { for (int m = 0; m < 3; m++) ptr[m].push_back((uint32_t)idx[m].size()); };
    const HFr one = HFr::one();
    for (auto& g : gates->arr) {
        const JVal *mt = g.get("mul_terms"), *at = g.get("add_terms"), *kt = g.get("constant_term");
        if (g.kind != JVal::OBJ || !mt || !at || !kt || mt->kind != JVal::ARR || at->kind != JVal::ARR || kt->kind != JVal::STR) return set_err(ZK_ERR_ARG, "RawR1CS JSON: malformed gate");
        std::vector<std::pair<uint32_t, HFr>> terms;
        auto add_term = [&](uint32_t x, const HFr& c) {
            for (auto& t : terms)
                if (t.first == x) { t.second = t.second + c; return; }
            terms.emplace_back(x, c);
        };
        for (auto& t : mt->arr) {
            const JVal* cj = t.get("coefficient");
            HFr c;
            uint32_t a, b;
            if (t.kind != JVal::OBJ || !cj || cj->kind != JVal::STR || !felt_from_hex(cj->str, &c) || !wire_of(t.get("multiplicand"), &a) || !wire_of(t.get("multiplier"), &b))
                return set_err(ZK_ERR_ARG, "RawR1CS JSON: malformed mul term");
            if (c.is_zero()) continue;
            const uint32_t p = (uint32_t)wv.size();
            wv.push_back(wv[a] * wv[b]);  // the solver's step for this internal variable
            idx[0].push_back(a); val[0].push_back(one);
            idx[1].push_back(b); val[1].push_back(one);
            idx[2].push_back(p); val[2].push_back(one);
            end_row();
            add_term(p, c);
        }
        for (auto& t : at->arr) {
            const JVal* cj = t.get("coefficient");
            HFr c;
            uint32_t x;
            if (t.kind != JVal::OBJ || !cj || cj->kind != JVal::STR || !felt_from_hex(cj->str, &c) || !wire_of(t.get("sum"), &x)) return set_err(ZK_ERR_ARG, "RawR1CS JSON: malformed add term");
            add_term(x, c);
        }
        HFr k;
        if (!felt_from_hex(kt->str, &k)) return set_err(ZK_ERR_ARG, "RawR1CS JSON: malformed constant term");
        if (!k.is_zero()) add_term(0, k);
        idx[0].push_back(0); val[0].push_back(one);
        for (auto& t : terms) { idx[1].push_back(t.first); val[1].push_back(t.second); }
        end_row();
    }
    zk_r1cs r;
    memset(&r, 0, sizeof r);
    r.n_constraints = ptr[0].size() - 1;
    r.n_wires = wv.size();
    r.n_public = npub;
    r.l_ptr = ptr[0].data(); r.l_idx = idx[0].data(); r.l_val = (const zk_fr*)val[0].data();
    r.r_ptr = ptr[1].data(); r.r_idx = idx[1].data(); r.r_val = (const zk_fr*)val[1].data();
    r.o_ptr = ptr[2].data(); r.o_idx = idx[2].data(); r.o_val = (const zk_fr*)val[2].data();
    ZK_TRY(zk_bn254_r1cs_load(&r, r1cs_handle));
    void* d = nullptr;
    int rc = zk_dev_alloc(&d, wv.size() * 32);
    if (rc == ZK_OK) rc = zk_dev_h2d(d, wv.data(), wv.size() * 32);
    if (rc != ZK_OK) {
        if (d) (void)zk_dev_free(d);
        (void)zk_bn254_r1cs_free(*r1cs_handle);
        return rc;
    }
    *d_witness = d;
    if (n_wires) *n_wires = wv.size();
    if (n_public) *n_public = npub;
    return ZK_OK;
}

// uniform-enough field elements for prover randomness / toxic waste (upstream: fr.SetRandom's rejection sampling over crypto/rand)
static int random_frs(HFr* out, int n, bool nonzero) {
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return set_err(ZK_ERR_ARG, "no randomness source");
    for (int i = 0; i < n; i++) {
        uint64_t t[4];
        do {
            if (fread(t, 1, 32, f) != 32) { fclose(f); return set_err(ZK_ERR_ARG, "no randomness source"); }
            t[3] &= 0x3fffffffffffffffULL;
        } while (HFr::geq_mod(t) || (nonzero && !(t[0] | t[1] | t[2] | t[3])));  // rejection: r > 2^253, at most ~1.3 draws on average
        out[i] = HFr{{t[0], t[1], t[2], t[3]}}.to_mont();
    }
    fclose(f);
    return ZK_OK;
}

namespace {
struct RawInstance {  // a RawR1CS lowered and resident; released on scope exit
    uint64_t r1cs = 0;
    void* d_w = nullptr;
    size_t n_wires = 0, n_public = 0;
    ~RawInstance() {
        if (d_w) (void)zk_dev_free(d_w);
        if (r1cs) (void)zk_bn254_r1cs_free(r1cs);
    }
};
void hex_of(const uint8_t* b, size_t n, char* o) {
    static const char* dg = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) { o[2 * i] = dg[b[i] >> 4]; o[2 * i + 1] = dg[b[i] & 15]; }
}
}  // namespace

// Preprocess of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:214-266): RawR1CS JSON -> groth16.Setup -> hex(ProvingKey.WriteTo),
// hex(VerifyingKey.WriteTo).  toxic: tau, alpha, beta, gamma, delta (Montgomery, non-zero) or NULL (/dev/urandom, as upstream draws them).
// pk_hex_out == NULL: only the sizes (the key is built to learn NbInfinityA / NbInfinityB).  pk_handle (optional) keeps the key resident.
int zk_groth16_preprocess(const char* raw_json, size_t raw_len, const zk_fr* toxic, char* pk_hex_out, size_t pk_cap, size_t* pk_len, char* vk_hex_out, size_t vk_cap,
                          size_t* vk_len, uint64_t* pk_handle) {
    if (!raw_json || !pk_len || !vk_len) return set_err(ZK_ERR_ARG, "null pointer");
    RawInstance I;
    ZK_TRY(zk_groth16_r1cs_from_raw(raw_json, raw_len, &I.r1cs, &I.d_w, &I.n_wires, &I.n_public));
    HFr tx[5];
    if (toxic) memcpy(tx, toxic, sizeof tx);
    else ZK_TRY(random_frs(tx, 5, true));
    std::vector<zk_g1_affine> vk_g1(1 + I.n_public);
    zk_g2_affine vk_g2[3];
    uint64_t h = 0;
    ZK_TRY(zk_bn254_groth16_setup(I.r1cs, (const zk_fr*)tx, 0, &h, vk_g1.data(), vk_g2));
    int rc = zk_bn254_groth16_pk_write(h, 1, pk_hex_out, pk_cap, pk_len);
    if (rc == ZK_OK) rc = zk_bn254_groth16_vk_write(h, vk_g1.data(), I.n_public, vk_g2, 1, pk_hex_out ? vk_hex_out : nullptr, vk_cap, vk_len);
    if (rc == ZK_OK && pk_handle && pk_hex_out) *pk_handle = h;
    else (void)zk_bn254_groth16_pk_free(h);
    return rc;
}

// ProveWithPK (r1cs.go:107-143): RawR1CS JSON + hex(ProvingKey.WriteTo) -> hex(Proof.WriteTo) (256 characters, no terminator).  pk_hex may be NULL when
// pk_handle names a resident key (the reference deserialises the key on every call).  rs: the prover's (r, s) or NULL (/dev/urandom).
int zk_groth16_prove_with_pk(const char* raw_json, size_t raw_len, const char* pk_hex, size_t pk_len, uint64_t pk_handle, const zk_fr* rs, char proof_hex_out[256]) {
    if (!raw_json || !proof_hex_out || (!pk_hex && !pk_handle)) return set_err(ZK_ERR_ARG, "null pointer");
    RawInstance I;
    ZK_TRY(zk_groth16_r1cs_from_raw(raw_json, raw_len, &I.r1cs, &I.d_w, &I.n_wires, &I.n_public));
    uint64_t h = pk_handle;
    if (pk_hex) ZK_TRY(zk_bn254_groth16_pk_read(pk_hex, pk_len, 1, 0, 0, &h));
    HFr r2[2];
    int rc = ZK_OK;
    if (rs) memcpy(r2, rs, sizeof r2);
    else rc = random_frs(r2, 2, false);
    uint8_t proof[128];
    if (rc == ZK_OK) rc = zk_bn254_groth16_prove_r1cs(I.r1cs, h, I.d_w, I.n_wires, (const zk_fr*)&r2[0], (const zk_fr*)&r2[1], 1, proof);
    if (pk_hex) (void)zk_bn254_groth16_pk_free(h);
    if (rc == ZK_OK) hex_of(proof, 128, proof_hex_out);
    return rc;
}

// ProveWithMeta (r1cs.go:74-105): Setup and Prove in one call (the proving key never leaves HBM).
int zk_groth16_prove_with_meta(const char* raw_json, size_t raw_len, const zk_fr* toxic, const zk_fr* rs, char proof_hex_out[256]) {
    if (!raw_json || !proof_hex_out) return set_err(ZK_ERR_ARG, "null pointer");
    RawInstance I;
    ZK_TRY(zk_groth16_r1cs_from_raw(raw_json, raw_len, &I.r1cs, &I.d_w, &I.n_wires, &I.n_public));
    HFr tx[5], r2[2];
    if (toxic) memcpy(tx, toxic, sizeof tx);
    else ZK_TRY(random_frs(tx, 5, true));
    if (rs) memcpy(r2, rs, sizeof r2);
    else ZK_TRY(random_frs(r2, 2, false));
    uint64_t h = 0;
    ZK_TRY(zk_bn254_groth16_setup(I.r1cs, (const zk_fr*)tx, 1, &h, nullptr, nullptr));  // one proof: window tables would cost more than they save
    uint8_t proof[128];
    int rc = zk_bn254_groth16_prove_r1cs(I.r1cs, h, I.d_w, I.n_wires, (const zk_fr*)&r2[0], (const zk_fr*)&r2[1], 1, proof);
    (void)zk_bn254_groth16_pk_free(h);
    if (rc == ZK_OK) hex_of(proof, 128, proof_hex_out);
    return rc;
}

// The lowering alone, for inspection / tests: gates of an ACIR circuit as the reference's BuildSparseR1CS emits them.  Any out pointer may be
// NULL; arrays need *n_constraints (first call with NULL arrays to size them) entries; coefficients come back as Montgomery fr.Elements.
int zk_acir_to_sparse_r1cs(const char* acir_json, size_t acir_len, size_t n_values, int layout, size_t* n_public, size_t* n_vars, size_t* n_constraints, zk_fr* ql,
                           zk_fr* qr, zk_fr* qo, zk_fr* qm, zk_fr* qk, uint32_t* xa, uint32_t* xb, uint32_t* xc, uint32_t* order) {
    if (!acir_json) return set_err(ZK_ERR_ARG, "null pointer");
    Gates G;
    ZK_TRY(lower_acir(acir_json, acir_len, n_values, layout, &G));
    if (n_public) *n_public = G.n_public;
    if (n_vars) *n_vars = G.n_vars;
    if (n_constraints) *n_constraints = G.xa.size();
    const size_t nc = G.xa.size();
    if (ql) memcpy(ql, G.ql.data(), nc * 32);
    if (qr) memcpy(qr, G.qr.data(), nc * 32);
    if (qo) memcpy(qo, G.qo.data(), nc * 32);
    if (qm) memcpy(qm, G.qm.data(), nc * 32);
    if (qk) memcpy(qk, G.qk.data(), nc * 32);
    if (xa) memcpy(xa, G.xa.data(), nc * 4);
    if (xb) memcpy(xb, G.xb.data(), nc * 4);
    if (xc) memcpy(xc, G.xc.data(), nc * 4);
    if (order) memcpy(order, G.order.data(), G.n_vars * 4);
    return ZK_OK;
}

}  // extern "C"
