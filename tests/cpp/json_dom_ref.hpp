// TEST INFRASTRUCTURE (not product code): the document-tree JSON reader and the two lowerings written on top of it that the product used until
// round 3, kept verbatim as the DIFFERENTIAL CHECKER of the streaming front end (noir_backend_using_gnark_amd/csrc/acir_host.hpp) in
// tests/cpp/parser_fuzz.cpp: for every mutated input both must agree on accept / reject, on the status code and on every output word.
// Semantics: /root/reference/gnark_backend_ffi/backend/plonk/sparse_r1cs.go:18-107, backend/common.go:45-76, backend/groth16/r1cs.go:9-72.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/zkmi.h"
#include "../../noir_backend_using_gnark_amd/csrc/host_ff.hpp"

namespace domref {
using zkmi::HFr;
static std::string g_err;
static int set_err(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define DOMREF_TRY(expr) do { int _rc = (expr); if (_rc != ZK_OK) return _rc; } while (0)

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

struct RawBuilt { std::vector<uint32_t> ptr[3], idx[3]; std::vector<HFr> val[3]; std::vector<HFr> wires; size_t n_public = 0; };
static int raw_r1cs_build(const char* raw_json, size_t len, RawBuilt* B) {
    JParser P{raw_json, raw_json + len, ""};
    JVal root;
    if (!P.value(&root) || root.kind != JVal::OBJ) return set_err(ZK_ERR_ARG, "RawR1CS JSON: %s", P.err.empty() ? "not an object" : P.err.c_str());
    const JVal *gates = root.get("gates"), *pubs = root.get("public_inputs"), *vals = root.get("values");
    if (!gates || gates->kind != JVal::ARR || !vals || vals->kind != JVal::STR) return set_err(ZK_ERR_ARG, "RawR1CS JSON: gates / values missing");
    // witness values: hex felt vector, decoded on the host here (they feed the host-side solver step for the product variables)
    const std::string& vh = vals->str;
    size_t n = 0;
    DOMREF_TRY(count_from_hex(vh.data(), vh.size(), &n));
    if (vh.size() != 8 + 64 * n) return set_err(ZK_ERR_LEN, "felt vector: %zu characters, the count says %zu felts", vh.size(), n);
    std::vector<HFr>& wv = B->wires;
    wv.assign(1, HFr::one());
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
    auto& ptr = B->ptr; auto& idx = B->idx; auto& val = B->val;
    for (int m = 0; m < 3; m++) { ptr[m].assign(1, 0); idx[m].clear(); val[m].clear(); }
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
    B->n_public = npub;
    return ZK_OK;
}

}  // namespace domref
