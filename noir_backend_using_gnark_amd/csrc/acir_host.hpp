// Host-only text front end of the path (no HIP in this file: it also compiles with plain g++ under -fsanitize=address,undefined for the mutation
// run of tests/cpp/parser_fuzz.cpp).  What the reference's Go shim does to the strings Noir hands over before gnark sees them:
//     acir.ACIR JSON                                  gnark_backend_ffi/acir/acir.go:17-75, opcode/arithmetic_opcode.go:18-83, term/mul_term.go:21-62,
//                                                     term/simple_term.go:20-51
//     BuildSparseR1CS / handleArithmeticOpcode        backend/plonk/sparse_r1cs.go:18-107 (one gate per arithmetic opcode: MulTerms[0] only;
//                                                     SimpleTerms of length 1 -> qO, 2 -> qL qR, 3 -> qL qR qO; directives / black boxes emit nothing)
//     HandleValues                                    backend/common.go:45-76
//     RawR1CS JSON (the intended Groth16 FFI)         src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60, backend/groth16/r1cs.go:9-72
// The circuits Noir produces are hundreds of MB of JSON (2^19 arithmetic opcodes = 240 MB), so the text is read ONCE by a pull tokenizer that is
// specialised to the two schemas: no document tree, no std::string per key or coefficient (a 64-character hex literal goes straight to four 64-bit
// limbs), numbers through a digit loop (strtod only for the spellings that need it).  The grammar accepted is exactly the one of the document-tree
// reader this replaces (kept as tests/cpp/json_dom_ref.hpp, the differential checker of the mutation run): same whitespace, escapes, literals,
// number spellings, nesting limit, first-key-wins for duplicate keys, nothing required after the root value.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <functional>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/zkmi.h"
#include "host_ff.hpp"

namespace zkmi {

// ------------------------------------------------------------------------------------------------ the tokenizer
struct JTok {
    const char* p;
    const char* end;
    const char* err = nullptr;
    std::string buf;  // the decoded form of the rare string that contains an escape

    static constexpr int MAX_DEPTH = 64;
    bool fail(const char* m) { if (!err) err = m; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    // every value starts here: the nesting limit, then the first character of the value
    bool enter(int depth) {
        if (depth > MAX_DEPTH) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        return true;
    }
    // p at the opening quote.  *s / *n: the characters between the quotes -- a view of the text, or of `buf` when the string had escapes
    bool str(const char** s, size_t* n) {
        if (p >= end || *p != '"') return fail("expected a string");
        const char* q = p + 1;
        while (q < end && *q != '"' && *q != '\\') q++;
        if (q >= end) return fail("unterminated string");
        if (*q == '"') { *s = p + 1; *n = (size_t)(q - p - 1); p = q + 1; return true; }
        buf.assign(p + 1, (size_t)(q - p - 1));
        p = q;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': buf.push_back('\n'); break;
                    case 't': buf.push_back('\t'); break;
                    case 'r': buf.push_back('\r'); break;
                    case 'b': buf.push_back('\b'); break;
                    case 'f': buf.push_back('\f'); break;
                    case 'u': {  // only the ASCII range can occur in these schemas
                        if (end - p < 5) return fail("bad \\u escape");
                        char h[5] = {p[1], p[2], p[3], p[4], 0};
                        buf.push_back((char)(strtoul(h, nullptr, 16) & 0x7f));
                        p += 4;
                        break;
                    }
                    default: buf.push_back(*p);
                }
                p++;
            } else {
                buf.push_back(*p++);
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        *s = buf.data();
        *n = buf.size();
        return true;
    }
    bool skip_str() {
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) return fail("bad escape");
                if (*p == 'u') {
                    if (end - p < 5) return fail("bad \\u escape");
                    p += 4;
                }
            }
            p++;
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    // a scalar that is not a string: true / false / null (reported as "not a number") or a number.  p at its first character.
    bool scalar(bool* is_num, double* num) {
        *is_num = false;
        // the length test comes first: the text is a GoString payload, not NUL-terminated
        if (end - p >= 4 && !memcmp(p, "true", 4)) { p += 4; return true; }
        if (end - p >= 5 && !memcmp(p, "false", 5)) { p += 5; return true; }
        if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return true; }
        {   // the spelling every index of a real circuit has: up to 18 digits and nothing a longer number could continue with
            const char* q = p;
            uint64_t v = 0;
            int nd = 0;
            while (q < end && *q >= '0' && *q <= '9' && nd < 19) { v = v * 10 + (uint64_t)(*q - '0'); q++; nd++; }
            if (nd > 0 && nd < 19 && (q == end || !(*q == '.' || *q == 'e' || *q == 'E' || *q == 'x' || *q == 'X' || *q == 'p' || *q == 'P'))) {
                p = q;
                *is_num = true;
                *num = (double)v;
                return true;
            }
        }
        char tmp[41];
        const size_t k = (size_t)(end - p) < 40 ? (size_t)(end - p) : 40;
        memcpy(tmp, p, k);
        tmp[k] = 0;
        char* e = nullptr;
        *num = strtod(tmp, &e);
        if (e == tmp) return fail("unexpected character");
        *is_num = true;
        p += e - tmp;
        return true;
    }
    // any value, validated and dropped
    bool skip(int depth) {
        if (!enter(depth)) return false;
        if (*p == '{') return object(depth, [&](const char*, size_t) { return skip(depth + 1); });
        if (*p == '[') return array(depth, [&](size_t) { return skip(depth + 1); });
        if (*p == '"') return skip_str();
        bool is_num;
        double d;
        return scalar(&is_num, &d);
    }
    // p at '{' (after enter(depth)).  member(key, key_len) consumes the member's value (a value at depth + 1).
    template <class F>
    bool object(int depth, F&& member) {
        p++;
        ws();
        if (p < end && *p == '}') { p++; return true; }
        for (;;) {
            ws();
            const char* k;
            size_t kn;
            if (!str(&k, &kn)) return false;
            std::string kept;
            if (k == buf.data()) { kept.assign(k, kn); k = kept.data(); }  // the member's value may reuse `buf`
            ws();
            if (p >= end || *p != ':') return fail("expected ':'");
            p++;
            if (!member(k, kn)) return false;
            ws();
            if (p < end && *p == ',') { p++; continue; }
            if (p < end && *p == '}') { p++; return true; }
            return fail("expected ',' or '}'");
        }
        (void)depth;
    }
    // p at '[' (after enter(depth)).  elem(i) consumes element i (a value at depth + 1).
    template <class F>
    bool array(int depth, F&& elem) {
        p++;
        ws();
        if (p < end && *p == ']') { p++; return true; }
        for (size_t i = 0;; i++) {
            if (!elem(i)) return false;
            ws();
            if (p < end && *p == ',') { p++; continue; }
            if (p < end && *p == ']') { p++; return true; }
            return fail("expected ',' or ']'");
        }
        (void)depth;
    }
};

static inline bool key_is(const char* k, size_t n, const char* lit) { return n == strlen(lit) && !memcmp(k, lit, n); }
static inline int hex_nibble(int c) { return (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; }

// fr.Element.SetString on a hex literal (FieldElement: up to 64 hex characters big-endian, canonical or not: reduced mod r) -> canonical limbs
static inline bool felt_from_hex(const char* s, size_t n, uint64_t t[4]) {
    if (n > 64 || n == 0) return false;
    t[0] = t[1] = t[2] = t[3] = 0;
    int bad = 0;
    for (size_t i = 0; i < n; i++) {  // character i from the END is nibble i
        const int d = hex_nibble((unsigned char)s[n - 1 - i]);
        bad |= d;
        t[i >> 4] |= (uint64_t)(d & 15) << (4 * (i & 15));
    }
    if (bad < 0) return false;
    while (HFr::geq_mod(t)) HFr::sub_mod(t);
    return true;
}
// a JSON number as a witness / variable index: an integer in [0, 2^32)
static inline bool as_index(double num, uint32_t* out) {
    if (!(num >= 0) || num > 4294967295.0 || num != (double)(uint64_t)num) return false;
    *out = (uint32_t)num;
    return true;
}

// canonical limbs -> Montgomery images, in place.  Circuits are mostly 0 / 1 / -1 coefficients: those skip the product.
static inline void to_mont_bulk(HFr* v, size_t n) {
    static const HFr one = HFr::one(), minus_one = HFr::zero() - HFr::one();
    static const uint64_t RM1[4] = {HFrParams::MOD[0] - 1, HFrParams::MOD[1], HFrParams::MOD[2], HFrParams::MOD[3]};
    auto run = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            uint64_t* l = v[i].l;
            if (!(l[1] | l[2] | l[3]) && l[0] <= 1) { if (l[0]) v[i] = one; continue; }
            if (l[0] == RM1[0] && l[1] == RM1[1] && l[2] == RM1[2] && l[3] == RM1[3]) { v[i] = minus_one; continue; }
            v[i] = v[i].to_mont();
        }
    };
    unsigned nt = n < ((size_t)1 << 15) ? 1 : std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt <= 1) { run(0, n); return; }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++) th.emplace_back(run, n * k / nt, n * (k + 1) / nt);
    for (auto& t : th) t.join();
}

// ------------------------------------------------------------------------------------------------ ACIR -> gates
struct Gates {
    size_t n_public = 0, n_vars = 0;
    bool with_coeffs = true;                 // false: only the wiring (what PlonkProveWithPK needs: the selectors come with the key)
    std::vector<HFr> ql, qr, qo, qm, qk;     // Montgomery
    std::vector<uint32_t> xa, xb, xc;
    std::vector<uint32_t> order;             // variable k holds witness order[k] (1-based witness index - 1): the gather that builds the solution
    size_t n_gates() const { return xa.size(); }
};

// HandleValues (common.go:45-76) without its |values| x |public inputs| loops.  Loop 1 there appends one PUBLIC variable per (witness w, public
// input equal to w) in witness order; loop 2, with public inputs, one SECRET variable per (w, public input NOT equal to w) -- i.e. |P| - c(w)
// copies of w, c(w) = how often w is listed -- and without public inputs one secret variable per witness; indexMap keeps the last index assigned.
// All copies appended for one w are equal, so only the counts matter: c(w) by one pass over the public inputs, then two passes over the witnesses.
// ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS: public witnesses first (witness order), then the others, one variable each.
static inline int handle_values(const std::vector<uint32_t>& pub, size_t n_values, int layout, Gates* G, std::vector<uint32_t>* index, std::string* err) {
    std::vector<uint32_t> cnt(n_values + 1, 0);
    for (uint32_t p : pub)
        if (p >= 1 && p <= n_values) cnt[p]++;
    index->assign(n_values + 1, 0);
    G->order.clear();
    if (layout == ZK_ACIR_LAYOUT_REFERENCE) {
        const size_t k = pub.size();
        size_t n_sec = 0;
        for (size_t w = 1; w <= n_values; w++) n_sec += k ? k - cnt[w] : 1;
        if (n_sec + n_values * k >= ((size_t)1 << 31)) {
            char m[160];
            snprintf(m, sizeof m, "HandleValues: %zu witnesses x %zu public inputs make too many variables", n_values, k);
            *err = m;
            return ZK_ERR_ARG;
        }
        size_t n_pub = 0;
        for (size_t w = 1; w <= n_values; w++) n_pub += cnt[w];
        G->order.resize(n_pub + n_sec);
        uint32_t* o = G->order.data();
        size_t at = 0;
        for (size_t w = 1; w <= n_values; w++)
            for (uint32_t c = 0; c < cnt[w]; c++) { o[at] = (uint32_t)(w - 1); (*index)[w] = (uint32_t)at++; }
        G->n_public = at;
        for (size_t w = 1; w <= n_values; w++) {
            const size_t copies = k ? k - cnt[w] : 1;
            for (size_t c = 0; c < copies; c++) { o[at] = (uint32_t)(w - 1); (*index)[w] = (uint32_t)at++; }
        }
    } else {
        G->order.resize(n_values);
        uint32_t* o = G->order.data();
        size_t at = 0;
        for (size_t w = 1; w <= n_values; w++)
            if (cnt[w]) { o[at] = (uint32_t)(w - 1); (*index)[w] = (uint32_t)at++; }
        G->n_public = at;
        for (size_t w = 1; w <= n_values; w++)
            if (!cnt[w]) { o[at] = (uint32_t)(w - 1); (*index)[w] = (uint32_t)at++; }
    }
    G->n_vars = G->order.size();
    return ZK_OK;
}

namespace acir_detail {
struct Term {       // one [coefficient, witness...] term as the text gave it
    bool ok = false;
    uint64_t c[4] = {0, 0, 0, 0};
    uint32_t w[2] = {0, 0};
};
// [hex, index x arity] at a value position.  ok = it is an array of exactly 1 + arity elements of the right kinds.
static inline bool read_term(JTok& T, int depth, int arity, bool want_coeff, Term* t) {
    t->ok = false;
    if (!T.enter(depth)) return false;
    if (*T.p != '[') return T.skip(depth);
    bool good = true;
    size_t count = 0;
    if (!T.array(depth, [&](size_t j) {
            count = j + 1;
            if (j > (size_t)arity) { good = false; return T.skip(depth + 1); }
            if (!T.enter(depth + 1)) return false;
            if (j == 0) {
                if (*T.p != '"') { good = false; return T.skip(depth + 1); }
                const char* s;
                size_t n;
                if (!T.str(&s, &n)) return false;
                if (want_coeff) good = felt_from_hex(s, n, t->c) && good;
                else {  // the wiring alone: the literal must still be one SetString accepts
                    bool hex = n >= 1 && n <= 64;
                    for (size_t i = 0; hex && i < n; i++) hex = hex_nibble((unsigned char)s[i]) >= 0;
                    good = hex && good;
                }
                return true;
            }
            if (*T.p == '{' || *T.p == '[' || *T.p == '"') { good = false; return T.skip(depth + 1); }
            bool is_num;
            double d;
            if (!T.scalar(&is_num, &d)) return false;
            good = is_num && as_index(d, &t->w[j - 1]) && good;
            return true;
        }))
        return false;
    t->ok = good && count == (size_t)arity + 1;
    return true;
}
}  // namespace acir_detail

namespace acir_detail {
// the gates one parser has emitted: wiring still in WITNESS numbers (`has` says which of xa / xb / xc name one), coefficients canonical
struct OpSink {
    std::vector<uint32_t> xa, xb, xc;
    std::vector<uint8_t> has;   // per gate, bit k: xa / xb / xc names a WITNESS still to be mapped to its variable (else: variable 0)
    std::vector<HFr> ql, qr, qo, qm, qk;
    void append(const OpSink& o) {
        xa.insert(xa.end(), o.xa.begin(), o.xa.end()); xb.insert(xb.end(), o.xb.begin(), o.xb.end()); xc.insert(xc.end(), o.xc.begin(), o.xc.end());
        has.insert(has.end(), o.has.begin(), o.has.end());
        ql.insert(ql.end(), o.ql.begin(), o.ql.end()); qr.insert(qr.end(), o.qr.begin(), o.qr.end()); qo.insert(qo.end(), o.qo.begin(), o.qo.end());
        qm.insert(qm.end(), o.qm.begin(), o.qm.end()); qk.insert(qk.end(), o.qk.begin(), o.qk.end());
    }
};
// One element of "opcodes" -> at most one gate (sparse_r1cs.go:28-107).  A parser of its own (tokenizer position, first semantic error, sink), so that
// several can work on different stretches of one opcodes array at once.
struct OpLower {
    JTok T;
    const char* sem = nullptr;  // the first semantic error (the text itself was well formed up to there)
    size_t n_values = 0;
    bool exact = true, with_coeffs = true;
    OpSink S;
    bool bad(const char* m) { if (!sem) sem = m; return false; }
    // a witness named by a gate: out of range -> variable 0 in the reference's map lookup (sparse_r1cs.go:53-54), an error in the other layout
    bool place(uint32_t w, uint32_t* x, uint8_t* mask, int bit) const {
        if (w < 1 || w > n_values) {
            if (!exact) return false;
            *x = 0;
            *mask &= (uint8_t)~(1u << bit);
            return true;
        }
        *x = w;
        *mask |= (uint8_t)(1u << bit);
        return true;
    }
    bool arithmetic(int depth) {  // p at '{' of the Arithmetic object; members are values at depth + 1
        Term mul0, lin[3], qc;
        bool have_mul = false, have_lin = false, have_qc = false, mul_arr = false, lin_arr = false, mul_empty = true;
        size_t nl = 0;
        if (!T.object(depth, [&](const char* k, size_t kn) {
                const int d = depth + 1;
                if (!have_mul && key_is(k, kn, "mul_terms")) {
                    have_mul = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '[') return T.skip(d);
                    mul_arr = true;
                    return T.array(d, [&](size_t i) {
                        mul_empty = false;
                        return i == 0 ? read_term(T, d + 1, 2, with_coeffs, &mul0) : T.skip(d + 1);
                    });
                }
                if (!have_lin && key_is(k, kn, "linear_combinations")) {
                    have_lin = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '[') return T.skip(d);
                    lin_arr = true;
                    return T.array(d, [&](size_t i) {
                        nl = i + 1;
                        return i < 3 ? read_term(T, d + 1, 1, with_coeffs, &lin[i]) : T.skip(d + 1);
                    });
                }
                if (!have_qc && key_is(k, kn, "q_c")) {
                    have_qc = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '"') return T.skip(d);
                    const char* s;
                    size_t n;
                    if (!T.str(&s, &n)) return false;
                    qc.ok = felt_from_hex(s, n, qc.c);
                    return true;
                }
                return T.skip(d);
            }))
            return false;
        if (!mul_arr || !lin_arr || !have_qc) return bad("ACIR JSON: malformed arithmetic opcode");
        uint32_t xa = 0, xb = 0, xc = 0;
        uint8_t mask = 0;
        const uint64_t* c_m = nullptr;
        const uint64_t *c_l = nullptr, *c_r = nullptr, *c_o = nullptr;
        if (!mul_empty) {  // qM * (xa * xb): only the first mul term
            if (!mul0.ok || !place(mul0.w[0], &xa, &mask, 0) || !place(mul0.w[1], &xb, &mask, 1)) return bad("ACIR JSON: malformed mul term");
            c_m = mul0.c;
        }
        bool ok = true;
        if (nl == 1) {
            ok = lin[0].ok && place(lin[0].w[0], &xc, &mask, 2);
            c_o = lin[0].c;
        } else if (nl == 2 || nl == 3) {
            ok = lin[0].ok && place(lin[0].w[0], &xa, &mask, 0) && lin[1].ok && place(lin[1].w[0], &xb, &mask, 1);
            c_l = lin[0].c;
            c_r = lin[1].c;
            if (ok && nl == 3) {
                ok = lin[2].ok && place(lin[2].w[0], &xc, &mask, 2);
                c_o = lin[2].c;
            }
        }
        if (!ok || !qc.ok) return bad("ACIR JSON: malformed linear combination / q_c");
        if (with_coeffs) {
            static const uint64_t Z[4] = {0, 0, 0, 0};
            auto put = [](std::vector<HFr>& v, const uint64_t* c) { v.push_back(HFr{{c[0], c[1], c[2], c[3]}}); };
            put(S.ql, c_l ? c_l : Z); put(S.qr, c_r ? c_r : Z); put(S.qo, c_o ? c_o : Z); put(S.qm, c_m ? c_m : Z); put(S.qk, qc.c);
        }
        S.xa.push_back(xa); S.xb.push_back(xb); S.xc.push_back(xc);
        S.has.push_back(mask);
        return true;
    }
    bool opcode(int depth) {  // one element of "opcodes"
        if (!T.enter(depth)) return false;
        if (*T.p != '{') return T.skip(depth) && bad("ACIR JSON: opcode is not an object");
        bool have_arith = false, arith_obj = false, other = false;
        // the Arithmetic member is lowered where it stands; a second one (the first wins) is only validated
        if (!T.object(depth, [&](const char* k, size_t kn) {
                const int d = depth + 1;
                if (!have_arith && key_is(k, kn, "Arithmetic")) {
                    have_arith = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '{') return T.skip(d);
                    arith_obj = true;
                    return arithmetic(d);
                }
                if (key_is(k, kn, "Directive") || key_is(k, kn, "BlackBoxFuncCall")) other = true;  // no constraints (sparse_r1cs.go:33-37)
                return T.skip(d);
            }))
            return false;
        if (!have_arith) return other ? true : bad("unknown opcode type");
        if (!arith_obj) return bad("ACIR JSON: malformed arithmetic opcode");
        return true;
    }
};

// ---- the elements of ONE opcodes array on several threads ---------------------------------------------------------------------------------------------
// The array of a 2^19-gate circuit is 190 MB of text and a fresh process (nargo runs one per proof) reads it while the HIP runtime starts: one core takes
// ~0.22 s for it, as long as everything else a cold PlonkProveWithPK does.  Split WITHOUT a structural pre-pass: candidate cut points are commas that look like
// the separator of two elements ('}' before, '{' after -- a guess: the same bytes can occur inside a string or deeper in the tree); parser k starts after
// candidate k as if it stood between two elements and runs until it stands exactly ON a later candidate, at the array's ']' or at an error.  Parser 0 starts
// at the real first element, so its state is the sequential reader's; a parser whose start the previous accepted one landed on inherits that property
// (induction) and the others are thrown away -- a wrong guess costs time, never a different result.  The accepted parsers' gates concatenate in text order;
// the first error in text order is the one reported, as in the sequential loop (JTok::array), whose accept / reject grammar and messages this repeats.
struct ParallelCfg {
    size_t min_bytes = (size_t)2 << 20;  // shorter texts: one thread
    unsigned threads = 0;                 // 0: hardware threads, at most 16
};
static inline ParallelCfg& parallel_cfg() {
    static ParallelCfg c;
    return c;
}
static inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }
static inline const char* find_element_comma(const char* from, const char* lo, const char* end) {
    const char* lim = end - from > (ptrdiff_t)(1 << 20) ? from + (1 << 20) : end;
    for (const char* q = from; q < lim; q++) {
        q = (const char*)memchr(q, ',', (size_t)(lim - q));
        if (!q) return nullptr;
        const char* a = q;
        while (a > lo && is_ws(a[-1])) a--;
        if (a == lo || a[-1] != '}') continue;
        const char* b = q + 1;
        while (b < end && is_ws(*b)) b++;
        if (b < end && *b == '{') return q;
    }
    return nullptr;
}
struct ChunkResult {
    OpSink S;
    int landed = -1;             // index of the candidate this parser stopped on
    const char* after = nullptr; // position after the array's ']' if it got there
    const char* terr = nullptr;  // tokenizer error / semantic error (either: the parse failed there)
    const char* sem = nullptr;
    bool failed = false;
};
// elements from `begin` (the first character of an element's value position) on; cuts[j] for j > first_cut are the candidates it may land on
static inline void run_chunk(const char* begin, const char* end, const std::vector<const char*>& cuts, size_t first_cut, const OpLower& proto, ChunkResult* R) {
    OpLower L;
    L.n_values = proto.n_values; L.exact = proto.exact; L.with_coeffs = proto.with_coeffs;
    L.T.p = begin; L.T.end = end;
    size_t nb = first_cut;
    {   // room for the gates of this stretch (an arithmetic opcode is >= ~150 characters of text): no reallocation while the parsers run side by side
        const char* stop = first_cut < cuts.size() ? cuts[first_cut] : end;
        const size_t est = (size_t)(stop - begin) / 128 + 16;
        L.S.xa.reserve(est); L.S.xb.reserve(est); L.S.xc.reserve(est); L.S.has.reserve(est);
        if (L.with_coeffs) { L.S.ql.reserve(est); L.S.qr.reserve(est); L.S.qo.reserve(est); L.S.qm.reserve(est); L.S.qk.reserve(est); }
    }
    for (;;) {
        if (!L.opcode(2)) { R->failed = true; break; }
        L.T.ws();
        const char* p = L.T.p;
        if (p < end && *p == ',') {
            while (nb < cuts.size() && cuts[nb] < p) nb++;
            if (nb < cuts.size() && cuts[nb] == p) { R->landed = (int)nb; break; }
            L.T.p++;
            continue;
        }
        if (p < end && *p == ']') { R->after = p + 1; break; }
        L.T.fail("expected ',' or ']'");
        R->failed = true;
        break;
    }
    R->terr = L.T.err;
    R->sem = L.sem;
    R->S = std::move(L.S);
}
// M.T.p at the first element of a non-empty opcodes array.  Returns like JTok::array's loop: true with M.T.p after the ']', or false with M.T.err / M.sem set.
static inline bool elements_parallel(OpLower& M, unsigned nt) {
    const char* begin = M.T.p;
    const char* end = M.T.end;
    std::vector<const char*> cuts;  // cuts[0]: a placeholder for the real start; cuts[j]: position of a candidate comma
    cuts.push_back(begin);
    const size_t span = (size_t)(end - begin);
    for (unsigned k = 1; k < nt; k++) {
        const char* from = begin + span / nt * k;
        if (from <= cuts.back()) continue;
        const char* c = find_element_comma(from, begin, end);
        if (c && c > cuts.back()) cuts.push_back(c);
    }
    std::vector<ChunkResult> res(cuts.size());
    std::vector<std::thread> th;
    for (size_t k = 1; k < cuts.size(); k++) {
        try {
            th.emplace_back(run_chunk, cuts[k] + 1, end, std::cref(cuts), k + 1, std::cref(M), &res[k]);
        } catch (const std::system_error&) {  // no more threads to be had: this stretch on the calling thread
            run_chunk(cuts[k] + 1, end, cuts, k + 1, M, &res[k]);
        }
    }
    run_chunk(begin, end, cuts, 1, M, &res[0]);
    for (auto& t : th) t.join();
    size_t total = M.S.xa.size();
    for (size_t cur = 0;;) {  // sizes first: one reservation
        total += res[cur].S.xa.size();
        if (res[cur].landed < 0) break;
        cur = (size_t)res[cur].landed;
    }
    M.S.xa.reserve(total); M.S.xb.reserve(total); M.S.xc.reserve(total); M.S.has.reserve(total);
    if (M.with_coeffs) { M.S.ql.reserve(total); M.S.qr.reserve(total); M.S.qo.reserve(total); M.S.qm.reserve(total); M.S.qk.reserve(total); }
    for (size_t cur = 0;;) {
        const ChunkResult& R = res[cur];
        M.S.append(R.S);
        if (R.failed) {
            if (R.terr) M.T.fail(R.terr);
            if (R.sem) M.bad(R.sem);
            return false;
        }
        if (R.after) { M.T.p = R.after; return true; }
        cur = (size_t)R.landed;
    }
}
}  // namespace acir_detail

// BuildSparseR1CS (sparse_r1cs.go:18-107) + HandleValues (common.go:45-76).  n_values = number of witness values handed over (witnesses 1..n).
// Returns ZK_OK or ZK_ERR_ARG with *err set.
static inline int lower_acir(const char* json, size_t len, size_t n_values, int layout, bool with_coeffs, Gates* G, std::string* err) {
    if (layout != ZK_ACIR_LAYOUT_REFERENCE && layout != ZK_ACIR_LAYOUT_ONE_VAR_PER_WITNESS) {
        *err = "unknown ACIR variable layout " + std::to_string(layout);
        return ZK_ERR_ARG;
    }
    *G = Gates();
    G->with_coeffs = with_coeffs;
    acir_detail::OpLower M;
    M.n_values = n_values;
    M.exact = layout == ZK_ACIR_LAYOUT_REFERENCE;
    M.with_coeffs = with_coeffs;
    M.T.p = json;
    M.T.end = json + len;
    JTok& T = M.T;
    auto bad = [&](const char* m) { return M.bad(m); };
    std::vector<uint32_t> pub;
    bool have_ops = false, have_pub = false;
    unsigned nt = acir_detail::parallel_cfg().threads ? acir_detail::parallel_cfg().threads : std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (len < acir_detail::parallel_cfg().min_bytes || nt < 2) nt = 1;

    bool root_obj = false, ops_arr = false;
    bool fine = T.enter(0);
    if (fine) {
        if (*T.p != '{') fine = T.skip(0);
        else {
            root_obj = true;
            fine = T.object(0, [&](const char* k, size_t kn) {
                if (!have_ops && key_is(k, kn, "opcodes")) {
                    have_ops = true;
                    if (!T.enter(1)) return false;
                    if (*T.p != '[') return T.skip(1);
                    ops_arr = true;
                    if (nt > 1) {  // JTok::array's prologue, then the elements on several threads
                        T.p++;
                        T.ws();
                        if (T.p < T.end && *T.p == ']') { T.p++; return true; }
                        return acir_detail::elements_parallel(M, nt);
                    }
                    return T.array(1, [&](size_t) { return M.opcode(2); });
                }
                if (!have_pub && key_is(k, kn, "public_inputs")) {
                    have_pub = true;
                    if (!T.enter(1)) return false;
                    if (*T.p != '[') return T.skip(1);
                    return T.array(1, [&](size_t) {
                        if (!T.enter(2)) return false;
                        if (*T.p == '{' || *T.p == '[' || *T.p == '"') return T.skip(2) && bad("ACIR JSON: bad public input");
                        bool is_num;
                        double d;
                        uint32_t w;
                        if (!T.scalar(&is_num, &d)) return false;
                        if (!is_num || !as_index(d, &w)) return bad("ACIR JSON: bad public input");
                        pub.push_back(w);
                        return true;
                    });
                }
                return T.skip(1);
            });
        }
    }
    const char* sem = M.sem;
    if (!fine || !root_obj) {
        *err = std::string("ACIR JSON: ") + (T.err ? T.err : sem ? sem : "not an object");
        if (sem && !T.err) *err = sem;
        return ZK_ERR_ARG;
    }
    if (!ops_arr) { *err = "ACIR JSON: no opcodes array"; return ZK_ERR_ARG; }
    G->xa = std::move(M.S.xa); G->xb = std::move(M.S.xb); G->xc = std::move(M.S.xc);
    G->ql = std::move(M.S.ql); G->qr = std::move(M.S.qr); G->qo = std::move(M.S.qo); G->qm = std::move(M.S.qm); G->qk = std::move(M.S.qk);
    const std::vector<uint8_t>& has = M.S.has;
    std::vector<uint32_t> index;
    const int rc = handle_values(pub, n_values, layout, G, &index, err);
    if (rc != ZK_OK) return rc;
    const size_t nc = G->xa.size();
    for (size_t i = 0; i < nc; i++) {
        const uint8_t m = has[i];
        if (m & 1) G->xa[i] = index[G->xa[i]];
        if (m & 2) G->xb[i] = index[G->xb[i]];
        if (m & 4) G->xc[i] = index[G->xc[i]];
    }
    if (with_coeffs)
        for (std::vector<HFr>* v : {&G->ql, &G->qr, &G->qo, &G->qm, &G->qk}) to_mont_bulk(v->data(), v->size());
    return ZK_OK;
}

// hex felt vector header: u32 big-endian count as 8 hex characters
static inline bool count_from_hex(const char* hex, size_t len, size_t* n) {
    if (len < 8) return false;
    size_t v = 0;
    for (int k = 0; k < 8; k++) {
        const int d = hex_nibble((unsigned char)hex[k]);
        if (d < 0) return false;
        v = (v << 4) | (size_t)d;
    }
    *n = v;
    return true;
}

// ------------------------------------------------------------------------------------------------ RawR1CS -> R1CS rows + wire values
// buildR1CS of the reference's intended Groth16 FFI (backend/groth16/r1cs.go:9-72, commented out there; payload RawR1CS of
// src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60: {"gates":[{"mul_terms":[{"coefficient","multiplicand","multiplier"}],"add_terms":
// [{"coefficient","sum"}],"constant_term"}],"public_inputs","values" (hex felt vector),"num_variables","num_constraints"}): every mul term gets an
// internal product variable p with (1 * multiplicand) * (1 * multiplier) = 1 * p; every gate ends in
// (1 * ONE) * (sum coefficient * p + sum coefficient * x + constant * ONE) = 0.  Made well-defined where the sketch is not: wires = [ONE, public
// witnesses in witness order, the other witnesses, product variables]; values[w - 1] is witness w; the product variable is the plain product (the
// sketch puts the coefficient on the product constraint's output AND on the term, which cancels it); the constant term IS in the sum (the sketch
// drops it); a mul term with coefficient 0 emits nothing.
// vectors whose resize() leaves new elements uninitialised (trivial element types only): the rows of a 2^20-constraint circuit are 0.15 GB that several
// threads fill right after -- zeroing them first, on one thread, was a fifth of the reader's time
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;
struct RawR1CSBuilt {
    RawVec<uint32_t> ptr[3], idx[3];
    RawVec<HFr> val[3];
    RawVec<HFr> wires;  // (want_wires) the full wire vector (Montgomery): [ONE, public..., secret..., products...]
    size_t n_public = 0;     // ONE included
    // How the wire vector follows from the values vector -- what a RESIDENT circuit keeps to assemble it on the device for every later proof:
    size_t n_values = 0;
    RawVec<uint32_t> order;           // wire 1 + k holds witness order[k] + 1 (k < n_values): the public ones first, each group in witness order
    RawVec<uint32_t> prod_a, prod_b;  // wire 1 + n_values + j = wire prod_a[j] * wire prod_b[j] (both operands are witness wires, never products)
    size_t values_at = 0, values_len = 0;  // the values string inside the text: offset of its first character and its length (values_at = 0: it had escapes)
};

namespace raw_detail {
struct MulT { bool ok; HFr c; uint32_t a, b; };
struct AddT { bool ok; HFr c; uint32_t x; };
struct Gate { bool shape_ok, k_ok; size_t m0, m1, a0, a1; HFr k; };
// One element of "gates" -> its terms, as the text gave them (kinds and ranges are judged afterwards, in text order).  A parser of its own, so that several
// can read different stretches of one gates array at once (the scheme of acir_detail::elements_parallel).
struct GateReader {
    JTok T{nullptr, nullptr, nullptr, {}};
    std::vector<MulT> muls;
    std::vector<AddT> adds;
    std::vector<Gate> gates;
    // {"coefficient": hex, <name1>: index[, <name2>: index]} ; ok only if every named member is present (first occurrence) with the right kind
    bool term(int depth, const char* n1, const char* n2, HFr* c, uint32_t* w1, uint32_t* w2, bool* ok) {
        *ok = false;
        if (!T.enter(depth)) return false;
        if (*T.p != '{') return T.skip(depth);
        bool hc = false, h1 = false, h2 = false, cs = false, i1 = false, i2 = false;
        if (!T.object(depth, [&](const char* k, size_t kn) {
                const int d = depth + 1;
                if (!hc && key_is(k, kn, "coefficient")) {
                    hc = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '"') return T.skip(d);
                    const char* s;
                    size_t n;
                    if (!T.str(&s, &n)) return false;
                    uint64_t t[4];
                    cs = felt_from_hex(s, n, t);
                    if (cs) *c = HFr{{t[0], t[1], t[2], t[3]}};  // canonical; to_mont_bulk afterwards (0 / 1 / -1 skip the product)
                    return true;
                }
                const bool m1 = !h1 && key_is(k, kn, n1), m2 = !m1 && n2 && !h2 && key_is(k, kn, n2);
                if (m1 || m2) {
                    (m1 ? h1 : h2) = true;
                    if (!T.enter(d)) return false;
                    if (*T.p == '{' || *T.p == '[' || *T.p == '"') return T.skip(d);
                    bool is_num;
                    double v;
                    if (!T.scalar(&is_num, &v)) return false;
                    (m1 ? i1 : i2) = is_num && as_index(v, m1 ? w1 : w2);
                    return true;
                }
                return T.skip(d);
            }))
            return false;
        *ok = cs && i1 && (!n2 || i2);
        return true;
    }
    bool element(int depth) {
        Gate g{false, false, muls.size(), muls.size(), adds.size(), adds.size(), HFr::zero()};
        if (!T.enter(depth)) return false;
        if (*T.p != '{') { gates.push_back(g); return T.skip(depth); }
        bool hm = false, ha = false, hk = false, m_arr = false, a_arr = false, k_str = false;
        if (!T.object(depth, [&](const char* k, size_t kn) {
                const int d = depth + 1;
                if (!hm && key_is(k, kn, "mul_terms")) {
                    hm = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '[') return T.skip(d);
                    m_arr = true;
                    g.m0 = muls.size();
                    const bool r = T.array(d, [&](size_t) {
                        MulT t{false, HFr::zero(), 0, 0};
                        if (!term(d + 1, "multiplicand", "multiplier", &t.c, &t.a, &t.b, &t.ok)) return false;
                        muls.push_back(t);
                        return true;
                    });
                    g.m1 = muls.size();
                    return r;
                }
                if (!ha && key_is(k, kn, "add_terms")) {
                    ha = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '[') return T.skip(d);
                    a_arr = true;
                    g.a0 = adds.size();
                    const bool r = T.array(d, [&](size_t) {
                        AddT t{false, HFr::zero(), 0};
                        if (!term(d + 1, "sum", nullptr, &t.c, &t.x, nullptr, &t.ok)) return false;
                        adds.push_back(t);
                        return true;
                    });
                    g.a1 = adds.size();
                    return r;
                }
                if (!hk && key_is(k, kn, "constant_term")) {
                    hk = true;
                    if (!T.enter(d)) return false;
                    if (*T.p != '"') return T.skip(d);
                    k_str = true;
                    const char* s;
                    size_t n;
                    if (!T.str(&s, &n)) return false;
                    uint64_t t[4];
                    g.k_ok = felt_from_hex(s, n, t);
                    if (g.k_ok) g.k = HFr{{t[0], t[1], t[2], t[3]}};
                    return true;
                }
                return T.skip(d);
            }))
            return false;
        g.shape_ok = m_arr && a_arr && k_str;
        gates.push_back(g);
        return true;
    }
    // after a read on several threads: the accepted readers of the stretches of the gates array, in text order (their gates index their OWN muls / adds;
    // nothing is merged: 0.11 GB of terms at 2^20 constraints would be copied by one thread); empty after a read on one thread (this reader holds everything)
    std::vector<GateReader> parts;
};
// A candidate cut between two GATES: a comma with '}' before and '{' after (acir_detail::find_element_comma) whose '{' opens an object that starts with one of a
// gate's own keys -- the same bytes `},{` also separate the terms INSIDE a gate's mul_terms / add_terms arrays (objects that start with "coefficient"), and a
// reader started there would stop at once at that array's ']', leaving the whole text to the first reader.  Still only a guess (see elements_parallel).
static inline const char* find_gate_comma(const char* from, const char* lo, const char* end) {
    const char* lim = end - from > (ptrdiff_t)(1 << 20) ? from + (1 << 20) : end;
    for (const char* q = from; q < lim;) {
        q = acir_detail::find_element_comma(q, lo, end);
        if (!q || q >= lim) return nullptr;
        const char* b = q + 1;
        while (b < end && acir_detail::is_ws(*b)) b++;
        b++;  // the '{'
        while (b < end && acir_detail::is_ws(*b)) b++;
        for (const char* key : {"\"mul_terms\"", "\"add_terms\"", "\"constant_term\""}) {
            const size_t kn = strlen(key);
            if ((size_t)(end - b) >= kn && !memcmp(b, key, kn)) return q;
        }
        q++;
    }
    return nullptr;
}
struct Chunk {
    GateReader R;
    int landed = -1;              // index of the candidate comma this parser stopped on
    const char* after = nullptr;  // position after the array's ']' if it got there
    bool failed = false;
};
static inline void run_chunk(const char* begin, const char* end, const std::vector<const char*>& cuts, size_t first_cut, Chunk* C) {
    GateReader& L = C->R;
    L.T.p = begin; L.T.end = end;
    size_t nb = first_cut;
    {
        const char* stop = first_cut < cuts.size() ? cuts[first_cut] : end;
        const size_t est = (size_t)(stop - begin) / 256 + 16;  // a gate with one term of each kind is ~300 characters
        L.gates.reserve(est); L.muls.reserve(est); L.adds.reserve(2 * est);
    }
    for (;;) {
        if (!L.element(2)) { C->failed = true; return; }
        L.T.ws();
        const char* p = L.T.p;
        if (p < end && *p == ',') {
            while (nb < cuts.size() && cuts[nb] < p) nb++;
            if (nb < cuts.size() && cuts[nb] == p) { C->landed = (int)nb; return; }
            L.T.p++;
            continue;
        }
        if (p < end && *p == ']') { C->after = p + 1; return; }
        L.T.fail("expected ',' or ']'");
        C->failed = true;
        return;
    }
}
// M.T.p at the first element of a non-empty gates array; like JTok::array's loop: true with M.T.p after the ']', or false with M.T.err set.  Same scheme and the
// same argument as acir_detail::elements_parallel: candidate cuts are GUESSED, a parser is accepted only if the accepted one before it stopped exactly on its
// cut while standing between two elements, so a wrong guess costs time and never changes the result; the first error in text order is the one reported.
static inline bool elements_parallel(GateReader& M, unsigned nt) {
    const char* begin = M.T.p;
    const char* end = M.T.end;
    std::vector<const char*> cuts;
    cuts.push_back(begin);
    const size_t span = (size_t)(end - begin);
    for (unsigned k = 1; k < nt; k++) {
        const char* from = begin + span / nt * k;
        if (from <= cuts.back()) continue;
        const char* c = find_gate_comma(from, begin, end);
        if (c && c > cuts.back()) cuts.push_back(c);
    }
    std::vector<Chunk> res(cuts.size());
    std::vector<std::thread> th;
    for (size_t k = 1; k < cuts.size(); k++) {
        try {
            th.emplace_back(run_chunk, cuts[k] + 1, end, std::cref(cuts), k + 1, &res[k]);
        } catch (const std::system_error&) {
            run_chunk(cuts[k] + 1, end, cuts, k + 1, &res[k]);
        }
    }
    run_chunk(begin, end, cuts, 1, &res[0]);
    for (auto& t : th) t.join();
    for (size_t cur = 0;;) {
        Chunk& C = res[cur];
        const char* terr = C.R.T.err;
        const bool failed = C.failed;
        const char* after = C.after;
        const int landed = C.landed;
        M.parts.push_back(std::move(C.R));
        if (failed) { M.T.fail(terr ? terr : "malformed gate"); return false; }
        if (after) { M.T.p = after; return true; }
        cur = (size_t)landed;
    }
}
}  // namespace raw_detail

// want_wires: also decode the values and run the solver's step for the product variables on the host (B->wires); without it only the circuit is built
// (callers that assemble the wire vector on the device from order / prod_a / prod_b).
static inline int raw_r1cs_build(const char* json, size_t len, RawR1CSBuilt* B, std::string* err, bool want_wires = true) {
    using namespace raw_detail;
    GateReader M;
    std::vector<double> pubs;      // numbers as read; kinds checked after (a non-number is NaN)
    const char* values = nullptr;  // a view of the text (or of `values_own` when the string had escapes)
    size_t values_n = 0;
    std::string values_own;
    bool have_gates = false, gates_arr = false, have_pub = false, pub_arr = false, have_vals = false, vals_str = false;
    M.T.p = json;
    M.T.end = json + len;
    JTok& T = M.T;
    const double NOT_NUM = -1;  // as_index rejects it like any other non-index
    unsigned nt = acir_detail::parallel_cfg().threads ? acir_detail::parallel_cfg().threads : std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (len < acir_detail::parallel_cfg().min_bytes || nt < 2) nt = 1;

    bool root_obj = false;
    bool fine = T.enter(0);
    if (fine) {
        if (*T.p != '{') fine = T.skip(0);
        else {
            root_obj = true;
            fine = T.object(0, [&](const char* k, size_t kn) {
                if (!have_gates && key_is(k, kn, "gates")) {
                    have_gates = true;
                    if (!T.enter(1)) return false;
                    if (*T.p != '[') return T.skip(1);
                    gates_arr = true;
                    if (nt > 1) {  // JTok::array's prologue, then the elements on several threads
                        T.p++;
                        T.ws();
                        if (T.p < T.end && *T.p == ']') { T.p++; return true; }
                        return elements_parallel(M, nt);
                    }
                    return T.array(1, [&](size_t) { return M.element(2); });
                }
                if (!have_pub && key_is(k, kn, "public_inputs")) {
                    have_pub = true;
                    if (!T.enter(1)) return false;
                    if (*T.p != '[') return T.skip(1);
                    pub_arr = true;
                    return T.array(1, [&](size_t) {
                        if (!T.enter(2)) return false;
                        if (*T.p == '{' || *T.p == '[' || *T.p == '"') { pubs.push_back(NOT_NUM); return T.skip(2); }
                        bool is_num;
                        double v;
                        if (!T.scalar(&is_num, &v)) return false;
                        pubs.push_back(is_num ? v : NOT_NUM);
                        return true;
                    });
                }
                if (!have_vals && key_is(k, kn, "values")) {
                    have_vals = true;
                    if (!T.enter(1)) return false;
                    if (*T.p != '"') return T.skip(1);
                    vals_str = true;
                    const char* s;
                    size_t n;
                    if (!T.str(&s, &n)) return false;
                    if (s >= json && s + n <= json + len) { values = s; values_n = n; }  // the usual case: 64 MB at 2^20 witnesses, not copied
                    else { values_own.assign(s, n); values = values_own.data(); values_n = n; }
                    return true;
                }
                return T.skip(1);
            });
        }
    }
    if (!fine || !root_obj) { *err = std::string("RawR1CS JSON: ") + (T.err ? T.err : "not an object"); return ZK_ERR_ARG; }
    if (!gates_arr || !vals_str) { *err = "RawR1CS JSON: gates / values missing"; return ZK_ERR_ARG; }
    // witness values: hex felt vector
    size_t n = 0;
    if (values_n < 8) { *err = "felt vector: " + std::to_string(values_n) + " characters cannot hold the 4-byte count"; return ZK_ERR_ARG; }
    if (!count_from_hex(values, values_n, &n)) { *err = "felt vector: invalid hex character in the count"; return ZK_ERR_ARG; }
    if (values_n != 8 + 64 * n) { *err = "felt vector: " + std::to_string(values_n) + " characters, the count says " + std::to_string(n) + " felts"; return ZK_ERR_LEN; }
    B->n_values = n;
    B->values_at = values_own.empty() && values >= json ? (size_t)(values - json) : 0;
    B->values_len = values_n;
    std::vector<bool> is_pub(n + 1, false);
    if (pub_arr)
        for (double v : pubs) {
            uint32_t w;
            if (!as_index(v, &w)) { *err = "RawR1CS JSON: bad public input"; return ZK_ERR_ARG; }
            if (w >= 1 && w <= n) is_pub[w] = true;
        }
    RawVec<HFr>& wv = B->wires;
    wv.clear();
    if (want_wires) { wv.reserve(1 + n); wv.push_back(HFr::one()); }
    std::vector<uint32_t> wire(n + 1, 0);
    B->order.clear();
    B->order.reserve(n);
    size_t npub = 1;
    for (int pass = 0; pass < 2; pass++)
        for (size_t w = 1; w <= n; w++)
            if (is_pub[w] == (pass == 0)) {
                wire[w] = (uint32_t)(1 + B->order.size());
                B->order.push_back((uint32_t)(w - 1));
                if (pass == 0) npub++;
                if (!want_wires) continue;
                uint64_t t[4] = {0, 0, 0, 0};
                int badc = 0;
                const char* s = values + 8 + 64 * (w - 1);
                for (int i = 0; i < 64; i++) {  // canonical values only, like fr.Vector.UnmarshalBinary
                    const int d = hex_nibble((unsigned char)s[63 - i]);
                    badc |= d;
                    t[i >> 4] |= (uint64_t)(d & 15) << (4 * (i & 15));
                }
                if (badc < 0 || HFr::geq_mod(t)) { *err = "felt vector: invalid hex character or fr.Element encoding"; return ZK_ERR_ARG; }
                wv.push_back(HFr{{t[0], t[1], t[2], t[3]}}.to_mont());
            }
    // From here on the gates are independent: the coefficients' Montgomery images, the rows and the solver's step run over stretches of the gate list on up to
    // sixteen threads -- a counting pass (rows, entries and product variables per stretch; the first malformed gate IN TEXT ORDER is the error, as in a loop over
    // all gates), then every stretch writes its rows where the counts say they go.  L has one entry per row (the multiplicand / ONE), O one per product row.
    // The gates stay in the readers that read them (M.parts after a read on several threads): a stretch is a range of ONE reader's gates.
    std::vector<GateReader*> readers;
    if (M.parts.empty()) readers.push_back(&M);
    else for (auto& r : M.parts) readers.push_back(&r);
    size_t G = 0;
    for (auto* r : readers) G += r->gates.size();
    const unsigned nt2 = (nt > 1 && (G >= 4096 || acir_detail::parallel_cfg().min_bytes == 0)) ? nt : 1;  // (min_bytes == 0: the mutation harness forces the threaded paths on short texts)
    struct Stretch {
        GateReader* R = nullptr;
        size_t g0 = 0, g1 = 0, rows = 0, nnz1 = 0, prods = 0;
        const char* bad = nullptr;
    };
    std::vector<Stretch> st;
    {   // about G / nt2 gates per stretch, never across readers
        const size_t per = G / nt2 + 1;
        for (auto* r : readers)
            for (size_t g0 = 0; g0 < r->gates.size(); g0 += per) {
                Stretch S;
                S.R = r;
                S.g0 = g0;
                S.g1 = g0 + per < r->gates.size() ? g0 + per : r->gates.size();
                st.push_back(S);
            }
        if (st.empty()) {  // no gate at all
            Stretch S;
            S.R = readers[0];
            st.push_back(S);
        }
    }
    const unsigned ns = (unsigned)st.size();
    const HFr one = HFr::one();
    static const HFr minus_one = HFr::zero() - HFr::one();
    static const uint64_t RM1[4] = {HFrParams::MOD[0] - 1, HFrParams::MOD[1], HFrParams::MOD[2], HFrParams::MOD[3]};
    auto mont = [&](HFr& v) {  // canonical -> Montgomery; circuits are mostly 0 / 1 / -1: those skip the product (to_mont_bulk's rule)
        uint64_t* l = v.l;
        if (!(l[1] | l[2] | l[3]) && l[0] <= 1) { if (l[0]) v = one; return; }
        if (l[0] == RM1[0] && l[1] == RM1[1] && l[2] == RM1[2] && l[3] == RM1[3]) { v = minus_one; return; }
        v = v.to_mont();
    };
    // one gate: its terms (equal wires merged, in order of first appearance).  Returns the first thing wrong with it, or null.
    typedef std::vector<std::pair<uint32_t, HFr>> Terms;
    auto gate_terms = [&](const GateReader& Rd, const Gate& g, size_t prod0, Terms& terms, size_t* n_prods) -> const char* {
        if (!g.shape_ok) return "RawR1CS JSON: malformed gate";
        terms.clear();
        auto add_term = [&](uint32_t x, const HFr& c) {
            for (auto& t : terms)
                if (t.first == x) { t.second = t.second + c; return; }
            terms.emplace_back(x, c);
        };
        size_t np = 0;
        for (size_t i = g.m0; i < g.m1; i++) {
            const MulT& t = Rd.muls[i];
            if (!t.ok || t.a < 1 || t.a > n || t.b < 1 || t.b > n) return "RawR1CS JSON: malformed mul term";
            if (t.c.is_zero()) continue;
            add_term((uint32_t)(1 + n + prod0 + np), t.c);
            np++;
        }
        for (size_t i = g.a0; i < g.a1; i++) {
            const AddT& t = Rd.adds[i];
            if (!t.ok || t.x < 1 || t.x > n) return "RawR1CS JSON: malformed add term";
            add_term(wire[t.x], t.c);
        }
        if (!g.k_ok) return "RawR1CS JSON: malformed constant term";
        if (!g.k.is_zero()) add_term(0, g.k);
        *n_prods = np;
        return nullptr;
    };
    auto count = [&](unsigned k) {
        Stretch& S = st[k];
        GateReader& Rd = *S.R;
        Terms terms;
        for (size_t gi = S.g0; gi < S.g1; gi++) {
            Gate& g = Rd.gates[gi];
            for (size_t i = g.m0; i < g.m1; i++) if (Rd.muls[i].ok) mont(Rd.muls[i].c);
            for (size_t i = g.a0; i < g.a1; i++) if (Rd.adds[i].ok) mont(Rd.adds[i].c);
            if (g.k_ok) mont(g.k);
            size_t np = 0;
            if (const char* e = gate_terms(Rd, g, 0, terms, &np)) { S.bad = e; return; }
            S.prods += np;
            S.rows += np + 1;
            S.nnz1 += np + terms.size();
        }
    };
    auto on_threads = [&](const std::function<void(unsigned)>& f) {
        if (ns == 1) { f(0); return; }
        std::vector<std::thread> th;
        for (unsigned k = 1; k < ns; k++) {
            try {
                th.emplace_back(f, k);
            } catch (const std::system_error&) {
                f(k);
            }
        }
        f(0);
        for (auto& t : th) t.join();
    };
    on_threads(count);
    for (unsigned k = 0; k < ns; k++)
        if (st[k].bad) { *err = st[k].bad; return ZK_ERR_ARG; }
    size_t rows = 0, nnz1 = 0, prods = 0;
    std::vector<size_t> row0(ns), nz0(ns), pr0(ns);
    for (unsigned k = 0; k < ns; k++) { row0[k] = rows; nz0[k] = nnz1; pr0[k] = prods; rows += st[k].rows; nnz1 += st[k].nnz1; prods += st[k].prods; }
    if (rows >= ((size_t)1 << 31) || nnz1 >= ((size_t)1 << 31) || 1 + n + prods >= ((size_t)1 << 31)) { *err = "RawR1CS JSON: too many constraints for 31-bit indices"; return ZK_ERR_ARG; }
    // (uninitialised: every element below is written by the stretch that owns it -- first touched, too, by the thread that fills it)
    for (int m = 0; m < 3; m++) { B->ptr[m].resize(rows + 1); B->ptr[m][0] = 0; }
    B->idx[0].resize(rows); B->val[0].resize(rows);
    B->idx[1].resize(nnz1); B->val[1].resize(nnz1);
    B->idx[2].resize(prods); B->val[2].resize(prods);
    B->prod_a.resize(prods);
    B->prod_b.resize(prods);
    if (want_wires) wv.resize(1 + n + prods);
    auto fill = [&](unsigned k) {
        const Stretch& S = st[k];
        const GateReader& Rd = *S.R;
        Terms terms;
        size_t r = row0[k], z = nz0[k], p = pr0[k];
        for (size_t gi = S.g0; gi < S.g1; gi++) {
            const Gate& g = Rd.gates[gi];
            size_t np = 0;
            (void)gate_terms(Rd, g, p, terms, &np);
            for (size_t i = g.m0; i < g.m1; i++) {  // one product constraint per mul term with a non-zero coefficient: (1 * a) * (1 * b) = 1 * p
                const MulT& t = Rd.muls[i];
                if (t.c.is_zero()) continue;
                const uint32_t a = wire[t.a], b = wire[t.b];
                B->prod_a[p] = a;
                B->prod_b[p] = b;
                if (want_wires) wv[1 + n + p] = wv[a] * wv[b];  // the solver's step for this internal variable (operands are witness wires)
                B->idx[0][r] = a; B->val[0][r] = one;
                B->idx[1][z] = b; B->val[1][z] = one;
                B->idx[2][p] = (uint32_t)(1 + n + p); B->val[2][p] = one;
                z++; p++; r++;
                B->ptr[0][r] = (uint32_t)r;
                B->ptr[1][r] = (uint32_t)z;
                B->ptr[2][r] = (uint32_t)p;
            }
            // the gate's sum constraint: (1 * ONE) * (sum of terms) = 0
            B->idx[0][r] = 0; B->val[0][r] = one;
            for (auto& t : terms) { B->idx[1][z] = t.first; B->val[1][z] = t.second; z++; }
            r++;
            B->ptr[0][r] = (uint32_t)r;
            B->ptr[1][r] = (uint32_t)z;
            B->ptr[2][r] = (uint32_t)p;
        }
    };
    on_threads(fill);
    B->n_public = npub;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------ content keys
// 128-bit content key of a text (the caches of decoded keys and lowered circuits are keyed by it; on the verify path it decides which circuit -- hence which
// values count as public inputs -- a text stands for).  Round 6: the home-made multiply-rotate mixer is gone; the key is a tree of SipHash-2-4 with 128-bit
// output (Aumasson & Bernstein, "SipHash: a fast short-input PRF", 2012 -- the reference implementation's outlen = 16 variant, checked below against the
// paper's test vector), keyed PER PROCESS from the OS generator (content keys never leave the process and are never compared across processes):
//     leaf    = SipHash-2-4-128_K (8 KB of text)                        the eight leaves of a 64 KB segment run interleaved in one loop (the rounds of one
//                                                                        state are a serial chain; eight independent states -- AVX2 lanes -- fill the core)
//     record  = the segment's eight leaf digests (a short last segment: one leaf over what is left, the other seven zero)
//     key     = SipHash-2-4-128_K' (record_0 | record_1 | ... | n)      K' = K with k1 complemented: leaves and root are different functions
// Guarantee: two different texts get the same key only through a collision of SipHash-2-4-128 under a key the caller does not know -- the PRF / MAC property the
// function was designed and analysed for (forgery probability ~2^-128 per attempt), not a property of an ad-hoc mixer.  Segments are hashed on up to 16
// threads (a proving key is 0.37 GB of hex text): the host's memory feeds them, not its cores.
struct ContentKey {
    uint64_t h[2] = {0, 0};
    uint64_t len = 0;
    bool operator<(const ContentKey& o) const { return h[0] != o.h[0] ? h[0] < o.h[0] : h[1] != o.h[1] ? h[1] < o.h[1] : len < o.len; }
    bool operator==(const ContentKey& o) const { return h[0] == o.h[0] && h[1] == o.h[1] && len == o.len; }
};
struct SipKey { uint64_t k0, k1; };
static inline uint64_t sip_rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
#define ZK_SIPROUND(v0, v1, v2, v3)                                                                                    \
    do {                                                                                                               \
        v0 += v1; v1 = sip_rotl(v1, 13); v1 ^= v0; v0 = sip_rotl(v0, 32);                                              \
        v2 += v3; v3 = sip_rotl(v3, 16); v3 ^= v2;                                                                     \
        v0 += v3; v3 = sip_rotl(v3, 21); v3 ^= v0;                                                                     \
        v2 += v1; v1 = sip_rotl(v1, 17); v1 ^= v2; v2 = sip_rotl(v2, 32);                                              \
    } while (0)
static inline uint64_t sip_le64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }  // (little-endian hosts only: x86-64)
// SipHash-2-4; out128 != nullptr: the 128-bit variant (out128[0], out128[1]); returns the 64-bit variant's value otherwise
static inline uint64_t siphash24(const uint8_t* in, size_t inlen, SipKey k, uint64_t* out128) {
    uint64_t v0 = 0x736f6d6570736575ULL ^ k.k0, v1 = 0x646f72616e646f6dULL ^ k.k1, v2 = 0x6c7967656e657261ULL ^ k.k0, v3 = 0x7465646279746573ULL ^ k.k1;
    if (out128) v1 ^= 0xee;
    const uint8_t* end = in + (inlen & ~(size_t)7);
    for (; in != end; in += 8) {
        const uint64_t m = sip_le64(in);
        v3 ^= m;
        ZK_SIPROUND(v0, v1, v2, v3);
        ZK_SIPROUND(v0, v1, v2, v3);
        v0 ^= m;
    }
    uint64_t b = (uint64_t)inlen << 56;
    for (size_t i = 0; i < (inlen & 7); i++) b |= (uint64_t)in[i] << (8 * i);
    v3 ^= b;
    ZK_SIPROUND(v0, v1, v2, v3);
    ZK_SIPROUND(v0, v1, v2, v3);
    v0 ^= b;
    v2 ^= out128 ? 0xee : 0xff;
    for (int i = 0; i < 4; i++) ZK_SIPROUND(v0, v1, v2, v3);
    const uint64_t r0 = v0 ^ v1 ^ v2 ^ v3;
    if (!out128) return r0;
    out128[0] = r0;
    v1 ^= 0xdd;
    for (int i = 0; i < 4; i++) ZK_SIPROUND(v0, v1, v2, v3);
    out128[1] = v0 ^ v1 ^ v2 ^ v3;
    return r0;
}
// four SipHash-2-4-128 of four inputs of the SAME length n (a multiple of 8), interleaved: out[j] = siphash24(p[j], n, k) -- the same function, four at a time
static inline void siphash24_128_x4(const uint8_t* const p[4], size_t n, SipKey k, uint64_t out[4][2]) {
    uint64_t v0[4], v1[4], v2[4], v3[4];
    for (int j = 0; j < 4; j++) {
        v0[j] = 0x736f6d6570736575ULL ^ k.k0; v1[j] = 0x646f72616e646f6dULL ^ k.k1 ^ 0xee; v2[j] = 0x6c7967656e657261ULL ^ k.k0; v3[j] = 0x7465646279746573ULL ^ k.k1;
    }
    for (size_t i = 0; i < n; i += 8) {
        uint64_t m[4];
        for (int j = 0; j < 4; j++) { m[j] = sip_le64(p[j] + i); v3[j] ^= m[j]; }
        for (int j = 0; j < 4; j++) ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        for (int j = 0; j < 4; j++) ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        for (int j = 0; j < 4; j++) v0[j] ^= m[j];
    }
    const uint64_t b = (uint64_t)n << 56;
    for (int j = 0; j < 4; j++) {
        v3[j] ^= b;
        ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        v0[j] ^= b;
        v2[j] ^= 0xee;
        for (int i = 0; i < 4; i++) ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        out[j][0] = v0[j] ^ v1[j] ^ v2[j] ^ v3[j];
        v1[j] ^= 0xdd;
        for (int i = 0; i < 4; i++) ZK_SIPROUND(v0[j], v1[j], v2[j], v3[j]);
        out[j][1] = v0[j] ^ v1[j] ^ v2[j] ^ v3[j];
    }
}
// The same eight at a time with AVX2 (four 64-bit lanes per register, two register sets interleaved): what the hashing threads run on x86-64 hosts that have
// it (every EPYC the GPU boxes use); one scalar state costs ~3 cycles per byte, this ~0.4, and sixteen threads are then fed by memory, not by arithmetic.
#if defined(__x86_64__)
#include <immintrin.h>
#define ZK_SIP_AVX2 1
__attribute__((target("avx2"))) static inline __m256i sip_rotl_v(__m256i x, int b) { return _mm256_or_si256(_mm256_slli_epi64(x, b), _mm256_srli_epi64(x, 64 - b)); }
#define ZK_SIPROUND_V(v0, v1, v2, v3)                                                                                                         \
    do {                                                                                                                                      \
        v0 = _mm256_add_epi64(v0, v1); v1 = sip_rotl_v(v1, 13); v1 = _mm256_xor_si256(v1, v0); v0 = _mm256_shuffle_epi32(v0, 0xb1);           \
        v2 = _mm256_add_epi64(v2, v3); v3 = sip_rotl_v(v3, 16); v3 = _mm256_xor_si256(v3, v2);                                                \
        v0 = _mm256_add_epi64(v0, v3); v3 = sip_rotl_v(v3, 21); v3 = _mm256_xor_si256(v3, v0);                                                \
        v2 = _mm256_add_epi64(v2, v1); v1 = sip_rotl_v(v1, 17); v1 = _mm256_xor_si256(v1, v2); v2 = _mm256_shuffle_epi32(v2, 0xb1);           \
    } while (0)
__attribute__((target("avx2"))) static inline void siphash24_128_x8_avx2(const uint8_t* const p[8], size_t n, SipKey k, uint64_t out[8][2]) {
    const __m256i k0 = _mm256_set1_epi64x((long long)k.k0), k1 = _mm256_set1_epi64x((long long)k.k1);
    __m256i a0 = _mm256_xor_si256(_mm256_set1_epi64x(0x736f6d6570736575LL), k0), a1 = _mm256_xor_si256(_mm256_set1_epi64x(0x646f72616e646f6dLL ^ 0xee), k1);
    __m256i a2 = _mm256_xor_si256(_mm256_set1_epi64x(0x6c7967656e657261LL), k0), a3 = _mm256_xor_si256(_mm256_set1_epi64x(0x7465646279746573LL), k1);
    __m256i b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (size_t i = 0; i < n; i += 8) {
        const __m256i ma = _mm256_set_epi64x((long long)sip_le64(p[3] + i), (long long)sip_le64(p[2] + i), (long long)sip_le64(p[1] + i), (long long)sip_le64(p[0] + i));
        const __m256i mb = _mm256_set_epi64x((long long)sip_le64(p[7] + i), (long long)sip_le64(p[6] + i), (long long)sip_le64(p[5] + i), (long long)sip_le64(p[4] + i));
        a3 = _mm256_xor_si256(a3, ma); b3 = _mm256_xor_si256(b3, mb);
        ZK_SIPROUND_V(a0, a1, a2, a3); ZK_SIPROUND_V(b0, b1, b2, b3);
        ZK_SIPROUND_V(a0, a1, a2, a3); ZK_SIPROUND_V(b0, b1, b2, b3);
        a0 = _mm256_xor_si256(a0, ma); b0 = _mm256_xor_si256(b0, mb);
    }
    const __m256i bl = _mm256_set1_epi64x((long long)((uint64_t)n << 56)), ee = _mm256_set1_epi64x(0xee), dd = _mm256_set1_epi64x(0xdd);
    __m256i* st[2][4] = {{&a0, &a1, &a2, &a3}, {&b0, &b1, &b2, &b3}};
    for (int h = 0; h < 2; h++) {
        __m256i v0 = *st[h][0], v1 = *st[h][1], v2 = *st[h][2], v3 = *st[h][3];
        v3 = _mm256_xor_si256(v3, bl);
        ZK_SIPROUND_V(v0, v1, v2, v3); ZK_SIPROUND_V(v0, v1, v2, v3);
        v0 = _mm256_xor_si256(v0, bl);
        v2 = _mm256_xor_si256(v2, ee);
        for (int i = 0; i < 4; i++) ZK_SIPROUND_V(v0, v1, v2, v3);
        alignas(32) uint64_t r0[4], r1[4];
        _mm256_store_si256((__m256i*)r0, _mm256_xor_si256(_mm256_xor_si256(v0, v1), _mm256_xor_si256(v2, v3)));
        v1 = _mm256_xor_si256(v1, dd);
        for (int i = 0; i < 4; i++) ZK_SIPROUND_V(v0, v1, v2, v3);
        _mm256_store_si256((__m256i*)r1, _mm256_xor_si256(_mm256_xor_si256(v0, v1), _mm256_xor_si256(v2, v3)));
        for (int j = 0; j < 4; j++) { out[4 * h + j][0] = r0[j]; out[4 * h + j][1] = r1[j]; }
    }
}
#endif
static inline bool& sip_scalar_only() { static bool v = false; return v; }  // tooling (tools/content_key_bench.cpp): time the scalar form on a CPU that has AVX2
// eight SipHash-2-4-128 of eight inputs of the same length (a multiple of 8): the vector form where the CPU has it, the scalar interleave otherwise -- the same values
static inline void siphash24_128_x8(const uint8_t* const p[8], size_t n, SipKey k, uint64_t out[8][2]) {
#ifdef ZK_SIP_AVX2
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && !sip_scalar_only()) { siphash24_128_x8_avx2(p, n, k, out); return; }
#endif
    siphash24_128_x4(p, n, k, out);
    siphash24_128_x4(p + 4, n, k, out + 4);
}
// the paper's test vector (appendix A: key 00 .. 0f, message 00 .. 0e -> a129ca6149be45e5), the reference implementation's first two 128-bit vectors, and both
// interleaved forms against the plain one; zk_selftest_host calls it
static inline bool siphash_selftest() {
    uint8_t key[16], msg[8 * 136];
    for (int i = 0; i < 16; i++) key[i] = (uint8_t)i;
    for (size_t i = 0; i < sizeof msg; i++) msg[i] = (uint8_t)(i * 7 + (i >> 8));
    for (int i = 0; i < 15; i++) msg[i] = (uint8_t)i;
    SipKey k{sip_le64(key), sip_le64(key + 8)};
    if (siphash24(msg, 15, k, nullptr) != 0xa129ca6149be45e5ULL) return false;
    uint64_t one[2];
    siphash24(msg, 0, k, one);
    if (one[0] != 0xe6a825ba047f81a3ULL || one[1] != 0x930255c71472f66dULL) return false;  // a3817f04ba25a8e6 6df67214c7550293 (vectors_sip128[0])
    siphash24(msg, 1, k, one);
    if (one[0] != 0x44af996bd8c187daULL || one[1] != 0x45fc229b11597634ULL) return false;  // da87c1d86b99af44 347659119b22fc45 (vectors_sip128[1])
    const uint8_t* p[8];
    for (int j = 0; j < 8; j++) p[j] = msg + 136 * j;
    uint64_t x8[8][2], x4[8][2];
    siphash24_128_x8(p, 136, k, x8);
    siphash24_128_x4(p, 136, k, x4);
    siphash24_128_x4(p + 4, 136, k, x4 + 4);
    for (int j = 0; j < 8; j++) {
        siphash24(p[j], 136, k, one);
        if (one[0] != x8[j][0] || one[1] != x8[j][1] || one[0] != x4[j][0] || one[1] != x4[j][1]) return false;
    }
    return true;
}
static inline const SipKey& ck_key() {
    static const SipKey key = [] {
        SipKey k{0, 0};
        bool ok = false;
        if (FILE* f = fopen("/dev/urandom", "rb")) {
            ok = fread(&k, 1, sizeof k, f) == sizeof k;
            fclose(f);
        }
        if (!ok) {  // no OS generator: what varies per process (address-space layout, clock, pid) -- weaker, and said so; the keys stay private to the process
            k.k0 = (uint64_t)(uintptr_t)&k ^ ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
            k.k1 = (uint64_t)(uintptr_t)&siphash_selftest ^ 0x9e3779b97f4a7c15ULL * (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
        }
        return k;
    }();
    return key;
}
static inline ContentKey content_key(const char* p_, size_t n) {
    const uint8_t* p = (const uint8_t*)p_;
    const size_t SEG = (size_t)1 << 16, LEAF = SEG / 8;
    const size_t nseg = (n + SEG - 1) / SEG;
    std::vector<uint64_t> rec(16 * (nseg ? nseg : 1), 0);  // per segment: eight leaf digests
    const SipKey key = ck_key();
    auto run = [&](size_t lo, size_t hi) {
        for (size_t s = lo; s < hi; s++) {
            const uint8_t* q = p + s * SEG;
            const size_t len = s + 1 == nseg ? n - s * SEG : SEG;
            uint64_t(*out)[2] = reinterpret_cast<uint64_t(*)[2]>(&rec[16 * s]);
            if (len == SEG) {
                const uint8_t* lanes[8];
                for (int j = 0; j < 8; j++) lanes[j] = q + j * LEAF;
                siphash24_128_x8(lanes, LEAF, key, out);
            } else {
                siphash24(q, len, key, out[0]);  // (the length is part of SipHash's last block and of the root's input: a short leaf cannot pass for a full one)
            }
        }
    };
    unsigned nt = nseg < 64 ? 1 : std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;  // (32 threads for the 0.26 + 0.37 GB of a 2^20-constraint call were measured: no faster -- the host's memory feeds them, not its cores)
    if (nt <= 1) run(0, nseg);
    else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++) th.emplace_back(run, nseg * k / nt, nseg * (k + 1) / nt);
        for (auto& t : th) t.join();
    }
    rec.push_back((uint64_t)n);
    ContentKey K;
    K.len = n;
    siphash24(reinterpret_cast<const uint8_t*>(rec.data()), rec.size() * 8, SipKey{key.k0, ~key.k1}, K.h);
    return K;
}

}  // namespace zkmi
