// C++ host-side mirror of the gnark-crypto / gnark interfaces that libzkmi replaces (header-only, over the C ABI of zkmi.h).
// The reference's host language is Go (no toolchain in this image), so this is what a C++ caller -- or the cgo shim of
// INTEGRATION.md, line for line -- writes: same names, same argument meaning, same error behaviour as
//   gnark-crypto v0.9.1  ecc/bn254  (*G1Affine).MultiExp / (*G2Affine).MultiExp, ecc.MultiExpConfig      [REF gnark_backend_ffi/go.mod:5]
//   gnark-crypto v0.9.1  ecc/bn254/fr/fft  NewDomain, (*Domain).FFT / FFTInverse, BitReverse, Decimation
//   gnark v0.8.0         groth16.Prove (with the prover randomness as arguments)                          [REF gnark_backend_ffi/main.go:131]
//   the reference's      DeserializeFelts                                                                 [REF internal/backend/helpers.go:24-33]
// Go's `(value, error)` returns become `Error` return values (nil == ok()); nothing throws.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "zkmi.h"

namespace zkmi {

struct Error {
    int code = ZK_OK;
    std::string msg;
    bool ok() const { return code == ZK_OK; }
    explicit operator bool() const { return code != ZK_OK; }  // `if (err)` reads like Go's `if err != nil`
};
inline Error make_error(int rc) {
    if (rc == ZK_OK) return Error{};
    const char* m = zk_last_error();
    return Error{rc, m ? m : ""};
}

namespace fr {
typedef zk_fr Element;  // Montgomery, 4 x u64 little-endian limbs: gnark-crypto's memory image
typedef std::vector<Element> Vector;
}  // namespace fr

namespace ecc {
// ecc.MultiExpConfig{NbTasks int; ScalarsMont bool}
struct MultiExpConfig {
    int NbTasks = 0;
    bool ScalarsMont = false;  // upstream's zero value: the scalars' limbs are taken in REGULAR form (gnark v0.8.0's prover calls
                               // FromMont() on the wire values and on h, then passes MultiExpConfig{}); true: Montgomery fr.Elements
};
}  // namespace ecc

namespace bn254 {

struct G1Affine : zk_g1_affine {
    // func (p *G1Affine) MultiExp(points []G1Affine, scalars []fr.Element, config ecc.MultiExpConfig) (*G1Affine, error)
    // errors: "len(points) != len(scalars)" (ZK_ERR_LEN), "invalid config: config.NbTasks > 1024" (ZK_ERR_NB_TASKS)
    Error MultiExp(const std::vector<G1Affine>& points, const fr::Vector& scalars, const ecc::MultiExpConfig& config = ecc::MultiExpConfig()) {
        zk_msm_cfg c = {config.NbTasks, config.ScalarsMont ? 1 : 0, 0, 0};
        return make_error(zk_bn254_g1_msm(points.data(), points.size(), scalars.data(), scalars.size(), &c, this));
    }
    bool IsInfinity() const {
        for (int i = 0; i < 4; i++)
            if (x.l[i] | y.l[i]) return false;
        return true;
    }
};

struct G2Affine : zk_g2_affine {
    Error MultiExp(const std::vector<G2Affine>& points, const fr::Vector& scalars, const ecc::MultiExpConfig& config = ecc::MultiExpConfig()) {
        zk_msm_cfg c = {config.NbTasks, config.ScalarsMont ? 1 : 0, 0, 0};
        return make_error(zk_bn254_g2_msm(points.data(), points.size(), scalars.data(), scalars.size(), &c, this));
    }
};

}  // namespace bn254

namespace fft {

enum Decimation { DIT = ZK_DIT, DIF = ZK_DIF };  // same iota order as gnark-crypto

// fft.Domain: only the cardinality travels; the twiddle / coset tables are device-resident and cached per size inside libzkmi
class Domain {
public:
    uint64_t Cardinality = 1;
    // fft.NewDomain(m): next power of two >= m
    static Domain NewDomain(uint64_t m) {
        Domain d;
        while (d.Cardinality < m) d.Cardinality <<= 1;
        return d;
    }
    // (*Domain).FFT(a, decimation, coset...): in place; DIF natural -> bit-reversed, DIT bit-reversed -> natural
    Error FFT(fr::Vector& a, Decimation decimation, bool coset = false) const { return run(a, 0, decimation, coset); }
    // (*Domain).FFTInverse: same data movement with the inverse twiddles, scaled by 1/N (and the inverse coset table)
    Error FFTInverse(fr::Vector& a, Decimation decimation, bool coset = false) const { return run(a, 1, decimation, coset); }

private:
    Error run(fr::Vector& a, int inverse, Decimation decimation, bool coset) const {
        if (a.size() != Cardinality) return Error{ZK_ERR_ARG, "len(a) != domain.Cardinality"};
        uint32_t log_n = 0;
        while ((uint64_t(1) << log_n) < Cardinality) log_n++;
        return make_error(zk_bn254_ntt(a.data(), log_n, inverse, (int)decimation, coset ? 1 : 0));
    }
};

// fft.BitReverse(a): len(a) must be a power of two
inline Error BitReverse(fr::Vector& a) {
    uint32_t log_n = 0;
    while ((size_t(1) << log_n) < a.size()) log_n++;
    if ((size_t(1) << log_n) != a.size()) return Error{ZK_ERR_ARG, "len(a) is not a power of two"};
    return make_error(zk_bn254_bit_reverse(a.data(), log_n));
}

}  // namespace fft

namespace groth16 {

// groth16.ProvingKey resident in HBM (RAII over zk_bn254_groth16_pk_load / _free); keeps the geometry the prover validates against
class ProvingKey {
public:
    ProvingKey() = default;
    ProvingKey(const ProvingKey&) = delete;
    ProvingKey& operator=(const ProvingKey&) = delete;
    ~ProvingKey() {
        if (handle_) zk_bn254_groth16_pk_free(handle_);
    }
    // pk.infinity_a / infinity_b (gnark's InfinityA / InfinityB) select gnark's compact A / B / G2.B layout
    Error Load(const zk_groth16_pk& pk) {
        Error e = make_error(zk_bn254_groth16_pk_load(&pk, &handle_));
        if (!e.ok()) return e;
        return make_error(zk_bn254_groth16_pk_info(handle_, &n_wires_, &n_public_, &log_domain_, nullptr));
    }
    // (*ProvingKey).ReadFrom on the bytes of WriteTo (hex = true: the text the reference's ProveWithPK receives, backend/groth16/r1cs.go:107-128);
    // every point is decompressed on the device
    Error ReadFrom(const void* data, size_t len, bool hex = false) {
        Error e = make_error(zk_bn254_groth16_pk_read(data, len, hex ? 1 : 0, 0, 0, &handle_));
        if (!e.ok()) return e;
        return make_error(zk_bn254_groth16_pk_info(handle_, &n_wires_, &n_public_, &log_domain_, nullptr));
    }
    // (*ProvingKey).WriteTo: compressed points, A / B / G2.B without their points at infinity, then InfinityA / InfinityB
    Error WriteTo(std::vector<uint8_t>* out, bool hex = false) const {
        size_t need = 0;
        Error e = make_error(zk_bn254_groth16_pk_write(handle_, hex ? 1 : 0, nullptr, 0, &need));
        if (!e.ok()) return e;
        out->resize(need);
        return make_error(zk_bn254_groth16_pk_write(handle_, hex ? 1 : 0, out->data(), out->size(), &need));
    }
    uint64_t handle() const { return handle_; }
    size_t NbWires() const { return n_wires_; }
    size_t NbPublic() const { return n_public_; }
    uint64_t DomainCardinality() const { return uint64_t(1) << log_domain_; }

private:
    uint64_t handle_ = 0;
    size_t n_wires_ = 0, n_public_ = 0;
    uint32_t log_domain_ = 0;
};

struct Proof {
    uint8_t bytes[128];  // Proof.WriteTo: Ar | Bs | Krs, gnark's compressed encodings
};

// groth16.Prove after the solver: a, b, c = evaluations of the constraint system, w = wire values, (r, s) = the prover's randomness.
// All four vectors hold Montgomery fr.Elements (the solver's containers).  Length errors come back as ZK_ERR_LEN -- the library
// reads exactly NbWires() elements of w and at most DomainCardinality() of a, b, c, so nothing is read out of bounds.
inline Error Prove(const ProvingKey& pk, const fr::Vector& a, const fr::Vector& b, const fr::Vector& c, const fr::Vector& w, const fr::Element& r,
                   const fr::Element& s, Proof* proof) {
    if (a.size() != b.size() || a.size() != c.size()) return Error{ZK_ERR_LEN, "len(a), len(b), len(c) differ"};
    if (a.size() > pk.DomainCardinality()) return Error{ZK_ERR_LEN, "len(a) exceeds the domain cardinality"};
    if (w.size() != pk.NbWires()) return Error{ZK_ERR_LEN, "len(w) != number of wires of the proving key"};
    return make_error(zk_bn254_groth16_prove(pk.handle(), a.data(), b.data(), c.data(), a.size(), w.data(), w.size(), &r, &s, 0, proof->bytes));
}

}  // namespace groth16

namespace kzg {

// kzg.SRS with the G1 side resident in HBM (RAII over the registered base array).  kzg.NewSRS / (*SRS).ReadFrom / WriteTo / kzg.Commit.
class SRS {
public:
    SRS() = default;
    SRS(const SRS&) = delete;
    SRS& operator=(const SRS&) = delete;
    ~SRS() {
        if (handle_) zk_bn254_bases_free(handle_);
    }
    // kzg.NewSRS(size, alpha): alpha^i * G1 generated on the device; alpha is a Montgomery fr.Element
    Error New(uint64_t size, const fr::Element& alpha) {
        void* d = nullptr;
        Error e = make_error(zk_dev_alloc(&d, size * 64));
        if (!e.ok()) return e;
        e = make_error(zk_bn254_kzg_new_srs_dev(d, size, &alpha, G2, nullptr));
        if (e.ok()) e = make_error(zk_bn254_bases_register_dev(d, size, 0, &handle_));
        zk_dev_free(d);
        if (e.ok()) size_ = size;
        return e;
    }
    // (*SRS).ReadFrom on the bytes of WriteTo (hex = true: the text of srs.hex); the G1 points are decompressed on the device
    Error ReadFrom(const void* data, size_t len, bool hex = false) {
        size_t n = 0;
        Error e = make_error(zk_bn254_kzg_srs_read(data, len, hex ? 1 : 0, 0, &handle_, &n, G2));
        if (e.ok()) size_ = n;
        return e;
    }
    Error WriteTo(std::vector<uint8_t>* out, bool hex = false) const {
        size_t need = (132 + 32 * size_) * (hex ? 2 : 1), n = 0;
        out->resize(need);
        return make_error(zk_bn254_kzg_srs_write(handle_, G2, hex ? 1 : 0, out->data(), out->size(), &n));
    }
    // kzg.Commit(p, srs) = MultiExp(srs.G1[:len(p)], p); p holds Montgomery fr.Elements
    Error Commit(const fr::Vector& p, zk_g1_affine* digest) const {
        if (p.size() > size_) return Error{ZK_ERR_LEN, "kzg: invalid polynomial size (larger than SRS or == 0)"};
        zk_msm_cfg c = {0, 1, 0, 0};
        return make_error(zk_bn254_msm_bases(handle_, 0, p.data(), p.size(), &c, digest));
    }
    // the three simultaneous commitments of plonk.Prove's rounds 1 and 3 in one call (zk_bn254_msm_bases_batch): digests[k] = Commit(*polys[k])
    Error CommitBatch(const std::vector<const fr::Vector*>& polys, zk_g1_affine* digests) const {
        std::vector<const zk_fr*> p;
        for (const fr::Vector* v : polys) {
            if (v->size() > size_ || v->size() != polys[0]->size()) return Error{ZK_ERR_LEN, "kzg: invalid polynomial size (larger than SRS, or the batch is ragged)"};
            p.push_back(v->data());
        }
        zk_msm_cfg c = {0, 1, 0, 0};
        return make_error(zk_bn254_msm_bases_batch(handle_, 0, p.data(), (uint32_t)p.size(), polys.empty() ? 0 : polys[0]->size(), &c, digests));
    }
    uint64_t handle() const { return handle_; }
    uint64_t Size() const { return size_; }
    zk_g2_affine G2[2] = {};

private:
    uint64_t handle_ = 0;
    uint64_t size_ = 0;
};

}  // namespace kzg

namespace plonk {

// plonk.ProvingKey resident in HBM; plonk.Setup / ReadFrom / WriteTo / plonk.Prove (the reference's live path: backend/plonk/plonk.go:21,67)
class ProvingKey {
public:
    ProvingKey() = default;
    ProvingKey(const ProvingKey&) = delete;
    ProvingKey& operator=(const ProvingKey&) = delete;
    ~ProvingKey() {
        if (handle_) zk_bn254_plonk_pk_free(handle_);
    }
    // plonk.Setup(spr, srs): the gates as BuildSparseR1CS emits them (backend/plonk/sparse_r1cs.go:44-107)
    Error Setup(const zk_plonk_circuit& spr, const kzg::SRS& srs) {
        n_vars_ = spr.n_vars;
        return make_error(zk_bn254_plonk_setup(&spr, srs.handle(), &handle_, &Vk));
    }
    // ProvingKey.ReadFrom on gnark's bytes (or their hex text) + the wire ids of the rebuilt spr
    Error ReadFrom(const void* data, size_t len, bool hex, size_t n_vars, const std::vector<uint32_t>& xa, const std::vector<uint32_t>& xb, const std::vector<uint32_t>& xc,
                   const kzg::SRS& srs) {
        if (xa.size() != xb.size() || xa.size() != xc.size()) return Error{ZK_ERR_LEN, "wire-id arrays differ in length"};
        n_vars_ = n_vars;
        return make_error(zk_bn254_plonk_pk_read(data, len, hex ? 1 : 0, n_vars, xa.size(), xa.data(), xb.data(), xc.data(), srs.handle(), &handle_));
    }
    Error WriteTo(std::vector<uint8_t>* out, bool hex = false) const {
        size_t n = 0;
        uint8_t dummy = 0;
        zk_bn254_plonk_pk_write(handle_, hex ? 1 : 0, &dummy, 0, &n);  // size query
        out->resize(n);
        return make_error(zk_bn254_plonk_pk_write(handle_, hex ? 1 : 0, out->data(), out->size(), &n));
    }
    uint64_t handle() const { return handle_; }
    size_t NbVariables() const { return n_vars_; }
    zk_plonk_vk Vk = {};

private:
    uint64_t handle_ = 0;
    size_t n_vars_ = 0;
};

struct Proof {
    uint8_t bytes[ZK_PLONK_PROOF_BYTES];  // Proof.WriteTo
};

// plonk.Prove after spr.Solve: solution = the values of all variables (public first), blinders = the nine fr.SetRandom draws of Blind(),
// challenges = nullptr (SHA-256 Fiat-Shamir as upstream) or five pinned values
inline Error Prove(const ProvingKey& pk, const fr::Vector& solution, const fr::Element blinders[9], Proof* proof, const fr::Element* challenges = nullptr) {
    if (solution.size() != pk.NbVariables()) return Error{ZK_ERR_LEN, "len(solution) != number of variables of the constraint system"};
    return make_error(zk_bn254_plonk_prove(pk.handle(), solution.data(), solution.size(), 0, blinders, challenges, proof->bytes));
}

// plonk.Verify(proof, vk, publicWitness) on the host: vk = the bytes of VerifyingKey.WriteTo, srs supplies the two G2 points InitKZG attaches.
// *accepted <- the verdict; malformed encodings come back as errors (what upstream's ReadFrom returns).
inline Error Verify(const Proof& proof, const std::vector<uint8_t>& vk, const kzg::SRS& srs, const fr::Vector& public_witness, bool* accepted) {
    int ok = 0;
    Error e = make_error(zk_bn254_plonk_verify(proof.bytes, vk.data(), vk.size(), 0, srs.G2, public_witness.data(), public_witness.size(), &ok));
    *accepted = ok != 0;
    return e;
}

}  // namespace plonk

namespace groth16 {
// groth16.Verify(proof, vk, publicWitness) on the host: vk = the bytes of VerifyingKey.WriteTo; the public witness excludes the constant wire
inline Error Verify(const Proof& proof, const std::vector<uint8_t>& vk, const fr::Vector& public_witness, bool* accepted) {
    int ok = 0;
    Error e = make_error(zk_bn254_groth16_verify(proof.bytes, vk.data(), vk.size(), 0, public_witness.data(), public_witness.size(), &ok));
    *accepted = ok != 0;
    return e;
}
}  // namespace groth16

// DeserializeFelts(encodedFelts string): hex(u32 BE count || count x 32 B BE) -> Montgomery vector, decoded on the device into d_out
inline Error DeserializeFelts(const std::string& encoded, void* d_out, size_t capacity, size_t* n) {
    return make_error(zk_bn254_felts_decode_hex(encoded.data(), encoded.size(), d_out, capacity, n));
}

}  // namespace zkmi
