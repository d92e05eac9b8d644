// Test program for include/zkmi.hpp (the C++ host mirror): reads a blob written by tests/test_gpu_parity.py
//   u64 n | n x 64 B G1 points | n x 32 B scalars (Montgomery) | 64 B expected MSM | n x 32 B the same scalars in regular form | u64 log_n | 2^log_n x 32 B input | 2^log_n x 32 B expected FFT(DIF)
// runs the gnark-crypto-shaped calls and checks results AND upstream's error behaviour.  Without arguments it only exercises the
// error paths that need no device (used by the CPU suite).
#include <cstdio>
#include <cstring>
#include <fstream>

#include "zkmi.hpp"

using namespace zkmi;

static int fail(const char* what) {
    std::printf("FAIL: %s\n", what);
    return 1;
}

int main(int argc, char** argv) {
    // upstream's two MultiExp errors come back before any device work
    {
        std::vector<bn254::G1Affine> pts(3);
        fr::Vector sc(2);
        bn254::G1Affine out;
        Error e = out.MultiExp(pts, sc);
        if (e.code != ZK_ERR_LEN) return fail("len(points) != len(scalars) must be ZK_ERR_LEN");
        sc.resize(3);
        ecc::MultiExpConfig cfg;
        cfg.NbTasks = 1025;
        e = out.MultiExp(pts, sc, cfg);
        if (e.code != ZK_ERR_NB_TASKS) return fail("NbTasks > 1024 must be ZK_ERR_NB_TASKS");
        fft::Domain d = fft::Domain::NewDomain(5);
        if (d.Cardinality != 8) return fail("NewDomain(5).Cardinality == 8");
        fr::Vector a(7);
        if (!d.FFT(a, fft::DIF)) return fail("len(a) != Cardinality must fail");
    }
    // the PLONK / KZG mirrors validate lengths before any pointer crosses the C ABI
    {
        kzg::SRS srs;
        fr::Vector p(4);
        zk_g1_affine d;
        if (srs.Commit(p, &d).code != ZK_ERR_LEN) return fail("kzg.Commit with a polynomial larger than the SRS must be ZK_ERR_LEN");
        plonk::ProvingKey pk;
        plonk::Proof pr;
        fr::Vector sol(3);
        fr::Element bl[9] = {};
        if (plonk::Prove(pk, sol, bl, &pr).code != ZK_ERR_LEN) return fail("plonk.Prove with a solution of the wrong length must be ZK_ERR_LEN");
        std::vector<uint32_t> xa(2), xb(1), xc(2);
        if (pk.ReadFrom("", 0, false, 3, xa, xb, xc, srs).code != ZK_ERR_LEN) return fail("ReadFrom with ragged wire-id arrays must be ZK_ERR_LEN");
    }
    // groth16.ProvingKey.ReadFrom: a truncated / malformed image is refused while its header is parsed, before any device work
    {
        groth16::ProvingKey pk;
        const uint8_t three[3] = {0, 0, 16};
        if (pk.ReadFrom(three, 3).code != ZK_ERR_LEN) return fail("ReadFrom of 3 bytes must be ZK_ERR_LEN");
        uint8_t head[300] = {};
        head[7] = 17;  // a cardinality that is not a power of two
        if (pk.ReadFrom(head, sizeof head).code != ZK_ERR_ARG) return fail("ReadFrom with cardinality 17 must be ZK_ERR_ARG");
        if (pk.ReadFrom("0g", 2, true).code != ZK_ERR_LEN) return fail("ReadFrom of a 2-character hex text must be ZK_ERR_LEN");
        fr::Vector a(2), w(5);
        fr::Element r = {}, s = {};
        groth16::Proof pr;
        if (groth16::Prove(pk, a, a, a, w, r, s, &pr).code != ZK_ERR_LEN) return fail("groth16.Prove with len(a) > the (unloaded) key's domain must be ZK_ERR_LEN");
    }
    // the verifiers run on the host: a truncated verifying key is an error, not a verdict
    {
        groth16::Proof gp = {};
        plonk::Proof pp = {};
        kzg::SRS srs;
        bool ok = true;
        std::vector<uint8_t> short_vk(100);
        fr::Vector none;
        if (groth16::Verify(gp, short_vk, none, &ok).code != ZK_ERR_LEN || ok) return fail("groth16.Verify with a 100-byte key must be ZK_ERR_LEN");
        if (plonk::Verify(pp, short_vk, srs, none, &ok).code != ZK_ERR_LEN || ok) return fail("plonk.Verify with a 100-byte key must be ZK_ERR_LEN");
    }
    if (argc < 2) {
        std::printf("ok (error paths only)\n");
        return 0;
    }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) return fail("cannot open blob");
    uint64_t n = 0, log_n = 0;
    f.read((char*)&n, 8);
    std::vector<bn254::G1Affine> pts(n);
    fr::Vector sc(n);
    bn254::G1Affine want, got;
    f.read((char*)pts.data(), n * 64);
    f.read((char*)sc.data(), n * 32);
    f.read((char*)&want, 64);
    ecc::MultiExpConfig mont;
    mont.ScalarsMont = true;
    Error e = got.MultiExp(pts, sc, mont);
    if (e) {
        std::printf("MultiExp: %s\n", e.msg.c_str());
        return 1;
    }
    if (std::memcmp(&got, &want, 64)) return fail("MultiExp result (ScalarsMont: true)");
    // ecc.MultiExpConfig{} -- upstream's zero value -- takes the limbs in regular form (what gnark's prover passes after FromMont())
    f.read((char*)sc.data(), n * 32);
    std::memset(&got, 0, sizeof got);
    if ((e = got.MultiExp(pts, sc))) return fail(e.msg.c_str());
    if (std::memcmp(&got, &want, 64)) return fail("MultiExp result (default config, regular-form scalars)");
    f.read((char*)&log_n, 8);
    const size_t N = size_t(1) << log_n;
    fr::Vector a(N), expect(N);
    f.read((char*)a.data(), N * 32);
    f.read((char*)expect.data(), N * 32);
    fr::Vector orig = a;
    fft::Domain dom = fft::Domain::NewDomain(N);
    if ((e = dom.FFT(a, fft::DIF))) return fail(e.msg.c_str());
    if (std::memcmp(a.data(), expect.data(), N * 32)) return fail("FFT(DIF) result");
    if ((e = dom.FFTInverse(a, fft::DIT))) return fail(e.msg.c_str());   // DIF then inverse DIT: back to the input, natural order
    if (std::memcmp(a.data(), orig.data(), N * 32)) return fail("FFTInverse(DIT) o FFT(DIF) != identity");
    std::printf("ok\n");
    return 0;
}
