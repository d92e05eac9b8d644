// R1CS on the device: the steps of gnark's Groth16 backend either side of the hot path
//     groth16.Setup(r1cs)                    gnark v0.8.0 internal/backend/bn254/groth16/setup.go   reached at /root/reference/gnark_backend_ffi/main.go:121
//     r1cs.Solve -> a, b, c (solution)       gnark constraint/bn254 solver                          reached inside groth16.Prove, main.go:131
//     buildR1CS(RawR1CS)                     the reference's intended (commented-out) Groth16 FFI    backend/groth16/r1cs.go:9-72,
//                                            payload src/gnark_backend_wrapper/groth16/acir_to_r1cs.rs:18-60
// The constraint system is three sparse matrices L, R, O (rows = constraints, columns = wires [ONE, public..., secret..., internal...]) with
// (L w) o (R w) = (O w).  On the device:
//   * a, b, c = L w, R w, O w: one lane per row of each matrix (CSR), coalesced over the row pointers -- the "sparse mat-vec" of SURVEY §8(f)3
//   * Setup from explicit toxic waste (tau, alpha, beta, gamma, delta -- upstream draws them; pinning them is what makes a key reproducible):
//     Lagrange basis at tau (closed form + one inversion per element), A_i(tau) / B_i(tau) / C_i(tau) as the TRANSPOSED products (one lane per
//     wire over the CSC form built on the host), K / Z scalars, then the fixed-base scalar multiplications [x]G1 / [x]G2 (util.hip: 8-bit windows over
//     a 32 x 255 table of the generator, one shared inversion per lane) straight into a resident proving key (window tables included) -- nothing is
//     staged through the host.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <vector>

#include "ctx.hpp"
#include "curve.hpp"
#include "ff.hpp"
#include "fixedbase.hpp"
#include "host_ff.hpp"
#include "ntt.hpp"

namespace zkmi {

static Fr todev(const HFr& h) {
    Fr r;
    memcpy(&r, &h, 32);
    return r;
}
__device__ __forceinline__ Fr ldf(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

struct Csr {
    uint32_t* ptr = nullptr;  // rows + 1
    uint32_t* idx = nullptr;  // nnz
    Fr* val = nullptr;        // nnz
    size_t nnz = 0;
};
struct R1csDev {
    size_t n_constraints = 0, n_wires = 0, n_public = 0;  // n_public includes the ONE wire
    Csr row[3];  // L, R, O by constraint (CSR)
    Csr col[3];  // the same by wire (CSC): Setup's transposed products -- built on the device when a Setup first asks for them (ensure_csc): a prover never does
    bool have_csc = false;
    std::mutex csc_mu;
    std::vector<void*> allocs;
};
static std::mutex g_r1cs_mu;
// shared ownership: a call that looked a system up keeps it alive even if another thread frees the handle meanwhile
static std::map<uint64_t, std::shared_ptr<R1csDev>> g_r1cs;
static uint64_t g_next_r1cs = 1;

static void r1cs_destroy(R1csDev* r) {
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}
template <class T>
static int dalloc(R1csDev* r, T** out, size_t n) {
    void* p = nullptr;
    ZK_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    r->allocs.push_back(p);
    *out = (T*)p;
    return ZK_OK;
}

// out[m * rows + i] = sum_k val[k] * x[idx[k]] over row i of matrix m  (a, b, c contiguous: 3 x rows)
struct Csr3 {
    const uint32_t* ptr[3];
    const uint32_t* idx[3];
    const Fr* val[3];
};
// Rows longer than SPMV_LONG entries are left to k_spmv3_long (listed through `long_rows`: [0] = count, then (matrix, row) pairs): in the by-wire form
// Setup multiplies with, the ONE wire's column holds an entry for every gate's sum constraint -- 2^19 of them at 2^20 constraints -- and one lane walking it alone
// took 0.52 s of a 0.59 s Setup (profiles/rnd5_e_groth16_export_2p20.json).
constexpr uint32_t SPMV_LONG = 1024, SPMV_LONG_MAX = 4096;
__global__ __launch_bounds__(256) void k_spmv3(Csr3 M, const Fr* __restrict__ x, size_t rows, Fr* __restrict__ out0, Fr* __restrict__ out1, Fr* __restrict__ out2,
                                               uint32_t* __restrict__ long_rows) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (i >= rows) return;
    const uint32_t b = M.ptr[m][i], e = M.ptr[m][i + 1];
    if (long_rows && e - b > SPMV_LONG) {
        const uint32_t slot = atomicAdd(&long_rows[0], 1u);
        if (slot < SPMV_LONG_MAX) {  // (more long rows than the list holds: they are summed here after all, slowly and correctly)
            long_rows[1 + 2 * slot] = (uint32_t)m;
            long_rows[2 + 2 * slot] = (uint32_t)i;
            return;
        }
    }
    Fr acc = Fr::zero();
    for (uint32_t k = b; k < e; k++) acc = acc + ldf(M.val[m] + k) * ldf(x + M.idx[m][k]);
    (m == 0 ? out0 : m == 1 ? out1 : out2)[i] = acc;
}
// one workgroup per listed row: 256 partial sums over strided entries, folded through LDS
__global__ __launch_bounds__(256) void k_spmv3_long(Csr3 M, const Fr* __restrict__ x, Fr* __restrict__ out0, Fr* __restrict__ out1, Fr* __restrict__ out2,
                                                    const uint32_t* __restrict__ long_rows) {
    __shared__ Fr part[256];
    const uint32_t cnt = long_rows[0] < SPMV_LONG_MAX ? long_rows[0] : SPMV_LONG_MAX;
    for (uint32_t r = blockIdx.x; r < cnt; r += gridDim.x) {
        const int m = (int)long_rows[1 + 2 * r];
        const size_t i = long_rows[2 + 2 * r];
        const uint32_t b = M.ptr[m][i], e = M.ptr[m][i + 1];
        Fr acc = Fr::zero();
        for (uint32_t k = b + threadIdx.x; k < e; k += 256) acc = acc + ldf(M.val[m] + k) * ldf(x + M.idx[m][k]);
        part[threadIdx.x] = acc;
        __syncthreads();
        for (unsigned d = 128; d > 0; d >>= 1) {
            if (threadIdx.x < d) part[threadIdx.x] = part[threadIdx.x] + part[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) (m == 0 ? out0 : m == 1 ? out1 : out2)[i] = part[0];
        __syncthreads();
    }
}
// Lagrange basis of the size-N domain at tau: lag_j = (tau^N - 1) / N * w^j / (tau - w^j)
__global__ __launch_bounds__(256) void k_lagrange_at(const Fr* __restrict__ tw, uint32_t N, Fr tau, Fr scale /* (tau^N - 1) / N */, Fr* __restrict__ out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t h = N >> 1;
    Fr w = (N == 1) ? Fr::one() : (j < h ? ldf(tw + j) : Fr::zero() - ldf(tw + (j - h)));
    out[j] = scale * w * (tau - w).inv();
}
// K-part scalars: (beta A_i + alpha B_i + C_i) * inv, inv = 1/gamma for the public wires and 1/delta for the others
__global__ void k_k_scalars(const Fr* __restrict__ A, const Fr* __restrict__ B, const Fr* __restrict__ Cc, size_t n, size_t n_public, Fr alpha, Fr beta, Fr inv_gamma, Fr inv_delta,
                            Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = beta * ldf(A + i) + alpha * ldf(B + i) + ldf(Cc + i);
    out[i] = v * (i < n_public ? inv_gamma : inv_delta);
}
// Z scalars in the order gnark stores pk.G1.Z (bit-reversed): out[bitrev(i)] = tau^i * (tau^N - 1) / delta
struct PowTab {
    Fr pw[28];
};
__global__ void k_z_scalars(PowTab tau_pw, uint32_t N, unsigned logN, Fr scale, Fr* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Fr acc = scale;
    for (unsigned b = 0; b < 28; b++)
        if ((i >> b) & 1) acc = acc * tau_pw.pw[b];
    uint32_t r = logN ? (__brev(i) >> (32 - logN)) : 0;
    out[r] = acc;
}
static unsigned gridn(size_t n) { return (unsigned)((n + 255) / 256); }

// CSR -> CSC on the device: count the entries of every wire, scan, scatter through per-wire cursors.  Inside a column the entries land in whatever order the
// lanes arrive -- the transposed product sums them in the field, where the order of additions does not exist.
// Column 0 is the ONE wire: every constraint with a constant names it (half a million times in R at 2^20 constraints of the export path's systems), and that
// many atomics on one address take 25 ms by themselves.  The lanes of a wave that hold it are counted with one ballot and served by one atomic.
__global__ void k_csc_count(const uint32_t* __restrict__ idx, size_t nnz, uint32_t* __restrict__ cnt) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t col = k < nnz ? idx[k] : 0xffffffffu;
    const unsigned long long one = __ballot(col == 0u);
    if (col == 0u) {
        if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)one) - 1)) atomicAdd(&cnt[0], (uint32_t)__popcll(one));
    } else if (k < nnz) {
        atomicAdd(&cnt[col], 1u);
    }
}
// cnt[0 .. n) -> exclusive prefix sums in ptr[0 .. n], one workgroup (a Setup step: 2^20 wires in ~0.1 ms; not worth a multi-pass scan)
__global__ __launch_bounds__(1024) void k_csc_scan(const uint32_t* __restrict__ cnt, size_t n, uint32_t* __restrict__ ptr) {
    __shared__ uint32_t part[1024];
    const size_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (size_t i = lo; i < hi; i++) sum += cnt[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (size_t i = lo; i < hi; i++) {
        ptr[i] = run;
        run += cnt[i];
    }
    if (threadIdx.x == 1023) ptr[n] = part[1023];
}
__global__ void k_csc_fill(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx, const Fr* __restrict__ val, size_t rows, uint32_t* __restrict__ cursor,
                           uint32_t* __restrict__ cidx, Fr* __restrict__ cval) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    uint32_t k = i < rows ? ptr[i] : 0u;
    const uint32_t end = i < rows ? ptr[i + 1] : 0u;
    while (__ballot(k < end)) {  // (the whole wave stays in step: the ONE wire's entries of one step share one atomic, as in k_csc_count)
        const bool act = k < end;
        const uint32_t col = act ? idx[k] : 0xffffffffu;
        const bool hot = col == 0u;
        const unsigned long long one = __ballot(hot);
        uint32_t pos = 0;
        if (one) {
            const int leader = __ffsll((long long)one) - 1;
            uint32_t base = 0;
            if (lane == (unsigned)leader) base = atomicAdd(&cursor[0], (uint32_t)__popcll(one));
            base = __shfl(base, leader, 64);
            pos = base + (uint32_t)__popcll(one & ((1ull << lane) - 1ull));
        }
        if (act && !hot) pos = atomicAdd(&cursor[col], 1u);
        if (act) {
            cidx[pos] = (uint32_t)i;
            cval[pos] = ldf(val + k);
            k++;
        }
    }
}

static int lookup_r1cs(uint64_t h, std::shared_ptr<R1csDev>* out) {
    std::lock_guard<std::mutex> lk(g_r1cs_mu);
    auto it = g_r1cs.find(h);
    if (it == g_r1cs.end()) return set_err(ZK_ERR_HANDLE, "unknown R1CS handle %llu", (unsigned long long)h);
    *out = it->second;
    return ZK_OK;
}

int groth16_pk_adopt(uint64_t handle);  // groth16.hip: the key takes ownership of its device arrays

// the by-wire (CSC) form of the three matrices, built once, on the device, when a Setup asks for it
static int ensure_csc(R1csDev* D, Slot* s, hipStream_t st) {
    std::lock_guard<std::mutex> lk(D->csc_mu);
    if (D->have_csc) return ZK_OK;
    const size_t nw = D->n_wires, nc = D->n_constraints;
    uint32_t* cnt = nullptr;
    ZK_HIP(hipMalloc((void**)&cnt, (nw + 1) * 4));
    struct FreeCnt { uint32_t* p; ~FreeCnt() { (void)hipFree(p); } } fc{cnt};
    for (int m = 0; m < 3; m++) {
        const Csr& R = D->row[m];
        Csr& Cc = D->col[m];
        Cc.nnz = R.nnz;
        ZK_TRY(dalloc(D, &Cc.ptr, nw + 1));
        ZK_TRY(dalloc(D, &Cc.idx, R.nnz));
        ZK_TRY(dalloc(D, &Cc.val, R.nnz));
        ZK_HIP(hipMemsetAsync(cnt, 0, (nw + 1) * 4, st));
        if (R.nnz) ZK_LAUNCH(s, st, "r1cs_csc", k_csc_count, dim3(gridn(R.nnz)), dim3(256), 0, (const uint32_t*)R.idx, R.nnz, cnt);
        ZK_LAUNCH(s, st, "r1cs_csc", k_csc_scan, dim3(1), dim3(1024), 0, (const uint32_t*)cnt, nw, Cc.ptr);
        ZK_HIP(hipMemcpyAsync(cnt, Cc.ptr, nw * 4, hipMemcpyDeviceToDevice, st));  // the cursors start at the column starts
        if (nc && R.nnz) ZK_LAUNCH(s, st, "r1cs_csc", k_csc_fill, dim3(gridn(nc)), dim3(256), 0, (const uint32_t*)R.ptr, (const uint32_t*)R.idx, (const Fr*)R.val, nc, cnt, Cc.idx, Cc.val);
    }
    ZK_TRY(slot_sync(s, st));
    D->have_csc = true;
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_bn254_r1cs_load(const zk_r1cs* r, uint64_t* handle) {
    if (!r || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    if (r->n_public > r->n_wires || r->n_public < 1) return set_err(ZK_ERR_ARG, "bad R1CS geometry (n_public counts the ONE wire)");
    if (r->n_wires >= ((size_t)1 << 31) || r->n_constraints >= ((size_t)1 << 28)) return set_err(ZK_ERR_ARG, "R1CS too large");
    const uint32_t* ptrs[3] = {r->l_ptr, r->r_ptr, r->o_ptr};
    const uint32_t* idxs[3] = {r->l_idx, r->r_idx, r->o_idx};
    const zk_fr* vals[3] = {r->l_val, r->r_val, r->o_val};
    for (int m = 0; m < 3; m++) {
        if (!ptrs[m]) return set_err(ZK_ERR_ARG, "null row pointers");
        if (ptrs[m][0] != 0) return set_err(ZK_ERR_ARG, "row pointers must start at 0");
        for (size_t i = 0; i < r->n_constraints; i++)
            if (ptrs[m][i + 1] < ptrs[m][i]) return set_err(ZK_ERR_ARG, "row pointers must not decrease");
        const size_t nnz = ptrs[m][r->n_constraints];
        if (nnz && (!idxs[m] || !vals[m])) return set_err(ZK_ERR_ARG, "null matrix entries");
        for (size_t k = 0; k < nnz; k++)
            if (idxs[m][k] >= r->n_wires) return set_err(ZK_ERR_ARG, "matrix %d names wire %u of %zu", m, idxs[m][k], r->n_wires);
    }
    const auto t_a = std::chrono::steady_clock::now();
    ZK_TRY(ensure_init());
    std::unique_ptr<R1csDev, void (*)(R1csDev*)> D(new R1csDev(), r1cs_destroy);
    D->n_constraints = r->n_constraints; D->n_wires = r->n_wires; D->n_public = r->n_public;
    SlotGuard g;  // a stream of its own for the uploads (the null stream would wait for every other stream's work)
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t up = g.s->stream;
    for (int m = 0; m < 3; m++) {
        const size_t nc = r->n_constraints, nnz = ptrs[m][nc];
        Csr& R = D->row[m];
        R.nnz = nnz;
        ZK_TRY(dalloc(D.get(), &R.ptr, nc + 1));
        ZK_TRY(dalloc(D.get(), &R.idx, nnz));
        ZK_TRY(dalloc(D.get(), &R.val, nnz));
        ZK_TRY(h2d_big(R.ptr, ptrs[m], (nc + 1) * 4, up));  // (big arrays go through the pinned ring of ctx.hip: 0.2 GB at 2^20 constraints)
        if (nnz) {
            ZK_TRY(h2d_big(R.idx, idxs[m], nnz * 4, up));
            ZK_TRY(h2d_big(R.val, vals[m], nnz * 32, up));
        }
    }
    ZK_TRY(slot_sync(g.s, up));
    prof_host("export.r1cs_load_upload", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
    std::lock_guard<std::mutex> lk(g_r1cs_mu);
    *handle = hmake(g_next_r1cs++);
    g_r1cs[*handle] = std::shared_ptr<R1csDev>(D.release(), r1cs_destroy);
    return ZK_OK;
}

int zk_bn254_r1cs_free(uint64_t handle) {
    ZK_ON_ENTRY_OF(handle);
    std::shared_ptr<R1csDev> D;  // destroyed here, or by the last call still using it
    {
        std::lock_guard<std::mutex> lk(g_r1cs_mu);
        auto it = g_r1cs.find(handle);
        if (it == g_r1cs.end()) return set_err(ZK_ERR_HANDLE, "unknown R1CS handle %llu", (unsigned long long)handle);
        D = it->second;
        g_r1cs.erase(it);
    }
    return ZK_OK;
}

// a = L w, b = R w, c = O w (n_constraints each), everything in HBM
int zk_bn254_r1cs_eval_abc_dev(uint64_t handle, const void* d_w, size_t n_wires, void* d_a, void* d_b, void* d_c, void* stream) {
    ZK_ON_ENTRY_OF(handle);
    std::shared_ptr<R1csDev> Dref;
    ZK_TRY(lookup_r1cs(handle, &Dref));
    R1csDev* D = Dref.get();
    if (n_wires != D->n_wires) return set_err(ZK_ERR_LEN, "len(w) = %zu != %zu wires of the constraint system", n_wires, D->n_wires);
    if (!d_w || !d_a || !d_b || !d_c) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    Csr3 M;
    for (int m = 0; m < 3; m++) { M.ptr[m] = D->row[m].ptr; M.idx[m] = D->row[m].idx; M.val[m] = D->row[m].val; }
    if (D->n_constraints)
        ZK_LAUNCH(g.s, st, "r1cs_spmv", k_spmv3, dim3(gridn(D->n_constraints), 3), dim3(256), 0, M, (const Fr*)d_w, D->n_constraints, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c,
                  (uint32_t*)nullptr);  // by constraint: rows are a handful of terms
    if (!stream || profiling_on()) ZK_TRY(slot_sync(g.s, st));
    return ZK_OK;
}

// groth16.Setup with the toxic waste as input: tau, alpha, beta, gamma, delta (Montgomery fr.Elements, all non-zero).  The proving key is
// built in HBM and loaded as a resident key (*pk_handle, usable with zk_bn254_groth16_prove); the verifying key comes back to the host:
// vk_g1 = [alpha]G1 followed by the n_public points K_i / gamma (gnark's vk.G1.K), vk_g2 = [beta]G2, [gamma]G2, [delta]G2.
int zk_bn254_groth16_setup(uint64_t r1cs_handle, const zk_fr toxic[5], int flags, uint64_t* pk_handle, zk_g1_affine* vk_g1, zk_g2_affine vk_g2[3]) {
    ZK_ON_ENTRY_OF(r1cs_handle);
    if (!toxic || !pk_handle) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<R1csDev> Dref;
    ZK_TRY(lookup_r1cs(r1cs_handle, &Dref));
    R1csDev* D = Dref.get();
    HFr tx[5];
    memcpy(tx, toxic, sizeof tx);
    for (int i = 0; i < 5; i++)
        if (tx[i].is_zero()) return set_err(ZK_ERR_ARG, "toxic-waste element %d is zero", i);
    const HFr tau = tx[0], alpha = tx[1], beta = tx[2], gamma = tx[3], delta = tx[4];
    unsigned logN = 0;
    while (((size_t)1 << logN) < D->n_constraints) logN++;
    const size_t N = (size_t)1 << logN, nw = D->n_wires, npub = D->n_public, nk = nw - npub;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    Slot* s = g.s;
    hipStream_t st = s->stream;
    Domain* dom;
    ZK_TRY(get_domain(s, st, logN, DOM_TW, &dom));
    HFr tauN = tau;
    for (unsigned i = 0; i < logN; i++) tauN = tauN.sqr();
    const HFr zt = tauN - HFr::one();
    if (zt.is_zero()) return set_err(ZK_ERR_ARG, "tau is a root of unity of the domain");
    ZK_TRY(s->reserve((N + 4 * nw + N) * sizeof(Fr) + 131072));
    Fr* lag = (Fr*)s->alloc(N * sizeof(Fr));
    Fr* abc = (Fr*)s->alloc(3 * nw * sizeof(Fr));
    Fr* ksc = (Fr*)s->alloc(nw * sizeof(Fr));
    Fr* zsc = (Fr*)s->alloc(N * sizeof(Fr));
    if (!lag || !abc || !ksc || !zsc) return set_err(ZK_ERR_HIP, "setup workspace");
    ZK_LAUNCH(s, st, "setup_lagrange", k_lagrange_at, dim3(gridn(N)), dim3(256), 0, (const Fr*)dom->tw, (uint32_t)N, todev(tau), todev(zt * dom->card_inv), lag);
    struct Lap {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        void lap(const char* name) {
            const auto t1 = std::chrono::steady_clock::now();
            prof_host(name, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    } lap;
    ZK_TRY(ensure_csc(D, s, st));
    lap.lap("export.setup_csc");
    {   // A_i, B_i, C_i = sum_j M[j][i] lag_j: the transposed products, one lane per wire over the CSC form
        Csr3 M;
        for (int m = 0; m < 3; m++) { M.ptr[m] = D->col[m].ptr; M.idx[m] = D->col[m].idx; M.val[m] = D->col[m].val; }
        uint32_t* long_rows = (uint32_t*)s->alloc((1 + 2 * SPMV_LONG_MAX) * 4);
        if (!long_rows) return set_err(ZK_ERR_HIP, "setup workspace");
        ZK_HIP(hipMemsetAsync(long_rows, 0, 4, st));
        ZK_LAUNCH(s, st, "setup_spmv_t", k_spmv3, dim3(gridn(nw), 3), dim3(256), 0, M, (const Fr*)lag, nw, abc, abc + nw, abc + 2 * nw, long_rows);
        ZK_LAUNCH(s, st, "setup_spmv_t_long", k_spmv3_long, dim3(512), dim3(256), 0, M, (const Fr*)lag, abc, abc + nw, abc + 2 * nw, (const uint32_t*)long_rows);
    }
    ZK_LAUNCH(s, st, "setup_k_scalars", k_k_scalars, dim3(gridn(nw)), dim3(256), 0, (const Fr*)abc, (const Fr*)(abc + nw), (const Fr*)(abc + 2 * nw), nw, npub, todev(alpha),
              todev(beta), todev(gamma.inv()), todev(delta.inv()), ksc);
    {
        PowTab pt;
        HFr p = tau;
        for (int b = 0; b < 28; b++) { pt.pw[b] = todev(p); p = p.sqr(); }
        ZK_LAUNCH(s, st, "setup_z_scalars", k_z_scalars, dim3(gridn(N)), dim3(256), 0, pt, (uint32_t)N, logN, todev(zt * delta.inv()), zsc);
    }
    // the five base arrays of the key (owned by the key once it is loaded) + the verifying key's K points
    void *d_a = nullptr, *d_b = nullptr, *d_k = nullptr, *d_z = nullptr, *d_b2 = nullptr, *d_ic = nullptr;
    struct Guard { void** p[6]; bool keep = false; ~Guard() { if (!keep) for (auto q : p) if (*q) (void)hipFree(*q); } } guard{{&d_a, &d_b, &d_k, &d_z, &d_b2, &d_ic}};
    ZK_HIP(hipMalloc(&d_a, (nw ? nw : 1) * 64));
    ZK_HIP(hipMalloc(&d_b, (nw ? nw : 1) * 64));
    ZK_HIP(hipMalloc(&d_k, (nk ? nk : 1) * 64));
    ZK_HIP(hipMalloc(&d_z, N * 64));
    ZK_HIP(hipMalloc(&d_b2, (nw ? nw : 1) * 128));
    ZK_HIP(hipMalloc(&d_ic, npub * 64));
    // fixed-base scalar multiplications by the generators (util.hip: 8-bit windows, shared inversion)
    ZK_TRY(fixed_base_mul_scalars(s, st, 0, abc, nw, d_a));
    ZK_TRY(fixed_base_mul_scalars(s, st, 0, abc + nw, nw, d_b));
    if (nk) ZK_TRY(fixed_base_mul_scalars(s, st, 0, ksc + npub, nk, d_k));
    ZK_TRY(fixed_base_mul_scalars(s, st, 0, ksc, npub, d_ic));
    ZK_TRY(fixed_base_mul_scalars(s, st, 0, zsc, N, d_z));
    ZK_TRY(fixed_base_mul_scalars(s, st, 1, abc + nw, nw, d_b2));
    // the handful of single points: host scalar multiplications
    auto mul1 = [](const HFr& k, Affine<HFp>* o) {
        Affine<Fp> gd = generator_g1();
        Affine<HFp> gh;
        memcpy(&gh, &gd, sizeof gh);
        uint32_t kk[8];
        HFr c = k.from_mont();
        memcpy(kk, c.l, 32);
        *o = scalar_mul(gh, kk).to_affine();
    };
    auto mul2 = [](const HFr& k, Affine<HFp2>* o) {
        Affine<Fp2> gd = generator_g2();
        Affine<HFp2> gh;
        memcpy(&gh, &gd, sizeof gh);
        uint32_t kk[8];
        HFr c = k.from_mont();
        memcpy(kk, c.l, 32);
        *o = scalar_mul(gh, kk).to_affine();
    };
    lap.lap("export.setup_enqueue");
    Affine<HFp> a1, b1, d1;
    Affine<HFp2> b2, g2, d2;
    mul1(alpha, &a1); mul1(beta, &b1); mul1(delta, &d1);
    mul2(beta, &b2); mul2(gamma, &g2); mul2(delta, &d2);
    lap.lap("export.setup_host_points");
    ZK_TRY(slot_sync(s, st));
    lap.lap("export.setup_device_wait");
    zk_groth16_pk pk;
    memset(&pk, 0, sizeof pk);
    pk.log_domain = logN;
    pk.n_wires = nw;
    pk.n_public = npub;
    pk.g1_alpha = (const zk_g1_affine*)&a1; pk.g1_beta = (const zk_g1_affine*)&b1; pk.g1_delta = (const zk_g1_affine*)&d1;
    pk.g1_a = (const zk_g1_affine*)d_a; pk.g1_b = (const zk_g1_affine*)d_b; pk.g1_k = (const zk_g1_affine*)d_k; pk.g1_z = (const zk_g1_affine*)d_z;
    pk.g2_beta = (const zk_g2_affine*)&b2; pk.g2_delta = (const zk_g2_affine*)&d2;
    pk.g2_b = (const zk_g2_affine*)d_b2;
    pk.bases_on_device = 1;
    pk.flags = flags & 1;
    ZK_TRY(zk_bn254_groth16_pk_load(&pk, pk_handle));
    lap.lap("export.setup_pk_load");
    ZK_TRY(groth16_pk_adopt(*pk_handle));
    d_a = d_b = d_k = d_z = d_b2 = nullptr;  // the key owns them now
    if (vk_g1) {
        memcpy(vk_g1, &a1, 64);
        ZK_HIP(hipMemcpy(vk_g1 + 1, d_ic, npub * 64, hipMemcpyDeviceToHost));
    }
    if (vk_g2) {
        memcpy(&vk_g2[0], &b2, 128);
        memcpy(&vk_g2[1], &g2, 128);
        memcpy(&vk_g2[2], &d2, 128);
    }
    return ZK_OK;
}

// gnark's Groth16 prover from the WITNESS: a, b, c = L w, R w, O w on the device (the solver's output for a system without hints), then
// zk_bn254_groth16_prove on resident data.  w: all wire values [ONE, public..., secret..., internal...] (Montgomery); on_device as there.
int zk_bn254_groth16_prove_r1cs(uint64_t r1cs_handle, uint64_t pk_handle, const void* w, size_t n_wires, const zk_fr* r, const zk_fr* s_, int on_device,
                                uint8_t proof_out[128]) {
    ZK_ON_ENTRY_OF(r1cs_handle);
    std::shared_ptr<R1csDev> Dref;
    ZK_TRY(lookup_r1cs(r1cs_handle, &Dref));
    R1csDev* D = Dref.get();
    if (n_wires != D->n_wires) return set_err(ZK_ERR_LEN, "len(w) = %zu != %zu wires of the constraint system", n_wires, D->n_wires);
    if (!w || !r || !s_ || !proof_out) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_TRY(ensure_init());
    const size_t nc = D->n_constraints;
    void *d_w = nullptr, *d_abc = nullptr;
    struct Free { void** p[2]; ~Free() { for (auto q : p) if (*q) (void)hipFree(*q); } } guard{{&d_w, &d_abc}};
    ZK_HIP(hipMalloc(&d_abc, (nc ? 3 * nc : 1) * 32));
    const void* dw = w;
    if (!on_device) {
        ZK_HIP(hipMalloc(&d_w, (n_wires ? n_wires : 1) * 32));
        ZK_HIP(hipMemcpy(d_w, w, n_wires * 32, hipMemcpyHostToDevice));
        dw = d_w;
    }
    Fr* abc = (Fr*)d_abc;
    ZK_TRY(zk_bn254_r1cs_eval_abc_dev(r1cs_handle, dw, n_wires, abc, abc + nc, abc + 2 * nc, nullptr));
    return zk_bn254_groth16_prove(pk_handle, abc, abc + nc, abc + 2 * nc, nc, dw, n_wires, r, s_, 1, proof_out);
}

}  // extern "C"
