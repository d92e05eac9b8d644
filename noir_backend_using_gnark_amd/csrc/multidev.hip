// One process, several GPUs: the range-sharded phases of SURVEY §8e behind the SAME entry points a single-GPU caller uses.
//
// The reference is one process (gnark_backend_ffi/main.go:24-37 -> backend/plonk/plonk.go:53-73): nargo -> Rust -> cgo cannot become "one process per
// GPU".  So the sharding that noir_backend_using_gnark_amd/parallel.py drives from torch.distributed ranks is also available INSIDE the library: after
// zk_init_devices(list) -- one ENTRY per listed device, ctx.hpp -- the calls below split their work over the entries on host threads of their own (one per
// entry, each bound to its entry for the duration of the call) and combine on the host exactly what the ranks combine after their all-gather:
//   zk_bn254_g1_msm / g2_msm (host slices)      points / scalars cut by range, one partial sum per entry (each entry uploads ITS slice over ITS PCIe link)
//   zk_bn254_bases_register* / zk_bn254_msm_bases*   a COMPOSITE handle: every entry keeps a range of the bases (+ its window-table slice) resident; a commit is
//                                               one partial per entry; scalars that live in another GPU's HBM travel by hipMemcpyPeerAsync (xGMI)
//   zk_bn254_groth16_pk_load / _prove           a composite key of per-entry range slices (A, B1, G2.B, K by wires; Z by blocks of h); prove = computeH
//                                               block-sharded over the entries (zk_bn254_groth16_h_shard_dev phases, the nine all-to-all transposes as peer
//                                               copies between the entries' streams), the five MSMs of every slice (zk_bn254_groth16_msm5_pk), the 768-byte
//                                               records combined on the host (zk_bn254_groth16_finalize)
//   zk_bn254_ntt (host slice)                   blocks of the array uploaded per entry, zk_bn254_ntt_shard_dev steps, two all-to-all transposes as peer copies
// Which entries: the device_mask of zk_msm_cfg / zk_groth16_pk (bit i = entry i), 0 = the process default -- every entry once zk_init_devices has been
// called, else only the calling thread's entry, so a process that never names devices behaves exactly as before.  An implicit (default) mask only spreads
// work that is large enough to pay for it; an explicit one is always honoured.  The exchange is point-to-point peer copies, not a collective library: every
// transfer has exactly one source and one destination stream (xGMI is point-to-point; RCCL's all_to_all is the same G*(G-1) copies).
// Listing ONE device several times gives virtual devices (their "peer copies" are device-local): that is how tests/test_gpu_multidev.py runs every path on a
// one-GPU box -- the bytes must equal the single-entry bytes for 2, 4 and 8 entries.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <thread>

#include "ctx.hpp"
#include "curve.hpp"
#include "ff.hpp"
#include "host_ff.hpp"
#include "msm.hpp"
#include "multidev.hpp"

namespace zkmi {

static std::atomic<uint32_t> g_default_mask{0};

void md_set_default_mask(uint32_t mask) { g_default_mask.store(mask); }
uint32_t md_default_mask() { return g_default_mask.load(); }

int md_entries_for(uint32_t requested, size_t units, size_t min_units_per_entry, std::vector<int>* out) {
    out->clear();
    const int n = n_entries();
    uint32_t mask = requested;
    bool implicit = false;
    if (!mask) {
        mask = g_default_mask.load();
        implicit = true;
    }
    if (!mask || n <= 1) { out->push_back(current_entry()); return ZK_OK; }
    for (int e = 0; e < 32; e++)
        if (mask & (1u << e)) {
            if (e >= n) return set_err(ZK_ERR_ARG, "device_mask names entry %d, %d entries exist (zk_init_devices)", e, n);
            out->push_back(e);
        }
    if (implicit && out->size() > 1 && units < min_units_per_entry * out->size()) {  // too small to pay for the spread
        out->clear();
        out->push_back(current_entry());
    }
    return ZK_OK;
}

struct Barrier {  // reusable; a failing member releases the others through `broken`
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0, phase = 0;
    bool broken = false;
    explicit Barrier(int n_) : n(n_) {}
    bool wait() {  // false: the team is broken, give up
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const int ph = phase;
        if (++waiting == n) { waiting = 0; phase++; cv.notify_all(); return !broken; }
        cv.wait(lk, [&] { return phase != ph || broken; });
        return !broken;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};
// ---- a team of host threads, one per entry; the first failure is the call's failure.  `bar` (optional): the barrier the members meet at -- EVERY member that ends
// with an error breaks it, wherever the error came from (its entry could not be entered, a slot timed out, a workspace did not fit: exits that never reach the
// member's own bar.fail()), so that no other member waits for it for ever.
static int team_run(const std::vector<int>& entries, const std::function<int(int)>& fn, Barrier* bar = nullptr) {
    const int G = (int)entries.size();
    std::vector<int> rcs(G, ZK_OK);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (int k = 0; k < G; k++)
        th.emplace_back([&, k] {
            CtxScope sc(entries[k]);
            rcs[k] = sc.rc != ZK_OK ? sc.rc : fn(k);
            if (rcs[k] != ZK_OK) {
                errs[k] = g_err;
                if (bar) bar->fail();
            }
        });
    for (auto& t : th) t.join();
    for (int k = 0; k < G; k++)
        if (rcs[k] != ZK_OK) return set_err(rcs[k], "entry %d: %s", entries[k], errs[k].c_str());
    return ZK_OK;
}

static int device_of_entry(int e) {
    CtxScope sc(e);
    return sc.rc == ZK_OK ? ctx().device : -1;
}
// where a caller's pointer lives: -1 = host memory, else the HIP device ordinal
static int device_of_pointer(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) return a.device;
    return -1;
}
static int copy_dev(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st) {
    if (!bytes) return ZK_OK;
    if (src_dev < 0) ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    else if (src_dev == dst_dev) ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    else ZK_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st));
    return ZK_OK;
}

template <class HF>
static void add_affine(XYZZ<HF>* acc, const void* aff) {
    Affine<HF> a;
    memcpy(&a, aff, sizeof a);
    if (!a.is_inf()) acc->madd(a);
}

// ------------------------------------------------------------------------------------------------ MSM over host slices
int md_msm_host(int g2, const void* points, const zk_fr* scalars, size_t n, const zk_msm_cfg* cfg, void* out, const std::vector<int>& entries) {
    const int G = (int)entries.size();
    const size_t psz = g2 ? 128 : 64;
    zk_msm_cfg c1 = cfg ? *cfg : zk_msm_cfg{0, 0, 0, 0};
    std::vector<uint8_t> part((size_t)G * 128, 0);
    ZK_TRY(team_run(entries, [&](int k) -> int {
        const size_t lo = n * k / G, hi = n * (k + 1) / G;
        zk_msm_cfg c = c1;
        c.device_mask = (int)(1u << entries[k]);  // this slice, on this entry
        return g2 ? zk_bn254_g2_msm((const zk_g2_affine*)((const char*)points + lo * psz), hi - lo, scalars + lo, hi - lo, &c, (zk_g2_affine*)&part[(size_t)k * 128])
                  : zk_bn254_g1_msm((const zk_g1_affine*)((const char*)points + lo * psz), hi - lo, scalars + lo, hi - lo, &c, (zk_g1_affine*)&part[(size_t)k * 128]);
    }));
    if (g2) {
        XYZZ<HFp2> t = XYZZ<HFp2>::inf();
        for (int k = 0; k < G; k++) add_affine(&t, &part[(size_t)k * 128]);
        Affine<HFp2> a = t.to_affine();
        memcpy(out, &a, sizeof a);
    } else {
        XYZZ<HFp> t = XYZZ<HFp>::inf();
        for (int k = 0; k < G; k++) add_affine(&t, &part[(size_t)k * 128]);
        Affine<HFp> a = t.to_affine();
        memcpy(out, &a, sizeof a);
    }
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------ composite resident bases
struct MdBases {
    int is_g2 = 0;
    size_t n = 0;
    std::vector<int> entries;
    std::vector<size_t> lo;       // G + 1 range bounds
    std::vector<uint64_t> sub;    // per-entry handles (zk_bn254_bases_register_cfg on that entry)
};
struct MdGroth16 {
    unsigned log_domain = 0, log_g = 0;
    size_t n_wires = 0, n_public = 0;
    std::vector<int> entries;
    std::vector<size_t> wlo;      // G + 1 wire-range bounds
    std::vector<uint64_t> sub;    // per-entry slice keys
    std::vector<void*> owned;     // device copies of slices made for entries on other GPUs than the caller's arrays
    std::shared_ptr<std::mutex> mu = std::make_shared<std::mutex>();  // one proof at a time per composite key (the entries' five-slot sessions)
};
static std::mutex g_md_mu;
static std::map<uint64_t, std::shared_ptr<MdBases>> g_md_bases;
static std::map<uint64_t, std::shared_ptr<MdGroth16>> g_md_keys;
static uint64_t g_md_next = 1;
static uint64_t md_handle(int first_entry) { return ((uint64_t)0xff << 56) | ((uint64_t)first_entry << 48) | g_md_next++; }

int md_bases_register(const void* points, size_t n, int is_g2, int on_device, int table_bits, const std::vector<int>& entries, uint64_t* handle) {
    const int G = (int)entries.size();
    const size_t psz = is_g2 ? 128 : 64;
    auto B = std::make_shared<MdBases>();
    B->is_g2 = is_g2 ? 1 : 0;
    B->n = n;
    B->entries = entries;
    B->sub.assign(G, 0);
    for (int k = 0; k <= G; k++) B->lo.push_back(n * k / G);
    const int src_dev = on_device ? device_of_pointer(points) : -1;
    int rc = team_run(entries, [&](int k) -> int {
        const size_t lo = B->lo[k], cnt = B->lo[k + 1] - lo;
        const void* src = (const char*)points + lo * psz;
        if (!on_device || src_dev == ctx().device) return bases_register_on_this_entry(src, cnt, is_g2, on_device, table_bits, &B->sub[k]);
        // the caller's array lives in another GPU's HBM: this entry's range comes over by a peer copy
        void* tmp = nullptr;
        ZK_HIP(hipMalloc(&tmp, cnt * psz + 16));
        int r = copy_dev(tmp, ctx().device, src, src_dev, cnt * psz, nullptr);
        if (r == ZK_OK && hipStreamSynchronize(nullptr) != hipSuccess) r = set_err(ZK_ERR_HIP, "peer copy of the bases failed");
        if (r == ZK_OK) r = bases_register_on_this_entry(tmp, cnt, is_g2, 1, table_bits, &B->sub[k]);
        (void)hipFree(tmp);
        return r;
    });
    if (rc != ZK_OK) {
        for (uint64_t h : B->sub)
            if (h) (void)zk_bn254_bases_free(h);
        return rc;
    }
    std::lock_guard<std::mutex> lk(g_md_mu);
    *handle = md_handle(entries[0]);
    g_md_bases[*handle] = B;
    return ZK_OK;
}
static int md_find_bases(uint64_t h, std::shared_ptr<MdBases>* out) {
    std::lock_guard<std::mutex> lk(g_md_mu);
    auto it = g_md_bases.find(h);
    if (it == g_md_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)h);
    *out = it->second;
    return ZK_OK;
}
int md_bases_build_table(uint64_t h, int table_bits) {
    std::shared_ptr<MdBases> B;
    ZK_TRY(md_find_bases(h, &B));
    return team_run(B->entries, [&](int k) -> int { return zk_bn254_bases_build_table(B->sub[k], table_bits); });
}
int md_bases_info(uint64_t h, size_t* n, int* is_g2) {
    std::shared_ptr<MdBases> B;
    ZK_TRY(md_find_bases(h, &B));
    if (n) *n = B->n;
    if (is_g2) *is_g2 = B->is_g2;
    return ZK_OK;
}
int md_bases_free(uint64_t h) {
    std::shared_ptr<MdBases> B;
    {
        std::lock_guard<std::mutex> lk(g_md_mu);
        auto it = g_md_bases.find(h);
        if (it == g_md_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)h);
        B = it->second;
        g_md_bases.erase(it);
    }
    for (uint64_t s : B->sub) (void)zk_bn254_bases_free(s);
    return ZK_OK;
}
int md_msm_bases(uint64_t h, size_t offset, const void* scalars, size_t n, const zk_msm_cfg* cfg, void* out, int on_device) {
    std::shared_ptr<MdBases> B;
    ZK_TRY(md_find_bases(h, &B));
    if (offset + n > B->n) return set_err(ZK_ERR_LEN, "len(points) != len(scalars): offset %zu + n %zu exceeds the %zu registered bases", offset, n, B->n);
    if (!out || (n && !scalars)) return set_err(ZK_ERR_ARG, "null pointer");
    const int G = (int)B->entries.size();
    const int src_dev = on_device ? device_of_pointer(scalars) : -1;
    zk_msm_cfg c1 = cfg ? *cfg : zk_msm_cfg{0, 0, 0, 0};
    c1.device_mask = 0;
    std::vector<uint8_t> part((size_t)G * 128, 0);
    std::vector<char> used(G, 0);
    ZK_TRY(team_run(B->entries, [&](int k) -> int {
        // scalars [0, n) pair with bases [offset, offset + n); this entry holds bases [lo, hi)
        const size_t lo = B->lo[k], hi = B->lo[k + 1];
        const size_t a = offset > lo ? offset : lo, b = offset + n < hi ? offset + n : hi;
        if (a >= b) return ZK_OK;
        used[k] = 1;
        const char* src = (const char*)scalars + (a - offset) * 32;
        if (!on_device) return zk_bn254_msm_bases(B->sub[k], a - lo, (const zk_fr*)src, b - a, &c1, &part[(size_t)k * 128]);
        if (src_dev == ctx().device) return zk_bn254_msm_bases_dev(B->sub[k], a - lo, src, b - a, &c1, &part[(size_t)k * 128]);
        void* tmp = nullptr;  // the polynomial lives in another GPU's HBM: this entry's range of it comes over xGMI
        ZK_HIP(hipMalloc(&tmp, (b - a) * 32 + 16));
        int r = copy_dev(tmp, ctx().device, src, src_dev, (b - a) * 32, nullptr);
        if (r == ZK_OK && hipStreamSynchronize(nullptr) != hipSuccess) r = set_err(ZK_ERR_HIP, "peer copy of the scalars failed");
        if (r == ZK_OK) r = zk_bn254_msm_bases_dev(B->sub[k], a - lo, tmp, b - a, &c1, &part[(size_t)k * 128]);
        (void)hipFree(tmp);
        return r;
    }));
    if (B->is_g2) {
        XYZZ<HFp2> t = XYZZ<HFp2>::inf();
        for (int k = 0; k < G; k++)
            if (used[k]) add_affine(&t, &part[(size_t)k * 128]);
        Affine<HFp2> a = t.to_affine();
        memcpy(out, &a, sizeof a);
    } else {
        XYZZ<HFp> t = XYZZ<HFp>::inf();
        for (int k = 0; k < G; k++)
            if (used[k]) add_affine(&t, &part[(size_t)k * 128]);
        Affine<HFp> a = t.to_affine();
        memcpy(out, &a, sizeof a);
    }
    return ZK_OK;
}

// ------------------------------------------------------------------------------------------------ all-to-all between the entries' blocks
// x[k] (entry k's block, G chunks of `chunk` bytes) -> y[s] chunk r = x[r] chunk s.  Every member enqueues ITS G sends on its own stream, waits for
// them, and the team meets at the barrier: after it every y is complete and every x may be overwritten.
struct Exchange {
    int G;
    std::vector<int> dev;
    Barrier bar;
    explicit Exchange(const std::vector<int>& entries) : G((int)entries.size()), bar((int)entries.size()) {
        for (int e : entries) dev.push_back(device_of_entry(e));
    }
    int run(int k, void* const* x, void* const* y, size_t chunk, hipStream_t st) {
        int rc = ZK_OK;
        for (int s = 0; s < G && rc == ZK_OK; s++)
            rc = copy_dev((char*)y[s] + (size_t)k * chunk, dev[s], (const char*)x[k] + (size_t)s * chunk, dev[k], chunk, st);
        if (rc == ZK_OK && hipStreamSynchronize(st) != hipSuccess) rc = set_err(ZK_ERR_HIP, "all-to-all: stream synchronisation failed");
        if (rc != ZK_OK) { bar.fail(); return rc; }
        if (!bar.wait()) return set_err(ZK_ERR_HIP, "all-to-all: another entry failed");
        return ZK_OK;
    }
};

// ------------------------------------------------------------------------------------------------ composite Groth16 key
static int ilog2_exact(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) l++;
    return ((size_t)1 << l) == v ? l : -1;
}
int md_groth16_pk_load(const zk_groth16_pk* pk, const std::vector<int>& entries, uint64_t* handle) {
    const int G = (int)entries.size(), lg = ilog2_exact((size_t)G);
    if (lg < 1 || lg > 3) return set_err(ZK_ERR_ARG, "a key can be spread over 2, 4 or 8 device entries (block-sharded computeH), not %d", G);
    if (pk->flags & 6) return set_err(ZK_ERR_ARG, "flags bits 1 and 2 describe slices of a key that is ALREADY sharded; they do not combine with a device_mask");
    const size_t N = (size_t)1 << pk->log_domain, nw = pk->n_wires, npub = pk->n_public;
    if (pk->log_domain < 2u * lg + 2) return set_err(ZK_ERR_ARG, "domain 2^%u is too small for %d entries", pk->log_domain, G);
    auto K = std::make_shared<MdGroth16>();
    K->log_domain = pk->log_domain;
    K->log_g = (unsigned)lg;
    K->n_wires = nw;
    K->n_public = npub;
    K->entries = entries;
    K->sub.assign(G, 0);
    for (int k = 0; k <= G; k++) K->wlo.push_back(nw * k / G);
    const bool compact = pk->infinity_a != nullptr;
    // gnark's compact A / B / G2.B: a wire range starts at (its first wire - the points at infinity before it) in the compact arrays
    std::vector<size_t> inf_a_before(G + 1, 0), inf_b_before(G + 1, 0);
    if (compact)
        for (int k = 0; k < G; k++) {
            size_t ca = 0, cb = 0;
            for (size_t i = K->wlo[k]; i < K->wlo[k + 1]; i++) { ca += pk->infinity_a[i] != 0; cb += pk->infinity_b[i] != 0; }
            inf_a_before[k + 1] = inf_a_before[k] + ca;
            inf_b_before[k + 1] = inf_b_before[k] + cb;
        }
    if (compact && (inf_a_before[G] != pk->nb_infinity_a || inf_b_before[G] != pk->nb_infinity_b)) return set_err(ZK_ERR_ARG, "NbInfinityA / NbInfinityB do not match the bitmaps");
    const int src_dev = pk->bases_on_device ? device_of_pointer(pk->g1_z) : -1;
    std::mutex own_mu;
    const size_t M = N >> lg;
    int rc = team_run(entries, [&](int k) -> int {
        const size_t lo = K->wlo[k], hi = K->wlo[k + 1], cnt = hi - lo;
        zk_groth16_pk s = *pk;
        s.device_mask = (int)(1u << entries[k]);
        s.log_domain = pk->log_domain - (unsigned)lg;       // Z: this entry's block of M = N / G points; all of them except in the last block
        s.flags = (pk->flags & 1) | (k + 1 < G ? 2 : 0);
        s.n_wires = cnt;
        s.n_public = npub > lo ? (npub - lo < cnt ? npub - lo : cnt) : 0;
        const size_t a_off = compact ? lo - inf_a_before[k] : lo, b_off = compact ? lo - inf_b_before[k] : lo;
        const size_t k_first = (lo > npub ? lo : npub) - npub;   // K[j] belongs to wire j + n_public
        s.g1_a = pk->g1_a + a_off;
        s.g1_b = pk->g1_b + b_off;
        s.g2_b = pk->g2_b + b_off;
        s.g1_k = pk->g1_k + k_first;
        s.g1_z = pk->g1_z + (size_t)k * M;
        if (compact) {
            s.infinity_a = pk->infinity_a + lo;
            s.infinity_b = pk->infinity_b + lo;
            s.nb_infinity_a = inf_a_before[k + 1] - inf_a_before[k];
            s.nb_infinity_b = inf_b_before[k + 1] - inf_b_before[k];
        }
        if (pk->bases_on_device && src_dev != ctx().device) {
            // the caller's arrays are in another GPU's HBM: this entry's slices come over by peer copies and belong to the composite key
            const size_t na = compact ? cnt - s.nb_infinity_a : cnt, nb = compact ? cnt - s.nb_infinity_b : cnt, nk = cnt - s.n_public;
            struct { const void** p; size_t bytes; } arr[5] = {{(const void**)&s.g1_a, na * 64}, {(const void**)&s.g1_b, nb * 64}, {(const void**)&s.g2_b, nb * 128},
                                                               {(const void**)&s.g1_k, nk * 64}, {(const void**)&s.g1_z, M * 64}};
            for (auto& a : arr) {
                void* d = nullptr;
                ZK_HIP(hipMalloc(&d, a.bytes + 16));
                { std::lock_guard<std::mutex> lk(own_mu); K->owned.push_back(d); }
                ZK_TRY(copy_dev(d, ctx().device, *a.p, src_dev, a.bytes, nullptr));
                *a.p = d;
            }
            ZK_HIP(hipStreamSynchronize(nullptr));
        }
        return zk_bn254_groth16_pk_load(&s, &K->sub[k]);
    });
    if (rc != ZK_OK) {
        for (uint64_t h : K->sub)
            if (h) (void)zk_bn254_groth16_pk_free(h);
        for (void* d : K->owned) (void)hipFree(d);
        return rc;
    }
    std::lock_guard<std::mutex> lk(g_md_mu);
    *handle = md_handle(entries[0]);
    g_md_keys[*handle] = K;
    return ZK_OK;
}
static int md_find_key(uint64_t h, std::shared_ptr<MdGroth16>* out) {
    std::lock_guard<std::mutex> lk(g_md_mu);
    auto it = g_md_keys.find(h);
    if (it == g_md_keys.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)h);
    *out = it->second;
    return ZK_OK;
}
bool md_is_groth16_key(uint64_t h) {
    std::lock_guard<std::mutex> lk(g_md_mu);
    return g_md_keys.count(h) != 0;
}
int md_groth16_pk_free(uint64_t h) {
    std::shared_ptr<MdGroth16> K;
    {
        std::lock_guard<std::mutex> lk(g_md_mu);
        auto it = g_md_keys.find(h);
        if (it == g_md_keys.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)h);
        K = it->second;
        g_md_keys.erase(it);
    }
    { std::lock_guard<std::mutex> lk(*K->mu); }  // a proof in flight finishes first
    int rc = ZK_OK;
    for (uint64_t s : K->sub) {
        const int r = zk_bn254_groth16_pk_free(s);
        if (r != ZK_OK) rc = r;
    }
    for (void* d : K->owned) (void)hipFree(d);
    return rc;
}
int md_groth16_pk_info(uint64_t h, size_t* n_wires, size_t* n_public, uint32_t* log_domain, int* has_tables, int* n_entries_out) {
    std::shared_ptr<MdGroth16> K;
    ZK_TRY(md_find_key(h, &K));
    if (n_wires) *n_wires = K->n_wires;
    if (n_public) *n_public = K->n_public;
    if (log_domain) *log_domain = K->log_domain;
    if (n_entries_out) *n_entries_out = (int)K->entries.size();
    if (has_tables) {
        *has_tables = 1;
        for (uint64_t s : K->sub) {
            int t = 0;
            ZK_TRY(zk_bn254_groth16_pk_info(s, nullptr, nullptr, nullptr, &t));
            if (!t) *has_tables = 0;
        }
    }
    return ZK_OK;
}
int md_groth16_finalize(uint64_t h, const uint64_t* partials, size_t n_partials, const zk_fr* r, const zk_fr* s, uint8_t proof_out[128]) {
    std::shared_ptr<MdGroth16> K;
    ZK_TRY(md_find_key(h, &K));
    return zk_bn254_groth16_finalize(K->sub[0], partials, n_partials, r, s, proof_out);  // alpha, beta, delta: the same in every slice
}

// groth16.Prove over the entries of a composite key.  Entry k owns block k of a, b, c / h and wires [wlo[k], wlo[k+1]).
//   1. every entry gets its block of a, b, c and its range of w (host: straight from the caller's slices, each over its own PCIe link; device: a view when the
//      data sits in that entry's GPU, else a peer copy) and starts preparing its wire scalars (zk_bn254_groth16_msm5_pk_begin)
//   2. computeH in the six-transform schedule of zk_bn254_groth16_h_shard_dev (phases 0, 1 / 6, 4, 7, 8; nine all-to-all transposes between the entries)
//   3. every entry runs the five MSMs of its slice against its block of h (zk_bn254_groth16_msm5_pk_end) -> 768-byte record
//   4. the records are combined on the host (zk_bn254_groth16_finalize): the same bytes as the single-GPU prover's
int md_groth16_prove(uint64_t h, const void* a, const void* b, const void* c, size_t n_constraints, const void* w, size_t n_wires, const zk_fr* r_, const zk_fr* s_,
                     int on_device, uint8_t proof_out[128]) {
    if (!r_ || !s_ || !proof_out) return set_err(ZK_ERR_ARG, "null pointer");
    std::shared_ptr<MdGroth16> K;
    ZK_TRY(md_find_key(h, &K));
    const size_t N = (size_t)1 << K->log_domain, nw = K->n_wires;
    if (n_wires != nw) return set_err(ZK_ERR_LEN, "len(w) = %zu != %zu wires of the proving key", n_wires, nw);
    if ((n_constraints && (!a || !b || !c)) || (nw && !w)) return set_err(ZK_ERR_ARG, "null pointer");
    if (n_constraints > N) return set_err(ZK_ERR_ARG, "n_constraints = %zu exceeds the domain size %zu", n_constraints, N);
    std::lock_guard<std::mutex> proof_lock(*K->mu);
    const int G = (int)K->entries.size();
    const unsigned lg = K->log_g;
    const size_t M = N >> lg, chunk = (M >> lg) * 32;
    const int dev_abc = on_device ? device_of_pointer(a) : -1, dev_w = on_device ? device_of_pointer(w) : -1;
    Exchange X(K->entries);
    // per-entry buffers: three blocks + one exchange target, allocated by their owner, visible to the whole team for the peer copies
    std::vector<void*> xa(G, nullptr), xb(G, nullptr), xc(G, nullptr), xt(G, nullptr), dw(G, nullptr);
    std::vector<uint64_t> recs((size_t)G * 96, 0);
    const void* src_abc[3] = {a, b, c};
    int rc = team_run(K->entries, [&](int k) -> int {
        const int dev = ctx().device;
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->hi();
        const size_t wlo = K->wlo[k], wcnt = K->wlo[k + 1] - wlo;
        ZK_TRY(g.s->reserve(4 * M * 32 + wcnt * 32 + 8192));
        void* blk[3];
        for (int i = 0; i < 3; i++) blk[i] = g.s->alloc(M * 32);
        xa[k] = blk[0]; xb[k] = blk[1]; xc[k] = blk[2];
        xt[k] = g.s->alloc(M * 32);
        void* d_w = g.s->alloc(wcnt * 32 + 16);
        auto fail = [&](int r) { X.bar.fail(); return r; };
        // block k of a, b, c: elements [k M, (k+1) M) of the padded vectors
        for (int i = 0; i < 3; i++) {
            const size_t lo = (size_t)k * M, have = n_constraints > lo ? (n_constraints - lo < M ? n_constraints - lo : M) : 0;
            int r = copy_dev(blk[i], dev, (const char*)src_abc[i] + lo * 32, dev_abc, have * 32, st);
            if (r == ZK_OK && have < M && hipMemsetAsync((char*)blk[i] + have * 32, 0, (M - have) * 32, st) != hipSuccess) r = set_err(ZK_ERR_HIP, "hipMemsetAsync failed");
            if (r != ZK_OK) return fail(r);
        }
        const void* w_here = (const char*)w + wlo * 32;
        if (on_device && dev_w == dev) d_w = const_cast<void*>(w_here);  // already in this GPU's HBM: read in place
        else if (int r = copy_dev(d_w, dev, w_here, dev_w, wcnt * 32, st)) return fail(r);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(set_err(ZK_ERR_HIP, "upload failed"));
        dw[k] = d_w;
        uint64_t session = 0;
        if (int r = zk_bn254_groth16_msm5_pk_begin(K->sub[k], d_w, &session)) return fail(r);
        struct Abort { uint64_t* s; ~Abort() { if (*s) (void)zk_bn254_groth16_msm5_pk_abort(*s); } } abort_guard{&session};
        if (!X.bar.wait()) return set_err(ZK_ERR_HIP, "another entry failed");  // every block and exchange target exists from here on
        // ---- computeH, six-transform schedule (include/zkmi.h, zk_bn254_groth16_h_shard_dev).  The nine transposes travel on the slot's SECOND stream, one
        // array ahead of the arithmetic: while the phase kernels of array i run on `st`, array i + 1 is already crossing the links on `sc` (round 4 ran both on
        // one stream and drained it at every transpose: exchange and arithmetic strictly alternated).  ready[i] = the last kernel that wrote array i.
        std::vector<void*>* arrs[3] = {&xa, &xb, &xc};
        hipStream_t sc = g.s->stream;
        hipEvent_t ready[3] = {nullptr, nullptr, nullptr};
        struct Ev { hipEvent_t* e; ~Ev() { for (int i = 0; i < 3; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ready};
        for (int i = 0; i < 3; i++) {
            if (hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(ready[i], st) != hipSuccess) return fail(set_err(ZK_ERR_HIP, "event setup failed"));
        }
        auto mark = [&](int i) -> int { return hipEventRecord(ready[i], st) == hipSuccess ? ZK_OK : set_err(ZK_ERR_HIP, "hipEventRecord failed"); };
        auto phase = [&](int ph, void* pa, void* pb, void* pc) { return zk_bn254_groth16_h_shard_dev(ph, pa, pb, pc, K->log_domain, lg, (uint32_t)k, (void*)st); };
        // one transpose of array i: into the spare block, which then IS the array (the old block becomes the spare).  All members swap alike.  The sends wait for
        // the array's last kernel, the host waits for the sends (the arithmetic of the other arrays goes on meanwhile), and after the barrier every member's new
        // block is complete: kernels enqueued on `st` from here on may read it.
        auto transpose = [&](int i) -> int {
            if (hipStreamWaitEvent(sc, ready[i], 0) != hipSuccess) return fail(set_err(ZK_ERR_HIP, "hipStreamWaitEvent failed"));
            ZK_TRY(X.run(k, arrs[i]->data(), xt.data(), chunk, sc));
            std::swap((*arrs[i])[k], xt[k]);
            if (!X.bar.wait()) return set_err(ZK_ERR_HIP, "another entry failed");  // every member has swapped before anyone sends again
            return ZK_OK;
        };
        auto one = [&](int ph, int i) -> int {  // a phase on array i alone
            if (int r = phase(ph, (*arrs[i])[k], nullptr, nullptr)) return fail(r);
            return mark(i) == ZK_OK ? ZK_OK : fail(ZK_ERR_HIP);
        };
        for (int i = 0; i < 3; i++) { if (int r = transpose(i)) return r; if (int r = one(0, i)) return r; }
        for (int i = 0; i < 3; i++) { if (int r = transpose(i)) return r; if (int r = one(i < 2 ? 1 : 6, i)) return r; }
        for (int i = 0; i < 2; i++) { if (int r = transpose(i)) return r; if (int r = one(4, i)) return r; }
        if (int r = phase(7, xa[k], xb[k], nullptr)) return fail(r);
        if (mark(0) != ZK_OK) return fail(ZK_ERR_HIP);
        if (int r = transpose(0)) return r;
        if (int r = phase(8, xa[k], nullptr, xc[k])) return fail(r);
        // ---- the five MSMs of this slice against this block of h (still in flight on st: the session orders itself behind it)
        const uint64_t sess = session;
        session = 0;  // _end consumes the session whatever it returns
        return zk_bn254_groth16_msm5_pk_end(sess, xa[k], &recs[(size_t)k * 96], (void*)st);
    }, &X.bar);
    ZK_TRY(rc);
    return zk_bn254_groth16_finalize(K->sub[0], recs.data(), (size_t)G, r_, s_, proof_out);
}

// ------------------------------------------------------------------------------------------------ (*Domain).FFT / FFTInverse over host slices
// Block k of the stored order goes to entry k (its own PCIe link), the steps of zk_bn254_ntt_shard_dev run with two all-to-all transposes between the
// entries, the block comes back.  Schedule: include/zkmi.h (zk_bn254_ntt_shard_dev), parallel.py ntt_sharded.
int md_ntt_host(zk_fr* a, uint32_t log_n, int inverse, int decimation, int coset, const std::vector<int>& entries) {
    const int G = (int)entries.size(), lg = ilog2_exact((size_t)G);
    if (lg < 1 || lg > 3) return set_err(ZK_ERR_ARG, "a transform can be spread over 2, 4 or 8 device entries, not %d", G);
    if (log_n < 2u * lg + 2 || log_n > 28) return set_err(ZK_ERR_ARG, "log_n = %u cannot be block-sharded over %d entries", log_n, G);
    const size_t N = (size_t)1 << log_n, M = N >> lg, chunk = (M >> lg) * 32;
    Exchange X(entries);
    std::vector<void*> x(G, nullptr), xt(G, nullptr);
    return team_run(entries, [&](int k) -> int {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->stream;
        auto fail = [&](int r) { X.bar.fail(); return r; };
        if (int r = g.s->reserve(2 * M * 32 + 4096)) return fail(r);
        x[k] = g.s->alloc(M * 32);
        xt[k] = g.s->alloc(M * 32);
        if (hipMemcpyAsync(x[k], a + (size_t)k * M, M * 32, hipMemcpyHostToDevice, st) != hipSuccess) return fail(set_err(ZK_ERR_HIP, "upload failed"));
        if (!X.bar.wait()) return set_err(ZK_ERR_HIP, "another entry failed");
        auto step = [&](int s) { return zk_bn254_ntt_shard_dev(s, x[k], log_n, (uint32_t)lg, (uint32_t)k, inverse, decimation, coset, (void*)st); };
        auto transpose = [&]() -> int {
            ZK_TRY(X.run(k, x.data(), xt.data(), chunk, st));
            std::swap(x[k], xt[k]);
            if (!X.bar.wait()) return set_err(ZK_ERR_HIP, "another entry failed");
            return ZK_OK;
        };
        if (decimation == ZK_DIF) {
            if (coset && !inverse) if (int r = step(2)) return fail(r);
            if (int r = transpose()) return r;
            if (int r = step(0)) return fail(r);
            if (int r = transpose()) return r;
            if (int r = step(1)) return fail(r);
        } else {
            if (int r = step(1)) return fail(r);
            if (int r = transpose()) return r;
            if (int r = step(0)) return fail(r);
            if (int r = transpose()) return r;
            if (coset && inverse) if (int r = step(2)) return fail(r);
        }
        if (hipMemcpyAsync(a + (size_t)k * M, x[k], M * 32, hipMemcpyDeviceToHost, st) != hipSuccess) return set_err(ZK_ERR_HIP, "download failed");
        return slot_sync(g.s, st);
    }, &X.bar);
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// The entries that calls without a device_mask of their own spread over (bit i = entry i; 0 = the calling thread's entry only).  zk_init_devices sets it
// to every entry it was given.
int zk_set_default_devices(uint32_t mask) {
    for (int e = 0; e < 32; e++)
        if ((mask & (1u << e)) && e >= n_entries()) return set_err(ZK_ERR_ARG, "device mask names entry %d, %d entries exist", e, n_entries());
    md_set_default_mask(mask);
    return ZK_OK;
}
uint32_t zk_default_devices(void) { return md_default_mask(); }

}  // extern "C"
