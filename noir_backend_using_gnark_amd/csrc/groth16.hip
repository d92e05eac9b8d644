// Groth16 prove on gfx950: gnark v0.8.0 `groth16.Prove` (internal/backend/bn254/groth16/prove.go; pinned at
// /root/reference/gnark_backend_ffi/go.mod:23; the reference's only live call is /root/reference/gnark_backend_ffi/main.go:131,
// and its intended FFI shape is the commented-out ProveWithPK at backend/groth16/r1cs.go:107-143) from the solver's
// output onwards:
//      h   = computeH(a, b, c)                                  7 NTTs of size N            (ntt.hip)
//      Ar  = MSM(pk.G1.A, w) + alpha + r*delta
//      Bs1 = MSM(pk.G1.B, w) + beta  + s*delta
//      Bs  = MSM(pk.G2.B, w) + beta2 + s*delta2
//      Krs = MSM(pk.G1.K, w[nPub:]) + MSM(pk.G1.Z, h[:N-1]) + s*Ar + r*Bs1 - rs*delta
//      proof bytes = Ar | Bs | Krs compressed (Proof.WriteTo)
// The prover randomness (r, s) is an INPUT here (upstream: crypto/rand), which is what makes proof bytes reproducible.
// The five MSMs and the NTTs run on the device; the O(1) tail (5 scalar multiplications, a few additions, 3 inversions,
// compression) runs on the host like upstream.
#include <stdlib.h>
#include <chrono>
#include <functional>
#include <memory>
#include <vector>
#include <string.h>

#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"
#include "keyio.hpp"
#include "msm.hpp"
#include "multidev.hpp"
#include "ntt.hpp"
#include "proofio.hpp"

namespace zkmi {

// k * P for a base fixed at key-load time (pk.G1.Delta, pk.G2.Delta): 8-bit windows, [w][b] = b * 2^(8w) * P kept as XYZZ (no
// inversions to build, 32 additions per product) -- the (r, s)-dependent terms of the host tail drop from ~0.36 ms to ~0.05 ms,
// which matters where they cannot hide under GPU work (the multi-GPU finalize).
template <class F>
struct FixedBase {
    std::vector<XYZZ<F>> tab;
    void build(const Affine<F>& p) {
        tab.resize(32 * 256);
        XYZZ<F> base = XYZZ<F>::from_affine(p);
        for (int w = 0; w < 32; w++) {
            XYZZ<F> acc = XYZZ<F>::inf();
            tab[w * 256] = acc;
            for (int b = 1; b < 256; b++) {
                acc.add(base);
                tab[w * 256 + b] = acc;
            }
            for (int i = 0; i < 8; i++) base.dbl();
        }
    }
    XYZZ<F> mul(const uint32_t k[8]) const {
        XYZZ<F> acc = XYZZ<F>::inf();
        for (int w = 0; w < 32; w++) {
            uint32_t b = (k[w >> 2] >> (8 * (w & 3))) & 255u;
            if (b) acc.add(tab[w * 256 + b]);
        }
        return acc;
    }
};

struct Groth16PK {
    uint32_t log_domain = 0;
    size_t n_wires = 0, n_public = 0, nz = 0;  // nz: entries of Z used (N - 1; N for a non-final shard of a range-sharded key)
    Affine<HFp> alpha, beta, delta;
    Affine<HFp2> beta2, delta2;
    std::shared_ptr<FixedBase<HFp>> fb_delta;
    std::shared_ptr<FixedBase<HFp2>> fb_delta2;
    void *d_a = nullptr, *d_b = nullptr, *d_k = nullptr, *d_z = nullptr, *d_b2 = nullptr;
    bool owns = true;       // all five base arrays are the key's own allocations
    bool owns_abb = false;  // (with !owns) only the expanded A / B / G2.B arrays are
    int sessions = 0;       // live msm5 sessions holding copies of the device pointers: pk_free refuses while > 0
    // precomputed window tables T[w][i] = 2^(c*w) * P_i for the five base arrays (resident; built once at load time):
    // every window of an MSM then shares one bucket set -- ceil(255/c) * n mixed additions with c ~ 20 instead of 16 windows
    // of c = 16, one bucket reduction instead of 16, no Horner.  The K table uses wire indexing (first n_public rows = infinity).
    unsigned proofs = 0;    // proofs asked of this key so far (pk_tables_for_tail)
    bool tables = false;
    MsmTable tab_w, tab_h;
    void *t_a = nullptr, *t_b = nullptr, *t_k = nullptr, *t_z = nullptr, *t_b2 = nullptr;
};
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::mutex g_pk_mu;
static std::map<uint64_t, Groth16PK> g_pks;
static uint64_t g_next_pk = 1;

}  // namespace zkmi

using namespace zkmi;

extern "C" {

// Everything pk_load allocates, freed again on any early return (with tables at 2^24 constraints that is ~90 GB of HBM).
struct PkAllocs {
    std::vector<void*> ptrs;
    bool keep = false;
    int dev_alloc(void** d, size_t bytes) {
        ZK_HIP(hipMalloc(d, bytes ? bytes : 16));
        ptrs.push_back(*d);
        return ZK_OK;
    }
    ~PkAllocs() {
        if (!keep)
            for (void* p : ptrs) (void)hipFree(p);
    }
};

// gnark's []bool image -> for every wire the index into the compact array (or ~0u for a point at infinity)
static int infinity_map(const uint8_t* inf, size_t n_wires, size_t nb_inf, std::vector<uint32_t>* map, const char* which) {
    map->resize(n_wires);
    size_t j = 0, cnt = 0;
    for (size_t i = 0; i < n_wires; i++) {
        if (inf[i]) { (*map)[i] = 0xffffffffu; cnt++; }
        else (*map)[i] = (uint32_t)j++;
    }
    if (cnt != nb_inf) return set_err(ZK_ERR_ARG, "NbInfinity%s = %zu but Infinity%s marks %zu wires", which, nb_inf, which, cnt);
    return ZK_OK;
}

// The window tables of a key whose five base arrays are resident (P->d_*): at load time, or later for a key that was loaded without them (a key read from its
// wire format for ONE proof is better off without: 0.15 s at 2^20 constraints against the 5 ms per proof they save -- zk_bn254_groth16_pk_build_tables).
// Allocations go through `mem` (freed again unless the caller keeps them).  Nothing is built -- and that is not an error -- when the tables do not fit half of
// the free HBM or ZKMI_TABLE_CAP_GB, unless the caller asked for a width or a window shard.
static int pk_build_tables(Groth16PK* Pp, PkAllocs* mem, int table_window_bits, bool win_shard, uint32_t shard_rank, uint32_t shard_count) {
    Groth16PK& P = *Pp;
    const size_t N = (size_t)1 << P.log_domain, nk = P.n_wires - P.n_public;
    if (P.n_wires == 0 || N <= 1) return ZK_OK;
    P.tab_w.c = table_window_bits ? (unsigned)table_window_bits : msm_pick_window_table(P.n_wires);
    P.tab_w.stride = P.n_wires;
    P.tab_h.c = table_window_bits ? (unsigned)table_window_bits : msm_pick_window_table(P.nz);
    {   // experiment: a narrower window for Z alone (its reduction tail is the exposed one: half the buckets per bit, 1 / 13 more additions per step)
        const long ch = ZK_EXP("ZKMI_TABLE_C_H_DELTA", 0);
        if (!table_window_bits && ch < 0 && (long)P.tab_h.c + ch >= 8) P.tab_h.c = (unsigned)((long)P.tab_h.c + ch);
    }
    P.tab_h.stride = N;
    if (win_shard) {
        P.tab_w.row_first = P.tab_h.row_first = shard_rank;
        P.tab_w.row_step = P.tab_h.row_step = shard_count;
    }
    P.tab_w.l1_m = 16;  // A, B1, K, G2.B: their reduction tails hide under the next accumulate -- less work beats lower latency
    P.tab_w.l2_m = (unsigned)ZK_EXP("ZKMI_L2_M", 8);  // measured: 8 -> -0.09 ms, 16 -> +0.15 ms, 32 -> +0.9 ms (the level gets too long to hide)
    P.tab_h.l1_m = (unsigned)ZK_EXP("ZKMI_L1H_M", 8);   // Z finishes last: its tail is exposed
    P.tab_h.l2_m = (unsigned)ZK_EXP("ZKMI_L2H_M", 0);
    const size_t Ww = P.tab_w.rows() ? P.tab_w.rows() : 1, Wh = P.tab_h.rows() ? P.tab_h.rows() : 1;  // rows held here (all of them unless window-sharded)
    const size_t bytes = Ww * P.n_wires * (3 * 64 + 128) + Wh * N * 64;
    size_t free_b = 0, total_b = 0;
    ZK_HIP(hipMemGetInfo(&free_b, &total_b));
    // measured (end of round 1): tables pay at every size that fits -- 2^23: 65.8 vs 76.3 ms per proof, 2^24 (84 GB of tables): 126.8 vs
    // 148.1 ms -- so the only limits are half of the free HBM and a 128 GB cap (an early measurement that showed the opposite at 2^24
    // was an artefact of the task-size heuristic fixed since)
    static const size_t cap_gb = (size_t)zk_env_bounded("ZKMI_TABLE_CAP_GB", 128, 0, 1024);  // 0 = never build tables; the result does not depend on it
    if (bytes < free_b / 2 && bytes <= (cap_gb << 30)) {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->stream;
        struct { void** t; const void* src; size_t n, stride, off, esz; unsigned c; size_t Wd; int g2; } jobs[5] = {
            {&P.t_a, P.d_a, P.n_wires, P.n_wires, 0, 64, P.tab_w.c, Ww, 0},  {&P.t_b, P.d_b, P.n_wires, P.n_wires, 0, 64, P.tab_w.c, Ww, 0},
            {&P.t_k, P.d_k, nk, P.n_wires, P.n_public, 64, P.tab_w.c, Ww, 0}, {&P.t_z, P.d_z, P.nz, N, 0, 64, P.tab_h.c, Wh, 0},
            {&P.t_b2, P.d_b2, P.n_wires, P.n_wires, 0, 128, P.tab_w.c, Ww, 1}};
        for (auto& j : jobs) {
            if (bg_cancelled()) {  // the process is exiting under a background build: what was queued is drained, nothing is published
                (void)slot_sync(g.s, st);
                return set_err(ZK_ERR_BUSY, "window tables: the process is exiting");
            }
            ZK_TRY(mem->dev_alloc(j.t, j.Wd * j.stride * j.esz));
            const MsmTable& tb = (&j == &jobs[3]) ? P.tab_h : P.tab_w;
            ZK_TRY(j.g2 ? msm_build_table_g2(g.s, st, j.src, j.n, j.stride, j.off, j.c, *j.t, tb.row_first, tb.row_step)
                        : msm_build_table_g1(g.s, st, j.src, j.n, j.stride, j.off, j.c, *j.t, tb.row_first, tb.row_step));
        }
        ZK_TRY(slot_sync(g.s, st));
        P.tables = true;
    } else if (table_window_bits || win_shard) {
        return set_err(ZK_ERR_HIP, "window tables of %zu bytes (c = %u) do not fit (free HBM %zu)", bytes, P.tab_w.c, free_b);
    }
    return ZK_OK;
}

int zk_bn254_groth16_pk_load(const zk_groth16_pk* pk, uint64_t* handle) {
    if (!pk || !handle) return set_err(ZK_ERR_ARG, "null pointer");
    // several device entries (pk->device_mask / the process default): a composite key of per-entry range slices (multidev.hip); one entry: a key on that entry.
    // Slices of a key that is already sharded by its caller (flags bits 1, 2: one process per GPU) stay where the caller's thread is.
    std::vector<int> ents;
    if (pk->flags & 6) ents.push_back(current_entry());
    else ZK_TRY(md_entries_for((uint32_t)pk->device_mask, pk->n_wires, (size_t)1 << 18, &ents));
    if (ents.size() > 1) return md_groth16_pk_load(pk, ents, handle);
    CtxScope _entry(ents[0]);
    if (_entry.rc != ZK_OK) return _entry.rc;
    if (pk->log_domain > 28 || pk->n_public > pk->n_wires) return set_err(ZK_ERR_ARG, "bad proving-key geometry");
    if (pk->n_wires >= ((size_t)1 << 31)) return set_err(ZK_ERR_ARG, "n_wires = %zu does not fit 31 bits", pk->n_wires);
    {   // experiment: the five slots of a proof session (and their high-priority streams) exist from the key's load on, not from the first proof on
        static const bool warm = ZK_EXP("ZKMI_PK_LOAD_WARMS_SESSION", 0) != 0;
        if (warm) {
            SlotsGuard<5> g;
            ZK_TRY(acquire_slots(5, g.s));
        }
    }
    if (!pk->g1_alpha || !pk->g1_beta || !pk->g1_delta || !pk->g2_beta || !pk->g2_delta) return set_err(ZK_ERR_ARG, "null pk element");
    if ((pk->infinity_a == nullptr) != (pk->infinity_b == nullptr)) return set_err(ZK_ERR_ARG, "InfinityA and InfinityB must be given together");
    if (!pk->infinity_a && (pk->nb_infinity_a || pk->nb_infinity_b)) return set_err(ZK_ERR_ARG, "NbInfinityA/B without the bitmaps");
    if (pk->table_window_bits && (pk->table_window_bits < 8 || pk->table_window_bits > 24))
        return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", pk->table_window_bits);
    const bool compact = pk->infinity_a != nullptr;
    const bool win_shard = (pk->flags & 4) != 0;
    if (win_shard && (pk->shard_count < 1 || pk->shard_count > 64 || pk->shard_rank >= pk->shard_count)) return set_err(ZK_ERR_ARG, "bad window-shard rank / count");
    if (win_shard && (pk->flags & 3)) return set_err(ZK_ERR_ARG, "a window-sharded key holds the whole key and needs its tables (flags bits 0 and 1 do not apply)");
    if (compact && (pk->nb_infinity_a > pk->n_wires || pk->nb_infinity_b > pk->n_wires)) return set_err(ZK_ERR_ARG, "NbInfinity exceeds the wire count");
    std::vector<uint32_t> map_a, map_b;
    if (compact) {  // validated before anything is allocated or read
        ZK_TRY(infinity_map(pk->infinity_a, pk->n_wires, pk->nb_infinity_a, &map_a, "A"));
        ZK_TRY(infinity_map(pk->infinity_b, pk->n_wires, pk->nb_infinity_b, &map_b, "B"));
    }
    ZK_TRY(ensure_init());
    Groth16PK P;
    PkAllocs mem;
    P.log_domain = pk->log_domain;
    P.n_wires = pk->n_wires;
    P.n_public = pk->n_public;
    memcpy(&P.alpha, pk->g1_alpha, 64);
    memcpy(&P.beta, pk->g1_beta, 64);
    memcpy(&P.delta, pk->g1_delta, 64);
    memcpy(&P.beta2, pk->g2_beta, 128);
    memcpy(&P.delta2, pk->g2_delta, 128);
    // (the 8-bit window tables of delta / delta2 for the host tail are NOT built here: 8-17 ms of one host core, which a process that makes one proof -- the
    // reference's use: nargo loads the library, proves once and exits -- never earns back at 0.3 ms per proof; pk_note_proof builds them at a key's second proof)
    const size_t N = (size_t)1 << pk->log_domain, nw = pk->n_wires, nk = pk->n_wires - pk->n_public;
    const size_t na = compact ? nw - pk->nb_infinity_a : nw, nbb = compact ? nw - pk->nb_infinity_b : nw;  // entries the caller's A / B arrays hold
    P.nz = (pk->flags & 2) ? N : N - 1;
    // ---- the five base arrays, wire-indexed, resident
    struct Up { void** d; const void* src; size_t n_src, esz; } up[5] = {{&P.d_a, pk->g1_a, na, 64}, {&P.d_b, pk->g1_b, nbb, 64}, {&P.d_k, pk->g1_k, nk, 64},
                                                                        {&P.d_z, pk->g1_z, N, 64},   {&P.d_b2, pk->g2_b, nbb, 128}};
    for (auto& u : up)
        if (u.n_src && !u.src) return set_err(ZK_ERR_ARG, "null pk base array");
    if (pk->bases_on_device && !compact) {
        P.owns = false;
        for (auto& u : up) *u.d = const_cast<void*>(u.src);
    } else if (pk->bases_on_device) {
        // compact A / B / G2.B already in HBM: only they are expanded into arrays of the key's own; K and Z stay the caller's
        P.owns = false;
        P.d_k = (void*)pk->g1_k;
        P.d_z = (void*)pk->g1_z;
    } else {
        for (int i = 2; i < 4; i++) {  // K, Z: as they are
            ZK_TRY(mem.dev_alloc(up[i].d, up[i].n_src * up[i].esz));
            if (up[i].n_src) ZK_HIP(hipMemcpy(*up[i].d, up[i].src, up[i].n_src * up[i].esz, hipMemcpyHostToDevice));
        }
        if (!compact)
            for (int i : {0, 1, 4}) {
                ZK_TRY(mem.dev_alloc(up[i].d, up[i].n_src * up[i].esz));
                if (up[i].n_src) ZK_HIP(hipMemcpy(*up[i].d, up[i].src, up[i].n_src * up[i].esz, hipMemcpyHostToDevice));
            }
    }
    if (compact) {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        hipStream_t st = g.s->stream;
        const size_t stage = pk->bases_on_device ? 0 : (na * 64 + nbb * 64 + nbb * 128 + 3 * 256);
        ZK_TRY(g.s->reserve(2 * (nw * 4 + 256) + stage + 4096));
        uint32_t* d_ma = (uint32_t*)g.s->alloc(nw * 4 + 16);
        uint32_t* d_mb = (uint32_t*)g.s->alloc(nw * 4 + 16);
        if (nw) {
            ZK_HIP(hipMemcpyAsync(d_ma, map_a.data(), nw * 4, hipMemcpyHostToDevice, st));
            ZK_HIP(hipMemcpyAsync(d_mb, map_b.data(), nw * 4, hipMemcpyHostToDevice, st));
        }
        const uint32_t* maps[5] = {d_ma, d_mb, nullptr, nullptr, d_mb};
        for (int i : {0, 1, 4}) {
            const void* d_src = up[i].src;
            if (!pk->bases_on_device) {
                void* d_stage = g.s->alloc(up[i].n_src * up[i].esz + 16);
                if (up[i].n_src) ZK_HIP(hipMemcpyAsync(d_stage, up[i].src, up[i].n_src * up[i].esz, hipMemcpyHostToDevice, st));
                d_src = d_stage;
            }
            ZK_TRY(mem.dev_alloc(up[i].d, nw * up[i].esz));
            ZK_TRY(msm_expand_bases(g.s, st, i == 4, d_src, maps[i], nw, *up[i].d));
        }
        ZK_TRY(slot_sync(g.s, st));
        P.owns_abb = true;
    }
    // ---- precomputed window tables (unless disabled or HBM is short)
    if (!(pk->flags & 1)) ZK_TRY(pk_build_tables(&P, &mem, pk->table_window_bits, win_shard, pk->shard_rank, pk->shard_count));
    if (win_shard && !P.tables) return set_err(ZK_ERR_ARG, "a window-sharded key needs its window tables");
    std::lock_guard<std::mutex> lk(g_pk_mu);
    *handle = hmake(g_next_pk++);
    g_pks[*handle] = P;
    mem.keep = true;
    return ZK_OK;
}

int zk_bn254_groth16_pk_free(uint64_t handle) {
    if (md_is_composite(handle)) return md_groth16_pk_free(handle);
    ZK_ON_ENTRY_OF(handle);
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
    if (it->second.sessions > 0)
        return set_err(ZK_ERR_HANDLE, "proving key %llu is still used by %d msm5 session(s): end or abort them first", (unsigned long long)handle, it->second.sessions);
    const Groth16PK& P = it->second;
    if (P.owns) {
        for (void* d : {P.d_a, P.d_b, P.d_k, P.d_z, P.d_b2}) (void)hipFree(d);
    } else if (P.owns_abb) {
        for (void* d : {P.d_a, P.d_b, P.d_b2}) (void)hipFree(d);
    }
    if (P.tables)
        for (void* d : {P.t_a, P.t_b, P.t_k, P.t_z, P.t_b2}) (void)hipFree(d);
    g_pks.erase(it);
    return ZK_OK;
}

// Window tables for a resident key that was loaded without them (flags bit 0) -- what a caller does when a key turns out to be used again: the export path
// reads a key text without tables for its first proof and builds them when the second one is asked for.  table_window_bits: 0 = the planner's choice.
// A key that has its tables, or whose tables do not fit (half of the free HBM, ZKMI_TABLE_CAP_GB), is left as it is: *built (optional) says which.
int zk_bn254_groth16_pk_build_tables(uint64_t handle, int table_window_bits, int* built) {
    if (built) *built = 0;
    if (md_is_composite(handle)) return set_err(ZK_ERR_ARG, "a key spread over several device entries gets its tables at load time");
    ZK_ON_ENTRY_OF(handle);
    if (table_window_bits && (table_window_bits < 8 || table_window_bits > 24)) return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", table_window_bits);
    Groth16PK P;
    {
        std::lock_guard<std::mutex> lk(g_pk_mu);
        auto it = g_pks.find(handle);
        if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
        if (it->second.tables) { if (built) *built = 1; return ZK_OK; }
        if (it->second.nz != ((size_t)1 << it->second.log_domain) - 1) return set_err(ZK_ERR_ARG, "a range-sharded slice of a key gets its tables at load time");
        P = it->second;
    }
    PkAllocs mem;
    ZK_TRY(pk_build_tables(&P, &mem, table_window_bits, false, 0, 1));
    if (!P.tables) return ZK_OK;
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end() || it->second.tables) return ZK_OK;  // freed, or built by another caller meanwhile: `mem` releases this copy
    Groth16PK& Q = it->second;
    Q.tab_w = P.tab_w; Q.tab_h = P.tab_h;
    Q.t_a = P.t_a; Q.t_b = P.t_b; Q.t_k = P.t_k; Q.t_z = P.t_z; Q.t_b2 = P.t_b2;
    Q.tables = true;
    mem.keep = true;
    if (built) *built = 1;
    return ZK_OK;
}

// HBM a resident key holds (base arrays it owns + window tables): what a cache of keys accounts for
int zk_bn254_groth16_pk_bytes(uint64_t handle, size_t* bytes) {
    if (!bytes) return set_err(ZK_ERR_ARG, "null pointer");
    if (md_is_composite(handle)) return set_err(ZK_ERR_ARG, "not available for a key spread over several device entries");
    ZK_ON_ENTRY_OF(handle);
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
    const Groth16PK& P = it->second;
    const size_t N = (size_t)1 << P.log_domain, nw = P.n_wires, nk = nw - P.n_public;
    size_t b = 0;
    if (P.owns) b += nw * (64 + 64 + 128) + nk * 64 + N * 64;
    else if (P.owns_abb) b += nw * (64 + 64 + 128);
    if (P.tables) b += (size_t)P.tab_w.rows() * nw * (3 * 64 + 128) + (size_t)P.tab_h.rows() * N * 64;
    *bytes = b;
    return ZK_OK;
}

}  // extern "C"
namespace zkmi {
// the key takes ownership of its five device base arrays (groth16.Setup builds them in HBM and hands them over)
int groth16_pk_adopt(uint64_t handle) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
    it->second.owns = true;
    it->second.owns_abb = false;
    return ZK_OK;
}
int groth16_pk_view(uint64_t handle, Groth16View* v) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
    const Groth16PK& P = it->second;
    v->log_domain = P.log_domain;
    v->n_wires = P.n_wires; v->n_public = P.n_public; v->nz = P.nz;
    v->alpha = P.alpha; v->beta = P.beta; v->delta = P.delta;
    v->beta2 = P.beta2; v->delta2 = P.delta2;
    v->d_a = P.d_a; v->d_b = P.d_b; v->d_k = P.d_k; v->d_z = P.d_z; v->d_b2 = P.d_b2;
    return ZK_OK;
}
}  // namespace zkmi
extern "C" {

int zk_bn254_groth16_pk_info(uint64_t handle, size_t* n_wires, size_t* n_public, uint32_t* log_domain, int* has_tables) {
    if (md_is_composite(handle)) return md_groth16_pk_info(handle, n_wires, n_public, log_domain, has_tables, nullptr);
    ZK_ON_ENTRY_OF(handle);
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(handle);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)handle);
    if (n_wires) *n_wires = it->second.n_wires;
    if (n_public) *n_public = it->second.n_public;
    if (log_domain) *log_domain = it->second.log_domain;
    if (has_tables) *has_tables = it->second.tables ? 1 : 0;
    return ZK_OK;
}

// The five MSMs of one proof (or of one rank's shard of it) on device-resident inputs; un-normalised XYZZ sums out:
// out[0..16) = MSM(G1.A, w), [16..32) = MSM(G1.B, w), [32..48) = MSM(G1.K, wk), [48..64) = MSM(G1.Z, h), [64..96) = MSM(G2.B, w).
// Each MSM runs on its own stream slot so that the latency-bound tail kernels of one (bucket reduction: a few waves)
// overlap with the throughput-bound accumulate kernels of the others.  `ev_w` / `ev_h` (optional) gate the streams on
// the producers of w and h.
struct Msm5Inputs {
    const void *d_a, *d_b, *d_b2, *d_w;
    size_t nw;
    const void *d_k, *d_wk;
    size_t nk;
    const void *d_z, *d_h;
    size_t nz;
    // optional precomputed tables (then d_a, d_b, d_b2, d_k, d_z are ignored)
    const MsmTable *tab_w = nullptr, *tab_h = nullptr;
    const void *t_a = nullptr, *t_b = nullptr, *t_b2 = nullptr, *t_k = nullptr, *t_z = nullptr;
};
static const zk_msm_cfg kMontCfg = {0, 1, 0, 0};  // scalars are Montgomery fr.Element images

// K pairs with w[j:] for some j (gnark: j = number of public wires): then A, B1, K and G2.B all share the scalar-side
// work (digits, sort, bucket bounds, task plan) of ONE preparation of w; K's bases are addressed as K[i - j].
static bool k_shares_w(const Msm5Inputs& in, size_t* j) {
    if (!in.nk || !in.nw) return false;
    ptrdiff_t d = (const char*)in.d_wk - (const char*)in.d_w;
    if (d < 0 || d % 32) return false;
    *j = (size_t)d / 32;
    return *j + in.nk == in.nw;
}

struct Msm5State {
    MsmJob jobs[5];
    MsmPrep prep_w, prep_h;
    hipStream_t chain = nullptr;  // stream on which the accumulate kernels are serialised
};

// Stream plan.  Every kernel of a proof is ALU-bound except the sorts, so nothing is gained by letting accumulate kernels
// share the machine (measured: the big G2 kernel is starved and finishes last, alone, with its tail exposed).  Instead the
// accumulate kernels are CHAINED through events -- G2.B -> A -> B1 -> K -> Z, each owning the GPU in turn -- and the
// latency-bound bucket-reduction tail of one MSM runs underneath the next MSM's accumulate:
//   slot 4 (high priority): prepare(w) -> G2.B accumulate -> reduce
//   slots 1, 2, 3          : A, B1, K accumulate -> reduce (all read prepare(w))
//   slot 0 (high priority): [computeH ->] prepare(h) -> Z accumulate -> reduce          (st0)
// The HOST enqueue order matters too (~10 us per launch, ~40 launches per preparation): the w-side work is enqueued first so
// that the GPU is busy while the host is still enqueuing computeH and prepare(h).
// scalar side of the four MSMs over the wire values (digits, sort, task plan) on slot 4's stream
static int msm5_prepare_w(Slot* sl[5], const Msm5Inputs& in, hipEvent_t ev_w, Msm5State* S) {
    static const bool low = ZK_EXP("ZKMI_PREPW_LOW", 0) == 1;  // experiment: normal priority under computeH
    hipStream_t st4 = low ? sl[4]->stream : sl[4]->hi();
    if (ev_w) ZK_HIP(hipStreamWaitEvent(st4, ev_w, 0));
    // The wire values of a real circuit are mostly 0 / 1 / small (booleans, bytes, range-check limbs): their zero digits never enter the sort (msm.hip
    // k_msm_digits_compact; the pair count stays on the device): 2^20 witness-like wires 6.5 -> 6.1 ms.  A vector without zero digits pays nothing for it
    // (one pass over the scalars either way).  ZKMI_W_DROP_ZERO_DIGITS=0 (experiments build): the ordinary recoding.
    static const bool drop = ZK_EXP("ZKMI_W_DROP_ZERO_DIGITS", 1) != 0;
    if (in.tab_w) return msm_prepare_scalars_table(sl[4], st4, in.d_w, in.nw, &kMontCfg, *in.tab_w, &S->prep_w, drop);
    return msm_prepare_scalars(sl[4], st4, in.d_w, in.nw, &kMontCfg, &S->prep_w, drop);
}
// `between` (optional): called once the G2.B accumulate is enqueued, with its completion event; it may enqueue work that must own the machine
// next (computeH when the inputs came from the host) and return the event the A accumulate has to wait for instead.
typedef std::function<int(hipEvent_t g2_done, hipEvent_t* gate_next)> BetweenFn;
static int msm5_accumulate_w(Slot* sl[5], const Msm5Inputs& in, hipEvent_t ev_w, Msm5State* S, hipEvent_t gate_first_acc, const BetweenFn* between = nullptr);
static int msm5_launch_w(Slot* sl[5], const Msm5Inputs& in, hipEvent_t ev_w, Msm5State* S, hipEvent_t gate_first_acc = nullptr) {
    ZK_TRY(msm5_prepare_w(sl, in, ev_w, S));
    return msm5_accumulate_w(sl, in, ev_w, S, gate_first_acc);
}
static int msm5_accumulate_w(Slot* sl[5], const Msm5Inputs& in, hipEvent_t ev_w, Msm5State* S, hipEvent_t gate_first_acc, const BetweenFn* between) {
    hipStream_t st4 = sl[4]->hi();
    size_t j = 0;
    const bool share_k = k_shares_w(in, &j);
    // accumulate kernels chained through events, each on its MSM's own stream (measured alternatives: one shared "chain" stream
    // for all accumulate kernels removes the ~0.15 ms event gaps but delays prepare(h) -- rocPRIM's onesweep sort spins on
    // look-back tiles that cannot get a wave slot under an accumulate kernel -- and ends up slower)
    // ZKMI_CHAIN=1 (experiment switch): all accumulate kernels back to back on ONE stream, tails on the jobs' own streams
    static const bool one_chain = ZK_EXP("ZKMI_CHAIN", 0) == 1;
    if (one_chain && (in.tab_w || share_k)) {
        hipStream_t chain = sl[4]->stream;
        S->jobs[4].gate_acc = gate_first_acc;
        S->jobs[4].chain = chain;
        ZK_TRY(msm_g2_accumulate(sl[4], st4, S->prep_w, in.tab_w ? in.t_b2 : in.d_b2, 0, &S->jobs[4]));
        S->jobs[1].chain = chain;
        ZK_TRY(msm_g1_accumulate(sl[1], sl[1]->stream, S->prep_w, in.tab_w ? in.t_a : in.d_a, 0, &S->jobs[1]));
        S->jobs[2].chain = chain;
        ZK_TRY(msm_g1_accumulate(sl[2], sl[2]->stream, S->prep_w, in.tab_w ? in.t_b : in.d_b, 0, &S->jobs[2]));
        S->jobs[3].chain = chain;
        S->jobs[3].want_done = true;
        if (in.tab_w) ZK_TRY(msm_g1_accumulate(sl[3], sl[3]->stream, S->prep_w, in.t_k, 0, &S->jobs[3]));
        else ZK_TRY(msm_g1_accumulate(sl[3], sl[3]->stream, S->prep_w, (const char*)in.d_k - j * 64, (uint32_t)j, &S->jobs[3]));
        return ZK_OK;
    }
    S->jobs[4].gate_acc = gate_first_acc;
    S->jobs[4].want_done = true;
    ZK_TRY(msm_g2_accumulate(sl[4], st4, S->prep_w, in.tab_w ? in.t_b2 : in.d_b2, 0, &S->jobs[4]));
    hipEvent_t prev = S->jobs[4].acc_done;
    if (between) ZK_TRY((*between)(S->jobs[4].acc_done, &prev));
    // ZKMI_BATCH_ACC=1 (experiment switch): A, B1, K -- which read the same sorted digits -- in ONE accumulate launch (grid.y = 3).
    // Measured 0.25 ms SLOWER per proof than the chained launches (11.2 vs 10.95 ms): the kernel itself runs at 0.70 of the madd
    // peak instead of 0.61, but the three reduction tails then all start late and pile up under Z instead of hiding one by one.
    static const bool batch_acc = ZK_EXP("ZKMI_BATCH_ACC", 0) == 1;
    if (batch_acc && (in.tab_w || share_k)) {
        Slot* bs[3] = {sl[1], sl[2], sl[3]};
        hipStream_t bst[3] = {sl[1]->stream, sl[2]->stream, sl[3]->stream};
        const void* bp[3] = {in.tab_w ? in.t_a : in.d_a, in.tab_w ? in.t_b : in.d_b, in.tab_w ? in.t_k : (const void*)((const char*)in.d_k - j * 64)};
        uint32_t bskip[3] = {0, 0, in.tab_w ? 0u : (uint32_t)j};
        MsmJob* bj[3] = {&S->jobs[1], &S->jobs[2], &S->jobs[3]};
        S->jobs[1].gate_acc = prev;
        S->jobs[3].want_done = true;
        return msm_g1_accumulate_batch(3, bs, bst, S->prep_w, bp, bskip, bj);
    }
    S->jobs[1].gate_acc = prev;
    S->jobs[1].want_done = true;
    ZK_TRY(msm_g1_accumulate(sl[1], sl[1]->stream, S->prep_w, in.tab_w ? in.t_a : in.d_a, 0, &S->jobs[1]));
    if (S->jobs[1].acc_done) prev = S->jobs[1].acc_done;
    S->jobs[2].gate_acc = prev;
    S->jobs[2].want_done = true;
    ZK_TRY(msm_g1_accumulate(sl[2], sl[2]->stream, S->prep_w, in.tab_w ? in.t_b : in.d_b, 0, &S->jobs[2]));
    if (S->jobs[2].acc_done) prev = S->jobs[2].acc_done;
    S->jobs[3].gate_acc = prev;
    S->jobs[3].want_done = true;
    if (in.tab_w) {
        ZK_TRY(msm_g1_accumulate(sl[3], sl[3]->stream, S->prep_w, in.t_k, 0, &S->jobs[3]));  // K table is wire-indexed
    } else if (share_k) {
        ZK_TRY(msm_g1_accumulate(sl[3], sl[3]->stream, S->prep_w, (const char*)in.d_k - j * 64, (uint32_t)j, &S->jobs[3]));
    } else {
        if (ev_w) ZK_HIP(hipStreamWaitEvent(sl[3]->stream, ev_w, 0));
        ZK_TRY(msm_g1_launch(sl[3], sl[3]->stream, in.d_k, in.d_wk, in.nk, &kMontCfg, &S->jobs[3]));
    }
    return ZK_OK;
}
// Z side, on st0 (after whatever produced h on that stream); its accumulate goes last in the chain
static int msm5_prepare_h(Slot* sl[5], hipStream_t st0, const Msm5Inputs& in, Msm5State* S) {
    if (in.tab_h) return msm_prepare_scalars_table(sl[0], st0, in.d_h, in.nz, &kMontCfg, *in.tab_h, &S->prep_h);
    return msm_prepare_scalars(sl[0], st0, in.d_h, in.nz, &kMontCfg, &S->prep_h);
}
static int msm5_launch_h(Slot* sl[5], hipStream_t st0, const Msm5Inputs& in, Msm5State* S, bool prepared = false) {
    // ZKMI_PREPH_LOW=1 (experiment switch): prepare(h) on slot 0's NORMAL-priority stream (it has ~7 ms of slack until Z needs it)
    static const bool preph_low = ZK_EXP("ZKMI_PREPH_LOW", 0) == 1;
    if (!prepared && preph_low && sl[0]->stream != st0) {
        hipEvent_t ev;
        ZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(ev, st0));
        ZK_HIP(hipStreamWaitEvent(sl[0]->stream, ev, 0));
        (void)hipEventDestroy(ev);
        ZK_TRY(msm5_prepare_h(sl, sl[0]->stream, in, S));
        prepared = true;
    }
    if (!prepared) ZK_TRY(msm5_prepare_h(sl, st0, in, S));
    // Z's accumulate goes last on the chain stream (which then waits for prepare(h) through the event inside msm_accumulate)
    S->jobs[0].chain = S->chain;
    if (!S->chain) {
        hipEvent_t prev = nullptr;
        for (int i : {4, 1, 2, 3})
            if (S->jobs[i].acc_done) prev = S->jobs[i].acc_done;
        S->jobs[0].gate_acc = prev;
    }
    // ZKMI_QUAD_TAIL=1 (experiment switch): wave levels of Z's reduction tail -- the only tail nothing can hide -- with four lanes per
    // point.  Measured: those levels shrink 0.39 -> 0.27 ms under the profiler but the proof does not (10.6-10.8 ms either way), and
    // level 1, which already fills every SIMD, gets 3x slower in that form; off by default.
    static const bool quad_tail = ZK_EXP("ZKMI_QUAD_TAIL", 0) == 1;
    S->jobs[0].quad_tail = quad_tail;
    ZK_TRY(msm_g1_accumulate(sl[0], st0, S->prep_h, in.tab_h ? in.t_z : in.d_z, 0, &S->jobs[0]));
    return ZK_OK;
}
// `early` (optional) is called as soon as the A and B1 sums are on the host -- long before K and Z finish -- so that the part of
// the host tail that only needs them runs under the remaining GPU work.
static int msm5_finish(Msm5State* S, uint64_t out[96], const std::function<void(const XYZZ<HFp>&, const XYZZ<HFp>&)>* early = nullptr) {
    XYZZ<HFp> m_a, m_b, m_k, m_z;
    XYZZ<HFp2> m_b2;
    // in completion order (the accumulate chain is G2.B -> A -> B1 -> K -> Z), so that host work overlaps the GPU work still in flight
    int rc = msm_g2_finish(S->jobs[4], &m_b2);
    if (rc == ZK_OK) rc = msm_g1_finish(S->jobs[1], &m_a);
    if (rc == ZK_OK) rc = msm_g1_finish(S->jobs[2], &m_b);
    if (rc == ZK_OK && early) (*early)(m_a, m_b);
    if (rc == ZK_OK) rc = msm_g1_finish(S->jobs[3], &m_k);
    if (rc == ZK_OK) rc = msm_g1_finish(S->jobs[0], &m_z);
    msm_prep_release(&S->prep_w);
    msm_prep_release(&S->prep_h);
    for (int i = 0; i < 5; i++)
        if (S->jobs[i].acc_done) { (void)hipEventDestroy(S->jobs[i].acc_done); S->jobs[i].acc_done = nullptr; }
    ZK_TRY(rc);
    memcpy(out, &m_a, 128);
    memcpy(out + 16, &m_b, 128);
    memcpy(out + 32, &m_k, 128);
    memcpy(out + 48, &m_z, 128);
    memcpy(out + 64, &m_b2, 256);
    return ZK_OK;
}
// arena reservations: slot 0 also carries `extra0` bytes of the caller's own buffers
static int msm5_reserve(Slot* sl[5], const Msm5Inputs& in, size_t extra0) {
    size_t need[5] = {0, 0, 0, 0, 0}, prep = 0, acc1 = 0, acc2 = 0, j = 0;
    if (in.tab_h) {
        size_t ph = 0, ah = 0;
        ZK_TRY(msm_prep_need_table(in.nz, *in.tab_h, sl[0]->stream, &ph, &ah, nullptr));
        need[0] = ph + ah;
    } else {
        ZK_TRY(msm_g1_need(in.nz, &kMontCfg, sl[0]->stream, &need[0]));
    }
    if (in.tab_w) ZK_TRY(msm_prep_need_table(in.nw, *in.tab_w, sl[1]->stream, &prep, &acc1, &acc2));
    else ZK_TRY(msm_prep_need(in.nw, &kMontCfg, sl[1]->stream, &prep, &acc1, &acc2));
    need[1] = acc1;
    need[2] = acc1;
    need[4] = prep + acc2 + msm_compact_need(in.nw, 1);
    if (in.tab_w || k_shares_w(in, &j)) need[3] = acc1;
    else ZK_TRY(msm_g1_need(in.nk, &kMontCfg, sl[3]->stream, &need[3]));
    need[0] += extra0;
    for (int i = 0; i < 5; i++) ZK_TRY(sl[i]->reserve(need[i] + 4096));
    return ZK_OK;
}

static int lookup_pk(uint64_t h, Groth16PK* P) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(h);
    if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)h);
    *P = it->second;
    return ZK_OK;
}
// Called once per proof, just before the host tail: from the key's SECOND proof on the tail multiplies delta / delta2 through 8-bit window tables (FixedBase:
// built here, once, 8-17 ms of one host core -- in the single-GPU prover that is under the GPU work the call has just enqueued).
static int pk_tables_for_tail(uint64_t h, Groth16PK* P) {
    bool build = false;
    {
        std::lock_guard<std::mutex> lk(g_pk_mu);
        auto it = g_pks.find(h);
        if (it == g_pks.end()) return set_err(ZK_ERR_HANDLE, "unknown proving-key handle %llu", (unsigned long long)h);
        build = ++it->second.proofs >= 2 && !it->second.fb_delta;
        P->fb_delta = it->second.fb_delta;
        P->fb_delta2 = it->second.fb_delta2;
    }
    if (!build) return ZK_OK;
    auto f1 = std::make_shared<FixedBase<HFp>>();
    auto f2 = std::make_shared<FixedBase<HFp2>>();
    f1->build(P->delta);
    f2->build(P->delta2);
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(h);
    if (it != g_pks.end() && !it->second.fb_delta) { it->second.fb_delta = f1; it->second.fb_delta2 = f2; }
    P->fb_delta = f1;
    P->fb_delta2 = f2;
    return ZK_OK;
}

// host tail of groth16.Prove.  The four scalar multiplications that do not depend on the MSM results
// (r*delta, s*delta, s*delta2, rs*delta) are computed while the GPU is still busy (tail_pre); what is left afterwards
// is s*(A+alpha) and r*(B+beta):   Krs = K + Z + s*Ar + r*Bs1 - rs*delta = K + Z + s*(A+alpha) + r*(B+beta) + rs*delta.
struct TailPre {
    uint32_t rk[8], sk[8];
    XYZZ<HFp> r_delta, s_delta, rs_delta;
    XYZZ<HFp2> s_delta2;
    bool have_early = false;  // s*(A + alpha) and r*(B1 + beta) already computed (single-GPU prove: under the K / Z accumulates)
    XYZZ<HFp> s_aalpha, r_bbeta;
};
static void tail_early(const Groth16PK& P, TailPre* T, const XYZZ<HFp>& m_a, const XYZZ<HFp>& m_b) {
    const double t0 = now_ms();
    XYZZ<HFp> a_alpha = m_a, b_beta = m_b;
    a_alpha.madd(P.alpha);
    b_beta.madd(P.beta);
    T->s_aalpha = scalar_mul(a_alpha.to_affine(), T->sk);
    T->r_bbeta = scalar_mul(b_beta.to_affine(), T->rk);
    T->have_early = true;
    prof_host("host_tail_early", now_ms() - t0);
}
static void tail_pre(const Groth16PK& P, const zk_fr* r_, const zk_fr* s_, TailPre* T) {
    const double t0 = now_ms();
    HFr r, s;
    memcpy(&r, r_, 32);
    memcpy(&s, s_, 32);
    HFr rs = r * s;
    uint32_t rsk[8];
    to_canonical_u32(r, T->rk);
    to_canonical_u32(s, T->sk);
    to_canonical_u32(rs, rsk);
    if (P.fb_delta && P.fb_delta2) {
        T->r_delta = P.fb_delta->mul(T->rk);
        T->s_delta = P.fb_delta->mul(T->sk);
        T->rs_delta = P.fb_delta->mul(rsk);
        T->s_delta2 = P.fb_delta2->mul(T->sk);
    } else {  // a key's first proof: plain double-and-add (0.36 ms instead of 0.05; hidden under the GPU's work in the single-GPU prover)
        T->r_delta = scalar_mul(P.delta, T->rk);
        T->s_delta = scalar_mul(P.delta, T->sk);
        T->rs_delta = scalar_mul(P.delta, rsk);
        T->s_delta2 = scalar_mul(P.delta2, T->sk);
    }
    prof_host("host_tail_pre", now_ms() - t0);
}
static void tail_post(const Groth16PK& P, const TailPre& T, const uint64_t* parts, size_t n_parts, uint8_t proof_out[128]) {
    const double t0 = now_ms();
    XYZZ<HFp> m_a = XYZZ<HFp>::inf(), m_b = m_a, m_k = m_a, m_z = m_a;
    XYZZ<HFp2> m_b2 = XYZZ<HFp2>::inf();
    for (size_t i = 0; i < n_parts; i++) {
        const uint64_t* p = parts + 96 * i;
        XYZZ<HFp> t;
        XYZZ<HFp2> t2;
        memcpy(&t, p, 128); m_a.add(t);
        memcpy(&t, p + 16, 128); m_b.add(t);
        memcpy(&t, p + 32, 128); m_k.add(t);
        memcpy(&t, p + 48, 128); m_z.add(t);
        memcpy(&t2, p + 64, 256); m_b2.add(t2);
    }
    XYZZ<HFp> a_alpha = m_a, b_beta = m_b;
    a_alpha.madd(P.alpha);
    b_beta.madd(P.beta);
    XYZZ<HFp> ar = a_alpha;
    ar.add(T.r_delta);
    XYZZ<HFp2> bs = m_b2;
    bs.madd(P.beta2);
    bs.add(T.s_delta2);
    XYZZ<HFp> krs = m_k;
    krs.add(m_z);
    if (T.have_early && n_parts == 1) {
        krs.add(T.s_aalpha);
        krs.add(T.r_bbeta);
    } else {
        krs.add(scalar_mul(a_alpha.to_affine(), T.sk));
        krs.add(scalar_mul(b_beta.to_affine(), T.rk));
    }
    krs.add(T.rs_delta);
    g1_compress(ar.to_affine(), proof_out);
    g2_compress(bs.to_affine(), proof_out + 32);
    g1_compress(krs.to_affine(), proof_out + 96);
    prof_host("host_tail_post", now_ms() - t0);
}
static void finalize(const Groth16PK& P, const uint64_t* parts, size_t n_parts, const zk_fr* r_, const zk_fr* s_, uint8_t proof_out[128]) {
    TailPre T;
    tail_pre(P, r_, s_, &T);
    tail_post(P, T, parts, n_parts, proof_out);
}

int zk_bn254_groth16_msm5_dev(const void* d_a, const void* d_b, const void* d_b2, const void* d_w, size_t nw, const void* d_k, const void* d_wk,
                              size_t nk, const void* d_z, const void* d_h, size_t nz, uint64_t out_xyzz[96], void* stream) {
    if (!out_xyzz) return set_err(ZK_ERR_ARG, "null pointer");
    if ((nw && (!d_a || !d_b || !d_b2 || !d_w)) || (nk && (!d_k || !d_wk)) || (nz && (!d_z || !d_h))) return set_err(ZK_ERR_ARG, "null pointer");
    SlotsGuard<5> g;
    ZK_TRY(acquire_slots(5, g.s));
    Msm5Inputs in = {d_a, d_b, d_b2, d_w, nw, d_k, d_wk, nk, d_z, d_h, nz};
    ZK_TRY(msm5_reserve(g.s, in, 0));
    hipEvent_t ev = nullptr;
    if (stream) {  // inputs are produced on the caller's stream: gate all five streams on it
        ZK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(ev, (hipStream_t)stream));
        ZK_HIP(hipStreamWaitEvent(g.s[0]->hi(), ev, 0));
    }
    Msm5State S;
    int rc = msm5_launch_w(g.s, in, ev, &S);
    if (rc == ZK_OK) rc = msm5_launch_h(g.s, g.s[0]->hi(), in, &S);
    if (rc == ZK_OK) rc = msm5_finish(&S, out_xyzz);
    else { msm_prep_release(&S.prep_w); msm_prep_release(&S.prep_h); }
    if (ev) (void)hipEventDestroy(ev);
    return rc;
}

// A rank's five MSMs in two calls, so that the preparation of the wire scalars (which does not need h) is already running while
// the host is still driving computeH and its exchanges.
struct Msm5Session {
    SlotsGuard<5> g;
    uint64_t pk_handle = 0;
    Groth16PK P;
    Msm5Inputs in;
    Msm5State S;
};
static std::mutex g_sess_mu;
static std::map<uint64_t, Msm5Session*> g_sessions;
static uint64_t g_next_session = 1;
// a live session keeps copies of the key's device pointers: the key counts its sessions and pk_free refuses while any is open
static void pk_session_ref(uint64_t h, int delta) {
    std::lock_guard<std::mutex> lk(g_pk_mu);
    auto it = g_pks.find(h);
    if (it != g_pks.end()) it->second.sessions += delta;
}
static void session_drain(Msm5Session* ss) {
    for (int i = 0; i < 5; i++) { (void)hipStreamSynchronize(ss->g.s[i]->stream); ss->g.s[i]->sync_hi(); }
    msm_prep_release(&ss->S.prep_w);
    msm_prep_release(&ss->S.prep_h);
    for (int i = 0; i < 5; i++)
        if (ss->S.jobs[i].acc_done) { (void)hipEventDestroy(ss->S.jobs[i].acc_done); ss->S.jobs[i].acc_done = nullptr; }
}

int zk_bn254_groth16_msm5_pk_begin(uint64_t pk_handle, const void* d_w, uint64_t* session) {
    ZK_ON_ENTRY_OF(pk_handle);
    if (!session) return set_err(ZK_ERR_ARG, "null pointer");
    std::unique_ptr<Msm5Session> ss(new Msm5Session());
    ZK_TRY(lookup_pk(pk_handle, &ss->P));
    const Groth16PK& P = ss->P;
    const size_t nw = P.n_wires, nk = P.n_wires - P.n_public;
    if (nw && !d_w) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_TRY(acquire_slots(5, ss->g.s));
    ss->in = Msm5Inputs{P.d_a, P.d_b, P.d_b2, d_w, nw, P.d_k, (const char*)d_w + P.n_public * 32, nk, P.d_z, nullptr, P.nz};
    if (P.tables) {
        ss->in.tab_w = &ss->P.tab_w; ss->in.tab_h = &ss->P.tab_h;
        ss->in.t_a = P.t_a; ss->in.t_b = P.t_b; ss->in.t_b2 = P.t_b2; ss->in.t_k = P.t_k; ss->in.t_z = P.t_z;
    }
    ZK_TRY(msm5_reserve(ss->g.s, ss->in, 0));
    int rc = msm5_prepare_w(ss->g.s, ss->in, nullptr, &ss->S);
    if (rc != ZK_OK) {
        session_drain(ss.get());
        return rc;
    }
    ss->pk_handle = pk_handle;
    pk_session_ref(pk_handle, +1);
    std::lock_guard<std::mutex> lk(g_sess_mu);
    *session = hmake(g_next_session++);
    g_sessions[*session] = ss.release();
    return ZK_OK;
}

int zk_bn254_groth16_msm5_pk_abort(uint64_t session) {
    ZK_ON_ENTRY_OF(session);
    std::unique_ptr<Msm5Session> ss;
    {
        std::lock_guard<std::mutex> lk(g_sess_mu);
        auto it = g_sessions.find(session);
        if (it == g_sessions.end()) return set_err(ZK_ERR_HANDLE, "unknown msm5 session %llu", (unsigned long long)session);
        ss.reset(it->second);
        g_sessions.erase(it);
    }
    session_drain(ss.get());
    pk_session_ref(ss->pk_handle, -1);
    return ZK_OK;  // ~Msm5Session releases the five slots
}

int zk_bn254_groth16_msm5_session_stream(uint64_t session, void** stream_out) {
    ZK_ON_ENTRY_OF(session);
    if (!stream_out) return set_err(ZK_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(g_sess_mu);
    auto it = g_sessions.find(session);
    if (it == g_sessions.end()) return set_err(ZK_ERR_HANDLE, "unknown msm5 session %llu", (unsigned long long)session);
    *stream_out = (void*)it->second->g.s[0]->hi();  // the stream prove() runs computeH on: prepare(h) and Z follow on it in order
    return ZK_OK;
}

int zk_bn254_groth16_msm5_pk_end(uint64_t session, const void* d_h, uint64_t out_xyzz[96], void* stream) {
    ZK_ON_ENTRY_OF(session);
    std::unique_ptr<Msm5Session> ss;
    {
        std::lock_guard<std::mutex> lk(g_sess_mu);
        auto it = g_sessions.find(session);
        if (it == g_sessions.end()) return set_err(ZK_ERR_HANDLE, "unknown msm5 session %llu", (unsigned long long)session);
        ss.reset(it->second);
        g_sessions.erase(it);
    }
    Slot** sl = ss->g.s;
    int rc = ZK_OK;
    if (!out_xyzz || (ss->P.nz && !d_h)) rc = set_err(ZK_ERR_ARG, "null pointer");
    ss->in.d_h = d_h;
    hipEvent_t ev = nullptr;
    if (rc == ZK_OK && stream) {  // h is being produced on the caller's stream (computeH)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, (hipStream_t)stream) != hipSuccess ||
            hipStreamWaitEvent(sl[0]->hi(), ev, 0) != hipSuccess)
            rc = set_err(ZK_ERR_HIP, "event setup failed");
    }
    // like prove(): the accumulate chain starts when computeH has left the machine
    if (rc == ZK_OK) rc = msm5_accumulate_w(sl, ss->in, nullptr, &ss->S, ev);
    if (rc == ZK_OK) rc = msm5_launch_h(sl, sl[0]->hi(), ss->in, &ss->S);
    if (rc == ZK_OK) rc = msm5_finish(&ss->S, out_xyzz);
    else session_drain(ss.get());
    if (ev) (void)hipEventDestroy(ev);
    pk_session_ref(ss->pk_handle, -1);
    return rc;
}

int zk_bn254_groth16_msm5_pk(uint64_t pk_handle, const void* d_w, const void* d_h, uint64_t out_xyzz[96], void* stream) {
    ZK_ON_ENTRY_OF(pk_handle);
    uint64_t session = 0;
    ZK_TRY(zk_bn254_groth16_msm5_pk_begin(pk_handle, d_w, &session));
    return zk_bn254_groth16_msm5_pk_end(session, d_h, out_xyzz, stream);
}

int zk_bn254_groth16_finalize(uint64_t pk_handle, const uint64_t* partials, size_t n_partials, const zk_fr* r, const zk_fr* s, uint8_t proof_out[128]) {
    if (md_is_composite(pk_handle)) return md_groth16_finalize(pk_handle, partials, n_partials, r, s, proof_out);
    ZK_ON_ENTRY_OF(pk_handle);
    if (!partials || !n_partials || !r || !s || !proof_out) return set_err(ZK_ERR_ARG, "null pointer");
    Groth16PK P;
    ZK_TRY(lookup_pk(pk_handle, &P));
    ZK_TRY(pk_tables_for_tail(pk_handle, &P));
    finalize(P, partials, n_partials, r, s, proof_out);
    return ZK_OK;
}

int zk_bn254_groth16_prove(uint64_t pk_handle, const void* a, const void* b, const void* c, size_t n_constraints, const void* w, size_t n_wires,
                           const zk_fr* r_, const zk_fr* s_, int on_device, uint8_t proof_out[128]) {
    if (md_is_composite(pk_handle)) return md_groth16_prove(pk_handle, a, b, c, n_constraints, w, n_wires, r_, s_, on_device, proof_out);
    ZK_ON_ENTRY_OF(pk_handle);
    if (!r_ || !s_ || !proof_out) return set_err(ZK_ERR_ARG, "null pointer");
    Groth16PK P;
    ZK_TRY(lookup_pk(pk_handle, &P));
    const size_t N = (size_t)1 << P.log_domain, nw = P.n_wires, nk = P.n_wires - P.n_public;
    if (n_wires != nw) return set_err(ZK_ERR_LEN, "len(w) = %zu != %zu wires of the proving key", n_wires, nw);
    if ((n_constraints && (!a || !b || !c)) || (nw && !w)) return set_err(ZK_ERR_ARG, "null pointer");
    if (n_constraints > N) return set_err(ZK_ERR_ARG, "n_constraints = %zu exceeds the domain size %zu", n_constraints, N);
    SlotsGuard<5> g;
    ZK_TRY(acquire_slots(5, g.s));
    Slot* s0 = g.s[0];
    hipStream_t st = s0->hi();
    Msm5Inputs in = {P.d_a, P.d_b, P.d_b2, nullptr, nw, P.d_k, nullptr, nk, P.d_z, nullptr, P.nz};
    in.d_wk = (const char*)in.d_w + P.n_public * 32;  // placeholder geometry for the reservation; real pointers below
    if (P.tables) {
        in.tab_w = &P.tab_w; in.tab_h = &P.tab_h;
        in.t_a = P.t_a; in.t_b = P.t_b; in.t_b2 = P.t_b2; in.t_k = P.t_k; in.t_z = P.t_z;
    }
    ZK_TRY(msm5_reserve(g.s, in, 3 * N * 32 + nw * 32 + 4096));
    hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // Order on the GPU:  [computeH  ||  upload + prepare(w)]  ->  G2.B acc -> A acc -> B1 acc -> K acc -> Z acc  (reduce tails and
    // prepare(h) run underneath the accumulate kernels).  computeH goes FIRST and alone with the bandwidth-bound sort of w:
    // measured, an NTT launched underneath an accumulate kernel is starved (1.3 ms -> 8 ms) because the long-running accumulate
    // workgroups never free enough wave slots, and the Z chain then finishes late.
    hipStream_t st4 = g.s[4]->hi();
    int rc = ZK_OK;
    Fr* d_abc[3] = {nullptr, nullptr, nullptr};
    const void* src[3] = {a, b, c};
    // full-length inputs already in HBM are read in place by the first NTT pass (they stay untouched); otherwise copy / pad first
    const bool direct = on_device && n_constraints == N && N > 1;
    // Host inputs (what a cgo caller has): 128 bytes per constraint cross PCIe inside the call.  Order: w first (32 B / wire), its scalar preparation
    // and the G2.B accumulate -- none of which needs a, b, c -- run while a, b, c (96 B / constraint) are still uploading; computeH then takes the
    // machine BETWEEN the G2.B and A accumulates (under an accumulate kernel it would be starved).  ZKMI_HOST_ORDER=0 restores computeH-first.
    static const bool host_order_on = (ZK_EXP("ZKMI_HOST_ORDER", 1) != 0);
    const bool host_order = !on_device && host_order_on && nw > 0 && N > 1;
    Fr* d_w = (Fr*)s0->alloc(nw * 32 + 16);
    for (int i = 0; i < 3; i++) d_abc[i] = (Fr*)s0->alloc(N * 32);
    if (on_device) d_w = (Fr*)const_cast<void*>(w);  // wire values already in HBM are only read: no staging copy
    auto upload_abc = [&]() -> int {
        for (int i = 0; i < 3; i++) {
            if (direct) continue;
            if (n_constraints && hipMemcpyAsync(d_abc[i], src[i], n_constraints * 32, kind, st) != hipSuccess) return set_err(ZK_ERR_HIP, "hipMemcpyAsync failed");
            if (n_constraints < N && hipMemsetAsync(d_abc[i] + n_constraints, 0, (N - n_constraints) * 32, st) != hipSuccess) return set_err(ZK_ERR_HIP, "hipMemsetAsync failed");
        }
        return ZK_OK;
    };
    // h = computeH(a, b, c), left in d_abc[0] (bit-reversed order, like upstream; pk.G1.Z is stored to match)
    const Fr* in_place_src[3] = {(const Fr*)a, (const Fr*)b, (const Fr*)c};
    // ZKMI_H_STREAMS=1 (experiment switch): the transforms of b and c on the (idle) high-priority streams of slots 1 and 2, next to a's
    static const bool h_streams = ZK_EXP("ZKMI_H_STREAMS", 0) == 1;
    const hipStream_t side[2] = {g.s[1]->hi(), g.s[2]->hi()};
    hipEvent_t ev_h = nullptr;
    auto run_compute_h = [&]() -> int {
        ZK_TRY(compute_h_inplace(s0, st, d_abc[0], d_abc[1], d_abc[2], P.log_domain, direct ? in_place_src : nullptr, h_streams ? side : nullptr));
        ZK_HIP(hipEventCreateWithFlags(&ev_h, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(ev_h, st));
        return ZK_OK;
    };
    Msm5State S;
    static const bool nogate = ZK_EXP("ZKMI_NOGATE", 0) == 1;  // experiment: G2.B accumulate does not wait for computeH
    static const bool preph_first = ZK_EXP("ZKMI_PREPH_FIRST", 0) == 1;  // experiment: prepare(h) alone, before G2.B
    static const bool preph_early = ZK_EXP("ZKMI_PREPH_EARLY", 0) == 1;  // experiment: prepare(h) ENQUEUED before the w-side accumulates (no extra gating)
    if (host_order) {
        if (hipMemcpyAsync(d_w, w, nw * 32, kind, st4) != hipSuccess) rc = set_err(ZK_ERR_HIP, "hipMemcpyAsync failed");
        in.d_w = d_w;
        in.d_wk = d_w + P.n_public;
        in.d_h = d_abc[0];
        if (rc == ZK_OK) rc = msm5_prepare_w(g.s, in, nullptr, &S);
        if (rc == ZK_OK) rc = upload_abc();  // the host blocks here while the GPU sorts the digits of w
        const BetweenFn between = [&](hipEvent_t g2_done, hipEvent_t* gate_next) -> int {
            if (g2_done) ZK_HIP(hipStreamWaitEvent(st, g2_done, 0));
            ZK_TRY(run_compute_h());
            *gate_next = ev_h;
            return ZK_OK;
        };
        if (rc == ZK_OK) rc = msm5_accumulate_w(g.s, in, nullptr, &S, nullptr, &between);
        if (rc == ZK_OK) rc = msm5_launch_h(g.s, st, in, &S);
    } else {
        // Order on the GPU:  [computeH  ||  upload + prepare(w)]  ->  G2.B acc -> A acc -> B1 acc -> K acc -> Z acc  (reduce tails and
        // prepare(h) run underneath the accumulate kernels).  computeH goes FIRST and alone with the bandwidth-bound sort of w:
        // measured, an NTT launched underneath an accumulate kernel is starved (1.3 ms -> 8 ms) because the long-running accumulate
        // workgroups never free enough wave slots, and the Z chain then finishes late.
        // ZKMI_H_UNDER_G2=1 (experiment switch): prepare(w) ALONE first, then computeH started together with the G2.B accumulate (to be combined with
        // ZKMI_ACC_WG_G2=1: that kernel at one wave per SIMD leaves half of the registers and the LDS to the transforms' workgroups)
        static const bool h_under_g2 = ZK_EXP("ZKMI_H_UNDER_G2", 0) == 1;
        if (h_under_g2) {
            if (!on_device && nw && hipMemcpyAsync(d_w, w, nw * 32, kind, st4) != hipSuccess) rc = set_err(ZK_ERR_HIP, "hipMemcpyAsync failed");
            in.d_w = d_w;
            in.d_wk = d_w + P.n_public;
            in.d_h = d_abc[0];
            if (rc == ZK_OK) rc = msm5_prepare_w(g.s, in, nullptr, &S);
            if (rc == ZK_OK) rc = upload_abc();
            if (rc == ZK_OK && S.prep_w.ready && hipStreamWaitEvent(st, S.prep_w.ready, 0) != hipSuccess) rc = set_err(ZK_ERR_HIP, "hipStreamWaitEvent failed");
            if (rc == ZK_OK) rc = run_compute_h();
            if (rc == ZK_OK) rc = msm5_accumulate_w(g.s, in, nullptr, &S, nullptr);
            if (rc == ZK_OK) rc = msm5_launch_h(g.s, st, in, &S);
        } else {
        rc = upload_abc();
        if (rc == ZK_OK) rc = run_compute_h();
        if (!on_device && rc == ZK_OK && nw && hipMemcpyAsync(d_w, w, nw * 32, kind, st4) != hipSuccess) rc = set_err(ZK_ERR_HIP, "hipMemcpyAsync failed");
        in.d_w = d_w;
        in.d_wk = d_w + P.n_public;
        in.d_h = d_abc[0];
        if (preph_first) {
            if (rc == ZK_OK) rc = msm5_prepare_h(g.s, st, in, &S);
            if (rc == ZK_OK) rc = msm5_launch_w(g.s, in, nullptr, &S, S.prep_h.ready ? S.prep_h.ready : ev_h);
            if (rc == ZK_OK) rc = msm5_launch_h(g.s, st, in, &S, true);
        } else if (preph_early) {
            // host enqueue order only: prepare(h) is enqueued (on computeH's stream, behind it) BEFORE the ~40 launches of the four w-side accumulates and their
            // tails -- with witness-like wires those are short and the GPU reaches the end of computeH before the host has enqueued prepare(h) behind them
            if (rc == ZK_OK) rc = msm5_prepare_w(g.s, in, nullptr, &S);
            if (rc == ZK_OK) rc = msm5_prepare_h(g.s, st, in, &S);
            if (rc == ZK_OK) rc = msm5_accumulate_w(g.s, in, nullptr, &S, nogate ? nullptr : ev_h);
            if (rc == ZK_OK) rc = msm5_launch_h(g.s, st, in, &S, true);
        } else {
            if (rc == ZK_OK) rc = msm5_launch_w(g.s, in, nullptr, &S, nogate ? nullptr : ev_h);
            if (rc == ZK_OK) rc = msm5_launch_h(g.s, st, in, &S);
        }
        }
    }
    if (ev_h) (void)hipEventDestroy(ev_h);
    uint64_t parts[96];
    TailPre T;
    if (rc == ZK_OK) rc = pk_tables_for_tail(pk_handle, &P);
    if (rc == ZK_OK) tail_pre(P, r_, s_, &T);  // host work hidden under the GPU's
    const std::function<void(const XYZZ<HFp>&, const XYZZ<HFp>&)> early = [&](const XYZZ<HFp>& ma, const XYZZ<HFp>& mb) { tail_early(P, &T, ma, mb); };
    if (rc == ZK_OK) rc = msm5_finish(&S, parts, &early);
    else {
        for (int i = 0; i < 5; i++) { (void)hipStreamSynchronize(g.s[i]->stream); g.s[i]->sync_hi(); }
        msm_prep_release(&S.prep_w);
        msm_prep_release(&S.prep_h);
    }
    if (rc != ZK_OK) return rc;
    tail_post(P, T, parts, 1, proof_out);
    return ZK_OK;
}

}  // extern "C"
