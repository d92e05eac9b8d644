// libzkmi runtime: device init, stream slots, arenas, profiling, device-memory plumbing entry points.
#include <vector>

#include "ctx.hpp"
#include "multidev.hpp"

#include <string.h>

#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>

namespace zkmi {

thread_local std::string g_err;

int set_err(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ---- the device list.  Entries are created once and never move (contexts are referenced from slots in flight); g_n_entries only grows.
static Ctx* g_entries[MAX_ENTRIES];
static std::atomic<int> g_n_entries{0};
static std::mutex g_entries_mu;
static thread_local int t_entry = 0;
static std::atomic<bool> g_lean_start{false};  // zk_init_flags(ZK_INIT_LEAN_STREAMS)
// After a lean start the high-priority streams are not created AT ALL until somebody asks for them (zk_warm_session_streams): hi() hands out the slot's own
// stream -- the fallback that has always existed for a failed hipStreamCreateWithPriority -- so a process's first proof pays no stream creation (five streams
// at 3.5-10 ms each, serialised with every other runtime call of the start-up: the cold ProveWithPK of profiles/rnd5_h_*) for a schedule refinement worth 1-3 %.
static std::atomic<bool> g_hi_streams_wanted{true};

Prof& prof() {
    static Prof p;
    return p;
}
int n_entries() { return g_n_entries.load(std::memory_order_acquire); }
int current_entry() { return t_entry; }
Ctx& ctx() {
    const int n = n_entries();
    if (t_entry < n) return *g_entries[t_entry];
    static Ctx unbound;  // before any device was named: not ready, device 0 (ensure_init creates entry 0)
    return unbound;
}

static int init_entry(Ctx& c) {  // under c.mu
    if (c.ready) return ZK_OK;
    ZK_HIP(hipSetDevice(c.device));
    hipDeviceProp_t prop;
    ZK_HIP(hipGetDeviceProperties(&prop, c.device));
    c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int lo = 0, hi = 0;  // numerically lower = higher priority
    ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    c.prio_lo = lo;
    c.prio_hi = hi;
    for (int i = 0; i < Ctx::NSLOTS; i++) c.slots[i].owner = &c;  // the streams: at a slot's first acquisition / first hi()
    // The own streams of the first five slots (a Groth16 proof's session) exist before anything else touches the runtime: the runtime gives the first four streams
    // of a priority a hardware queue each and lets later ones -- the null stream's included -- share them, and WHICH streams share is part of the measured
    // schedule of the 2^20 proof (created after the first copies and launches instead: 9.7 -> 11.1 ms, profiles/r05_c_stream_creation_order.jsonl).  The other
    // eleven are created when first used: 3.5-10 ms each (tools/hip_start_bench.hip), and a process that makes one PLONK proof never needs them.
    // Their high-priority streams too, interleaved with them, on the process's first entry -- created later (at the first proof session or at a key's load) the
    // same proof takes 1 % longer (9.48-9.53 against 9.58-9.64 ms, profiles/r05_z_hi_streams_at_init.jsonl) -- unless the caller asked for a lean start
    // (zk_init_flags: the export shim, whose one PLONK proof uses three streams and counts every 10 ms of its cold call).
    {
        static const int knob = ZK_EXP("ZKMI_INIT_STREAMS", -1);  // experiment: 0 none; 1 all sixteen, interleaved (rounds 1-3); 2 slots 0-4 both; 3 the eight own; 4 five own
        const int at_init = knob >= 0 ? knob : (g_lean_start.load() || c.entry != 0) ? 4 : 2;
        const int ns = at_init == 1 || at_init == 3 ? Ctx::NSLOTS : (at_init == 2 || at_init == 4) ? 5 : 0;
        // (Round 5 measured a lean start with two or three own streams, slots 2-4 borrowing them until the second proof: the runtime's start 24 / 15 ms shorter,
        // the key reader and the circuit's upload 10-20 ms longer on the shared streams, and every WARM proof 0.9 ms slower for streams created late --
        // profiles/rnd5_x_lean_stream_count.txt.  Five it stays.)
        for (int i = 0; i < ns; i++) {
            ZK_HIP(hipStreamCreateWithPriority(&c.slots[i].stream, hipStreamNonBlocking, lo));
            if (at_init <= 2) ZK_HIP(hipStreamCreateWithPriority(&c.slots[i].stream_hi_, hipStreamNonBlocking, hi));
        }
        // experiment: slot b runs on slot a's stream (explicit sharing instead of the runtime's hardware-queue sharing): ZKMI_ALIAS_LO / _HI = 10 * a + b + 100
        static const int alias_lo = ZK_EXP("ZKMI_ALIAS_LO", 0), alias_hi = ZK_EXP("ZKMI_ALIAS_HI", 0);
        if (alias_lo >= 100 && ns >= 5) c.slots[(alias_lo - 100) % 10].stream = c.slots[(alias_lo - 100) / 10].stream;
        if (alias_hi >= 100) {
            for (int i = 0; i < 5; i++)
                if (!c.slots[i].stream_hi_) ZK_HIP(hipStreamCreateWithPriority(&c.slots[i].stream_hi_, hipStreamNonBlocking, hi));
            c.slots[(alias_hi - 100) % 10].stream_hi_ = c.slots[(alias_hi - 100) / 10].stream_hi_;
        }
    }
    c.ready = true;
    return ZK_OK;
}

int init_devices(const int* devices, int n) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_err(ZK_ERR_NO_DEVICE, "no HIP device visible (hipGetDeviceCount: %s); libzkmi has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    std::vector<int> all;
    if (n <= 0 || !devices) {
        for (int d = 0; d < count; d++) all.push_back(d);
        devices = all.data();
        n = count;
    }
    if (n > MAX_ENTRIES) return set_err(ZK_ERR_ARG, "%d device entries, at most %d", n, MAX_ENTRIES);
    for (int i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= count) return set_err(ZK_ERR_ARG, "device %d out of range (0..%d)", devices[i], count - 1);
    std::lock_guard<std::mutex> lk(g_entries_mu);
    const int have = n_entries();
    for (int i = 0; i < have && i < n; i++)
        if (g_entries[i]->device != devices[i]) return set_err(ZK_ERR_ARG, "entry %d is already bound to device %d", i, g_entries[i]->device);
    for (int i = have; i < n; i++) {
        Ctx* c = new Ctx();
        c->entry = i;
        c->device = devices[i];
        g_entries[i] = c;
        g_n_entries.store(i + 1, std::memory_order_release);
    }
    // peers see each other's memory where the hardware allows it (xGMI); a refusal only means copies are staged by the runtime
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (devices[i] != devices[j]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[i]);
                    (void)hipDeviceEnablePeerAccess(devices[j], 0);
                    (void)hipGetLastError();  // "already enabled" is not an error worth keeping
                }
            }
    (void)hipSetDevice(g_entries[t_entry < n_entries() ? t_entry : 0]->device);
    return ZK_OK;
}

CtxScope::CtxScope(int entry) : prev(t_entry) {
    if (entry < 0 || entry >= n_entries()) {
        if (entry == 0) { rc = ensure_init(); return; }  // handles made before any explicit list: entry 0
        rc = set_err(ZK_ERR_HANDLE, "device entry %d does not exist (%d entries)", entry, n_entries());
        return;
    }
    t_entry = entry;
    rc = ensure_init();
}
CtxScope::~CtxScope() {
    if (t_entry != prev) {
        t_entry = prev;
        if (prev < n_entries()) (void)hipSetDevice(g_entries[prev]->device);
    }
}

// Experiment: a pair of CU-masked streams per slot.  ZKMI_CU_SPLIT=k gives the scalar preparation (digits, sort, plan: bandwidth-bound, needs wave slots to
// make progress) k CUs of its own -- bits i with i % (256 / k) == 0 of the CU mask -- and the accumulate kernels the others.
int masked_streams(Slot* s) {
    static const int k = ZK_EXP("ZKMI_CU_SPLIT", 0);
    if (k <= 0 || s->stream_prep) return ZK_OK;
    const int ncu = ctx().num_cus, words = (ncu + 31) / 32;
    static const int mode = ZK_EXP("ZKMI_CU_SPLIT_MODE", 0);  // 0: every (ncu/k)-th bit; 1: the first k bits
    std::vector<uint32_t> mp(words, 0), ma(words, 0);
    const int step = ncu / k > 0 ? ncu / k : 1;
    for (int i = 0; i < ncu; i++) {
        const bool prep = mode == 1 ? i < k : (i % step == 0 && i / step < k);
        (prep ? mp : ma)[i >> 5] |= 1u << (i & 31);
    }
    ZK_HIP(hipExtStreamCreateWithCUMask(&s->stream_prep, (uint32_t)words, mp.data()));
    ZK_HIP(hipExtStreamCreateWithCUMask(&s->stream_acc, (uint32_t)words, ma.data()));
    return ZK_OK;
}

int ensure_init() {
    if (n_entries() == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        const int dev0 = 0;
        ZK_TRY(init_devices(&dev0, 1));
        Ctx& c0 = *g_entries[0];
        int rc;
        {
            std::lock_guard<std::mutex> lk(c0.mu);
            rc = init_entry(c0);
        }
        prof_host("export.hip_init", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());  // runtime start + streams, once per process
        ZK_TRY(rc);
    }
    if (t_entry >= n_entries()) return set_err(ZK_ERR_ARG, "this thread is on device entry %d, %d entries exist", t_entry, n_entries());
    Ctx& c = *g_entries[t_entry];
    if (!c.ready) {
        std::lock_guard<std::mutex> lk(c.mu);
        ZK_TRY(init_entry(c));
    }
    // other host threads (goroutine-backed OS threads, the per-entry workers of multidev.hip) must also target the entry's device
    hipError_t e = hipSetDevice(c.device);
    if (e != hipSuccess) return set_err(ZK_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    return ZK_OK;
}

// A caller that already holds slots (an msm5 session pins 5 of the 8) and asks for more than are left would wait forever:
// waiting is bounded (ZKMI_SLOT_TIMEOUT_S, default 120 s) and ends in ZK_ERR_BUSY instead of a silent hang.
static double slot_timeout_s() {
    static const double t = (double)zk_env_bounded("ZKMI_SLOT_TIMEOUT_S", 120, 1, 86400);  // whole seconds, 1 s .. 1 day
    return t;
}
static bool slot_wait_expired(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > slot_timeout_s();
}

// the slot's own stream, created on the entry's device at the slot's first acquisition (under the entry's mutex: the slot is not visible to anybody else yet)
static int slot_stream(Slot* s) {
    if (s->stream) return ZK_OK;
    hipError_t e = hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, s->owner->prio_lo);
    if (e != hipSuccess) {
        s->stream = nullptr;
        s->busy = false;
        return set_err(ZK_ERR_HIP, "hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    }
    return ZK_OK;
}
hipStream_t Slot::hi() {
    std::lock_guard<std::mutex> lk(owner->mu);
    return hi_locked();
}
hipStream_t Slot::hi_locked() {  // under owner->mu
    // The answer is STICKY for one acquisition of the slot: zk_warm_session_streams() may flip g_hi_streams_wanted (another thread, a key's second proof in a
    // lean process) while a proof in flight holds this slot -- if that proof's earlier hi() calls got the slot's own stream, its later ones must too, or work it
    // ordered by stream (an upload followed by prepare(w)) would be split over two unordered streams.  begin_acquisition() clears the latch.
    if (hi_latched_) return hi_latched_;
    if (!stream_hi_ && !g_hi_streams_wanted.load()) return hi_latched_ = stream;
    if (!stream_hi_) {
        int cur = owner->device;
        (void)hipGetDevice(&cur);
        if (cur != owner->device) (void)hipSetDevice(owner->device);
        hipStream_t st = nullptr;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, owner->prio_hi) != hipSuccess) st = stream;  // no second stream: the chain runs on the slot's own
        stream_hi_ = st;
        if (cur != owner->device) (void)hipSetDevice(cur);
    }
    return hi_latched_ = stream_hi_;
}

int acquire_slot(Slot** out) {
    ZK_TRY(ensure_init());
    Ctx& c = ctx();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        if ((spin & 1023) == 1023 && slot_wait_expired(t0))
            return set_err(ZK_ERR_BUSY, "no stream slot became free within %.0f s (an unfinished msm5 session? call zk_bn254_groth16_msm5_pk_abort)", slot_timeout_s());
        {
            std::lock_guard<std::mutex> lk(c.mu);
            for (int i = 0; i < Ctx::NSLOTS; i++)
                if (!c.slots[i].busy) {
                    c.slots[i].busy = true;
                    c.slots[i].begin_acquisition();
                    *out = &c.slots[i];
                    return slot_stream(&c.slots[i]);
                }
        }
        std::this_thread::yield();
    }
}

int acquire_slots(int k, Slot** out) {
    ZK_TRY(ensure_init());
    Ctx& c = ctx();
    if (k > Ctx::NSLOTS) return set_err(ZK_ERR_ARG, "asked for %d stream slots, only %d exist", k, Ctx::NSLOTS);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        if ((spin & 1023) == 1023 && slot_wait_expired(t0))
            return set_err(ZK_ERR_BUSY, "%d stream slots did not become free within %.0f s (an unfinished msm5 session? call zk_bn254_groth16_msm5_pk_abort)", k,
                           slot_timeout_s());
        {
            std::lock_guard<std::mutex> lk(c.mu);
            int nfree = 0;
            for (int i = 0; i < Ctx::NSLOTS; i++) nfree += !c.slots[i].busy;
            if (nfree >= k) {
                int got = 0;
                for (int i = 0; i < Ctx::NSLOTS && got < k; i++)
                    if (!c.slots[i].busy) {
                        c.slots[i].busy = true;
                        c.slots[i].begin_acquisition();
                        out[got++] = &c.slots[i];
                    }
                for (int i = 0; i < k; i++) {
                    const int rc = slot_stream(out[i]);
                    if (rc != ZK_OK) {
                        for (int j = 0; j < k; j++) out[j]->busy = false;
                        return rc;
                    }
                }
                // A session of several slots (a Groth16 proof's five) uses their high-priority streams too, and WHICH of them share a hardware queue is part of
                // the measured schedule: the runtime gives the first four streams of a priority a hardware queue each and lets the later ones share, in order of
                // creation (tools/hip_start_bench.hip: 10 ms for streams 1-8, 3.5 ms after).  Created here in slot order, as when all sixteen streams were created
                // with the entry, and not in order of first use (init_4 against init_2 in profiles/r05_c_stream_creation_order.jsonl: 9.80 against 9.65 ms).
                static const int hi_order = ZK_EXP("ZKMI_HI_ORDER", 1234);  // experiment: creation order of the five as decimal digits (01234 = slot order)
                if (hi_order != 1234 && k == 5)
                    for (int i = 0, div = 10000; i < 5; i++, div /= 10) out[(hi_order / div) % 10 % 5]->hi_locked();
                for (int i = 0; i < k; i++) out[i]->hi_locked();
                return ZK_OK;
            }
        }
        std::this_thread::yield();
    }
}

// The streams a first proof will take, created ahead of time (7-14 ms each, the process's first one 40-160 ms: tools/hip_start_bench.hip) -- for a caller that
// has something else to do meanwhile (the export shim reads srs.hex).  n slots of the calling thread's entry, at most all of them.
// Process-wide start-up choices, read when a device entry is first used (call it before anything that touches a device).  ZK_INIT_LEAN_STREAMS: create only the
// streams every caller needs with the entry and the others on first use -- for a process that makes one proof and exits.
extern "C" int zk_init_flags(uint32_t flags) {
    if (flags & ~(uint32_t)ZK_INIT_LEAN_STREAMS) return set_err(ZK_ERR_ARG, "unknown start-up flags 0x%x", flags);
    g_lean_start.store((flags & ZK_INIT_LEAN_STREAMS) != 0);
    g_hi_streams_wanted.store((flags & ZK_INIT_LEAN_STREAMS) == 0);
    return ZK_OK;
}
extern "C" int zk_warm_streams(int n) {
    ZK_TRY(ensure_init());
    if (n < 1) return ZK_OK;
    if (n > Ctx::NSLOTS) n = Ctx::NSLOTS;
    Slot* sl[Ctx::NSLOTS];
    int got = 0, rc = ZK_OK;
    for (; got < n && rc == ZK_OK; got++) rc = acquire_slot(&sl[got]);  // one by one: the slots' own streams only (acquire_slots is a proof session's)
    if (rc != ZK_OK) got--;
    for (int i = 0; i < got; i++) release_slot(sl[i]);
    return rc;
}

// The five slots of a Groth16 proof session with their high-priority streams (acquire_slots creates them: 3.5-10 ms each), ahead of the first proof: for a caller
// that has other start-up work in flight -- the export shim's first ProveWithPK runs this beside the key's decoding.
extern "C" int zk_warm_session_streams(void) {
    g_hi_streams_wanted.store(true);  // (after a lean start: from now on hi() creates them)
    SlotsGuard<5> g;
    return acquire_slots(5, g.s);
}

// ------------------------------------------------------------------------------------------------ background work
// One worker thread per process for work that pays off only if the process goes on proving and must not sit on a call's critical path: the window tables of a
// key (or SRS) that proves again -- 113 ms at 2^20 constraints -- and the high-priority streams a lean start withheld -- 39 ms.  The call that notices the
// second proof queues the job and proves WITHOUT the tables; whoever comes after the build finds them (both builds publish their result under the registry's
// mutex, and running multi-exps keep the geometry they started with).  A process that exits meanwhile cancels: exit() runs bg_stop_at_exit (registered AFTER the
// HIP runtime's own handlers, so it runs before them), the job in flight stops at its next check, the thread is joined.
namespace {
struct Background {
    std::mutex mu;
    std::condition_variable cv, idle_cv;
    std::deque<std::function<void()>> q;
    std::thread th;
    bool started = false, stop = false;
    int running = 0;
};
Background& bg() {
    static Background* b = new Background();  // (never destroyed: the worker may outlive static destruction order otherwise)
    return *b;
}
std::atomic<bool> g_bg_cancel{false};
// A background job starts when no call is in flight -- no export holds the library (zk_background_hold) and no stream slot is taken -- or after g_bg_yield_ms
// whatever is running (a process that proves back to back still gets its tables).  The call that queued the job is usually still proving, and both the stream
// creation (runtime locks, host CPUs: the GPU boxes give a process 16) and the table kernels (the machine) would cost it: at 2^20 the second ProveWithPK took
// 31.9 ms beside them against 21 alone.
std::atomic<int> g_bg_holds{0};
std::atomic<int> g_bg_yield_ms{250};
void bg_yield_to_calls_in_flight() {
    const auto t0 = std::chrono::steady_clock::now();
    const auto cap = std::chrono::milliseconds(g_bg_yield_ms.load());
    for (int quiet = 0;;) {
        if (g_bg_cancel.load()) return;
        bool busy = g_bg_holds.load() > 0;
        const int n = n_entries();
        for (int e = 0; e < n && !busy; e++) {
            Ctx* c = g_entries[e];
            if (!c || !c->ready) continue;
            std::lock_guard<std::mutex> lk(c->mu);
            for (int i = 0; i < Ctx::NSLOTS; i++) busy = busy || c->slots[i].busy;
        }
        quiet = busy ? 0 : quiet + 1;
        if (quiet >= 4 || std::chrono::steady_clock::now() - t0 > cap) return;  // four looks in a row (1 ms): the gap between two phases of one call does not count
        std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
}
void bg_loop() {
    Background& b = bg();
    std::unique_lock<std::mutex> lk(b.mu);
    for (;;) {
        b.cv.wait(lk, [&] { return b.stop || !b.q.empty(); });
        if (b.stop) return;
        std::function<void()> job = std::move(b.q.front());
        b.q.pop_front();
        b.running++;
        lk.unlock();
        bg_yield_to_calls_in_flight();
        job();
        lk.lock();
        b.running--;
        if (b.q.empty() && !b.running) b.idle_cv.notify_all();
    }
}
void bg_stop_at_exit() {
    Background& b = bg();
    g_bg_cancel.store(true);
    {
        std::lock_guard<std::mutex> lk(b.mu);
        b.stop = true;
        b.q.clear();
    }
    b.cv.notify_all();
    if (b.th.joinable()) b.th.join();
}
}  // namespace
bool bg_cancelled() { return g_bg_cancel.load(std::memory_order_relaxed); }
void bg_submit(std::function<void()> job) {
    Background& b = bg();
    {
        std::lock_guard<std::mutex> lk(b.mu);
        if (b.stop) return;
        if (!b.started) {
            b.started = true;
            b.th = std::thread(bg_loop);
            (void)atexit(bg_stop_at_exit);
        }
        b.q.push_back(std::move(job));
    }
    b.cv.notify_one();
}
// A caller that is about to make (or is inside) a call of several phases says so: +1 on entry, -1 on exit.  Background jobs wait for the count to reach zero
// (at most zk_background_set_yield_ms, default 250 ms).  The export shim brackets every export with it; the library's own export entry points do too.
extern "C" void zk_background_hold(int delta) { g_bg_holds.fetch_add(delta); }
// How long a background job waits for calls in flight before it starts anyway (0: at once).  A tuning knob for long-lived servers; it changes no result.
extern "C" int zk_background_set_yield_ms(int ms) {
    if (ms < 0 || ms > 60000) return set_err(ZK_ERR_ARG, "yield of %d ms outside [0, 60000]", ms);
    g_bg_yield_ms.store(ms);
    return ZK_OK;
}
// 1 when no background job is queued or running (waits up to timeout_ms for that; < 0 = as long as it takes), else 0.  For callers that want the tables before
// they measure or compare (tests, bench.py) -- the product never waits.
extern "C" int zk_background_wait(int timeout_ms) {
    Background& b = bg();
    std::unique_lock<std::mutex> lk(b.mu);
    auto idle = [&] { return b.q.empty() && !b.running; };
    if (timeout_ms < 0) { b.idle_cv.wait(lk, idle); return 1; }
    return b.idle_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), idle) ? 1 : 0;
}

// The high-priority streams of the five slots a proof session takes first, created ONE SLOT AT A TIME from the background: each slot is held only while its
// stream is created (3.5-10 ms) and the creation itself runs outside the entry's mutex, so a proof that arrives meanwhile still finds the five slots it needs
// (zk_warm_session_streams takes all five at once for 39 ms).  hi() keeps answering a slot's own stream until all five exist.
static int warm_hi_streams_one_by_one() {
    ZK_TRY(ensure_init());
    Ctx& c = ctx();
    for (int i = 0; i < 5 && i < Ctx::NSLOTS; i++) {
        Slot* s = &c.slots[i];
        for (;;) {
            if (bg_cancelled()) return ZK_ERR_BUSY;
            {
                std::lock_guard<std::mutex> lk(c.mu);
                if (!s->busy) {
                    s->busy = true;
                    s->begin_acquisition();
                    break;
                }
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        int rc = slot_stream(s);  // (clears busy itself when it fails)
        if (rc != ZK_OK) return rc;
        hipStream_t st = nullptr;
        if (!s->stream_hi_ && hipStreamCreateWithPriority(&st, hipStreamNonBlocking, c.prio_hi) != hipSuccess) st = nullptr;
        std::lock_guard<std::mutex> lk(c.mu);
        if (st) s->stream_hi_ = st;
        s->busy = false;
    }
    g_hi_streams_wanted.store(true);
    return ZK_OK;
}
extern "C" int zk_warm_session_streams_background(void) {
    ZK_TRY(ensure_init());
    const int entry = current_entry();
    bg_submit([entry] {
        CtxScope sc(entry);
        const auto t0 = std::chrono::steady_clock::now();
        if (sc.rc == ZK_OK) (void)warm_hi_streams_one_by_one();
        prof_host("export.session_streams", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    });
    return ZK_OK;
}

void release_slot(Slot* s) {
    std::lock_guard<std::mutex> lk((s->owner ? *s->owner : ctx()).mu);  // whatever entry the releasing thread is on
    s->busy = false;
}

int Slot::reserve(size_t bytes) {
    bytes = align_up(bytes + 4096, 1 << 20);
    if (bytes <= arena_cap) return ZK_OK;
    if (arena) {
        ZK_HIP(hipStreamSynchronize(stream));
        ZK_HIP(hipFree(arena));
        arena = nullptr;
        arena_cap = 0;
    }
    hipError_t e = hipMalloc((void**)&arena, bytes);
    if (e != hipSuccess) return set_err(ZK_ERR_HIP, "hipMalloc(%zu bytes of workspace): %s", bytes, hipGetErrorString(e));
    arena_cap = bytes;
    arena_off = 0;
    return ZK_OK;
}

void* Slot::alloc(size_t bytes) {
    size_t off = align_up(arena_off, 256);
    if (off + bytes > arena_cap) return nullptr;
    arena_off = off + bytes;
    return arena + off;
}

int Slot::pinned_reserve(size_t bytes) {
    if (bytes <= pinned_cap) return ZK_OK;
    if (pinned) ZK_HIP(hipHostFree(pinned));
    pinned = nullptr;
    pinned_cap = 0;
    ZK_HIP(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    pinned_cap = bytes;
    return ZK_OK;
}

void prof_begin(Slot* s, hipStream_t st, const char* name) {
    Slot::Pending p;
    p.name = name;
    for (hipEvent_t* e : {&p.e0, &p.e1}) {
        if (!s->free_events.empty()) {
            *e = s->free_events.back();
            s->free_events.pop_back();
        } else {
            (void)hipEventCreate(e);
        }
    }
    (void)hipEventRecord(p.e0, st);  // profiling only: a failed record shows up as a failed elapsed-time query when the profile is read
    s->pending.push_back(p);
}
void prof_end(Slot* s, hipStream_t st) { (void)hipEventRecord(s->pending.back().e1, st); }
void prof_host(const char* name, double ms) {
    Prof& c = prof();
    if (!c.profiling) return;
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.prof.find(name);
    if (it == c.prof.end()) {
        c.prof_names.push_back(name);
        it = c.prof.emplace(name, ProfEntry{}).first;
    }
    it->second.launches++;
    it->second.total_ms += ms;
}

// Folds the event pairs whose kernels have completed into the profile; pairs still in flight (an asynchronous call on a
// caller's stream returned without synchronising) stay attached to the slot until a later fold.
static void fold_pending(Slot* s) {
    if (s->pending.empty()) return;
    Prof& c = prof();
    std::lock_guard<std::mutex> lk(c.mu);
    std::vector<Slot::Pending> keep;
    for (auto& p : s->pending) {
        if (hipEventQuery(p.e1) != hipSuccess) {
            keep.push_back(p);
            continue;
        }
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            auto it = c.prof.find(p.name);
            if (it == c.prof.end()) {
                c.prof_names.push_back(p.name);
                it = c.prof.emplace(p.name, ProfEntry{}).first;
            }
            it->second.launches++;
            it->second.total_ms += ms;
        }
        s->free_events.push_back(p.e0);
        s->free_events.push_back(p.e1);
    }
    s->pending.swap(keep);
}

int slot_sync(Slot* s, hipStream_t st) {
    ZK_HIP(hipStreamSynchronize(st));
    fold_pending(s);
    return ZK_OK;
}

// Pageable host memory -> HBM for the big one-off uploads of a process's first call (a key text of 0.37 GB, a constraint system of 0.2 GB).  hipMemcpy pins the
// caller's pages on first touch: 14.5 GB/s for a text that was just read, a third of that when two threads upload at once (tools/h2d_bench.hip,
// profiles/rnd5_p_h2d_bench.jsonl: the cold ProveWithPK at 2^20 spent 90-108 ms on 0.53 GB).  Here four threads copy 8 MB pieces into a ring of pinned buffers
// and enqueue them on `st` themselves -- 24-25 GB/s, no pinning of the caller's pages -- one upload at a time per device entry.  Returns when `src` has been read
// and every piece is enqueued: like hipMemcpyAsync from pageable memory, the data is in place in stream order.  Copies below 1 MB take the runtime's own path.
int h2d_big(void* dst, const void* src, size_t bytes, hipStream_t st) {
    constexpr size_t PIECE = (size_t)8 << 20;
    constexpr int T = 4;
    if (bytes < ((size_t)1 << 20)) {
        if (bytes) ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return ZK_OK;
    }
    struct Ring {
        std::mutex mu;
        char* buf[T] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev[T] = {nullptr, nullptr, nullptr, nullptr};
        bool in_flight[T] = {false, false, false, false};
    };
    static Ring* rings = new Ring[MAX_ENTRIES];  // (never destroyed: the runtime may be gone before a static destructor would free into it)
    const int entry = current_entry();
    Ring& R = rings[entry];
    const auto t_in = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(R.mu);
    const auto t_lock = std::chrono::steady_clock::now();
    for (int t = 0; t < T; t++)
        if (!R.buf[t]) {
            ZK_HIP(hipHostMalloc((void**)&R.buf[t], PIECE, hipHostMallocPortable));
            ZK_HIP(hipEventCreateWithFlags(&R.ev[t], hipEventDisableTiming));
        }
    const auto t_ring = std::chrono::steady_clock::now();
    const size_t npieces = (bytes + PIECE - 1) / PIECE;
    hipError_t err[T] = {hipSuccess, hipSuccess, hipSuccess, hipSuccess};
    int scope_rc[T] = {ZK_OK, ZK_OK, ZK_OK, ZK_OK};
    std::string scope_msg[T];
    auto work = [&](int t) {
        CtxScope sc(entry);
        if (sc.rc != ZK_OK) { scope_rc[t] = sc.rc; scope_msg[t] = g_err; return; }  // (g_err is thread-local: the text travels with the code)
        for (size_t p = (size_t)t; p < npieces; p += T) {
            if (R.in_flight[t] && (err[t] = hipEventSynchronize(R.ev[t])) != hipSuccess) return;  // this buffer's previous piece has left
            const size_t from = p * PIECE, len = from + PIECE <= bytes ? PIECE : bytes - from;
            memcpy(R.buf[t], (const char*)src + from, len);
            if ((err[t] = hipMemcpyAsync((char*)dst + from, R.buf[t], len, hipMemcpyHostToDevice, st)) != hipSuccess) return;
            if ((err[t] = hipEventRecord(R.ev[t], st)) != hipSuccess) return;
            R.in_flight[t] = true;
        }
    };
    std::thread team[T - 1];
    const int nt = npieces < (size_t)T ? (int)npieces : T;  // (a few MB: one piece, this thread alone)
    for (int t = 1; t < nt; t++) team[t - 1] = std::thread(work, t);
    work(0);
    for (int t = 1; t < nt; t++) team[t - 1].join();
    {
        const auto t_out = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        prof_host("export.h2d_ring_wait", ms(t_in, t_lock));
        prof_host("export.h2d_ring_make", ms(t_lock, t_ring));
        prof_host("export.h2d_ring_copy", ms(t_ring, t_out));
    }
    for (int t = 0; t < T; t++) {
        if (scope_rc[t] != ZK_OK) return set_err(scope_rc[t], "staged upload: %s", scope_msg[t].c_str());  // the worker's message, on the CALLER's thread
        if (err[t] != hipSuccess) return set_err(ZK_ERR_HIP, "staged upload: %s", hipGetErrorString(err[t]));
    }
    return ZK_OK;
}

void fold_all_slots() {
    for (int e = 0; e < n_entries(); e++) {
        Ctx& c = *g_entries[e];
        if (!c.ready) continue;
        CtxScope sc(e);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < Ctx::NSLOTS; i++) fold_pending(&c.slots[i]);
    }
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zk_init(int device) {
    if (n_entries() > 0) return g_entries[0]->device == device ? ensure_init() : set_err(ZK_ERR_ARG, "already bound to device %d", g_entries[0]->device);
    ZK_TRY(init_devices(&device, 1));
    return ensure_init();
}
// The process's device list: one entry per listed HIP device, in order (n == 0 or devices == NULL: every visible device).  A device may be listed more
// than once -- each listing is an entry of its own with its own streams and workspaces.  Calling it again may only extend the list.
int zk_init_devices(const int* devices, size_t n) {
    const bool first = n_entries() == 0;
    const auto t0 = std::chrono::steady_clock::now();
    ZK_TRY(init_devices(devices, (int)n));
    const int e = n_entries();
    md_set_default_mask(e >= 32 ? 0xffffffffu : ((1u << e) - 1u));  // calls without a device_mask of their own spread over every entry from here on
    const int rc = ensure_init();
    if (first) prof_host("export.hip_init", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());  // runtime start + streams, once per process
    return rc;
}
int zk_device_entries(int* devices_out, size_t cap) {
    const int n = n_entries();
    for (int i = 0; i < n && (size_t)i < cap && devices_out; i++) devices_out[i] = g_entries[i]->device;
    return n;
}
// The calling THREAD's entry for the single-device entry points that take no handle (zk_dev_alloc, zk_bn254_ntt_dev, ...); entry 0 by default.
int zk_set_entry(int entry) {
    if (entry < 0 || entry >= (n_entries() ? n_entries() : 1)) return set_err(ZK_ERR_ARG, "device entry %d does not exist (%d entries)", entry, n_entries());
    t_entry = entry;
    return ensure_init();
}

const char* zk_last_error(void) { return g_err.c_str(); }
const char* zk_version(void) { return "libzkmi 0.1 (gfx950; BN254 G1/G2 MSM + Fr NTT + Groth16 prove)"; }

int zk_dev_alloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return set_err(ZK_ERR_ARG, "null out pointer");
    ZK_TRY(ensure_init());
    ZK_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return ZK_OK;
}
int zk_dev_free(void* d_ptr) {
    ZK_TRY(ensure_init());
    ZK_HIP(hipFree(d_ptr));
    return ZK_OK;
}
int zk_dev_h2d(void* d_dst, const void* h_src, size_t bytes) {
    ZK_TRY(ensure_init());
    ZK_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ZK_OK;
}
int zk_dev_d2h(void* h_dst, const void* d_src, size_t bytes) {
    ZK_TRY(ensure_init());
    ZK_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ZK_OK;
}
int zk_dev_sync(void) {
    ZK_TRY(ensure_init());
    ZK_HIP(hipDeviceSynchronize());
    return ZK_OK;
}

int zk_profile_enable(int on) {
    prof().profiling = on != 0;
    return ZK_OK;
}
int zk_profile_host(const char* name, double ms) {
    if (!name) return set_err(ZK_ERR_ARG, "null pointer");
    prof_host(name, ms);
    return ZK_OK;
}
int zk_profile_reset(void) {
    Prof& c = prof();
    std::lock_guard<std::mutex> lk(c.mu);
    c.prof.clear();
    c.prof_names.clear();
    return ZK_OK;
}
int zk_profile_count(void) {
    fold_all_slots();  // asynchronous calls may have left event pairs in flight
    return (int)prof().prof_names.size();
}
int zk_profile_get(int idx, char* name_out, size_t name_cap, uint64_t* launches, double* total_ms) {
    Prof& c = prof();
    std::lock_guard<std::mutex> lk(c.mu);
    if (idx < 0 || idx >= (int)c.prof_names.size()) return set_err(ZK_ERR_ARG, "profile index out of range");
    const std::string& n = c.prof_names[idx];
    if (name_out && name_cap) {
        strncpy(name_out, n.c_str(), name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    const ProfEntry& e = c.prof[n];
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.total_ms;
    return ZK_OK;
}

}  // extern "C"
