// Runtime context of libzkmi: device ENTRIES (one context per entry of the process's device list: stream slots with bump-allocated HBM workspaces),
// per-kernel hipEvent timing, thread-local error text.  One process drives any number of GPUs: zk_init_devices(list) creates one entry per listed
// device -- the same device may be listed several times (virtual devices: that is how the multi-device paths are tested on a one-GPU box) -- every
// thread works on ITS current entry (entry 0 unless inside a CtxScope / after zk_set_entry), every resident object (bases, keys, sessions) carries
// the entry it lives on in the top byte of its handle, and csrc/multidev.hip runs the range-sharded phases over the entries from host threads of its own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zkmi.h"

// ---- environment variables ---------------------------------------------------------------------------------------------------------
// The shipped library is loaded into the prover's process (nargo, through libgnark_backend.so): nothing a stray environment variable says may change a
// result or crash it.  It reads exactly two variables, both validated and neither able to affect a result: ZKMI_TABLE_CAP_GB (groth16.hip: upper bound of
// the window tables of a resident key) and ZKMI_SLOT_TIMEOUT_S (ctx.hip).  Every A/B switch of the measurements in DESIGN.md is a ZK_EXP(name, default):
// the default, as a constant, unless the library is built with -DZKMI_EXPERIMENTS (`make EXPERIMENTS=1` -> libzkmi_exp.so, never shipped), where the
// variable is read.  tests/test_cabi_cpu.py asserts that libzkmi.so contains no other ZKMI_* string.
#ifdef ZKMI_EXPERIMENTS
#include <stdlib.h>
static inline long zk_exp_env(const char* name, long dflt) {
    const char* v = getenv(name);
    return v && *v ? strtol(v, nullptr, 0) : dflt;
}
#define ZK_EXP(name, dflt) zk_exp_env(name, (long)(dflt))
#else
#define ZK_EXP(name, dflt) (dflt)
#endif
// Wave priority of the short kernels (scalar preparation, transforms, reduction tails).  A SIMD arbitrates VALU issue by priority, then AGE
// (MI355X_MICROARCH.md "Two waves per SIMD"): next to the long-lived waves of an accumulate kernel -- always older -- a young wave only gets the issue
// slots they leave -- tools/corun_bench.hip: at equal priority a sort makes NO progress under a compute kernel, at any occupancy.  -DZKMI_PRIO_HI=n (1..3)
// raises them above the accumulate kernels (priority 0); 0 compiles the call away.  The shipped build uses 3 here and 1 for the transforms (Makefile PRIO):
// priority alone does not make a short kernel fast next to an accumulate kernel at full occupancy (it also needs free registers: 6-7 x its solo time),
// but it keeps it from starving, and the memory-bound preparation gets through computeH's phase ahead of the transforms.
#ifndef ZKMI_PRIO_HI
#define ZKMI_PRIO_HI 0
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void prio_hi() {
#if ZKMI_PRIO_HI
    __builtin_amdgcn_s_setprio(ZKMI_PRIO_HI);
#endif
}
// the transforms: above the accumulate kernels, below the memory-bound preparation kernels they share computeH's phase with (-DZKMI_PRIO_MID=n)
#ifndef ZKMI_PRIO_MID
#define ZKMI_PRIO_MID ZKMI_PRIO_HI
#endif
__device__ __forceinline__ void prio_mid() {
#if ZKMI_PRIO_MID
    __builtin_amdgcn_s_setprio(ZKMI_PRIO_MID);
#endif
}
#endif
// a validated setting: an integer in [lo, hi], anything else (unset, not a number, out of range) is the default
static inline long zk_env_bounded(const char* name, long dflt, long lo, long hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* e = nullptr;
    const long x = strtol(v, &e, 10);
    return (e && *e == 0 && x >= lo && x <= hi) ? x : dflt;
}

namespace zkmi {

extern thread_local std::string g_err;
int set_err(int code, const char* fmt, ...);

#define ZK_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return zkmi::set_err(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define ZK_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != ZK_OK) return _rc; \
    } while (0)

struct ProfEntry {
    uint64_t launches = 0;
    double total_ms = 0;
};

// A stream slot: one HIP stream + one growing HBM arena + pinned host staging + pending profile events.
struct Slot {
    // Most streams are created when first needed, not with the entry: hipStreamCreateWithPriority costs 3.5-14 ms each on MI355X / ROCm 7.2
    // (tools/hip_start_bench.hip: 113-229 ms for the 16 of one entry) and a process that makes ONE proof -- nargo's -- uses three or four of them.
    hipStream_t stream = nullptr;     // normal priority; slots 0-4: created with the entry (ctx.hip init_entry says why), the others at their first acquisition
    hipStream_t stream_hi_ = nullptr; // high priority (critical-path chains of a proof): through hi()
    hipStream_t hi_latched_ = nullptr; // what hi() answered first during THIS acquisition of the slot: it keeps answering that until the slot is released
    hipStream_t hi();                 // creates it on first use
    hipStream_t hi_locked();          // the same under the owner's mutex (acquire_slots)
    void sync_hi() { if (stream_hi_) (void)hipStreamSynchronize(stream_hi_); }
    hipStream_t stream_prep = nullptr, stream_acc = nullptr;  // experiment (ZKMI_CU_SPLIT=k): CU-masked pair -- k CUs for scalar preparation, the rest for accumulates
    char* arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    bool busy = false;
    struct Ctx* owner = nullptr;  // the entry whose device the stream and the arena live on
    struct Pending {
        const char* name;
        hipEvent_t e0, e1;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;

    int reserve(size_t bytes);              // make sure the arena holds `bytes` (call before any alloc of a request)
    void* alloc(size_t bytes);              // bump allocate (256-B aligned); nullptr if reserve() was too small
    void reset() { arena_off = 0; }
    void begin_acquisition() { arena_off = 0; hi_latched_ = nullptr; }  // acquire_slot / acquire_slots, under the owner's mutex
    int pinned_reserve(size_t bytes);
};

struct Ctx {
    bool ready = false;
    int entry = 0;    // position in the process's device list
    int device = 0;   // HIP device ordinal (several entries may name the same device)
    int num_cus = 256;
    int prio_lo = 0, prio_hi = 0;  // hipDeviceGetStreamPriorityRange: numerically lower = higher priority
    std::mutex mu;
#ifndef ZKMI_NSLOTS
#define ZKMI_NSLOTS 8  // (measurement builds may ask for more: make EXPERIMENTS=1 EXTRA=-DZKMI_NSLOTS=12 -- two proof sessions at once, tools/throughput_bench.py)
#endif
    static constexpr int NSLOTS = ZKMI_NSLOTS;
    Slot slots[NSLOTS];
};
// the profile is one per process (kernels of every entry fold into it)
struct Prof {
    std::mutex mu;
    bool profiling = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_names;
};
Prof& prof();
static inline bool profiling_on() { return prof().profiling; }

static constexpr int MAX_ENTRIES = 64;
Ctx& ctx();                         // the calling thread's current entry
int n_entries();                    // entries created so far (0 before the first call that needs a device)
int current_entry();
int ensure_init();                  // lazy init of the current entry (entry 0 on device 0 if the process named no devices); ZK_ERR_NO_DEVICE if there is no GPU
int init_devices(const int* devices, int n);  // the process's device list (n == 0: every visible device); extends an existing list, never rebinds an entry
// the calling thread works on `entry` for the lifetime of the scope (hipSetDevice included); rc != ZK_OK if there is no such entry
struct CtxScope {
    int prev = 0, rc = ZK_OK;
    explicit CtxScope(int entry);
    ~CtxScope();
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
};
// handles of resident objects: the entry they live on in the top byte, a per-kind counter below
static inline uint64_t hmake(uint64_t counter) { return ((uint64_t)current_entry() << 56) | counter; }
static inline int hentry(uint64_t handle) { return (handle >> 56) == 0xff ? (int)((handle >> 48) & 0xff) : (int)(handle >> 56); }  // 0xff: a composite of multidev.hip, first entry next
// first line of every entry point that takes a handle: the call runs on the handle's entry whatever entry the calling thread was on
#define ZK_ON_ENTRY_OF(handle)                   \
    zkmi::CtxScope _scope(zkmi::hentry(handle)); \
    if (_scope.rc != ZK_OK) return _scope.rc
// background work (ctx.hip): one worker thread per process for what must not sit on a call's critical path -- window tables of keys that prove again, the
// streams a lean start withheld.  Jobs run in submission order; bg_cancelled() turns true when the process is exiting (long jobs poll it between launches).
void bg_submit(std::function<void()> job);
bool bg_cancelled();
int acquire_slot(Slot** out);       // blocks (spins) until a slot is free
int acquire_slots(int k, Slot** out); // k slots at once (all or nothing: no partial holds, hence no deadlock)
void release_slot(Slot* s);
int masked_streams(Slot* s);         // creates stream_prep / stream_acc on first use; ZK_OK with both left null when ZKMI_CU_SPLIT is unset
int slot_sync(Slot* s, hipStream_t st);  // synchronize + fold pending profile events
int h2d_big(void* dst, const void* src, size_t bytes, hipStream_t st);  // pageable host memory -> HBM through a ring of pinned buffers (ctx.hip); src is free on return

struct SlotGuard {
    Slot* s = nullptr;
    ~SlotGuard() { if (s) release_slot(s); }
};
template <int K>
struct SlotsGuard {
    Slot* s[K] = {};
    ~SlotsGuard() { for (int i = 0; i < K; i++) if (s[i]) release_slot(s[i]); }
};

// launch wrapper: optional event pair around the kernel, on the stream it is launched on
void prof_begin(Slot* s, hipStream_t st, const char* name);
void prof_end(Slot* s, hipStream_t st);
void prof_host(const char* name, double ms);  // host-side section of the path (wall clock), reported beside the kernels

#define ZK_LAUNCH(slot, st, name, kernel, grid, block, shmem, ...)                  \
    do {                                                                            \
        if (zkmi::profiling_on()) zkmi::prof_begin(slot, st, name);                 \
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);            \
        if (zkmi::profiling_on()) zkmi::prof_end(slot, st);                         \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace zkmi
