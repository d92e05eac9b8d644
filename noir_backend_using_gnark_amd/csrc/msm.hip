// Pippenger multi-scalar multiplication over BN254 G1 / G2 on gfx950.
//
// Replaces gnark-crypto v0.9.1 `(*G1Jac).MultiExp` / `(*G2Jac).MultiExp` (ecc/bn254/multiexp.go; pinned at
// /root/reference/gnark_backend_ffi/go.mod:5), the dominant cost of groth16.Prove
// (/root/reference/gnark_backend_ffi/main.go:131) and of every kzg.Commit inside plonk.Prove / plonk.Setup
// (/root/reference/gnark_backend_ffi/backend/plonk/plonk.go:67,21).  Same contract: out = sum_i s_i * P_i, errors for
// len(points) != len(scalars) and NbTasks > 1024; the AFFINE result is canonical, so it is bit-identical to upstream's
// whatever window size either side picks.
//
// Device pipeline (all on one stream, no host round trip until the window sums come back):
//   1. k_msm_digits      scalars (Montgomery or canonical) -> signed c-bit digits (gnark's partitionScalars recoding);
//                        one (bucket key, point index|sign) pair per non-zero digit, zero digits get a sentinel key
//   2. radix sort        radix.hpp: LSD passes of <= 8 key bits over the ~log2(W * 2^(c-1)) key bits, per-tile histograms + row scans + ballot-ranked
//                        scatters; no workgroup waits for another one (rounds 1-2 called rocPRIM's onesweep here, whose look-back tiles spin)
//   3. k_bucket_bounds   bucket -> [start, end) in the sorted array
//   4. k_task_plan       buckets are cut into tasks of <= L points so that one giant bucket (scalars 0/1 dominate real
//                        witnesses) cannot serialise the GPU; exclusive scan -> task offsets
//   5. k_accumulate      one thread per task: XYZZ accumulator += affine points (madd-2008-s); gathers of 64-B / 128-B
//                        points by sorted index; buckets with several tasks are folded by k_fold_multi (one WG each)
//   6. k_reduce_l1 / k_reduce_wave   sum_k k * B_k per window: thread-serial running sums over 8 buckets, then
//                        wave-cooperative suffix scans (64 lanes, DPP/bpermute shuffles of whole points)
//   7. host              Horner over the <= 32 window sums + one inversion (HField, host_ff.hpp), like gnark's final step.
// Roofline: algorithmic bytes = 96 B (G1) / 160 B (G2) per scalar-mul.  The kernel is VALU-bound on CDNA4: each mixed
// addition is 10 Fp products of ~136 quarter-rate v_mad_u64_u32 each -- see DESIGN.md for both fractions.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <tuple>
#include <type_traits>

#ifdef ZKMI_EXPERIMENTS  // rocPRIM headers (A/B only: ZKMI_SORT=0 / ZKMI_SCAN=0 select the library sort / scan)
#include "experiments/msm_rocprim_include.inc"
#endif

#include <chrono>
#include "ctx.hpp"
#include "curve.hpp"
#include "ff29.hpp"
#include "host_ff.hpp"
#include "msm.hpp"
#include "multidev.hpp"
#include "radix.hpp"

namespace zkmi {

// two buffers and which one is current (what a sort pass swaps)
struct PingPong {
    uint32_t* b[2];
    int cur = 0;
    PingPong(uint32_t* x, uint32_t* y) : b{x, y} {}
    uint32_t* current() const { return b[cur]; }
    uint32_t* alternate() const { return b[cur ^ 1]; }
    void swap() { cur ^= 1; }
};
#ifdef ZKMI_EXPERIMENTS  // rocPRIM sort / scan configuration of the A/B runs
#include "experiments/msm_rocprim_config.inc"
#endif

// The hand-written sort of radix.hpp (no workgroup ever waits for another one), same contract as sort_pairs: sorted pairs end up in kb / vb's current buffers.
// d_n (may be null): the number of pairs when only the device knows it (at most n).
static int rs_sort_pairs(Slot* s, hipStream_t st, void* tmp, PingPong& kb, PingPong& vb, size_t n, unsigned key_bits, const uint32_t* d_n = nullptr) {
    const RsPlan R = rs_plan(n, key_bits);
    uint32_t* tile_hist = (uint32_t*)tmp;
    uint32_t* ghist = tile_hist + (size_t)RS_MAX_BINS * R.ntiles;
    uint32_t* gbase = ghist + RS_MAX_PASSES * RS_MAX_BINS;
    RsArgs A;
    A.npass = R.npass;
    for (unsigned p = 0; p < RS_MAX_PASSES; p++) { A.shift[p] = R.shift[p]; A.bits[p] = R.bits[p]; }
    const unsigned nt = (unsigned)R.ntiles;
    ZK_HIP(hipMemsetAsync(ghist, 0, RS_MAX_PASSES * RS_MAX_BINS * 4, st));
    ZK_LAUNCH(s, st, "msm_sort_hist", k_rs_hist, dim3(nt), dim3(RS_THREADS), 0, (const uint32_t*)kb.current(), (uint32_t)n, A, nt, ghist, tile_hist, d_n);
    ZK_LAUNCH(s, st, "msm_sort_bases", k_rs_bases, dim3(1), dim3(RS_MAX_BINS), 0, (const uint32_t*)ghist, gbase, R.npass);
    for (unsigned p = 0; p < R.npass; p++) {
        if (p) ZK_LAUNCH(s, st, "msm_sort_hist", k_rs_tile_hist, dim3(nt), dim3(RS_THREADS), 0, (const uint32_t*)kb.current(), (uint32_t)n, R.shift[p], R.bits[p], nt, tile_hist, d_n);
        ZK_LAUNCH(s, st, "msm_sort_scan", k_rs_scan_rows, dim3(1u << R.bits[p]), dim3(256), 0, tile_hist, nt);
        ZK_LAUNCH(s, st, "msm_sort_pass", k_rs_scatter, dim3(nt), dim3(RS_THREADS), 0, (const uint32_t*)kb.current(), (const uint32_t*)vb.current(), kb.alternate(), vb.alternate(),
                  (uint32_t)n, R.shift[p], R.bits[p], nt, (const uint32_t*)tile_hist, (const uint32_t*)(gbase + p * RS_MAX_BINS), d_n);
        kb.swap();
        vb.swap();
    }
    return ZK_OK;
}

// exclusive scan of n u32 counters (radix.hpp): out may alias in
static int xs_exclusive_scan(Slot* s, hipStream_t st, void* tmp, const uint32_t* in, uint32_t* out, size_t n) {
    const unsigned nt = (unsigned)((n + XS_TILE - 1) / XS_TILE);
    uint32_t* sums = (uint32_t*)tmp;
    ZK_LAUNCH(s, st, "msm_task_scan", k_xs_sums, dim3(nt), dim3(XS_THREADS), 0, in, (uint32_t)n, sums);
    ZK_LAUNCH(s, st, "msm_task_scan", k_xs_scan_sums, dim3(1), dim3(XS_THREADS), 0, sums, nt);
    ZK_LAUNCH(s, st, "msm_task_scan", k_xs_apply, dim3(nt), dim3(XS_THREADS), 0, in, (uint32_t)n, (const uint32_t*)sums, out);
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------- wide global loads/stores
template <class T>
__device__ __forceinline__ T gload(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples only");
    T r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = q[i];
    return r;
}
template <class T>
__device__ __forceinline__ void gstore(T* p, const T& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint4* d = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) q[i] = d[i];
}
template <class T>
__device__ __forceinline__ T shfl_down_t(const T& v, unsigned d) {
    T r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* o = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) o[i] = __shfl_down(s[i], d, 64);
    return r;
}
template <class T>
__device__ __forceinline__ T shfl_xor_t(const T& v, unsigned d) {
    T r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* o = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) o[i] = __shfl_xor(s[i], d, 64);
    return r;
}

// ---------------------------------------------------------------------------------------- 1. digits
// gnark-crypto partitionScalars: digit = bits [w*c, (w+1)*c) + carry; if digit > 2^(c-1): digit -= 2^c, carry = 1.
// Table mode (table_stride != 0; resident bases with precomputed 2^(c*w) * P_i): every window shares ONE bucket set, so the
// key is the digit magnitude alone and the value indexes the table entry (w, i).
// Window-sharded tables (row_step > 1): every window is still recoded (the carry runs through all of them) but only the digits of the
// rows this rank owns -- w = row_first + k * row_step -- are emitted, as row k of the rank's table.
// Several scalar vectors against ONE table (sets > 1; table mode only: PLONK commits l, r, o -- and h1, h2, h3 -- against the same SRS): blockIdx.y is the
// vector, its digits go to a bucket set of their own (key = v * B + digit - 1) and to its own rows of the key / value arrays, so that ONE sort, ONE task
// plan and ONE accumulate launch serve all of them and the reduction yields one sum per vector.
struct DigitSrc {
    const Fr* p[3];
};
__global__ void k_msm_digits(DigitSrc src, uint32_t n, int mont, unsigned c, unsigned W, uint32_t* keys, uint32_t* vals,
                             uint32_t table_stride, unsigned row_first, unsigned row_step, unsigned rows_per_set) {
    prio_hi();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned vec = blockIdx.y, sets = gridDim.y;
    const Fr* scalars = src.p[vec];
    Fr s;
    {
        const uint4* q = reinterpret_cast<const uint4*>(scalars + i);
        uint4 a = q[0], b = q[1];
        s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w;
        s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
    }
    if (mont) s = s.from_mont();
    const uint32_t B = 1u << (c - 1);
    const uint32_t sentinel = table_stride ? sets * B : W * B;
    uint32_t carry = 0;
    unsigned next_row = row_first, wl = 0;  // table mode: the next window this rank owns and its row in the rank's table
    for (unsigned w = 0; w < W; w++) {
        unsigned bit = w * c, limb = bit >> 5, off = bit & 31;
        uint64_t v = 0;
        if (limb < 8) {
            v = s.l[limb];
            if (limb + 1 < 8) v |= (uint64_t)s.l[limb + 1] << 32;
            v >>= off;
        }
        uint32_t d = ((uint32_t)v & ((1u << c) - 1)) + carry;
        carry = 0;
        uint32_t neg = 0, mag = d;
        if (d > B) {  // d - 2^c is negative: magnitude 2^c - d
            mag = (1u << c) - d;
            neg = 1;
            carry = 1;
        }
        size_t o = (size_t)w * n + i;
        if (table_stride) {
            if (w != next_row) continue;
            o = ((size_t)vec * rows_per_set + wl) * n + i;
            keys[o] = mag ? (vec * B + (mag - 1)) : sentinel;
            vals[o] = ((wl * table_stride + i) << 1) | neg;
            next_row += row_step;
            wl++;
        } else {
            keys[o] = mag ? (w * B + (mag - 1)) : sentinel;
            vals[o] = (i << 1) | neg;
        }
    }
}

// ---- the zero digits dropped BEFORE the sort (opt-in: callers whose scalars are wire values -- bits, bytes, words -- where two thirds of the digits are zero and
// the sort would move their sentinel keys through every pass): count the non-zero digits per scalar, scan, write them contiguously.  The TOTAL stays on the
// device: the scan's last entry is what the sort, the bucket bounds and everything after read as their length (radix.hpp rs_len), so the host never waits for
// it (round 4 copied it back and synchronised -- fine for PLONK's round 1, not for a Groth16 proof whose five multi-exps are enqueued in one go).
// Same buckets, same sums: within a bucket the points arrive in another order, and the group law does not care.
__device__ __forceinline__ Fr canonical_scalar(const Fr* __restrict__ scalars, uint32_t i, int mont) {
    Fr s;
    const uint4* q = reinterpret_cast<const uint4*>(scalars + i);
    uint4 a = q[0], b = q[1];
    s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w;
    s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
    return mont ? s.from_mont() : s;
}
template <class Emit>
__device__ __forceinline__ void digits_of(const Fr& s, unsigned c, unsigned W, bool table, unsigned row_first, unsigned row_step, Emit&& emit) {
    const uint32_t B = 1u << (c - 1);
    uint32_t carry = 0;
    unsigned next_row = row_first, wl = 0;
    for (unsigned w = 0; w < W; w++) {  // the recoding of k_msm_digits, digit for digit
        unsigned bit = w * c, limb = bit >> 5, off = bit & 31;
        uint64_t v = 0;
        if (limb < 8) {
            v = s.l[limb];
            if (limb + 1 < 8) v |= (uint64_t)s.l[limb + 1] << 32;
            v >>= off;
        }
        uint32_t d = ((uint32_t)v & ((1u << c) - 1)) + carry;
        carry = 0;
        uint32_t neg = 0, mag = d;
        if (d > B) {
            mag = (1u << c) - d;
            neg = 1;
            carry = 1;
        }
        if (!table) { emit(w, mag, neg); continue; }  // one bucket set per window: `w` is the window
        if (w != next_row) continue;
        emit(wl, mag, neg);                           // one bucket set for all windows: `wl` is the row of the table
        next_row += row_step;
        wl++;
    }
}
// One pass: a workgroup recodes its 256 scalars, ranks their non-zero digits with a scan over the workgroup, stages the pairs in LDS in that order, reserves
// room for all of them with ONE atomic add on the global pair counter (one counter even for a batch of vectors: the bucket keys tell the vectors apart) and
// copies them out with consecutive lanes writing consecutive pairs.  The workgroups land in whatever order their atomics arrive: a sort's input has no order
// to keep.  (Round 5's earlier forms, both measured on the uniform 2^20 proof, whose scalar preparation is on the critical path beside computeH: count + scan
// over all scalars + write = two passes and three launches more than k_msm_digits, +0.16 ms; one pass with every lane writing its own run of pairs to global
// memory -- 52-byte strides between lanes -- still +0.14 ms; staged through LDS but with the scalar read and converted twice +0.1 ms; this form -- read and
// converted once, recoded twice, staged -- costs a uniform vector nothing (9.67-9.73 against 9.75 ms on one box) and a witness-like one 0.3-0.4 ms less:
// profiles/rnd5_d_*, rnd5_e_*, rnd5_f_*, rnd5_g_*.)  Dynamic LDS: 256 * W pairs of 8 bytes.
__global__ __launch_bounds__(256) void k_msm_digits_compact(DigitSrc src, uint32_t n, int mont, unsigned c, unsigned W, unsigned row_first, unsigned row_step,
                                                           uint32_t table_stride, uint32_t* __restrict__ total, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    prio_hi();
    extern __shared__ uint32_t stage[];  // [0, 256 W): keys, [256 W, 512 W): values
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t base;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned vec = blockIdx.y;
    const uint32_t B = 1u << (c - 1), cap = 256u * W;
    const bool live = i < n;
    uint32_t k = 0;
    Fr sc = Fr::zero();
    if (live) {
        sc = canonical_scalar(src.p[vec], i, mont);  // read and converted once; recoded twice (count, then write)
        digits_of(sc, c, W, table_stride != 0, row_first, row_step, [&](unsigned, uint32_t mag, uint32_t) { k += mag != 0; });
    }
    // exclusive scan of k over the workgroup: inside the wave by shuffles, across the four waves through LDS
    uint32_t incl = k;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= (unsigned)d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (unsigned w2 = 0; w2 < wave; w2++) before += wsum[w2];
    const uint32_t wg_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (threadIdx.x == 0) base = wg_total ? atomicAdd(total, wg_total) : 0;
    if (live && k) {
        uint32_t o = before + incl - k;
        digits_of(sc, c, W, table_stride != 0, row_first, row_step, [&](unsigned wl, uint32_t mag, uint32_t neg) {
            if (!mag) return;
            if (table_stride) {
                stage[o] = vec * B + (mag - 1);
                stage[cap + o] = ((wl * table_stride + i) << 1) | neg;
            } else {
                stage[o] = wl * B + (mag - 1);
                stage[cap + o] = (i << 1) | neg;
            }
            o++;
        });
    }
    __syncthreads();
    const uint32_t g0 = base;
    for (uint32_t j = threadIdx.x; j < wg_total; j += 256) {
        keys[g0 + j] = stage[j];
        vals[g0 + j] = stage[cap + j];
    }
}

// ---------------------------------------------------------------------------------------- 3. bucket boundaries
// start[b] = first sorted position whose key >= b (lower bound), for b in [0, nb]; key nb is the zero-digit sentinel.
// One lane per bucket, ~log2(total) dependent L2 hits each: no serial gap-filling loops whatever the key distribution
// (the top window leaves ~20k empty buckets in a row for uniform scalars).
__global__ void k_bucket_bounds(const uint32_t* __restrict__ keys, uint32_t total, uint32_t nb, uint32_t* __restrict__ start, const uint32_t* __restrict__ d_total) {
    prio_hi();
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    total = rs_len(total, d_total);
    uint32_t lo = 0, hi = total;  // answer in [lo, hi]
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

// ---------------------------------------------------------------------------------------- 4. task plan
// The task length is chosen ON THE DEVICE from what the sorted digits look like (ctl[1] = largest bucket, ctl[2] <- the length): the host's L -- sized for
// uniform scalars -- is only the maximum.  Real witnesses are mostly 0 / 1 / small values: few non-zero digits, a handful of giant buckets.  The accumulate
// kernel then is not throughput- but LATENCY-bound by its longest tasks (one lane adds L points one after the other, ~5 us each), so the length shrinks with
// the work there is -- 4x what would give every resident lane one task -- but never so far that the largest bucket falls into more than 2048 partial sums
// (one wave folds them, 64 per round).  Uniform inputs keep the host's L.
// Giant buckets (0 / 1-heavy witnesses put a quarter of all points into ONE bucket): their partial sums are folded by up to 64 waves over segments
// (k_fold_giant) before k_fold_multi folds the segment heads -- which is what lets the task length drop to 32 for sparse inputs without a serial fold of
// thousands of partial sums.  ctl layout (the 256-byte header of the bin counters): [0] split buckets, [1] largest bucket, [2] task length, [3] buckets above
// GIANT_POINTS points, [4] giant buckets listed, [8 .. 8 + GIANT_MAX) their ids.
constexpr uint32_t GIANT_T = 256;         // partial sums above which a bucket counts as giant
constexpr uint32_t GIANT_MAX = 48;        // giants listed; any further ones fold serially as before
constexpr uint32_t GIANT_POINTS = 8192;   // = GIANT_T tasks of the shortest length
__device__ __forceinline__ uint32_t giant_seg(uint32_t cnt) {  // segment length: a multiple of 64 such that there are at most 64 segments
    const uint32_t s = ((cnt + 63) / 64 + 63) / 64 * 64;
    return s < 64 ? 64 : s;
}
__global__ __launch_bounds__(256) void k_bucket_stats(const uint32_t* __restrict__ start, uint32_t nb, uint32_t* __restrict__ ctl) {
    prio_hi();
    __shared__ uint32_t wmax[4];
    uint32_t len = 0;
    uint32_t big = 0;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        const uint32_t l = start[b + 1] - start[b];
        len = max(len, l);
        big += l > GIANT_POINTS ? 1u : 0u;
    }
    if (big) atomicAdd(&ctl[3], big);
    for (int d = 32; d > 0; d >>= 1) len = max(len, (uint32_t)__shfl_down((int)len, d, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        len = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (len) atomicMax(&ctl[1], len);
    }
}
// flat_keeps: when NO bucket is longer than the host's task length nothing serialises a lane, and every bucket stays one task.  (Cutting them anyway -- to have
// more tasks than lanes at 2^17..2^19 points, where there are fewer buckets than lanes -- made half of the buckets two-task buckets and cost a fold pass as long
// as the accumulate kernel itself: PLONK at 2^19 gates 21.6 -> 17.1 ms, 2^18 16.7 -> 12.3, 2^17 10.6 -> 9.4; profiles/r04_h_plonk_window_sweep.jsonl.)
__global__ void k_pick_len(const uint32_t* __restrict__ start, uint32_t nb, uint32_t Lmax, uint32_t Lmin, uint32_t lanes, uint32_t factor, uint32_t cap,
                           uint32_t flat_keeps, uint32_t* __restrict__ ctl) {
    prio_hi();
    const uint32_t nnz = start[nb], biggest = ctl[1];
    const uint32_t q = (nnz + lanes - 1) / lanes;
    uint32_t lw = factor * q;
    if (2 * lw < Lmax) lw = (factor / 2 ? factor / 2 : 1) * q;  // sparse digits: the accumulate kernel is latency-bound -- go shorter still
    if (ctl[3] > GIANT_MAX) cap = cap > 2048 ? 2048 : cap;       // more giants than k_fold_giant takes: keep their serial folds short
    uint32_t lg = (biggest + cap - 1) / cap;
    uint32_t L = max(max(lw, lg), Lmin);
    if (flat_keeps && biggest <= Lmax && nb * 8 >= lanes) L = Lmax;  // (with very few buckets -- 2^16 points -- cutting them is still the better deal: 3.5 vs 3.7 ms)
    ctl[2] = min(L, Lmax);
}
__global__ void k_task_plan(const uint32_t* start, uint32_t nb, const uint32_t* __restrict__ ctl, uint32_t* ntasks, uint32_t* multi_list, uint32_t* num_multi) {
    prio_hi();
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    if (b == nb) { ntasks[b] = 0; return; }
    const uint32_t L = ctl[2];
    uint32_t cnt = start[b + 1] - start[b];
    uint32_t t = (cnt + L - 1) / L;
    ntasks[b] = t;
    if (t > 1) multi_list[atomicAdd(num_multi, 1u)] = b;
    if (t > GIANT_T) {
        const uint32_t gi = atomicAdd(&num_multi[4], 1u);
        if (gi < GIANT_MAX) num_multi[8 + gi] = b;
    }
}

// Hand-off between the workgroups of ONE launch without anybody waiting (the serial step that used to be a launch of its own -- 10-17 us start to start on a
// dependent chain, whatever it computes -- is run by the workgroup that finishes LAST): every workgroup stores its results, releases them at agent scope and adds
// 1 to a counter; the one whose add brought it to `expected` acquires and goes on.  Nobody spins, so the launch cannot deadlock however few of its workgroups
// are resident at a time.  Form per MI355X_MICROARCH.md "inter-workgroup visibility": stores -> barrier (the workgroup's stores are issued) -> release fence ->
// agent-scope atomic by one lane; the last arriver: acquire fence -> plain loads.  All lanes of the workgroup get the same answer.
__device__ __forceinline__ bool block_arrive_is_last(uint32_t* counter, uint32_t expected) {
    __shared__ uint32_t is_last;
    // every lane drains ITS OWN stores and no-return atomics first (waves other than lane 0's may still have them in flight: the barrier alone does not
    // wait for vector memory operations): only then does lane 0's release + arrival say "this workgroup's writes are visible"
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = (atomicAdd(counter, 1u) + 1 == expected) ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return false;
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}

// k_bucket_stats + k_pick_len in one launch: the last workgroup to arrive picks the task length (ctl[5] counts arrivals)
__global__ __launch_bounds__(256) void k_bucket_stats_pick(const uint32_t* __restrict__ start, uint32_t nb, uint32_t Lmax, uint32_t Lmin, uint32_t lanes, uint32_t factor,
                                                           uint32_t cap, uint32_t flat_keeps, uint32_t* __restrict__ ctl) {
    prio_hi();
    __shared__ uint32_t wmax[4];
    uint32_t len = 0;
    uint32_t big = 0;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        const uint32_t l = start[b + 1] - start[b];
        len = max(len, l);
        big += l > GIANT_POINTS ? 1u : 0u;
    }
    if (big) atomicAdd(&ctl[3], big);
    for (int d = 32; d > 0; d >>= 1) len = max(len, (uint32_t)__shfl_down((int)len, d, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        len = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (len) atomicMax(&ctl[1], len);
    }
    if (!block_arrive_is_last(&ctl[5], gridDim.x)) return;
    if (threadIdx.x == 0) {  // = k_pick_len
        const uint32_t nnz = start[nb], biggest = atomicAdd(&ctl[1], 0u), nbig = atomicAdd(&ctl[3], 0u);
        const uint32_t q = (nnz + lanes - 1) / lanes;
        uint32_t lw = factor * q;
        if (2 * lw < Lmax) lw = (factor / 2 ? factor / 2 : 1) * q;
        if (nbig > GIANT_MAX) cap = cap > 2048 ? 2048 : cap;
        const uint32_t lg = (biggest + cap - 1) / cap;
        uint32_t L = max(max(lw, lg), Lmin);
        if (flat_keeps && biggest <= Lmax && nb * 8 >= lanes) L = Lmax;  // see k_pick_len
        ctl[2] = min(L, Lmax);
    }
}

// k_task_plan + the first two launches of the exclusive scan of its output (radix.hpp: k_xs_sums, k_xs_scan_sums) in one: a workgroup plans one scan tile
// of buckets, leaves the tile's task count, and the last workgroup to arrive scans the tile counts (ctl[6] counts arrivals).  k_xs_apply follows as before.
__global__ __launch_bounds__(XS_THREADS) void k_task_plan_scan(const uint32_t* __restrict__ start, uint32_t nb, uint32_t* __restrict__ ctl, uint32_t* __restrict__ ntasks,
                                                               uint32_t* __restrict__ multi_list, uint32_t* __restrict__ sums) {
    prio_hi();
    const uint32_t L = ctl[2];
    const uint32_t base = blockIdx.x * XS_TILE + threadIdx.x * XS_IPT;
    uint32_t v = 0;
#pragma unroll
    for (unsigned k = 0; k < XS_IPT; k++) {
        const uint32_t b = base + k;
        if (b > nb) continue;
        uint32_t t = 0;
        if (b < nb) {
            const uint32_t cnt = start[b + 1] - start[b];
            t = (cnt + L - 1) / L;
            if (t > 1) multi_list[atomicAdd(&ctl[0], 1u)] = b;
            if (t > GIANT_T) {
                const uint32_t gi = atomicAdd(&ctl[4], 1u);
                if (gi < GIANT_MAX) ctl[8 + gi] = b;
            }
        }
        ntasks[b] = t;
        v += t;
    }
    uint32_t total;
    (void)xs_block_exclusive(v, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
    if (!block_arrive_is_last(&ctl[6], gridDim.x)) return;
    const uint32_t ntiles = gridDim.x;  // = k_xs_scan_sums
    uint32_t carry = 0;
    for (uint32_t lo = 0; lo < ntiles; lo += XS_THREADS) {
        const uint32_t i = lo + threadIdx.x;
        const uint32_t x = i < ntiles ? sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = xs_block_exclusive(x, &tot);
        if (i < ntiles) sums[i] = carry + ex;
        carry += tot;
    }
}

// One record per task (thread t < total tasks): where its points start in the sorted array, and a sort key that orders
// tasks by DECREASING length so that the 64 lanes of a wave run tasks of (nearly) equal length -- bucket loads are
// Poisson-distributed, and a wave otherwise waits for its longest lane.
//
// The ordering is a counting sort on the length key (k_task_fill histograms, k_task_bins scans, k_task_scatter places): three
// small launches instead of a library sort's ~20 merge passes.  Keys above TS_BINS are quantised (bshift) -- the order only
// balances lanes, results never depend on it (every task writes its own partial sum).
constexpr uint32_t TS_BINS = 2048;

__device__ __forceinline__ void task_fill_body(const uint32_t* __restrict__ start, const uint32_t* __restrict__ task_off, uint32_t nb, uint32_t Lmax,
                                               const uint32_t* __restrict__ ctl, uint32_t max_tasks, uint32_t bshift, uint32_t nbins,
                                               uint32_t* __restrict__ task_begin, uint32_t* __restrict__ len_key, uint32_t* __restrict__ hist) {
    const uint32_t L = ctl[2];  // the length the plan cut the buckets with; keys stay relative to Lmax (the accumulate kernel's argument)
    __shared__ uint32_t h[TS_BINS];
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) h[b] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total_tasks = task_off[nb];
    if (t < max_tasks) {
        if (t >= total_tasks) {
            len_key[t] = 0xffffffffu;  // padding: stays behind the real tasks
            task_begin[t] = 0;
        } else {
            // bucket = last b with task_off[b] <= t  (empty buckets have task_off[b] == task_off[b+1] and are skipped)
            uint32_t lo = 0, hi = nb;  // invariant: task_off[lo] <= t < task_off[hi]
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (task_off[mid] <= t) lo = mid; else hi = mid;
            }
            uint32_t b = lo;
            uint32_t begin = start[b] + (t - task_off[b]) * L;
            uint32_t end = min(begin + L, start[b + 1]);
            task_begin[t] = begin;
            uint32_t key = Lmax - (end - begin);  // 0 = longest possible
            len_key[t] = key;
            atomicAdd(&h[key >> bshift], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

__global__ __launch_bounds__(256) void k_task_fill(const uint32_t* __restrict__ start, const uint32_t* __restrict__ task_off, uint32_t nb, uint32_t Lmax,
                                                   const uint32_t* __restrict__ ctl, uint32_t max_tasks, uint32_t bshift, uint32_t nbins,
                                                   uint32_t* __restrict__ task_begin, uint32_t* __restrict__ len_key, uint32_t* __restrict__ hist) {
    prio_hi();
    task_fill_body(start, task_off, nb, Lmax, ctl, max_tasks, bshift, nbins, task_begin, len_key, hist);
}

// exclusive scan of the <= TS_BINS bin counts, in place (one workgroup)
__device__ __forceinline__ void task_bins_body(uint32_t* __restrict__ hist, uint32_t nbins) {
    __shared__ uint32_t part[256];
    constexpr uint32_t PER = TS_BINS / 256;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        uint32_t b = threadIdx.x * PER + k;
        v[k] = b < nbins ? hist[b] : 0;
        sum += v[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        uint32_t b = threadIdx.x * PER + k;
        if (b < nbins) hist[b] = run;
        run += v[k];
    }
}
__global__ __launch_bounds__(256) void k_task_bins(uint32_t* __restrict__ hist, uint32_t nbins) {
    prio_hi();
    task_bins_body(hist, nbins);
}
// k_task_fill + k_task_bins in one launch: the last workgroup to arrive scans the bin counts (ctl[7] counts arrivals)
__global__ __launch_bounds__(256) void k_task_fill_bins(const uint32_t* __restrict__ start, const uint32_t* __restrict__ task_off, uint32_t nb, uint32_t Lmax,
                                                        uint32_t* __restrict__ ctl, uint32_t max_tasks, uint32_t bshift, uint32_t nbins,
                                                        uint32_t* __restrict__ task_begin, uint32_t* __restrict__ len_key, uint32_t* __restrict__ hist) {
    prio_hi();
    task_fill_body(start, task_off, nb, Lmax, ctl, max_tasks, bshift, nbins, task_begin, len_key, hist);
    if (!block_arrive_is_last(&ctl[7], gridDim.x)) return;
    task_bins_body(hist, nbins);
}

__global__ __launch_bounds__(256) void k_task_scatter(const uint32_t* __restrict__ len_key, uint32_t max_tasks, uint32_t bshift, uint32_t nbins,
                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ len_key_sorted,
                                                      uint32_t* __restrict__ task_sorted) {
    prio_hi();
    __shared__ uint32_t h[TS_BINS], base[TS_BINS];
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x) h[b] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = t < max_tasks ? len_key[t] : 0xffffffffu, rank = 0;
    bool real = key != 0xffffffffu;
    if (real) rank = atomicAdd(&h[key >> bshift], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += blockDim.x)
        if (h[b]) base[b] = atomicAdd(&cursor[b], h[b]);
    __syncthreads();
    if (t >= max_tasks) return;
    uint32_t pos = real ? base[key >> bshift] + rank : t;  // real tasks fill [0, total_tasks); padding keeps its own slot behind them
    len_key_sorted[pos] = key;
    task_sorted[pos] = t;
}

// ---------------------------------------------------------------------------------------- 5. accumulate
// Up to three MSMs over the SAME prepared scalars (Groth16: A, B1, K all pair with the wire values) go into one launch --
// blockIdx.y selects the base array / output -- so that the machine stays full across their seams.
struct AccBatch {
    const void* pts[3];
    void* partial[3];
    uint32_t skip_below[3];
};
// one task: the points vals[begin .. end) summed into partial[t]
template <class F>
__device__ __forceinline__ void acc_task(const Affine<F>* __restrict__ pts, XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ vals, uint32_t skip_below,
                                         uint32_t t, uint32_t begin, uint32_t end) {
    if constexpr (sizeof(F) == sizeof(Fp)) {
        // G1: unsaturated 9 x 29-bit accumulator (ff29.hpp); converted back to gnark's image once per task
        Acc29 acc;
        acc.inf = true;
        for (uint32_t j = begin; j < end; j++) {
            uint32_t v = vals[j];
            if ((v >> 1) < skip_below) continue;  // scalars shared with an MSM whose first bases do not exist (pk.G1.K vs w)
            Affine<F> p = gload(pts + (v >> 1));
            if (v & 1) p.y = p.y.neg();
            xyzz_madd29(acc, p.x, p.y);
        }
        gstore(partial + t, acc29_to_packed(acc));  // G1 interchange format (ff29.hpp): no multiplication to leave the 2^261 domain
    } else {
        // G2: the same representation per Fp2 component
        Acc29G2 acc;
        acc.inf = true;
        for (uint32_t j = begin; j < end; j++) {
            uint32_t v = vals[j];
            if ((v >> 1) < skip_below) continue;
            Affine<F> p = gload(pts + (v >> 1));
            if (v & 1) p.y = p.y.neg();
            xyzz_madd29(acc, p.x, p.y);
        }
        gstore(partial + t, acc29g2_to_xyzz(acc));
    }
}
#ifdef ZKMI_EXPERIMENTS  // accumulate with LDS-DMA prefetch of the next point (round 3: neutral)
#include "experiments/msm_accumulate_prefetch.inc"
#endif
// four waves per SIMD (G1) / two (G2) are part of the kernel's design: say so, so that a change that wants a few registers more shows up as spills in
// -Rpass-analysis=kernel-resource-usage instead of silently dropping a wave (a shorter formula for the second point of a task asked for 134: DESIGN.md 8)
#define ZK_ACC_BOUNDS __launch_bounds__(256, (sizeof(F) == sizeof(Fp) ? 4 : 2))  // (measured equal to __launch_bounds__(256) in four alternating pairs: unlike the NTT passes, ntt.hip)
template <class F>
__global__ ZK_ACC_BOUNDS void k_accumulate(AccBatch batch, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ task_begin,
                                                    const uint32_t* __restrict__ len_key_sorted, const uint32_t* __restrict__ task_sorted, uint32_t L,
                                                    uint32_t max_tasks) {
    const Affine<F>* __restrict__ pts = (const Affine<F>*)batch.pts[blockIdx.y];
    XYZZ<F>* __restrict__ partial = (XYZZ<F>*)batch.partial[blockIdx.y];
    const uint32_t skip_below = batch.skip_below[blockIdx.y];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_tasks) return;
    uint32_t key = len_key_sorted[i];
    if (key == 0xffffffffu) return;
    uint32_t t = task_sorted[i];
    uint32_t begin = task_begin[t];
    acc_task<F>(pts, partial, vals, skip_below, t, begin, begin + (L - key));
}
#ifdef ZKMI_EXPERIMENTS  // G2 accumulate with ZZ / ZZZ in LDS, three waves per SIMD (round 6: 6 % slower at 2^20, 5 % at 2^24, 3 % witness-like -- DESIGN.md 8)
#include "experiments/msm_accumulate_g2_lds.inc"
#endif
#ifdef ZKMI_EXPERIMENTS  // accumulate from a resident grid with a chunk counter (round 3: 3-7 % slower)
#include "experiments/msm_accumulate_resident.inc"
#endif

// ---------------------------------------------------------------------------------------- tail arithmetic
// The reduction tail adds XYZZ points to XYZZ points.  G1 uses the 29-bit-limb form (ff29.hpp: 1.4x fewer instructions per
// product); G2 keeps the canonical saturated form.  TailPt<F> hides the difference from the three tail kernels.
template <class F>
struct TailPt {
    static constexpr unsigned LPP = 1;  // lanes per point
    XYZZ<F> v;
    static __device__ __forceinline__ TailPt inf() { return TailPt{XYZZ<F>::inf()}; }
    static __device__ __forceinline__ TailPt load(const XYZZ<F>* p) { return TailPt{gload(p)}; }
    __device__ __forceinline__ void store(XYZZ<F>* p) const { gstore(p, v); }
    __device__ __forceinline__ void add(const TailPt& o) { v.add(o.v); }
    __device__ __forceinline__ void dbl() { v.dbl(); }
    __device__ __forceinline__ TailPt shfl_down(unsigned d) const { return TailPt{shfl_down_t(v, d)}; }
    __device__ __forceinline__ TailPt shfl_xor(unsigned d) const { return TailPt{shfl_xor_t(v, d)}; }
};
template <>
struct TailPt<Fp> {
    static constexpr unsigned LPP = 1;
    Acc29 v;
    static __device__ __forceinline__ TailPt inf() { TailPt t; t.v.inf = true; for (int i = 0; i < 9; i++) t.v.x.l[i] = t.v.y.l[i] = t.v.zz.l[i] = t.v.zzz.l[i] = 0; return t; }
    static __device__ __forceinline__ TailPt load(const XYZZ<Fp>* p) { TailPt t; acc29_load(t.v, gload(p)); return t; }
    __device__ __forceinline__ void store(XYZZ<Fp>* p) const { gstore(p, acc29_to_packed(v)); }
    __device__ __forceinline__ void add(const TailPt& o) { acc29_add(v, o.v); }
    __device__ __forceinline__ void dbl() { acc29_dbl(v); }
    __device__ __forceinline__ TailPt shfl_down(unsigned d) const {
        TailPt r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 36; i++) o[i] = __shfl_down(s[i], d, 64);
        r.v.inf = __shfl_down((int)v.inf, d, 64) != 0;
        return r;
    }
    __device__ __forceinline__ TailPt shfl_xor(unsigned d) const {
        TailPt r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 36; i++) o[i] = __shfl_xor(s[i], d, 64);
        r.v.inf = __shfl_xor((int)v.inf, d, 64) != 0;
        return r;
    }
};

// G1, four lanes per point (acc29_add_quad / acc29_dbl_quad): every lane of a quad carries the same point; 16 points per wave.
struct TailPtQ {
    static constexpr unsigned LPP = 4;
    Acc29 v;
    static __device__ __forceinline__ TailPtQ inf() { TailPtQ t; t.v.inf = true; for (int i = 0; i < 9; i++) t.v.x.l[i] = t.v.y.l[i] = t.v.zz.l[i] = t.v.zzz.l[i] = 0; return t; }
    static __device__ __forceinline__ TailPtQ load(const XYZZ<Fp>* p) { TailPtQ t; acc29_load(t.v, gload(p)); return t; }
    __device__ __forceinline__ void store(XYZZ<Fp>* p) const { gstore(p, acc29_to_packed(v)); }
    __device__ __forceinline__ void add(const TailPtQ& o) { acc29_add_quad(v, o.v, threadIdx.x & 3u); }
    __device__ __forceinline__ void dbl() { acc29_dbl_quad(v, threadIdx.x & 3u); }
    __device__ __forceinline__ TailPtQ shfl_down(unsigned d) const {  // d counted in points
        TailPtQ r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 36; i++) o[i] = __shfl_down(s[i], d * 4, 64);
        r.v.inf = __shfl_down((int)v.inf, d * 4, 64) != 0;
        return r;
    }
    __device__ __forceinline__ TailPtQ shfl_xor(unsigned d) const {
        TailPtQ r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 36; i++) o[i] = __shfl_xor(s[i], d * 4, 64);
        r.v.inf = __shfl_xor((int)v.inf, d * 4, 64) != 0;
        return r;
    }
};

// G2 in the same form (fused multi-product add / dbl of ff29.hpp, model-checked).  With the latency-oriented tail structure it measured
// no faster than the canonical saturated form (the kernels hold 256 VGPRs + 142 AGPRs at one wave per SIMD); with the work-oriented
// structure of the hidden tails (16 buckets per lane, lane-serial second level) its 1.4x lower instruction count is worth ~0.07 ms per
// proof.  -DZKMI_G2_TAIL_SAT selects the canonical form (A/B).
#ifndef ZKMI_G2_TAIL_SAT
template <>
struct TailPt<Fp2> {
    static constexpr unsigned LPP = 1;
    Acc29G2 v;
    static __device__ __forceinline__ TailPt inf() {
        TailPt t;
        t.v.inf = true;
        uint32_t* o = reinterpret_cast<uint32_t*>(&t.v);
        for (unsigned i = 0; i < 72; i++) o[i] = 0;
        return t;
    }
    static __device__ __forceinline__ TailPt load(const XYZZ<Fp2>* p) { TailPt t; acc29g2_load(t.v, gload(p)); return t; }
    __device__ __forceinline__ void store(XYZZ<Fp2>* p) const { gstore(p, acc29g2_to_xyzz(v)); }
    __device__ __forceinline__ void add(const TailPt& o) { acc29g2_add(v, o.v); }
    __device__ __forceinline__ void dbl() { acc29g2_dbl(v); }
    __device__ __forceinline__ TailPt shfl_down(unsigned d) const {
        TailPt r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 72; i++) o[i] = __shfl_down(s[i], d, 64);
        r.v.inf = __shfl_down((int)v.inf, d, 64) != 0;
        return r;
    }
    __device__ __forceinline__ TailPt shfl_xor(unsigned d) const {
        TailPt r;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
        uint32_t* o = reinterpret_cast<uint32_t*>(&r.v);
#pragma unroll
        for (unsigned i = 0; i < 72; i++) o[i] = __shfl_xor(s[i], d, 64);
        r.v.inf = __shfl_xor((int)v.inf, d, 64) != 0;
        return r;
    }
};
#endif

// buckets cut into several tasks: one WAVE folds the bucket's partials into the first one.  Most split buckets have 2 - 4 partials (the dense
// population under a narrow top window), so the butterfly only runs the levels the count needs: 2 additions for 3 partials instead of 6.
// (Measured earlier in round 2: EVERY split bucket folded by up to 64 waves in segments + a second launch over the segment heads -- the 64x larger grid costs
// more than the serial rounds it saves, PLONK 2^22 87.6 -> 90.9 ms.  What runs now is the targeted form: only the listed giants, a fixed grid of 768 workgroups.)
// first pass over the listed giant buckets: wave `seg` of giant `gi` folds its segment of the bucket's partial sums into the segment's first slot
template <class F>
__global__ __launch_bounds__(256) void k_fold_giant(XYZZ<F>* partial, const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ ctl) {
    prio_hi();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gi = blockIdx.x / 16, seg = (blockIdx.x % 16) * 4 + wave;
    const uint32_t ng = min(ctl[4], GIANT_MAX);
    if (gi >= ng) return;
    const uint32_t b = ctl[8 + gi];
    const uint32_t t0 = task_off[b], t1 = task_off[b + 1];
    const uint32_t S = giant_seg(t1 - t0);
    const uint32_t lo = t0 + seg * S, hi = min(lo + S, t1);
    if (lo >= t1 || hi - lo < 2) return;  // nothing, or a single partial sum: it already sits in the segment's first slot
    TailPt<F> acc = TailPt<F>::inf();
    for (uint32_t t = lo + lane; t < hi; t += 64) acc.add(TailPt<F>::load(partial + t));
    const uint32_t cnt = min(hi - lo, 64u);
    unsigned first = 32;
    while (first >= cnt && first > 0) first >>= 1;
    for (unsigned d = first; d > 0; d >>= 1) {
        TailPt<F> o = acc.shfl_down(d);
        if (lane < d) acc.add(o);
    }
    if (lane == 0) acc.store(partial + lo);
}
template <class F>
__global__ __launch_bounds__(256) void k_fold_multi(XYZZ<F>* partial, const uint32_t* task_off, const uint32_t* multi_list,
                                                    const uint32_t* num_multi) {
    prio_hi();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nm = *num_multi;
    for (uint32_t m = blockIdx.x * 4 + wave; m < nm; m += gridDim.x * 4) {
        uint32_t b = multi_list[m];
        uint32_t t0 = task_off[b], t1 = task_off[b + 1];
        TailPt<F> acc = TailPt<F>::inf();
        uint32_t nheld = t1 - t0;  // partial sums this wave folds
        bool listed = false;
        if (nheld > GIANT_T) {  // a giant that k_fold_giant has reduced to its segment heads?
            const uint32_t ng = min(num_multi[4], GIANT_MAX);
            listed = __ballot(lane < ng && num_multi[8 + lane] == b) != 0;
        }
        if (listed) {
            const uint32_t S = giant_seg(nheld);
            nheld = (nheld + S - 1) / S;  // <= 64 heads
            if (lane < nheld) acc = TailPt<F>::load(partial + t0 + lane * S);
        } else {
            for (uint32_t t = t0 + lane; t < t1; t += 64) acc.add(TailPt<F>::load(partial + t));
        }
        const uint32_t cnt = min(nheld, 64u);  // lanes that hold something (uniform over the wave)
        unsigned first = 32;
        while (first >= cnt && first > 0) first >>= 1;  // largest power of two below cnt
        for (unsigned d = first; d > 0; d >>= 1) {
            TailPt<F> o = acc.shfl_down(d);
            if (lane < d) acc.add(o);
        }
        if (lane == 0) acc.store(partial + t0);
    }
}

// ---------------------------------------------------------------------------------------- 6. bucket reduce
// Invariant carried through the levels, per window:  value = sum_j A[j] + 2^sh * sum_j j * S[j]   (j = 0..N-1).
// Level 1 (thread-serial, m buckets per thread): buckets X_k (weight k+1) -> A'[q] = sum_l (l+1) X_{qm+l},
// S'[q] = sum_l X_{qm+l}; then value = sum A' + m * sum q S'.
template <class F, class PT = TailPt<F>>
__global__ __launch_bounds__(256) void k_reduce_l1(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ task_off, uint32_t B,
                                                   uint32_t W, uint32_t m, XYZZ<F>* __restrict__ A_out, XYZZ<F>* __restrict__ S_out) {
    prio_hi();
    uint32_t N1 = B / m;
    uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / PT::LPP;  // PT::LPP lanes share one chunk of m buckets
    if (g >= W * N1) return;
    uint32_t w = g / N1, q = g % N1;
    uint32_t b0 = w * B + q * m;
    PT run = PT::inf(), acc = PT::inf();
    for (int l = (int)m - 1; l >= 0; l--) {
        uint32_t b = b0 + l;
        uint32_t t0 = task_off[b];
        if (task_off[b + 1] > t0) run.add(PT::load(partial + t0));
        acc.add(run);
    }
    if (threadIdx.x % PT::LPP == 0) {
        acc.store(A_out + g);
        run.store(S_out + g);
    }
}

// Thread-serial level over m entries (A_l, S_l) per lane: 3 additions per entry instead of the ~15 of a wave level -- for tails whose
// latency hides under another kernel.   S' = sum_l S_l ;  A' = sum_l A_l + 2^sh * sum_l l * S_l ;  next shift = sh + log2(m).
template <class F>
__global__ __launch_bounds__(256) void k_reduce_l2(const XYZZ<F>* __restrict__ A_in, const XYZZ<F>* __restrict__ S_in, uint32_t N, uint32_t W, uint32_t sh,
                                                   uint32_t m, XYZZ<F>* __restrict__ A_out, XYZZ<F>* __restrict__ S_out) {
    prio_hi();
    typedef TailPt<F> PT;
    uint32_t Nout = (N + m - 1) / m;
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= W * Nout) return;
    uint32_t w = g / Nout, q = g % Nout;
    PT run = PT::inf(), t = PT::inf(), asum = PT::inf();
    for (int l = (int)m - 1; l >= 0; l--) {
        uint32_t j = q * m + (uint32_t)l;
        if (j >= N) continue;
        asum.add(PT::load(A_in + (size_t)w * N + j));
        run.add(PT::load(S_in + (size_t)w * N + j));  // suffix sum S_l + ... + S_(m-1)
        if (l >= 1) t.add(run);                        // sum over l >= 1 of the suffix sums = sum_l l * S_l
    }
    for (uint32_t i = 0; i < sh; i++) t.dbl();
    asum.add(t);
    asum.store(A_out + g);
    run.store(S_out + g);
}

// Wave-cooperative level: 64 consecutive entries per wave (lane = l).
//   S' = sum_l S_l ;  A' = sum_l A_l + 2^sh * sum_l l * S_l ;   next shift = sh + 6
// suffix scan of S (6 shuffle steps), per-lane doublings, one butterfly reduction.
// FAN = 64 / PT::LPP entries per wave; next shift = sh + log2(FAN).
template <class F, class PT = TailPt<F>>
__global__ __launch_bounds__(64) void k_reduce_wave(const XYZZ<F>* __restrict__ A_in, const XYZZ<F>* __restrict__ S_in, uint32_t N, uint32_t W,
                                                    uint32_t sh, XYZZ<F>* __restrict__ A_out, XYZZ<F>* __restrict__ S_out) {
    prio_hi();
    constexpr uint32_t FAN = 64 / PT::LPP;
    uint32_t Nout = (N + FAN - 1) / FAN;
    uint32_t chunk = blockIdx.x;  // W * Nout chunks
    uint32_t w = chunk / Nout, q = chunk % Nout;
    uint32_t lane = threadIdx.x / PT::LPP;
    uint32_t j = q * FAN + lane;
    PT s = PT::inf(), a = PT::inf();
    if (j < N) {
        s = PT::load(S_in + (size_t)w * N + j);
        a = PT::load(A_in + (size_t)w * N + j);
    }
    // inclusive suffix sums: s_l = sum_{u >= l} S_u
    for (unsigned d = 1; d < FAN; d <<= 1) {
        PT t = s.shfl_down(d);
        if (lane + d < FAN) s.add(t);
    }
    PT y = s;  // lane 0 holds the plain sum S'
    if (lane == 0) y = PT::inf();
    for (uint32_t i = 0; i < sh; i++) y.dbl();  // 2^sh * suffix_l, in parallel on the lanes
    y.add(a);
    for (unsigned d = FAN / 2; d > 0; d >>= 1) {
        PT t = y.shfl_xor(d);
        y.add(t);
    }
    if (threadIdx.x == 0) {
        s.store(S_out + chunk);
        y.store(A_out + chunk);
    }
}

#ifdef ZKMI_EXPERIMENTS  // single-launch fold / reduction tail (round 3: neutral / +0.6 ms per 2^20 proof)
#include "experiments/msm_tail_fused.inc"
#endif

// ---------------------------------------------------------------------------------------- host side
template <class HF> struct HostOf;
template <> struct HostOf<Fp> { typedef HFp type; };
template <> struct HostOf<Fp2> { typedef HFp2 type; };

unsigned msm_pick_window(size_t n) {
    // GPU cost model in Fp products: accumulate n*W mixed adds (10 each) + bucket reduce W*2^(c-1) * ~2 full adds (14 each)
    // + fixed per-bucket overhead of the plan/scan kernels.
    unsigned best = 8;
    double bc = 1e300;
    for (unsigned c = 6; c <= 22; c++) {
        unsigned W = (255 + c - 1) / c;
        double cost = (double)n * W * 10.0 + (double)W * (double)((size_t)1 << (c - 1)) * 34.0;
        if (cost < bc) { bc = cost; best = c; }
    }
    return best;
}

// Sizes every buffer of one MSM call; `need` is what the caller must reserve in the slot's arena BEFORE it carves
// anything else out of it (the arena cannot grow while allocations are live).
// window size for resident bases with precomputed tables: all windows share one bucket set, so only 2^(c-1) buckets are reduced
// Cost model: mixed additions (n W, 10 units each) + bucket reduction (34 units per bucket).  Round 4: the additions run at full rate only when there is a bucket
// -- one task, one lane -- for every lane the machine holds (256 CUs x 1,024); with fewer buckets the accumulate kernel's rate falls like (buckets / lanes)^(1/4)
// (measured at 2^19 points: 7.5 / 8.3 / 10.4 / 9.8 G additions/s with 2^16 / 2^17 / 2^18 / 2^19 buckets).  Without that term the model picked c = 17 at 2^19
// points and c = 16 at 2^16, where 19 and 17 are 23-29 % and 17 % faster (Groth16 2^19: 8.4 -> 6.0 ms; profiles/r04_j_groth16_window_sweep.jsonl).
unsigned msm_pick_window_table(size_t n) {
    unsigned best = 8;
    double bc = 1e300;
    const double lanes = 256.0 * 1024.0;
    for (unsigned c = 8; c <= 22; c++) {
        unsigned W = (255 + c - 1) / c;
        const double buckets = (double)((size_t)1 << (c - 1));
        const double eff = buckets >= lanes ? 1.0 : sqrt(sqrt(buckets / lanes));
        double cost = (double)n * W * 10.0 / eff + buckets * 34.0;
        if (cost < bc) { bc = cost; best = c; }
    }
    return best;
}

template <class F>
static int msm_plan_uncached(size_t n, const zk_msm_cfg* cfg, hipStream_t st, MsmPlan* P, const MsmTable* tab, unsigned sets);

// plans are pure functions of (n, window choice, table geometry, group): memoised, because the rocPRIM workspace-size
// queries inside cost ~0.1 ms of host time per proof otherwise
template <class F>
static int msm_plan(size_t n, const zk_msm_cfg* cfg, hipStream_t st, MsmPlan* P, const MsmTable* tab = nullptr, unsigned sets = 1) {
    struct Key {
        size_t n, stride;
        unsigned c_cfg, c_tab, g2, l1, rows, sets;
        bool operator<(const Key& o) const { return std::tie(n, stride, c_cfg, c_tab, g2, l1, rows, sets) < std::tie(o.n, o.stride, o.c_cfg, o.c_tab, o.g2, o.l1, o.rows, o.sets); }
    };
    static std::mutex mu;
    static std::map<Key, MsmPlan> memo;
    Key k{n, tab ? tab->stride : 0, (cfg && cfg->window_bits) ? (unsigned)cfg->window_bits : 0u, tab ? tab->c : 0u, (unsigned)(sizeof(F) != 32), tab ? (tab->l1_m | (tab->l2_m << 8)) : 0u, tab ? (tab->row_first | (tab->row_step << 8)) : 0u, sets};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(k);
        if (it != memo.end()) { *P = it->second; return ZK_OK; }
    }
    ZK_TRY(msm_plan_uncached<F>(n, cfg, st, P, tab, sets));
    std::lock_guard<std::mutex> lk(mu);
    memo[k] = *P;
    return ZK_OK;
}

template <class F>
static int msm_plan_uncached(size_t n, const zk_msm_cfg* cfg, hipStream_t st, MsmPlan* P, const MsmTable* tab, unsigned sets) {
    typedef XYZZ<F> Pt;
    memset(P, 0, sizeof *P);
    if (n == 0) return ZK_OK;
    if (sets < 1 || sets > 3 || (sets > 1 && !tab)) return set_err(ZK_ERR_ARG, "a batch of %u scalar vectors needs a window table and at most three vectors", sets);
    if (n > ((size_t)1 << 27)) return set_err(ZK_ERR_ARG, "n = %zu exceeds the per-call limit 2^27 (shard the MSM)", n);
    unsigned c = tab ? tab->c : ((cfg && cfg->window_bits) ? (unsigned)cfg->window_bits : msm_pick_window(n));
    // plain method: c <= 22 (one bucket set PER WINDOW: 12 x 2^21 partial sums already); against a window table all windows share one bucket set and c = 23, 24
    // -- 11 digits per scalar instead of 12 at 2^24 points -- are admissible (round 5 measured them: DESIGN.md 8)
    if (c < 2 || c > (tab ? 24u : 22u)) return set_err(ZK_ERR_ARG, "window_bits = %u outside [2, %u]", c, tab ? 24u : 22u);
    P->c = c;
    P->Wd = (255 + c - 1) / c;
    P->W = tab ? sets : P->Wd;  // bucket sets: one per window without a table, one per scalar vector with one
    P->row_first = tab ? tab->row_first : 0;
    P->row_step = tab ? (tab->row_step ? tab->row_step : 1) : 1;
    P->Wrows = tab ? tab->rows() : P->Wd;
    if (tab && P->Wrows == 0) { memset(P, 0, sizeof *P); return ZK_OK; }  // this rank owns no window: the MSM is empty
    P->table_stride = tab ? (uint32_t)tab->stride : 0;
    P->B = 1u << (c - 1);
    P->nb = P->W * P->B;
    P->total = n * P->Wrows * (tab ? sets : 1);
    if (P->total >= ((size_t)1 << 31)) return set_err(ZK_ERR_ARG, "n * windows = %zu overflows 31-bit positions", P->total);
    if (tab && (size_t)tab->stride * P->Wrows >= ((size_t)1 << 31)) return set_err(ZK_ERR_ARG, "table too large for 31-bit indices");
    // tasks of at most L points: 2x the mean bucket load of a uniform input, at least 32.  The top window only sees digits up to
    // (r-1) >> (c*(W-1)), so its buckets (window-per-bucket-set mode) / the lowest buckets (table mode: one bucket set) are denser
    // than that: they are simply cut into several tasks and folded by k_fold_multi.  Only when the dense population is mild (<= 4x)
    // is L raised to keep it whole -- one task per bucket is cheaper than a fold.
    size_t mean = P->total / P->nb + 1;
    {
        // r - 1 = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000; its top 64 bits:
        const uint64_t r_top64 = 0x30644e72e131a029ULL;
        unsigned shift = c * (P->Wd - 1);  // bits below the top window
        uint64_t top_max = shift >= 192 ? (r_top64 >> (shift - 192)) : ~0ULL;
        if (top_max < P->B && top_max > 0) {
            size_t dense = n / (size_t)top_max + 1 + (tab ? mean : 0);  // table mode: the top window's points come on top of the others
            static const size_t dense_max = (size_t)ZK_EXP("ZKMI_L_DENSE", 4);  // experiment switch: 0 = always cut dense buckets
            if (dense > mean && dense <= dense_max * mean + 8 && dense_max) mean = dense;
        }
    }
    P->L = (uint32_t)(mean * 2 < 32 ? 32 : mean * 2);
    // one wave folds the partials of a split bucket: bound their number (a bucket can hold up to n points -- real witnesses are
    // mostly 0/1) by growing the task size with n
    if (P->L < (n >> 16)) P->L = (uint32_t)(n >> 16);
    // the device may shorten the tasks down to Lmin when the digits turn out sparse / skewed (k_pick_len): the task arrays are sized for that
    static const bool adaptive = (ZK_EXP("ZKMI_ADAPTIVE_L", 1) != 0);
    static const uint32_t lmin_div = (uint32_t)std::max<long>(1, ZK_EXP("ZKMI_L_MIN_DIV", 8));  // experiment switch
    P->Lmin = adaptive ? (P->L / lmin_div > 32 ? P->L / lmin_div : 32) : P->L;
    if (P->Lmin > P->L) P->Lmin = P->L;
    P->max_tasks = (size_t)P->nb + P->total / P->Lmin + 1;
    static const unsigned l1_env = (unsigned)ZK_EXP("ZKMI_L1_M", 0);  // experiment switch (power of two)
    unsigned l1_m = l1_env ? l1_env : (tab && tab->l1_m ? tab->l1_m : 8);
    while (l1_m & (l1_m - 1)) l1_m &= l1_m - 1;  // a power of two (N1 = B / m1 must be exact)
    P->m1 = P->B >= l1_m ? l1_m : P->B;  // level-1 serial chunk
    P->m2 = (tab && tab->l2_m) ? tab->l2_m : 0;
    P->N1 = P->B / P->m1;
    P->key_bits = 1;
    while (((uint64_t)1 << P->key_bits) <= P->nb) P->key_bits++;
    P->sort_tmp_bytes = rs_plan(P->total, P->key_bits).tmp_bytes;
    P->scan_tmp_bytes = xs_tmp_bytes((size_t)P->nb + 1);
#ifdef ZKMI_EXPERIMENTS  // which sort the preparation calls
#include "experiments/msm_sort_select.inc"
#endif
    P->lvl_elems = (size_t)P->W * P->N1;  // level-1 outputs; later levels are 64x smaller
    P->tsort_tmp_bytes = TS_BINS * 4;  // bin counters of the task counting sort
    P->need_prep = 5 * align_up(P->max_tasks * 4, 256) + align_up(P->tsort_tmp_bytes + 16, 256) + 4 * align_up(P->total * 4, 256) +
                   align_up(P->sort_tmp_bytes + 16, 256) + align_up(P->scan_tmp_bytes + 16, 256) + 3 * align_up(((size_t)P->nb + 2) * 4, 256) +
                   align_up((size_t)P->nb * 4, 256) + 256 + 32768;
    P->need_acc = align_up(P->max_tasks * sizeof(Pt), 256) + 4 * align_up((P->lvl_elems + 64) * sizeof(Pt), 256) + 32768;
    P->need = P->need_prep + P->need_acc;
    return ZK_OK;
}

// Scalar-side half of an MSM on stream `st`: digits, sort, bucket bounds, task plan.  The result only depends on the
// scalars, so several MSMs over the same scalar vector (Groth16: A, B1, K and G2.B all pair with the wire values) share it.
static int msm_prepare(Slot* s, hipStream_t st, const MsmPlan& P, const Fr* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmPrep* out, const DigitSrc* batch = nullptr,
                       bool drop_zero_digits = false) {
    out->P = P;
    out->n = n;
    out->empty = (n == 0) || P.total == 0;  // no points, or a window-sharded table of which this rank owns no row
    out->ready = nullptr;
    if (out->empty) return ZK_OK;
    const unsigned c = P.c, W = P.W, key_bits = P.key_bits;
    const uint32_t nb = P.nb, L = P.L;
    size_t total = P.total;
    const size_t max_tasks = P.max_tasks;
    size_t sort_tmp_bytes = P.sort_tmp_bytes, scan_tmp_bytes = P.scan_tmp_bytes;
    uint32_t* keys0 = (uint32_t*)s->alloc(total * 4);
    uint32_t* keys1 = (uint32_t*)s->alloc(total * 4);
    uint32_t* vals0 = (uint32_t*)s->alloc(total * 4);
    uint32_t* vals1 = (uint32_t*)s->alloc(total * 4);
    void* sort_tmp = s->alloc(sort_tmp_bytes + 16);
    void* scan_tmp = s->alloc(scan_tmp_bytes + 16);
    uint32_t* start = (uint32_t*)s->alloc(((size_t)nb + 2) * 4);
    uint32_t* ntasks = (uint32_t*)s->alloc(((size_t)nb + 2) * 4);
    uint32_t* task_off = (uint32_t*)s->alloc(((size_t)nb + 2) * 4);
    uint32_t* multi_list = (uint32_t*)s->alloc((size_t)nb * 4);
    uint32_t* task_begin = (uint32_t*)s->alloc(max_tasks * 4);
    uint32_t* lkey0 = (uint32_t*)s->alloc(max_tasks * 4);
    uint32_t* lkey1 = (uint32_t*)s->alloc(max_tasks * 4);
    uint32_t* tid1 = (uint32_t*)s->alloc(max_tasks * 4);
    uint32_t* bins = (uint32_t*)s->alloc(256 + TS_BINS * 4);  // [0] = num_multi, [64..] = bin counters: one memset clears both
    if (!keys0 || !keys1 || !vals0 || !vals1 || !sort_tmp || !scan_tmp || !start || !ntasks || !task_off || !multi_list || !task_begin ||
        !lkey0 || !lkey1 || !tid1 || !bins)
        return set_err(ZK_ERR_HIP, "MSM workspace was not reserved up front (%zu bytes needed)", P.need_prep);
    uint32_t* num_multi = bins;
    uint32_t* hist = bins + 64;

    // the counters of step 4 are cleared here, ahead of the chain of dependent launches they would otherwise lengthen
    ZK_HIP(hipMemsetAsync(bins, 0, 256 + TS_BINS * 4, st));
    // ---- 1. digits
    const uint32_t* d_total = nullptr;  // non-null: the zero digits were dropped and only the device knows how many pairs there are (at most `total`)
    {
        DigitSrc src = batch ? *batch : DigitSrc{{d_scalars, nullptr, nullptr}};
        const unsigned sets = P.table_stride ? P.W : 1;
        const int mont = (cfg && cfg->scalars_mont) ? 1 : 0;
        const dim3 grid((unsigned)((n + 255) / 256), sets);
        if (drop_zero_digits && 256u * P.Wd * 8u > 65536u) drop_zero_digits = false;  // windows narrower than 8 bits (tests): more pairs per workgroup than the stage holds
        if (drop_zero_digits) {  // only the non-zero digits are written (k_msm_digits_compact); their number stays on the device
            uint32_t* cnt = (uint32_t*)s->alloc(256);
            if (!cnt) return set_err(ZK_ERR_HIP, "MSM workspace was not reserved up front (zero-digit compaction)");
            ZK_HIP(hipMemsetAsync(cnt, 0, 4, st));
            ZK_LAUNCH(s, st, "msm_digits", k_msm_digits_compact, grid, dim3(256), 256u * P.Wd * 8u, src, (uint32_t)n, mont, c, P.Wd, P.row_first, P.row_step, P.table_stride, cnt,
                      keys0, vals0);
            d_total = cnt;  // the number of pairs, where the kernels below read it
        } else {
            ZK_LAUNCH(s, st, "msm_digits", k_msm_digits, grid, dim3(256), 0, src, (uint32_t)n, mont, c, P.Wd, keys0, vals0, P.table_stride, P.row_first, P.row_step,
                      P.Wrows);
        }
    }
    // ---- 2. sort (bucket key -> point index|sign)
    PingPong kb(keys0, keys1), vb(vals0, vals1);
    bool own_sort = true, own_scan = true;
#ifdef ZKMI_EXPERIMENTS  // the rocPRIM radix-sort call of the A/B runs
#include "experiments/msm_sort_rocprim_call.inc"
#endif
    if (own_sort) ZK_TRY(rs_sort_pairs(s, st, sort_tmp, kb, vb, total, key_bits, d_total));
    const uint32_t* keys = kb.current();
    // ---- 3. bucket bounds
    ZK_LAUNCH(s, st, "msm_bucket_bounds", k_bucket_bounds, dim3((nb + 1 + 255) / 256), dim3(256), 0, keys, (uint32_t)total, nb, start, d_total);
    // ---- 4. plan.  The steps that are one workgroup's work (task length, scan of the tile counts, scan of the bin counts) are run by the LAST workgroup of
    // the launch before them (block_arrive_is_last): eleven dependent launches of 4-30 us each were 0.26 ms start to end, on the critical path of every MSM
    // (DESIGN.md 3.4); seven remain.  ZKMI_PREP_MERGE=0 (A/B switch): one launch per step.
    static const bool merged = ZK_EXP("ZKMI_PREP_MERGE", 1) != 0;
    static const uint32_t l_factor = (uint32_t)ZK_EXP("ZKMI_L_FACTOR", 4);     // experiment switches
    static const uint32_t l_cap = (uint32_t)ZK_EXP("ZKMI_L_GIANT_CAP", 16384);
    static const uint32_t l_flat = (uint32_t)ZK_EXP("ZKMI_L_FLAT_KEEPS", 1);   // 0 = the rule of rounds 1-3 (A/B)
    const unsigned stats_grid = nb / 4096 ? (nb / 4096 > 512 ? 512 : nb / 4096) : 1;
    const unsigned scan_tiles = (unsigned)(((size_t)nb + 1 + XS_TILE - 1) / XS_TILE);
    if (merged && P.Lmin < L) {
        ZK_LAUNCH(s, st, "msm_bucket_stats", k_bucket_stats_pick, dim3(stats_grid), dim3(256), 0, (const uint32_t*)start, nb, L, P.Lmin, (uint32_t)(ctx().num_cus * 1024), l_factor,
                  l_cap, l_flat, bins);
    } else {
        if (P.Lmin < L) ZK_LAUNCH(s, st, "msm_bucket_stats", k_bucket_stats, dim3(stats_grid), dim3(256), 0, (const uint32_t*)start, nb, bins);
        ZK_LAUNCH(s, st, "msm_pick_len", k_pick_len, dim3(1), dim3(1), 0, (const uint32_t*)start, nb, L, P.Lmin, (uint32_t)(ctx().num_cus * 1024), l_factor, l_cap, l_flat, bins);
    }
    if (merged && own_scan) {
        uint32_t* sums = (uint32_t*)scan_tmp;
        ZK_LAUNCH(s, st, "msm_task_plan", k_task_plan_scan, dim3(scan_tiles), dim3(XS_THREADS), 0, (const uint32_t*)start, nb, bins, ntasks, multi_list, sums);
        ZK_LAUNCH(s, st, "msm_task_scan", k_xs_apply, dim3(scan_tiles), dim3(XS_THREADS), 0, (const uint32_t*)ntasks, nb + 1, (const uint32_t*)sums, task_off);
    } else {
    ZK_LAUNCH(s, st, "msm_task_plan", k_task_plan, dim3((nb + 1 + 255) / 256), dim3(256), 0, (const uint32_t*)start, nb, (const uint32_t*)bins, ntasks, multi_list, num_multi);
#ifdef ZKMI_EXPERIMENTS  // the rocPRIM scan call of the A/B runs
#include "experiments/msm_scan_rocprim_call.inc"
#endif
    if (own_scan) ZK_TRY(xs_exclusive_scan(s, st, scan_tmp, ntasks, task_off, (size_t)nb + 1));
    }
    // ---- 4b. per-task records, ordered by decreasing length (counting sort on L - len)
    uint32_t bshift = 0;
    while ((L >> bshift) >= TS_BINS) bshift++;
    const uint32_t nbins = (L >> bshift) + 1;
    const unsigned tgrid = (unsigned)((max_tasks + 255) / 256);
    // (the same hand-off for fill + bins was measured and is off: one release fence per workgroup is one write-back of the L2 -- 128 / 257 workgroups above do
    // not notice, the 3,713 of k_task_fill at 2^20 turn a 30 us kernel into 99 us, and the 2^26 MSM loses 3.6 ms.  ZKMI_PREP_MERGE=2 selects it for A/B.)
    static const bool merged_fill = ZK_EXP("ZKMI_PREP_MERGE", 1) == 2;
    if (merged_fill) {
        ZK_LAUNCH(s, st, "msm_task_fill", k_task_fill_bins, dim3(tgrid), dim3(256), 0, (const uint32_t*)start, (const uint32_t*)task_off, nb, L, bins, (uint32_t)max_tasks, bshift,
                  nbins, task_begin, lkey0, hist);
    } else {
        ZK_LAUNCH(s, st, "msm_task_fill", k_task_fill, dim3(tgrid), dim3(256), 0, (const uint32_t*)start, (const uint32_t*)task_off, nb, L, (const uint32_t*)bins,
                  (uint32_t)max_tasks, bshift, nbins, task_begin, lkey0, hist);
        ZK_LAUNCH(s, st, "msm_task_bins", k_task_bins, dim3(1), dim3(256), 0, hist, nbins);
    }
    ZK_LAUNCH(s, st, "msm_task_scatter", k_task_scatter, dim3(tgrid), dim3(256), 0, (const uint32_t*)lkey0, (uint32_t)max_tasks, bshift, nbins, hist,
              lkey1, tid1);
    out->vals = vb.current();
    out->start = start;
    out->task_off = task_off;
    out->task_begin = task_begin;
    out->lkeys = lkey1;
    out->tids = tid1;
    out->multi_list = multi_list;
    out->num_multi = num_multi;
    ZK_HIP(hipEventCreateWithFlags(&out->ready, hipEventDisableTiming));
    ZK_HIP(hipEventRecord(out->ready, st));
    return ZK_OK;
}

// Base-side half: bucket accumulation, bucket reduction, window sums to pinned memory.  `nb` jobs (<= 3) over the same prepared
// scalars share ONE accumulate launch on the first job's stream (which waits for the preparation when it ran elsewhere); each
// job's reduction tail then runs on its own stream.  `d_pts[b]` is indexed by the scalar index; indices < skip_below[b] are skipped.
template <class F>
static int msm_accumulate_batch(int nb, Slot* const* sl, const hipStream_t* sts, const MsmPrep& R, const void* const* d_pts, const uint32_t* skip_below,
                                MsmJob* const* jobs) {
    // jobs[0]->gate_acc (in, optional): the accumulate kernel waits for it; any ->want_done (in): record an event after it
    // (returned in the LAST job's acc_done, which owns it).
    typedef XYZZ<F> Pt;
    const MsmPlan& P = R.P;
    for (int b = 0; b < nb; b++) {
        jobs[b]->s = sl[b];
        jobs[b]->st = sts[b];
        jobs[b]->c = P.c;
        jobs[b]->W = P.W;
        jobs[b]->empty = R.empty;
    }
    if (R.empty) return ZK_OK;
    const unsigned W = P.W;
    const uint32_t B = P.B, L = P.L, m1 = P.m1, N1 = P.N1;
    const size_t max_tasks = P.max_tasks, lvl_elems = P.lvl_elems;
    Pt *partial[3], *lvlA[3][2], *lvlS[3][2];
    AccBatch batch = {};
    for (int b = 0; b < nb; b++) {
        Slot* s = sl[b];
        partial[b] = (Pt*)s->alloc(max_tasks * sizeof(Pt));
        for (int k = 0; k < 2; k++) {
            lvlA[b][k] = (Pt*)s->alloc((lvl_elems + 64) * sizeof(Pt));
            lvlS[b][k] = (Pt*)s->alloc((lvl_elems + 64) * sizeof(Pt));
        }
        if (!partial[b] || !lvlA[b][0] || !lvlA[b][1] || !lvlS[b][0] || !lvlS[b][1])
            return set_err(ZK_ERR_HIP, "MSM workspace was not reserved up front (%zu bytes needed)", P.need_acc);
        batch.pts[b] = d_pts[b];
        batch.partial[b] = partial[b];
        batch.skip_below[b] = skip_below[b];
    }
    // The throughput-bound accumulate kernel runs on `jobs[0]->chain` when given (the caller serialises the accumulate kernels of
    // several MSMs back to back on one stream, no event round trips between them); everything after it -- the latency-bound
    // reduction tail -- runs on each job's own stream, gated on an event.
    MsmJob* j0 = jobs[0];
    hipStream_t sa = j0->chain ? j0->chain : sts[0];
    if (R.ready) ZK_HIP(hipStreamWaitEvent(sa, R.ready, 0));
    if (j0->gate_acc) ZK_HIP(hipStreamWaitEvent(sa, j0->gate_acc, 0));
    // Independent callers (the MultiExp calls upstream issues from concurrent goroutines) pass through a turnstile per device entry: their accumulate kernels
    // -- each of which fills the machine -- run one at a time in arrival order, gated stream to stream (no host wait), while preparation and reduction tails of
    // the others run underneath: the schedule of the fused prover, for callers that do not know of each other.
    struct Turnstile {
        std::mutex mu;
        hipEvent_t ring[16] = {};
        int next = 0;
        hipEvent_t last = nullptr;
    };
    static Turnstile g_turn[MAX_ENTRIES];
    Turnstile* T = j0->turnstile ? &g_turn[current_entry()] : nullptr;
    std::unique_lock<std::mutex> turn_lock;
    if (T) {
        turn_lock = std::unique_lock<std::mutex>(T->mu);
        if (T->last) ZK_HIP(hipStreamWaitEvent(sa, T->last, 0));
    }
    // ---- 5. accumulate
    const char* acc_name = sizeof(F) == 32 ? "msm_accumulate_g1" : "msm_accumulate_g2";
    const unsigned full_grid = (unsigned)((max_tasks + 255) / 256);
    bool launched = false;
#ifdef ZKMI_EXPERIMENTS  // launch of the prefetch / resident accumulate variants
#include "experiments/msm_accumulate_variants_launch.inc"
#endif
    // occupancy cap (experiment ZKMI_ACC_LDS_KB): an unused dynamic-LDS request of k KB per workgroup lets only floor(160 / k) workgroups -- that many waves per
    // SIMD -- of this kernel be resident per CU, so that the short kernels of other streams (sort, transforms) find registers and wave slots beside it
    static const unsigned acc_lds_kb = (unsigned)ZK_EXP("ZKMI_ACC_LDS_KB", 0);
    static const unsigned acc_lds_min_tasks = (unsigned)ZK_EXP("ZKMI_ACC_LDS_MIN_TASKS", 0);
    unsigned acc_shmem = (acc_lds_kb && max_tasks >= acc_lds_min_tasks) ? acc_lds_kb * 1024u : 0u;
    if (acc_shmem > 65536u) {
        static std::mutex attr_mu;
        static uint64_t attr_done = 0;
        std::lock_guard<std::mutex> lk(attr_mu);
        if (!(attr_done & ((uint64_t)1 << current_entry()))) {
            ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_accumulate<Fp>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ZK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_accumulate<Fp2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_done |= (uint64_t)1 << current_entry();
        }
    }
#ifdef ZKMI_EXPERIMENTS
    if constexpr (sizeof(F) == sizeof(Fp2)) {
        static const bool g2_lds = ZK_EXP("ZKMI_G2_LDS", 0) != 0;  // experiment (round 6): ZZ / ZZZ of the G2 accumulator in LDS, three waves per SIMD
        if (g2_lds && !launched) {
            ZK_LAUNCH(sl[0], sa, acc_name, k_accumulate_g2_lds, dim3(full_grid, (unsigned)nb), dim3(256), 0, batch, R.vals, R.task_begin, R.lkeys, R.tids, L, (uint32_t)max_tasks);
            launched = true;
        }
    }
#endif
    if (!launched) ZK_LAUNCH(sl[0], sa, acc_name, (k_accumulate<F>), dim3(full_grid, (unsigned)nb), dim3(256), acc_shmem, batch, R.vals, R.task_begin, R.lkeys, R.tids, L, (uint32_t)max_tasks);
    if (T) {
        hipEvent_t& e = T->ring[T->next];
        if (!e) ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(e, sa));  // re-recording is safe: a stream that already waits on the event waits for the recording it saw
        T->last = e;
        T->next = (T->next + 1) % 16;
        turn_lock.unlock();
    }
    bool want = j0->chain != nullptr || nb > 1;
    for (int b = 0; b < nb; b++) want = want || jobs[b]->want_done;
    if (want) {
        hipEvent_t done;
        ZK_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        ZK_HIP(hipEventRecord(done, sa));
        jobs[nb - 1]->acc_done = done;
        for (int b = 0; b < nb; b++)
            if (sts[b] != sa) ZK_HIP(hipStreamWaitEvent(sts[b], done, 0));
    }
    for (int b = 0; b < nb; b++) {
        Slot* s = sl[b];
        hipStream_t st = sts[b];
        bool folded = false;
#ifdef ZKMI_EXPERIMENTS  // launch of the fused fold / tail
#include "experiments/msm_tail_fused_launch.inc"
#endif
        if (!folded) {
            ZK_LAUNCH(s, st, "msm_fold_giant", (k_fold_giant<F>), dim3(GIANT_MAX * 16), dim3(256), 0, partial[b], R.task_off, (const uint32_t*)R.num_multi);  // exits at once without giants
            ZK_LAUNCH(s, st, "msm_fold_multi", (k_fold_multi<F>), dim3(1024), dim3(256), 0, partial[b], R.task_off, R.multi_list, R.num_multi);
        }
        // ---- 6. bucket reduce
        bool quad = false;
        if constexpr (sizeof(F) == sizeof(Fp)) quad = jobs[b]->quad_tail;
        const uint32_t lpp = quad ? 4 : 1, fan = 64 / lpp, fan_log = quad ? 4 : 6;
        // level 1 fills every SIMD with one wave already (it is throughput-bound: measured, four lanes per point make it 3x SLOWER);
        // only the wave levels after it, which cannot fill the machine, gain from the shorter serial chain
        ZK_LAUNCH(s, st, "msm_reduce_l1", (k_reduce_l1<F>), dim3((unsigned)(((size_t)W * N1 + 255) / 256)), dim3(256), 0, (const Pt*)partial[b], R.task_off, B, W,
                  m1, lvlA[b][0], lvlS[b][0]);
        uint32_t N = N1, sh = 0;
        for (uint32_t mm = m1; mm > 1; mm >>= 1) sh++;  // value = sum A + 2^sh * sum j S_j
        int cur = 0;
        if (P.m2 > 1 && N > 64 * P.m2) {  // second thread-serial level (tails that hide under the next accumulate)
            uint32_t Nout = (N + P.m2 - 1) / P.m2;
            ZK_LAUNCH(s, st, "msm_reduce_l2", (k_reduce_l2<F>), dim3((unsigned)(((size_t)W * Nout + 255) / 256)), dim3(256), 0, (const Pt*)lvlA[b][cur],
                      (const Pt*)lvlS[b][cur], N, W, sh, P.m2, lvlA[b][cur ^ 1], lvlS[b][cur ^ 1]);
            cur ^= 1;
            N = Nout;
            for (uint32_t mm = P.m2; mm > 1; mm >>= 1) sh++;
        }
        // the last level (<= 64 entries per window) is cheaper on the host than one more latency-bound launch, as long as the host
        // has few windows to do (table mode: one)
        const uint32_t host_n = (W * 64 <= 64) ? 64 : 1;
        while (N > host_n) {
            uint32_t Nout = (N + fan - 1) / fan;
            if constexpr (sizeof(F) == sizeof(Fp)) {
                if (quad) ZK_LAUNCH(s, st, "msm_reduce_wave", (k_reduce_wave<F, TailPtQ>), dim3(W * Nout), dim3(64), 0, (const Pt*)lvlA[b][cur], (const Pt*)lvlS[b][cur], N, W, sh, lvlA[b][cur ^ 1], lvlS[b][cur ^ 1]);
            }
            if (!quad) ZK_LAUNCH(s, st, "msm_reduce_wave", (k_reduce_wave<F>), dim3(W * Nout), dim3(64), 0, (const Pt*)lvlA[b][cur], (const Pt*)lvlS[b][cur], N, W, sh, lvlA[b][cur ^ 1], lvlS[b][cur ^ 1]);
            cur ^= 1;
            N = Nout;
            sh += fan_log;
        }
        // ---- 7. last-level entries -> pinned host memory (final combination and Horner happen in msm_finish)
        jobs[b]->n_final = N;
        jobs[b]->sh_final = sh;
        const size_t cnt = (size_t)W * N;
        ZK_TRY(s->pinned_reserve(2 * cnt * sizeof(Pt)));
        ZK_HIP(hipMemcpyAsync(s->pinned, lvlA[b][cur], cnt * sizeof(Pt), hipMemcpyDeviceToHost, st));
        if (N > 1) ZK_HIP(hipMemcpyAsync((char*)s->pinned + cnt * sizeof(Pt), lvlS[b][cur], cnt * sizeof(Pt), hipMemcpyDeviceToHost, st));
    }
    return ZK_OK;
}

template <class F>
static int msm_accumulate(Slot* s, hipStream_t st, const MsmPrep& R, const Affine<F>* d_pts, uint32_t skip_below, MsmJob* job) {
    const void* pts = d_pts;
    return msm_accumulate_batch<F>(1, &s, &st, R, &pts, &skip_below, &job);
}

// Whole MSM on one stream.
template <class F>
static int msm_launch(Slot* s, hipStream_t st, const MsmPlan& P, const Affine<F>* d_pts, const Fr* d_scalars, size_t n, const zk_msm_cfg* cfg,
                      MsmJob* job) {
    MsmPrep R;
    ZK_TRY(msm_prepare(s, st, P, d_scalars, n, cfg, &R));
    int rc = msm_accumulate<F>(s, st, R, d_pts, 0, job);
    if (R.ready) (void)hipEventDestroy(R.ready);  // same stream: ordering is implicit; the wait above was a no-op
    return rc;
}

template <class HF>
static int msm_finish(const MsmJob& job, XYZZ<HF>* total_out, XYZZ<HF>* per_set = nullptr) {
    *total_out = XYZZ<HF>::inf();
    if (per_set)
        for (unsigned w = 0; w < (job.empty ? 3u : job.W); w++) per_set[w] = XYZZ<HF>::inf();
    if (job.empty) return ZK_OK;
    ZK_TRY(slot_sync(job.s, job.st));
    const unsigned N = job.n_final;
    if constexpr (sizeof(HF) == sizeof(HFp)) {
        // G1 sums arrive in the kernels' interchange format x * 2^261 mod p: one multiplication by 2^-5 per coordinate gives gnark's image
        static const HFp inv32 = HFp{{32, 0, 0, 0}}.to_mont().inv();
        XYZZ<HFp>* w = reinterpret_cast<XYZZ<HFp>*>(job.s->pinned);
        const size_t cnt = (size_t)job.W * N * (N > 1 ? 2 : 1);
        for (size_t i = 0; i < cnt; i++) {
            w[i].x = w[i].x * inv32;
            w[i].y = w[i].y * inv32;
            w[i].zz = w[i].zz * inv32;
            w[i].zzz = w[i].zzz * inv32;
        }
    }
    const XYZZ<HF>* wa = reinterpret_cast<const XYZZ<HF>*>(job.s->pinned);
    const XYZZ<HF>* wsum = wa + (size_t)job.W * N;
    XYZZ<HF> tot = XYZZ<HF>::inf();
    for (int w = (int)job.W - 1; w >= 0; w--) {
        for (unsigned k = 0; k < job.c; k++) tot.dbl();
        // window value = sum_j A_j + 2^sh * sum_j j S_j  (N = 1: just A_0)
        XYZZ<HF> val = wa[(size_t)w * N];
        if (N > 1) {
            XYZZ<HF> run = XYZZ<HF>::inf(), acc = XYZZ<HF>::inf();
            for (unsigned j = N - 1; j >= 1; j--) {
                run.add(wsum[(size_t)w * N + j]);
                acc.add(run);
                val.add(wa[(size_t)w * N + j]);
            }
            for (unsigned k = 0; k < job.sh_final; k++) acc.dbl();
            val.add(acc);
        }
        if (per_set) per_set[w] = val;  // a batch of scalar vectors: the bucket sets are separate sums, not windows of one
        tot.add(val);
    }
    *total_out = tot;
    return ZK_OK;
}

int msm_g1_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P));
    *need = P.need;
    return ZK_OK;
}
int msm_g2_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp2>(n, cfg, st, &P));
    *need = P.need;
    return ZK_OK;
}
// The slot's arena must already hold msm_g?_need() free bytes.
int msm_g1_launch(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmJob* job) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P));
    return msm_launch<Fp>(s, st, P, (const Affine<Fp>*)d_pts, (const Fr*)d_scalars, n, cfg, job);
}
int msm_g2_launch(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmJob* job) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp2>(n, cfg, st, &P));
    return msm_launch<Fp2>(s, st, P, (const Affine<Fp2>*)d_pts, (const Fr*)d_scalars, n, cfg, job);
}
int msm_prep_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need_prep, size_t* need_acc_g1, size_t* need_acc_g2) {
    MsmPlan P1, P2;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P1));
    ZK_TRY(msm_plan<Fp2>(n, cfg, st, &P2));
    if (need_prep) *need_prep = P1.need_prep;
    if (need_acc_g1) *need_acc_g1 = P1.need_acc;
    if (need_acc_g2) *need_acc_g2 = P2.need_acc;
    return ZK_OK;
}
int msm_prep_need_table(size_t n, const MsmTable& tab, hipStream_t st, size_t* need_prep, size_t* need_acc_g1, size_t* need_acc_g2) {
    MsmPlan P1, P2;
    ZK_TRY(msm_plan<Fp>(n, nullptr, st, &P1, &tab));
    ZK_TRY(msm_plan<Fp2>(n, nullptr, st, &P2, &tab));
    if (need_prep) *need_prep = P1.need_prep;
    if (need_acc_g1) *need_acc_g1 = P1.need_acc;
    if (need_acc_g2) *need_acc_g2 = P2.need_acc;
    return ZK_OK;
}
// up to three scalar vectors of n elements each against one table: one recoding, one accumulate launch, one sum per vector (msm_g1_finish_batch)
int msm_prep_need_table_batch(size_t n, unsigned sets, const MsmTable& tab, hipStream_t st, size_t* need_prep, size_t* need_acc_g1) {
    MsmPlan P1;
    ZK_TRY(msm_plan<Fp>(n, nullptr, st, &P1, &tab, sets));
    if (need_prep) *need_prep = P1.need_prep;
    if (need_acc_g1) *need_acc_g1 = P1.need_acc;
    return ZK_OK;
}
size_t msm_compact_need(size_t, unsigned) { return 1024; }  // what dropping the zero digits adds to the preparation's workspace: the pair counter
int msm_prepare_scalars_table_batch(Slot* s, hipStream_t st, const void* const* d_scalars, unsigned sets, size_t n, const zk_msm_cfg* cfg, const MsmTable& tab, MsmPrep* out,
                                    bool drop_zero_digits) {
    if (sets < 1 || sets > 3) return set_err(ZK_ERR_ARG, "a batch holds one to three scalar vectors");
    MsmPlan P;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P, &tab, sets));
    DigitSrc src = {{nullptr, nullptr, nullptr}};
    for (unsigned v = 0; v < sets; v++) src.p[v] = (const Fr*)d_scalars[v];
    return msm_prepare(s, st, P, src.p[0], n, cfg, out, &src, drop_zero_digits);
}
int msm_prepare_scalars_table(Slot* s, hipStream_t st, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, const MsmTable& tab, MsmPrep* out, bool drop_zero_digits) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P, &tab));
    return msm_prepare(s, st, P, (const Fr*)d_scalars, n, cfg, out, nullptr, drop_zero_digits);
}

// T[k * stride + i] = 2^(c*w_k) * P_i as affine points for the rows w_k = row_first + k * row_step < Wd (all rows: row_first = 0, row_step = 1);
// rows are `stride` entries apart, entries [n, stride) of a row are left untouched (the caller zeroes the buffer: (0,0) = infinity).
// One-time cost per resident base array.
template <class F>
__global__ __launch_bounds__(256) void k_build_table(const Affine<F>* __restrict__ pts, uint32_t n, uint32_t stride, uint32_t offset, unsigned c,
                                                     unsigned Wd, unsigned row_first, unsigned row_step, Affine<F>* __restrict__ table) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = gload(pts + i);
    if (p.is_inf()) return;  // the caller zeroed the table: (0, 0) in every row
    // The doubling chain runs through all rows in XYZZ and the rows this table holds are normalised TOGETHER: one inversion per point instead of one per
    // entry (Montgomery's trick over the <= 32 rows of a lane; the per-row coordinates live in private memory) -- 20 doublings + ~37 products per entry
    // instead of 20 doublings + ~385.
    constexpr int MAXR = 32;  // ceil(255 / c), c >= 8
    F xs[MAXR], ys[MAXR], zs[MAXR], ws[MAXR], pre[MAXR];
    int k = 0;
    if constexpr (std::is_same<F, Fp>::value) {
        // G1: the chain in the 29-bit-limb form of the bucket reductions (ff29.hpp acc29_dbl, class invariant closed under repetition); a row leaves the chain
        // through one conversion per coordinate.  1,000,000 points x 13 rows: 22.7 -> 17.0 ms (profiles/r05_w_table_build_29bit_chain.txt)
        {
            Acc29 A;
            acc29_from_xyzz(A, XYZZ<Fp>::from_affine(p));
            for (unsigned w = 0; w < Wd; w++) {
                if (w) {
                    for (unsigned b = 0; b < c; b++) acc29_dbl(A);
                }
                if (w < row_first || (w - row_first) % row_step) continue;
                const XYZZ<Fp> r = acc29_to_xyzz(A);
                xs[k] = r.x; ys[k] = r.y; zs[k] = r.zz; ws[k] = r.zzz;
                k++;
            }
        }
    }
    if constexpr (std::is_same<F, Fp2>::value) {  // G2: the same with the Fp2 form of the chain
        Acc29G2 A;
        acc29g2_from_xyzz(A, XYZZ<Fp2>::from_affine(p));
        for (unsigned w = 0; w < Wd; w++) {
            if (w) {
                for (unsigned b = 0; b < c; b++) acc29g2_dbl(A);
            }
            if (w < row_first || (w - row_first) % row_step) continue;
            const XYZZ<Fp2> r = acc29g2_to_xyzz(A);
            xs[k] = r.x; ys[k] = r.y; zs[k] = r.zz; ws[k] = r.zzz;
            k++;
        }
    }
    if (k == 0) {
        XYZZ<F> acc = XYZZ<F>::from_affine(p);
        for (unsigned w = 0; w < Wd; w++) {
            if (w) {
                for (unsigned b = 0; b < c; b++) acc.dbl();
            }
            if (w < row_first || (w - row_first) % row_step) continue;
            xs[k] = acc.x; ys[k] = acc.y; zs[k] = acc.zz; ws[k] = acc.zzz;
            k++;
        }
    }
    F run = F::one();
    for (int j = 0; j < k; j++) {
        pre[j] = run;
        run = run * (zs[j] * ws[j]);  // non-zero for points of odd order: a doubling cannot reach infinity
    }
    if (run.is_zero()) {  // a caller-supplied point of even order (outside the prime-order subgroup): row by row
        for (int j = 0; j < k; j++) gstore(table + (size_t)j * stride + offset + i, XYZZ<F>{xs[j], ys[j], zs[j], ws[j]}.to_affine());
        return;
    }
    F inv = run.inv();
    for (int j = k - 1; j >= 0; j--) {
        const F t = inv * pre[j];  // 1 / (zz * zzz) of row j
        inv = inv * (zs[j] * ws[j]);
        Affine<F> a{xs[j] * (t * ws[j]), ys[j] * (t * zs[j])};  // x / zz, y / zzz
        gstore(table + (size_t)j * stride + offset + i, a);
    }
}
template <class F>
static int build_table(Slot* s, hipStream_t st, const void* d_pts, size_t n, size_t stride, size_t offset, unsigned c, void* d_table, unsigned row_first,
                       unsigned row_step) {
    MsmTable t;
    t.c = c; t.row_first = row_first; t.row_step = row_step ? row_step : 1;
    unsigned Wd = (255 + c - 1) / c, rows = t.rows();
    ZK_HIP(hipMemsetAsync(d_table, 0, (size_t)(rows ? rows : 1) * stride * sizeof(Affine<F>), st));
    if (n && rows)
        ZK_LAUNCH(s, st, sizeof(F) == 32 ? "msm_build_table_g1" : "msm_build_table_g2", (k_build_table<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (const Affine<F>*)d_pts, (uint32_t)n, (uint32_t)stride, (uint32_t)offset, c, Wd, row_first, t.row_step, (Affine<F>*)d_table);
    return ZK_OK;
}
int msm_build_table_g1(Slot* s, hipStream_t st, const void* d_pts, size_t n, size_t stride, size_t offset, unsigned c, void* d_table, unsigned row_first,
                       unsigned row_step) {
    return build_table<Fp>(s, st, d_pts, n, stride, offset, c, d_table, row_first, row_step);
}
int msm_build_table_g2(Slot* s, hipStream_t st, const void* d_pts, size_t n, size_t stride, size_t offset, unsigned c, void* d_table, unsigned row_first,
                       unsigned row_step) {
    return build_table<Fp2>(s, st, d_pts, n, stride, offset, c, d_table, row_first, row_step);
}

// gnark keeps pk.G1.A / pk.G1.B / pk.G2.B WITHOUT their points at infinity (InfinityA / InfinityB bitmaps, setup.go); the resident key
// is wire-indexed, so the compact array is scattered once at load time: out[i] = src_idx[i] == ~0 ? infinity (0,0) : compact[src_idx[i]].
template <class F>
__global__ __launch_bounds__(256) void k_expand_bases(const Affine<F>* __restrict__ compact, const uint32_t* __restrict__ src_idx, uint32_t n,
                                                      Affine<F>* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = src_idx[i];
    Affine<F> p;
    uint4* d = reinterpret_cast<uint4*>(&p);
    if (j == 0xffffffffu) {
#pragma unroll
        for (unsigned k = 0; k < sizeof(p) / 16; k++) d[k] = make_uint4(0, 0, 0, 0);
    } else {
        p = gload(compact + j);
    }
    gstore(out + i, p);
}
int msm_expand_bases(Slot* s, hipStream_t st, int is_g2, const void* d_compact, const uint32_t* d_src_idx, size_t n, void* d_out) {
    if (!n) return ZK_OK;
    if (is_g2)
        ZK_LAUNCH(s, st, "pk_expand_bases_g2", (k_expand_bases<Fp2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const Affine<Fp2>*)d_compact, d_src_idx,
                  (uint32_t)n, (Affine<Fp2>*)d_out);
    else
        ZK_LAUNCH(s, st, "pk_expand_bases_g1", (k_expand_bases<Fp>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const Affine<Fp>*)d_compact, d_src_idx,
                  (uint32_t)n, (Affine<Fp>*)d_out);
    return ZK_OK;
}

int msm_prepare_scalars(Slot* s, hipStream_t st, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmPrep* out, bool drop_zero_digits) {
    MsmPlan P;
    ZK_TRY(msm_plan<Fp>(n, cfg, st, &P));  // the scalar-side plan does not depend on the group
    return msm_prepare(s, st, P, (const Fr*)d_scalars, n, cfg, out, nullptr, drop_zero_digits);
}
void msm_prep_release(MsmPrep* R) {
    if (R->ready) (void)hipEventDestroy(R->ready);
    R->ready = nullptr;
}
int msm_g1_accumulate(Slot* s, hipStream_t st, const MsmPrep& R, const void* d_pts, uint32_t skip_below, MsmJob* job) {
    return msm_accumulate<Fp>(s, st, R, (const Affine<Fp>*)d_pts, skip_below, job);
}
int msm_g1_accumulate_batch(int nb, Slot* const* sl, const hipStream_t* sts, const MsmPrep& R, const void* const* d_pts, const uint32_t* skip_below,
                            MsmJob* const* jobs) {
    if (nb < 1 || nb > 3) return set_err(ZK_ERR_ARG, "accumulate batch of %d", nb);
    return msm_accumulate_batch<Fp>(nb, sl, sts, R, d_pts, skip_below, jobs);
}
int msm_g2_accumulate(Slot* s, hipStream_t st, const MsmPrep& R, const void* d_pts, uint32_t skip_below, MsmJob* job) {
    return msm_accumulate<Fp2>(s, st, R, (const Affine<Fp2>*)d_pts, skip_below, job);
}
int msm_g1_finish(const MsmJob& job, XYZZ<HFp>* out) { return msm_finish<HFp>(job, out); }
int msm_g1_finish_batch(const MsmJob& job, XYZZ<HFp> out[3]) {
    XYZZ<HFp> unused;
    return msm_finish<HFp>(job, &unused, out);
}
int msm_g2_finish(const MsmJob& job, XYZZ<HFp2>* out) { return msm_finish<HFp2>(job, out); }
int msm_g1_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp>* out) {
    MsmJob job;
    ZK_TRY(msm_g1_launch(s, st, d_pts, d_scalars, n, cfg, &job));
    return msm_g1_finish(job, out);
}
int msm_g2_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp2>* out) {
    MsmJob job;
    ZK_TRY(msm_g2_launch(s, st, d_pts, d_scalars, n, cfg, &job));
    return msm_g2_finish(job, out);
}

static int check_cfg(const zk_msm_cfg* cfg) {
    if (cfg && cfg->nb_tasks > 1024) return set_err(ZK_ERR_NB_TASKS, "invalid config: config.NbTasks > 1024");
    return ZK_OK;
}

// bases registry -----------------------------------------------------------------------------------------------------
struct Bases {
    void* d = nullptr;
    size_t n = 0;
    int is_g2 = 0;
    // precomputed window table T[w * n + i] = 2^(c*w) * P_i (as for a resident Groth16 key): built at registration when it fits, so that
    // every commit against the bases feeds ONE bucket set (KZG: ~10 commits per PLONK proof against the same SRS)
    void* d_table = nullptr;
    MsmTable tab;
};
static std::mutex g_bases_mu;
static std::map<uint64_t, Bases> g_bases;
static uint64_t g_next_handle = 1;

int bases_ptr(uint64_t handle, const void** d, size_t* n, int* is_g2) {
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
    if (d) *d = it->second.d;
    if (n) *n = it->second.n;
    if (is_g2) *is_g2 = it->second.is_g2;
    return ZK_OK;
}
// window table of a registered base array (null when it was registered without one)
int bases_table(uint64_t handle, const void** d_table, MsmTable* tab, size_t* n) {
    if (md_is_composite(handle)) {  // spread over several GPUs: no single table; commits go through zk_bn254_msm_bases_dev, one partial per entry
        if (d_table) *d_table = nullptr;
        return md_bases_info(handle, n, nullptr);
    }
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
    if (d_table) *d_table = it->second.d_table;
    if (tab) *tab = it->second.tab;
    if (n) *n = it->second.n;
    return ZK_OK;
}
int bases_info(uint64_t handle, size_t* n, int* is_g2) {
    if (md_is_composite(handle)) return md_bases_info(handle, n, is_g2);
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
    if (n) *n = it->second.n;
    if (is_g2) *is_g2 = it->second.is_g2;
    return ZK_OK;
}

}  // namespace zkmi

using namespace zkmi;

template <class HF, class OUT>
static void write_affine(const XYZZ<HF>& t, OUT* out) {
    Affine<HF> a = t.to_affine();
    memcpy(out, &a, sizeof(a));
}

template <int G2>
static int msm_host_impl(const void* points, size_t n_points, const zk_fr* scalars, size_t n_scalars, const zk_msm_cfg* cfg, void* out) {
    const size_t psz = G2 ? 128 : 64;
    if (n_points != n_scalars) return set_err(ZK_ERR_LEN, "len(points) != len(scalars)");
    ZK_TRY(check_cfg(cfg));
    if (!out || (n_points && (!points || !scalars))) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    size_t n = n_points, need = 0;
    ZK_TRY(G2 ? msm_g2_need(n, cfg, st, &need) : msm_g1_need(n, cfg, st, &need));
    ZK_TRY(g.s->reserve(n * (psz + 32) + 1024 + need));
    void* d_p = g.s->alloc(n * psz + 16);
    void* d_s = g.s->alloc(n * 32 + 16);
    if (n) {
        ZK_HIP(hipMemcpyAsync(d_p, points, n * psz, hipMemcpyHostToDevice, st));
        ZK_HIP(hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, st));
    }
    if (G2) {
        XYZZ<HFp2> t;
        ZK_TRY(msm_g2_xyzz(g.s, st, d_p, d_s, n, cfg, &t));
        write_affine(t, (zk_g2_affine*)out);
    } else {
        XYZZ<HFp> t;
        ZK_TRY(msm_g1_xyzz(g.s, st, d_p, d_s, n, cfg, &t));
        write_affine(t, (zk_g1_affine*)out);
    }
    return ZK_OK;
}

template <int G2>
static int msm_dev_impl(const void* d_points, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, void* out, void* stream, int partial) {
    ZK_TRY(check_cfg(cfg));
    if (!out || (n && (!d_points || !d_scalars))) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = stream ? (hipStream_t)stream : g.s->stream;
    size_t need = 0;
    ZK_TRY(G2 ? msm_g2_need(n, cfg, st, &need) : msm_g1_need(n, cfg, st, &need));
    ZK_TRY(g.s->reserve(need));
    if (G2) {
        XYZZ<HFp2> t;
        ZK_TRY(msm_g2_xyzz(g.s, st, d_points, d_scalars, n, cfg, &t));
        if (partial) memcpy(out, &t, sizeof(t)); else write_affine(t, (zk_g2_affine*)out);
    } else {
        XYZZ<HFp> t;
        ZK_TRY(msm_g1_xyzz(g.s, st, d_points, d_scalars, n, cfg, &t));
        if (partial) memcpy(out, &t, sizeof(t)); else write_affine(t, (zk_g1_affine*)out);
    }
    return ZK_OK;
}

// several device entries (cfg->device_mask / the process default): the slices go to the entries by range (multidev.hip); one entry: that entry
template <int G2>
static int msm_host_entry(const void* points, size_t n_points, const zk_fr* scalars, size_t n_scalars, const zk_msm_cfg* cfg, void* out) {
    if (n_points != n_scalars) return set_err(ZK_ERR_LEN, "len(points) != len(scalars)");
    ZK_TRY(check_cfg(cfg));
    if (!out || (n_points && (!points || !scalars))) return set_err(ZK_ERR_ARG, "null pointer");
    std::vector<int> ents;
    ZK_TRY(md_entries_for(cfg ? (uint32_t)cfg->device_mask : 0u, n_points, (size_t)1 << 17, &ents));
    if (ents.size() > 1 && n_points >= ents.size()) return md_msm_host(G2, points, scalars, n_points, cfg, out, ents);
    CtxScope sc(ents[0]);
    if (sc.rc != ZK_OK) return sc.rc;
    return msm_host_impl<G2>(points, n_points, scalars, n_scalars, cfg, out);
}
extern "C" {

// what the planner picks for n points: the window width c and the number of c-bit digits per scalar (= mixed additions per scalar-mul)
int zk_bn254_msm_plan_info(size_t n, int window_tables, uint32_t* window_bits, uint32_t* digits) {
    if (!window_bits || !digits) return set_err(ZK_ERR_ARG, "null pointer");
    unsigned c = window_tables ? msm_pick_window_table(n) : msm_pick_window(n);
    *window_bits = c;
    *digits = (255 + c - 1) / c;
    return ZK_OK;
}

int zk_bn254_g1_msm(const zk_g1_affine* points, size_t n_points, const zk_fr* scalars, size_t n_scalars, const zk_msm_cfg* cfg, zk_g1_affine* out) {
    return msm_host_entry<0>(points, n_points, scalars, n_scalars, cfg, out);
}
int zk_bn254_g2_msm(const zk_g2_affine* points, size_t n_points, const zk_fr* scalars, size_t n_scalars, const zk_msm_cfg* cfg, zk_g2_affine* out) {
    return msm_host_entry<1>(points, n_points, scalars, n_scalars, cfg, out);
}
int zk_bn254_g1_msm_dev(const void* d_points, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, zk_g1_affine* out_host, void* stream) {
    return msm_dev_impl<0>(d_points, d_scalars, n, cfg, out_host, stream, 0);
}
int zk_bn254_g2_msm_dev(const void* d_points, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, zk_g2_affine* out_host, void* stream) {
    return msm_dev_impl<1>(d_points, d_scalars, n, cfg, out_host, stream, 0);
}
int zk_bn254_g1_msm_partial_dev(const void* d_points, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, uint64_t out_xyzz[16], void* stream) {
    return msm_dev_impl<0>(d_points, d_scalars, n, cfg, out_xyzz, stream, 1);
}
int zk_bn254_g2_msm_partial_dev(const void* d_points, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, uint64_t out_xyzz[32], void* stream) {
    return msm_dev_impl<1>(d_points, d_scalars, n, cfg, out_xyzz, stream, 1);
}
int zk_bn254_g1_sum_xyzz(const uint64_t* partials, size_t n_partials, zk_g1_affine* out) {
    if (!out || (n_partials && !partials)) return set_err(ZK_ERR_ARG, "null pointer");
    XYZZ<HFp> t = XYZZ<HFp>::inf();
    for (size_t i = 0; i < n_partials; i++) {
        XYZZ<HFp> p;
        memcpy(&p, partials + 16 * i, sizeof(p));
        t.add(p);
    }
    write_affine(t, out);
    return ZK_OK;
}
int zk_bn254_g2_sum_xyzz(const uint64_t* partials, size_t n_partials, zk_g2_affine* out) {
    if (!out || (n_partials && !partials)) return set_err(ZK_ERR_ARG, "null pointer");
    XYZZ<HFp2> t = XYZZ<HFp2>::inf();
    for (size_t i = 0; i < n_partials; i++) {
        XYZZ<HFp2> p;
        memcpy(&p, partials + 32 * i, sizeof(p));
        t.add(p);
    }
    write_affine(t, out);
    return ZK_OK;
}

// table_c: 0 = planner's choice (tables for >= 4096 bases when they fit), -1 = no tables, else an explicit window width (any n)
static int bases_register_here(const void* points, size_t n, int is_g2, uint64_t* handle, hipMemcpyKind kind, int table_c);
static int bases_register(const void* points, size_t n, int is_g2, uint64_t* handle, hipMemcpyKind kind, int table_c) {
    if (!handle || (n && !points)) return set_err(ZK_ERR_ARG, "null pointer");
    if (table_c != 0 && table_c != -1 && (table_c < 8 || table_c > 24)) return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", table_c);
    ZK_TRY(ensure_init());
    if (md_default_mask()) {  // the process named several GPUs: bases of a size that pays for it are kept by range on all of them (a composite handle)
        std::vector<int> ents;
        ZK_TRY(md_entries_for(0, n, (size_t)1 << 16, &ents));
        if (ents.size() > 1) return md_bases_register(points, n, is_g2, kind == hipMemcpyDeviceToDevice, table_c, ents, handle);
    }
    return bases_register_here(points, n, is_g2, handle, kind, table_c);
}
}  // extern "C"
namespace zkmi {
int bases_register_on_this_entry(const void* points, size_t n, int is_g2, int on_device, int table_c, uint64_t* handle) {
    return bases_register_here(points, n, is_g2, handle, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, table_c);
}
}  // namespace zkmi
// the window table of a base array that is in HBM already: b->d_table / b->tab (left empty when the planner's table does not fit and none was asked for by width)
static int bases_make_table(Bases* b, int table_c) {
    const size_t n = b->n;
    const int is_g2 = b->is_g2;
    MsmTable tab;
    tab.c = table_c > 0 ? (unsigned)table_c : msm_pick_window_table(n);
    tab.stride = n;
    // experiment switches: level sizes of the reduction tail of commits against registered bases (0 = the latency-structured default)
    tab.l1_m = (unsigned)ZK_EXP("ZKMI_BASES_L1M", 0);
    tab.l2_m = (unsigned)ZK_EXP("ZKMI_BASES_L2M", 0);
    const size_t Wd = (255 + tab.c - 1) / tab.c, tbytes = Wd * n * (is_g2 ? 128 : 64);
    size_t free_b = 0, total_b = 0;
    ZK_HIP(hipMemGetInfo(&free_b, &total_b));
    if (tbytes < free_b / 2 && tbytes <= ((size_t)64 << 30)) {
        SlotGuard g;
        ZK_TRY(acquire_slot(&g.s));
        void* d_table = nullptr;
        ZK_HIP(hipMalloc(&d_table, tbytes));
        int rc = is_g2 ? msm_build_table_g2(g.s, g.s->stream, b->d, n, n, 0, tab.c, d_table) : msm_build_table_g1(g.s, g.s->stream, b->d, n, n, 0, tab.c, d_table);
        if (rc == ZK_OK) rc = slot_sync(g.s, g.s->stream);
        if (rc != ZK_OK) { (void)hipFree(d_table); return rc; }
        b->tab = tab;
        b->d_table = d_table;
    } else if (table_c > 0) {
        return set_err(ZK_ERR_HIP, "window tables of %zu bytes do not fit (free HBM %zu)", tbytes, free_b);
    }
    return ZK_OK;
}
extern "C" {
static int bases_register_here(const void* points, size_t n, int is_g2, uint64_t* handle, hipMemcpyKind kind, int table_c) {
    if (!handle || (n && !points)) return set_err(ZK_ERR_ARG, "null pointer");
    if (table_c != 0 && table_c != -1 && (table_c < 8 || table_c > 24)) return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", table_c);
    ZK_TRY(ensure_init());
    Bases b;
    b.n = n;
    b.is_g2 = is_g2 ? 1 : 0;
    size_t bytes = n * (is_g2 ? 128 : 64);
    ZK_HIP(hipMalloc(&b.d, bytes ? bytes : 16));
    struct Cleanup {  // an early return must not leak the bases / the table
        Bases* b;
        ~Cleanup() { if (b) { (void)hipFree(b->d); if (b->d_table) (void)hipFree(b->d_table); } }
    } cleanup{&b};
    if (bytes) ZK_HIP(hipMemcpy(b.d, points, bytes, kind));
    if (n && table_c != -1 && (table_c > 0 || n >= 4096)) ZK_TRY(bases_make_table(&b, table_c));
    std::lock_guard<std::mutex> lk(g_bases_mu);
    *handle = hmake(g_next_handle++);
    g_bases[*handle] = b;
    cleanup.b = nullptr;
    return ZK_OK;
}
int zk_bn254_bases_register(const void* points, size_t n, int is_g2, uint64_t* handle) { return bases_register(points, n, is_g2, handle, hipMemcpyHostToDevice, 0); }
int zk_bn254_bases_register_dev(const void* d_points, size_t n, int is_g2, uint64_t* handle) {
    return bases_register(d_points, n, is_g2, handle, hipMemcpyDeviceToDevice, 0);
}
int zk_bn254_bases_register_cfg(const void* points, size_t n, int is_g2, int on_device, int table_window_bits, uint64_t* handle) {
    return bases_register(points, n, is_g2, handle, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, table_window_bits);
}
int zk_bn254_bases_free(uint64_t handle) {
    if (md_is_composite(handle)) return md_bases_free(handle);
    ZK_ON_ENTRY_OF(handle);
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(handle);
    if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
    (void)hipFree(it->second.d);
    if (it->second.d_table) (void)hipFree(it->second.d_table);
    g_bases.erase(it);
    return ZK_OK;
}
// Window tables for a base array that was registered without them (table_window_bits as in zk_bn254_bases_register_cfg; 0 = the planner's width, and nothing
// happens below 4096 bases).  Multi-exps that are running keep the geometry they started with; later ones find the table.  A handle that has one: ZK_OK, untouched.
int zk_bn254_bases_build_table(uint64_t handle, int table_window_bits) {
    if (table_window_bits != 0 && (table_window_bits < 8 || table_window_bits > 24)) return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", table_window_bits);
    if (md_is_composite(handle)) return md_bases_build_table(handle, table_window_bits);
    ZK_ON_ENTRY_OF(handle);
    Bases b;
    {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        auto it = g_bases.find(handle);
        if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
        b = it->second;
    }
    if (b.d_table || !b.n || (table_window_bits == 0 && b.n < 4096)) return ZK_OK;
    ZK_TRY(bases_make_table(&b, table_window_bits));
    if (!b.d_table) return ZK_OK;
    std::lock_guard<std::mutex> lk(g_bases_mu);
    auto it = g_bases.find(handle);
    if (it == g_bases.end() || it->second.d_table) {  // freed, or built by another caller meanwhile
        (void)hipFree(b.d_table);
        return it == g_bases.end() ? set_err(ZK_ERR_HANDLE, "bases handle %llu was freed while its table was built", (unsigned long long)handle) : ZK_OK;
    }
    it->second.d_table = b.d_table;
    it->second.tab = b.tab;
    return ZK_OK;
}
int zk_bn254_bases_build_table_background(uint64_t handle, int table_window_bits) {
    if (table_window_bits != 0 && (table_window_bits < 8 || table_window_bits > 24)) return set_err(ZK_ERR_ARG, "table_window_bits = %d outside [8, 24]", table_window_bits);
    ZK_TRY(ensure_init());
    bg_submit([handle, table_window_bits] {
        if (bg_cancelled()) return;
        const auto t0 = std::chrono::steady_clock::now();
        (void)zk_bn254_bases_build_table(handle, table_window_bits);  // (a handle freed meanwhile: ZK_ERR_HANDLE, dropped)
        prof_host("export.srs_window_tables", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    });
    return ZK_OK;
}
static int msm_bases(uint64_t handle, size_t offset, const void* scalars, size_t n, const zk_msm_cfg* cfg, void* out, hipMemcpyKind kind) {
    if (md_is_composite(handle)) {
        ZK_TRY(check_cfg(cfg));
        return md_msm_bases(handle, offset, scalars, n, cfg, out, kind == hipMemcpyDeviceToDevice);
    }
    ZK_ON_ENTRY_OF(handle);
    Bases b;
    {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        auto it = g_bases.find(handle);
        if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
        b = it->second;
    }
    if (offset + n > b.n) return set_err(ZK_ERR_LEN, "len(points) != len(scalars): offset %zu + n %zu exceeds the %zu registered bases", offset, n, b.n);
    ZK_TRY(check_cfg(cfg));
    if (!out || (n && !scalars)) return set_err(ZK_ERR_ARG, "null pointer");
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    if (b.d_table && n && !(cfg && cfg->window_bits)) {  // window-table path (an explicit window width in cfg selects the plain method)
        size_t np = 0, na1 = 0, na2 = 0;
        ZK_TRY(msm_prep_need_table(n, b.tab, st, &np, &na1, &na2));
        ZK_TRY(g.s->reserve(n * 32 + 1024 + np + (b.is_g2 ? na2 : na1)));
        void* d_sc = g.s->alloc(n * 32 + 16);
        if (kind == hipMemcpyDeviceToDevice) d_sc = const_cast<void*>(scalars);  // already in HBM: read in place
        else ZK_HIP(hipMemcpyAsync(d_sc, scalars, n * 32, kind, st));
        MsmPrep prep;
        ZK_TRY(msm_prepare_scalars_table(g.s, st, d_sc, n, cfg, b.tab, &prep));
        MsmJob job;
        static const bool dev_turnstile = ZK_EXP("ZKMI_DEV_TURNSTILE", 0) != 0;  // A/B: the PLONK prover's thread-per-commit MSMs through the turnstile too
        job.turnstile = kind == hipMemcpyHostToDevice || dev_turnstile;  // host-slice callers are upstream's goroutines; device-pointer callers (the PLONK prover) schedule themselves
        int rc;
        if (b.is_g2) {
            XYZZ<HFp2> t;
            rc = msm_g2_accumulate(g.s, st, prep, (const char*)b.d_table + offset * 128, 0, &job);
            if (rc == ZK_OK) rc = msm_g2_finish(job, &t);
            if (rc == ZK_OK) write_affine(t, (zk_g2_affine*)out);
        } else {
            XYZZ<HFp> t;
            rc = msm_g1_accumulate(g.s, st, prep, (const char*)b.d_table + offset * 64, 0, &job);
            if (rc == ZK_OK) rc = msm_g1_finish(job, &t);
            if (rc == ZK_OK) write_affine(t, (zk_g1_affine*)out);
        }
        if (rc != ZK_OK) (void)hipStreamSynchronize(st);
        msm_prep_release(&prep);
        return rc;
    }
    size_t need = 0;
    ZK_TRY(b.is_g2 ? msm_g2_need(n, cfg, st, &need) : msm_g1_need(n, cfg, st, &need));
    ZK_TRY(g.s->reserve(n * 32 + 1024 + need));
    void* d_s = g.s->alloc(n * 32 + 16);
    if (kind == hipMemcpyDeviceToDevice) d_s = const_cast<void*>(scalars);
    else if (n) ZK_HIP(hipMemcpyAsync(d_s, scalars, n * 32, kind, st));
    if (b.is_g2) {
        XYZZ<HFp2> t;
        ZK_TRY(msm_g2_xyzz(g.s, st, (const char*)b.d + offset * 128, d_s, n, cfg, &t));
        write_affine(t, (zk_g2_affine*)out);
    } else {
        XYZZ<HFp> t;
        ZK_TRY(msm_g1_xyzz(g.s, st, (const char*)b.d + offset * 64, d_s, n, cfg, &t));
        write_affine(t, (zk_g1_affine*)out);
    }
    return ZK_OK;
}
// ---- scalars kept resident and recoded ONCE for every base array they pair with -------------------------------------------------------------------
// groth16.Prove issues five MultiExp calls and four of them -- A, B1, K, G2.B -- pair with the SAME wire values (gnark v0.8.0 groth16 prove.go; reached from
// gnark_backend_ffi/main.go:131).  Through zk_bn254_msm_bases each of them uploads the 32 B x n scalars and recodes them (digits, radix sort, task plan).
// A scalars handle uploads once and keeps one recoding per table geometry (window width, row stride, first scalar): A, B1 and G2.B -- registered over the
// same wires -- share one; K (registered over the n - n_public private wires) gets a second from the resident copy, no second upload.
// Lifetimes (round 5; the advisor's finding on round 4's layout, where all recodings lived in ONE pool slot's arena that a third geometry rebuilt in place under
// readers that had already dropped the lock): every recoding is an allocation of its OWN, shared -- a multi-exp keeps its shared_ptr from the lookup until its sums
// are back on the host, so neither zk_bn254_scalars_free nor a new geometry can free or move what a kernel in flight reads; the scalars handle holds no pool slot
// (preparations borrow one for the time it takes to enqueue them), so any number of registrations can be live next to a proof session.
struct PrepGeo {
    unsigned c = 0;
    size_t stride = 0, skip = 0;
    Slot ws;        // a private workspace (never one of the entry's pool slots): its arena IS this recoding's arrays
    MsmPrep prep;
    ~PrepGeo() {
        msm_prep_release(&prep);
        for (auto& p : ws.pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
        for (hipEvent_t e : ws.free_events) (void)hipEventDestroy(e);
        if (ws.arena) (void)hipFree(ws.arena);
        if (ws.pinned) (void)hipHostFree(ws.pinned);
    }
};
struct PreparedScalars {
    size_t n = 0;
    int mont = 0;
    void* d_sc = nullptr;
    std::mutex mu;  // the list below; a geometry being prepared (callers that arrive together wait for the first one's)
    std::vector<std::shared_ptr<PrepGeo>> geos;
    ~PreparedScalars() {
        geos.clear();
        if (d_sc) (void)hipFree(d_sc);
    }
};
static std::mutex g_ps_mu;
static std::map<uint64_t, std::shared_ptr<PreparedScalars>> g_ps;
static uint64_t g_next_ps = 1;

int zk_bn254_scalars_register(const zk_fr* scalars, size_t n, const zk_msm_cfg* cfg, uint64_t* handle) {
    if (!handle || (n && !scalars)) return set_err(ZK_ERR_ARG, "null pointer");
    ZK_TRY(check_cfg(cfg));
    ZK_TRY(ensure_init());
    auto S = std::make_shared<PreparedScalars>();
    S->n = n;
    S->mont = (cfg && cfg->scalars_mont) ? 1 : 0;
    ZK_HIP(hipMalloc(&S->d_sc, n * 32 + 16));
    {
        SlotGuard g;  // borrowed for the upload only
        ZK_TRY(acquire_slot(&g.s));
        if (n) ZK_HIP(hipMemcpyAsync(S->d_sc, scalars, n * 32, hipMemcpyHostToDevice, g.s->stream));
        ZK_HIP(hipStreamSynchronize(g.s->stream));  // the caller's slice is not ours after the call
    }
    std::lock_guard<std::mutex> lk(g_ps_mu);
    *handle = hmake(g_next_ps++);
    g_ps[*handle] = S;
    return ZK_OK;
}
// The handle is gone at once; the resident copy and its recodings go when the last multi-exp still reading them has its result.
int zk_bn254_scalars_free(uint64_t handle) {
    ZK_ON_ENTRY_OF(handle);
    std::shared_ptr<PreparedScalars> S;
    {
        std::lock_guard<std::mutex> lk(g_ps_mu);
        auto it = g_ps.find(handle);
        if (it == g_ps.end()) return set_err(ZK_ERR_HANDLE, "unknown scalars handle %llu", (unsigned long long)handle);
        S = it->second;
        g_ps.erase(it);
    }
    return ZK_OK;  // ~PreparedScalars runs here unless a call in flight still holds S
}
// out = sum_{i >= skip} scalars[i] * bases[bases_offset + (i - skip)]   (skip: Groth16's K pairs with the wire values from the first private wire on)
int zk_bn254_msm_bases_prepared(uint64_t bases, size_t bases_offset, uint64_t scalars_handle, size_t skip, const zk_msm_cfg* cfg, void* out) {
    if (!out) return set_err(ZK_ERR_ARG, "null pointer");
    if (md_is_composite(bases)) return set_err(ZK_ERR_ARG, "prepared scalars pair with bases on ONE device entry (register them with the process default on a single entry)");
    ZK_ON_ENTRY_OF(bases);
    if (hentry(scalars_handle) != hentry(bases)) return set_err(ZK_ERR_ARG, "the scalars live on device entry %d, the bases on entry %d", hentry(scalars_handle), hentry(bases));
    ZK_TRY(check_cfg(cfg));
    Bases b;
    {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        auto it = g_bases.find(bases);
        if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)bases);
        b = it->second;
    }
    std::shared_ptr<PreparedScalars> S;
    {
        std::lock_guard<std::mutex> lk(g_ps_mu);
        auto it = g_ps.find(scalars_handle);
        if (it == g_ps.end()) return set_err(ZK_ERR_HANDLE, "unknown scalars handle %llu", (unsigned long long)scalars_handle);
        S = it->second;
    }
    if (skip > S->n) return set_err(ZK_ERR_LEN, "len(points) != len(scalars): the first scalar %zu is beyond the %zu registered", skip, S->n);
    const size_t n = S->n - skip;
    if (bases_offset + n > b.n) return set_err(ZK_ERR_LEN, "len(points) != len(scalars): offset %zu + n %zu exceeds the %zu registered bases", bases_offset, n, b.n);
    zk_msm_cfg c1 = cfg ? *cfg : zk_msm_cfg{0, 0, 0, 0};
    c1.scalars_mont = S->mont;  // the form they were registered in
    const size_t esz = b.is_g2 ? 128 : 64;
    if (!n) { memset(out, 0, esz); return ZK_OK; }
    if (!b.d_table || c1.window_bits) {  // no window table to share a recoding through: the plain method on the resident scalars (still no upload)
        return msm_bases(bases, bases_offset, (const char*)S->d_sc + skip * 32, n, &c1, out, hipMemcpyDeviceToDevice);
    }
    // the recoding for this table geometry: found, or made now from the resident copy (callers that arrive together wait for the first one's)
    std::shared_ptr<PrepGeo> G;  // held until this call's sums are on the host: what the accumulate kernel reads cannot go away under it
    uint32_t skip_below = 0;
    const char* table = nullptr;
    {
        std::lock_guard<std::mutex> lk(S->mu);
        // one recoding per (window width, row stride, first scalar): the recoded values are table indices w * stride + i, i counted from the first scalar, so
        // every base array registered with that stride shares it -- whatever its bases_offset, which only moves the table pointer.  (The accumulate kernels'
        // skip_below compares whole table indices: it cannot drop the first scalars of rows w > 0, so a `skip` is a recoding of its own.)
        for (auto& g : S->geos)
            if (g->c == b.tab.c && g->stride == b.tab.stride && g->skip == skip) G = g;
        if (!G) {
            const size_t cnt = S->n - skip;
            auto N = std::make_shared<PrepGeo>();
            N->c = b.tab.c;
            N->stride = b.tab.stride;
            N->skip = skip;
            SlotGuard borrowed;  // its stream carries the preparation; the arrays live in N->ws
            ZK_TRY(acquire_slot(&borrowed.s));
            hipStream_t pst = borrowed.s->stream;
            N->ws.stream = pst;
            N->ws.owner = &ctx();
            size_t np = 0, na1 = 0, na2 = 0;
            ZK_TRY(msm_prep_need_table(cnt, b.tab, pst, &np, &na1, &na2));
            ZK_TRY(N->ws.reserve(np + 16384));
            ZK_TRY(msm_prepare_scalars_table(&N->ws, pst, (const char*)S->d_sc + skip * 32, cnt, &c1, b.tab, &N->prep, true));  // registered scalars are wire values: zero digits dropped
            if (profiling_on()) ZK_TRY(slot_sync(&N->ws, pst));  // the event pairs of these launches sit in the private workspace: folded here
            N->ws.stream = nullptr;  // (not ours; prep.ready orders the readers behind the preparation)
            S->geos.push_back(N);
            G = N;
        }
        table = (const char*)b.d_table + bases_offset * esz;  // index i of the recoding is base bases_offset + i
    }
    const MsmPrep& prep = G->prep;
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    size_t np = 0, na1 = 0, na2 = 0;
    ZK_TRY(msm_prep_need_table(prep.n, b.tab, st, &np, &na1, &na2));
    ZK_TRY(g.s->reserve((b.is_g2 ? na2 : na1) + 4096));
    MsmJob job;
    job.turnstile = true;
    int rc;
    if (b.is_g2) {
        XYZZ<HFp2> t;
        rc = msm_g2_accumulate(g.s, st, prep, table, skip_below, &job);
        if (rc == ZK_OK) rc = msm_g2_finish(job, &t);
        if (rc == ZK_OK) write_affine(t, (zk_g2_affine*)out);
    } else {
        XYZZ<HFp> t;
        rc = msm_g1_accumulate(g.s, st, prep, table, skip_below, &job);
        if (rc == ZK_OK) rc = msm_g1_finish(job, &t);
        if (rc == ZK_OK) write_affine(t, (zk_g1_affine*)out);
    }
    if (rc != ZK_OK) (void)hipStreamSynchronize(st);
    return rc;
}

int zk_bn254_msm_bases(uint64_t handle, size_t offset, const zk_fr* scalars, size_t n, const zk_msm_cfg* cfg, void* out) {
    return msm_bases(handle, offset, scalars, n, cfg, out, hipMemcpyHostToDevice);
}
int zk_bn254_msm_bases_dev(uint64_t handle, size_t offset, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, void* out) {
    return msm_bases(handle, offset, d_scalars, n, cfg, out, hipMemcpyDeviceToDevice);
}

// ---- several scalar vectors against ONE registered base array ------------------------------------------------------------------------------------------
// plonk.Prove commits l, r, o -- and later h1, h2, h3 -- against the same SRS at the same moment (gnark v0.8.0 backend/plonk/bn254/prove.go, reached from
// gnark_backend_ffi/backend/plonk/plonk.go:53-73).  With a G1 window table on one device entry the `count` (<= 3) multi-exps are ONE multi-scalar
// multiplication with one bucket set per vector: one digit pass over the vectors, one sort, one task plan, ONE accumulate launch, one reduction per set.
// Everything else (G2, no table, a composite handle, an explicit window width, count > 3) runs the vectors one after the other through msm_bases: same points out.
static int msm_bases_batch(uint64_t handle, size_t offset, const void* const* scalars, unsigned count, size_t n, const zk_msm_cfg* cfg, void* out, hipMemcpyKind kind,
                           bool sparse = false) {
    if (!count) return ZK_OK;
    if (!scalars || !out) return set_err(ZK_ERR_ARG, "null pointer");
    for (unsigned k = 0; k < count; k++)
        if (n && !scalars[k]) return set_err(ZK_ERR_ARG, "null scalar vector %u", k);
    ZK_TRY(check_cfg(cfg));
    bool batched = count >= 2 && count <= 3 && n && !md_is_composite(handle) && !(cfg && cfg->window_bits);
    Bases b;
    if (batched) {
        std::lock_guard<std::mutex> lk(g_bases_mu);
        auto it = g_bases.find(handle);
        if (it == g_bases.end()) return set_err(ZK_ERR_HANDLE, "unknown bases handle %llu", (unsigned long long)handle);
        b = it->second;
        batched = b.d_table && !b.is_g2;
    }
    if (!batched) {
        int is_g2 = 0;
        size_t nb = 0;
        ZK_TRY(bases_info(handle, &nb, &is_g2));
        const size_t osz = is_g2 ? sizeof(zk_g2_affine) : sizeof(zk_g1_affine);
        for (unsigned k = 0; k < count; k++) ZK_TRY(msm_bases(handle, offset, scalars[k], n, cfg, (char*)out + k * osz, kind));
        return ZK_OK;
    }
    ZK_ON_ENTRY_OF(handle);
    if (offset + n > b.n) return set_err(ZK_ERR_LEN, "len(points) != len(scalars): offset %zu + n %zu exceeds the %zu registered bases", offset, n, b.n);
    SlotGuard g;
    ZK_TRY(acquire_slot(&g.s));
    hipStream_t st = g.s->stream;
    size_t np = 0, na = 0;
    ZK_TRY(msm_prep_need_table_batch(n, count, b.tab, st, &np, &na));
    const bool up = kind != hipMemcpyDeviceToDevice;
    ZK_TRY(g.s->reserve((up ? count * (n * 32 + 1024) : 0) + np + na + 65536 + (sparse ? msm_compact_need(n, count) : 0)));
    const void* sc[3] = {nullptr, nullptr, nullptr};
    for (unsigned k = 0; k < count; k++) {
        if (up) {
            void* d = g.s->alloc(n * 32 + 16);
            ZK_HIP(hipMemcpyAsync(d, scalars[k], n * 32, kind, st));
            sc[k] = d;
        } else sc[k] = scalars[k];
    }
    MsmPrep prep;
    ZK_TRY(msm_prepare_scalars_table_batch(g.s, st, sc, count, n, cfg, b.tab, &prep, sparse));
    MsmJob job;
    job.turnstile = up;  // as in msm_bases: host-slice callers are upstream's goroutines
    XYZZ<HFp> t[3];
    int rc = msm_g1_accumulate(g.s, st, prep, (const char*)b.d_table + offset * 64, 0, &job);
    if (rc == ZK_OK) rc = msm_g1_finish_batch(job, t);
    if (rc != ZK_OK) (void)hipStreamSynchronize(st);
    msm_prep_release(&prep);
    ZK_TRY(rc);
    for (unsigned k = 0; k < count; k++) write_affine(t[k], (zk_g1_affine*)out + k);
    return ZK_OK;
}
}  // extern "C"
namespace zkmi {
// the same for scalars known to be mostly small (the PLONK prover's wire values): the zero digits are dropped before the sort (k_msm_digit_count)
int msm_bases_batch_dev_sparse(uint64_t handle, size_t offset, const void* const* d_scalars, unsigned count, size_t n, const zk_msm_cfg* cfg, void* out) {
    return msm_bases_batch(handle, offset, d_scalars, count, n, cfg, out, hipMemcpyDeviceToDevice, true);
}
}  // namespace zkmi
extern "C" {
int zk_bn254_msm_bases_batch(uint64_t handle, size_t offset, const zk_fr* const* scalars, uint32_t count, size_t n, const zk_msm_cfg* cfg, void* out) {
    return msm_bases_batch(handle, offset, (const void* const*)scalars, count, n, cfg, out, hipMemcpyHostToDevice);
}
int zk_bn254_msm_bases_batch_dev(uint64_t handle, size_t offset, const void* const* d_scalars, uint32_t count, size_t n, const zk_msm_cfg* cfg, void* out) {
    return msm_bases_batch(handle, offset, d_scalars, count, n, cfg, out, hipMemcpyDeviceToDevice);
}

}  // extern "C"
