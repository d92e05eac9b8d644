// Radix sort of (bucket key, point index | sign) pairs for the MSM's scalar preparation -- hand-written for gfx950, no library call.
// Part of the replacement of gnark-crypto v0.9.1 `MultiExp` (ecc/bn254/multiexp.go: partitionScalars + one goroutine per window walking ALL points;
// reached from /root/reference/gnark_backend_ffi/main.go:131 and backend/plonk/plonk.go:21,67): here every bucket becomes a contiguous run instead.
//
// Why not rocPRIM's radix_sort_pairs (rounds 1-2): its onesweep passes chain the tiles of a pass through decoupled look-back -- a tile SPINS until its
// predecessor has published a prefix.  Alone on the GPU that is the fastest known scheme (0.39 ms for the 13.6 M digits of a 2^20-point MSM); inside a proof
// the sort shares every SIMD with the long-lived waves of an accumulate kernel, the spinning tiles and their predecessors fight for the issue slots those
// waves leave, and the same sort takes 3-5 ms.  Nothing here waits for another workgroup:
//
//   k_rs_hist      ONE pass over the keys: the digit histograms of ALL passes (global, 3 x 256 counters) and the per-tile histogram of pass 0
//   k_rs_bases     exclusive scan of every pass's bins (tiny)
//   per pass p:    k_rs_tile_hist (p > 0: per-tile histogram of digit p, bin-major)  ->  k_rs_scan_rows (exclusive scan along the tiles, one workgroup
//                  per bin)  ->  k_rs_scatter (rank inside the tile by wave-wide digit matching -- ballots, no atomics -- stage the tile in LDS in digit
//                  order, write runs of equal digits to consecutive addresses)
//
// LSD, stable: inside a tile items keep memory order (wave, round, lane), tiles keep theirs through the row scans.  Digit widths are balanced over
// ceil(key_bits / 8) passes (19 bits: 7 + 6 + 6).  Traffic per pass: 4 B (tile histogram) + 8 B in + 8 B out per pair -- 25 % more than onesweep's, paid
// for kernels whose progress does not depend on when their neighbours get scheduled.  Every kernel raises its wave priority (prio_hi, ctx.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctx.hpp"

namespace zkmi {

constexpr unsigned RS_THREADS = 512, RS_WAVES = RS_THREADS / 64, RS_IPT = 16, RS_TILE = RS_THREADS * RS_IPT;  // 8192 pairs per tile
constexpr unsigned RS_MAX_BITS = 8, RS_MAX_BINS = 1u << RS_MAX_BITS, RS_MAX_PASSES = 4;

struct RsPlan {
    unsigned npass = 0;
    unsigned shift[RS_MAX_PASSES] = {}, bits[RS_MAX_PASSES] = {};
    size_t ntiles = 0;
    size_t tmp_bytes = 0;  // tile histograms + global histograms + bases
};
static inline RsPlan rs_plan(size_t n, unsigned key_bits) {
    RsPlan P;
    if (key_bits == 0) key_bits = 1;
    P.npass = (key_bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
    unsigned at = 0;
    for (unsigned p = 0; p < P.npass; p++) {  // balanced widths, the wider digits first (least significant)
        const unsigned left = key_bits - at, passes_left = P.npass - p;
        P.bits[p] = (left + passes_left - 1) / passes_left;
        P.shift[p] = at;
        at += P.bits[p];
    }
    P.ntiles = (n + RS_TILE - 1) / RS_TILE;
    P.tmp_bytes = ((size_t)RS_MAX_BINS * P.ntiles + 2 * RS_MAX_PASSES * RS_MAX_BINS) * 4 + 256;
    return P;
}
struct RsArgs {
    unsigned npass;
    unsigned shift[RS_MAX_PASSES], bits[RS_MAX_PASSES];
};

// all digit histograms in one pass over the keys; tile_hist is bin-major: [bin * ntiles + tile]
// d_n (every kernel that takes it; may be null): the number of pairs as the DEVICE knows it -- a preparation that drops the zero digits learns its length from a
// scan and must not make the host wait for it.  The launch is sized for `n` (the most there can be); tiles past *d_n find nothing to do.
__device__ __forceinline__ uint32_t rs_len(uint32_t n, const uint32_t* __restrict__ d_n) {
    if (!d_n) return n;
    const uint32_t m = *d_n;
    return m < n ? m : n;
}
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, RsArgs A, uint32_t ntiles, uint32_t* __restrict__ ghist,
                                                        uint32_t* __restrict__ tile_hist0, const uint32_t* __restrict__ d_n) {
    prio_hi();
    n = rs_len(n, d_n);
    __shared__ uint32_t h[RS_MAX_PASSES][RS_MAX_BINS];
    for (unsigned i = threadIdx.x; i < RS_MAX_PASSES * RS_MAX_BINS; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll
    for (unsigned k = 0; k < RS_IPT; k++) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t key = keys[i];
            for (unsigned p = 0; p < A.npass; p++) atomicAdd(&h[p][(key >> A.shift[p]) & ((1u << A.bits[p]) - 1)], 1u);
        }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < RS_MAX_BINS; i += RS_THREADS) {
        for (unsigned p = 0; p < A.npass; p++)
            if (h[p][i]) atomicAdd(&ghist[p * RS_MAX_BINS + i], h[p][i]);
        if (i < (1u << A.bits[0])) tile_hist0[(size_t)i * ntiles + blockIdx.x] = h[0][i];
    }
}

// gbase[p][b] = number of keys whose digit p is below b
__global__ __launch_bounds__(RS_MAX_BINS) void k_rs_bases(const uint32_t* __restrict__ ghist, uint32_t* __restrict__ gbase, unsigned npass) {
    prio_hi();
    __shared__ uint32_t s[RS_MAX_BINS];
    for (unsigned p = 0; p < npass; p++) {
        const uint32_t v = ghist[p * RS_MAX_BINS + threadIdx.x];
        s[threadIdx.x] = v;
        __syncthreads();
        for (unsigned d = 1; d < RS_MAX_BINS; d <<= 1) {
            const uint32_t add = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        gbase[p * RS_MAX_BINS + threadIdx.x] = s[threadIdx.x] - v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_tile_hist(const uint32_t* __restrict__ keys, uint32_t n, unsigned shift, unsigned bits, uint32_t ntiles,
                                                             uint32_t* __restrict__ tile_hist, const uint32_t* __restrict__ d_n) {
    prio_hi();
    n = rs_len(n, d_n);
    __shared__ uint32_t h[RS_MAX_BINS];
    for (unsigned i = threadIdx.x; i < RS_MAX_BINS; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE, mask = (1u << bits) - 1;
#pragma unroll
    for (unsigned k = 0; k < RS_IPT; k++) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i <= mask; i += RS_THREADS) tile_hist[(size_t)i * ntiles + blockIdx.x] = h[i];
}

// in-place exclusive scan of row blockIdx.x (one bin) along the tiles
__global__ __launch_bounds__(256) void k_rs_scan_rows(uint32_t* __restrict__ tile_hist, uint32_t ntiles) {
    prio_hi();
    __shared__ uint32_t part[256];
    uint32_t* row = tile_hist + (size_t)blockIdx.x * ntiles;
    const uint32_t per = (ntiles + 255) / 256, lo = threadIdx.x * per, hi = min(lo + per, ntiles);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += row[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
}

// One tile: rank, stage in LDS in digit order, write out.  Wave w owns the tile's pairs [w * 1024, (w + 1) * 1024) in 16 rounds of 64 consecutive ones, so the
// order (wave, round, lane) IS memory order.  In a round the lanes with equal digits find each other with one ballot per digit bit; the rank inside the wave is
// the digit's running count (an LDS word per wave and digit, written by one lane of the group) plus the lanes of the group below this one.
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, uint32_t n, unsigned shift, unsigned bits, uint32_t ntiles,
                                                           const uint32_t* __restrict__ tile_prefix, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ d_n) {
    prio_hi();
    n = rs_len(n, d_n);
    if (blockIdx.x * RS_TILE >= n) return;  // (uniform over the workgroup: before any barrier)
    __shared__ uint32_t cnt[RS_WAVES][RS_MAX_BINS];
    __shared__ uint32_t off[RS_MAX_BINS];   // first staged slot of a digit; then (global destination of the digit's first pair of this tile) - off
    __shared__ uint32_t scan[RS_MAX_BINS];
    __shared__ uint32_t stage_k[RS_TILE], stage_v[RS_TILE];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nbins = 1u << bits, mask = nbins - 1;
    for (unsigned i = threadIdx.x; i < RS_WAVES * RS_MAX_BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile_base = blockIdx.x * RS_TILE, wbase = tile_base + wave * (RS_TILE / RS_WAVES);
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t key[RS_IPT], val[RS_IPT], rank[RS_IPT];
#pragma unroll
    for (unsigned r = 0; r < RS_IPT; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0u;
        val[r] = valid ? vals_in[i] : 0u;
    }
#pragma unroll
    for (unsigned r = 0; r < RS_IPT; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (key[r] >> shift) & mask;
        uint64_t m = __ballot(valid);
        for (unsigned b = 0; b < bits; b++) {
            const bool bit = (d >> b) & 1;
            const uint64_t bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        rank[r] = 0;
        if (valid) {
            // volatile: the word is rewritten by ANOTHER lane of this wave between rounds -- the compiler must neither keep it in a register nor reorder the pair
            volatile uint32_t* c = &cnt[wave][d];
            const uint32_t prev = *c;
            rank[r] = prev + (uint32_t)__popcll(m & lt);
            if ((m & lt) == 0) *c = prev + (uint32_t)__popcll(m);  // the group's lowest lane; every lane of the group has read `prev` (same wave, program order)
        }
    }
    __syncthreads();
    // per digit: counts of the waves -> exclusive prefix over the waves, total -> exclusive scan over the digits
    uint32_t total = 0;
    if (threadIdx.x < nbins) {
        for (unsigned w = 0; w < RS_WAVES; w++) {
            const uint32_t c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = total;
            total += c;
        }
    }
    if (threadIdx.x < RS_MAX_BINS) scan[threadIdx.x] = threadIdx.x < nbins ? total : 0;
    __syncthreads();
    for (unsigned d = 1; d < RS_MAX_BINS; d <<= 1) {
        uint32_t add = 0;
        if (threadIdx.x < RS_MAX_BINS && threadIdx.x >= d) add = scan[threadIdx.x - d];
        __syncthreads();
        if (threadIdx.x < RS_MAX_BINS) scan[threadIdx.x] += add;
        __syncthreads();
    }
    if (threadIdx.x < nbins) off[threadIdx.x] = scan[threadIdx.x] - total;
    __syncthreads();
#pragma unroll
    for (unsigned r = 0; r < RS_IPT; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t lp = off[d] + cnt[wave][d] + rank[r];
            stage_k[lp] = key[r];
            stage_v[lp] = val[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < nbins) off[threadIdx.x] = gbase[threadIdx.x] + tile_prefix[(size_t)threadIdx.x * ntiles + blockIdx.x] - off[threadIdx.x];
    __syncthreads();
    const uint32_t count = min((uint32_t)RS_TILE, n - tile_base);
#pragma unroll
    for (unsigned k = 0; k < RS_IPT; k++) {
        const uint32_t lp = k * RS_THREADS + threadIdx.x;
        if (lp < count) {
            const uint32_t kk = stage_k[lp];
            const uint32_t dst = off[(kk >> shift) & mask] + lp;
            keys_out[dst] = kk;
            vals_out[dst] = stage_v[lp];
        }
    }
}

// ---------------------------------------------------------------------------------------- exclusive scan (u32), three short launches
// out[i] = sum of in[0 .. i): per-tile sums -> one workgroup scans them -> per-tile scan with its base.  (The library alternative, rocPRIM's single-pass scan,
// chains its tiles through look-back like the sort did.)  in == out is allowed.
constexpr unsigned XS_THREADS = 256, XS_IPT = 8, XS_TILE = XS_THREADS * XS_IPT;
static inline size_t xs_tmp_bytes(size_t n) { return ((n + XS_TILE - 1) / XS_TILE + 1) * 4 + 256; }

__device__ __forceinline__ uint32_t xs_block_exclusive(uint32_t v, uint32_t* total) {  // exclusive scan of one value per thread over the workgroup
    __shared__ uint32_t sh[XS_THREADS];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (unsigned d = 1; d < XS_THREADS; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t incl = sh[threadIdx.x];
    if (total) *total = sh[XS_THREADS - 1];
    __syncthreads();
    return incl - v;
}
__global__ __launch_bounds__(XS_THREADS) void k_xs_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums) {
    prio_hi();
    const uint32_t base = blockIdx.x * XS_TILE + threadIdx.x * XS_IPT;
    uint32_t v = 0;
#pragma unroll
    for (unsigned k = 0; k < XS_IPT; k++)
        if (base + k < n) v += in[base + k];
    uint32_t total;
    (void)xs_block_exclusive(v, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(XS_THREADS) void k_xs_scan_sums(uint32_t* __restrict__ sums, uint32_t ntiles) {
    prio_hi();
    uint32_t carry = 0;
    for (uint32_t lo = 0; lo < ntiles; lo += XS_THREADS) {
        const uint32_t i = lo + threadIdx.x;
        const uint32_t v = i < ntiles ? sums[i] : 0;
        uint32_t total;
        const uint32_t ex = xs_block_exclusive(v, &total);
        if (i < ntiles) sums[i] = carry + ex;
        carry += total;
    }
}
__global__ __launch_bounds__(XS_THREADS) void k_xs_apply(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ sums, uint32_t* __restrict__ out) {
    prio_hi();
    const uint32_t base = blockIdx.x * XS_TILE + threadIdx.x * XS_IPT;
    uint32_t x[XS_IPT], v = 0;
#pragma unroll
    for (unsigned k = 0; k < XS_IPT; k++) {
        x[k] = base + k < n ? in[base + k] : 0;
        v += x[k];
    }
    uint32_t run = sums[blockIdx.x] + xs_block_exclusive(v, nullptr);
#pragma unroll
    for (unsigned k = 0; k < XS_IPT; k++) {
        if (base + k < n) out[base + k] = run;
        run += x[k];
    }
}

}  // namespace zkmi
