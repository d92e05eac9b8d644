// Internal interface of the MSM module (msm.hip) used by the Groth16 prover (groth16.hip).
#pragma once
#include "ctx.hpp"
#include "curve.hpp"
#include "host_ff.hpp"

namespace zkmi {

unsigned msm_pick_window(size_t n);
// Bytes of arena an MSM of n points needs; reserve them (plus anything else the call carves out) up front.
int msm_g1_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need);
int msm_g2_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need);
// Precomputed-table mode for resident bases: T[w * stride + i] = 2^(c*w) * P_i; all windows then share one bucket set.
struct MsmTable {
    unsigned c = 0;
    size_t stride = 0;
    // window (table-row) sharding across GPUs: this table holds only the rows w = row_first + k * row_step (k = 0, 1, ...) of the ceil(255/c)
    // digit windows, for ALL points; the MSM over it is the partial sum of those windows (the others belong to other ranks)
    unsigned row_first = 0, row_step = 1;
    unsigned rows() const {
        unsigned Wd = (255 + c - 1) / c;
        return row_first < Wd ? (Wd - 1 - row_first) / row_step + 1 : 0;
    }
    unsigned l2_m = 0;  // > 0: a second thread-serial level over l2_m entries (3 additions per entry instead of ~15 in a wave level)
    unsigned l1_m = 0;  // buckets per lane in the first reduction level (0: default 8).  16 does 22 % less tail work at twice the level-1
                        // latency: right for MSMs whose tail hides under the next accumulate, wrong for the last one of a proof
};
unsigned msm_pick_window_table(size_t n);
int msm_build_table_g1(Slot* s, hipStream_t st, const void* d_pts, size_t n, size_t stride, size_t offset, unsigned c, void* d_table, unsigned row_first = 0,
                       unsigned row_step = 1);
int msm_build_table_g2(Slot* s, hipStream_t st, const void* d_pts, size_t n, size_t stride, size_t offset, unsigned c, void* d_table, unsigned row_first = 0,
                       unsigned row_step = 1);

// out[i] = src_idx[i] == ~0u ? (0,0) : compact[src_idx[i]]  (gnark's InfinityA / InfinityB compaction undone at key load)
int msm_expand_bases(Slot* s, hipStream_t st, int is_g2, const void* d_compact, const uint32_t* d_src_idx, size_t n, void* d_out);

// number of points / group of a registered base array (zk_bn254_bases_register*)
int bases_info(uint64_t handle, size_t* n, int* is_g2);
struct MsmTable;
int bases_table(uint64_t handle, const void** d_table, MsmTable* tab, size_t* n);  // window table of a registered base array (null: none)

struct MsmPlan {
    unsigned c, W, Wd, key_bits;   // W bucket sets (windows that are reduced separately); Wd digit windows (== W unless table mode)
    unsigned Wrows, row_first, row_step;  // table mode: rows of the table this MSM feeds (all Wd unless window-sharded)
    uint32_t table_stride;
    uint32_t B, nb, L, m1, N1, m2 = 0;
    uint32_t Lmin = 0;  // the shortest task length the device may choose (k_pick_len); the task arrays are sized for it
    size_t total, max_tasks, sort_tmp_bytes, scan_tmp_bytes, tsort_tmp_bytes, lvl_elems, need, need_prep, need_acc;
};
// Result of the scalar-side half of an MSM (digits, sort, bucket bounds, task plan); device arrays live in the
// preparing slot's arena and stay valid until that slot's arena is reset.
struct MsmPrep {
    MsmPlan P;
    size_t n = 0;
    bool empty = true;
    const uint32_t *vals = nullptr, *start = nullptr, *task_off = nullptr, *task_begin = nullptr, *lkeys = nullptr, *tids = nullptr,
                   *multi_list = nullptr, *num_multi = nullptr;
    hipEvent_t ready = nullptr;  // recorded on the preparing stream
};

// Asynchronous form: *_launch enqueues the device pipeline on `st` (window sums land in the slot's pinned buffer),
// *_finish synchronises that stream and does the host Horner.  One job per slot at a time.
struct MsmJob {
    Slot* s = nullptr;
    hipStream_t st = nullptr;
    unsigned c = 0, W = 0;
    unsigned n_final = 1, sh_final = 0;  // entries (A_j, S_j) per window left for the host: value = sum A_j + 2^sh * sum j S_j
    bool empty = true;
    // scheduling hooks (set by the caller BEFORE launch): the throughput-bound accumulate kernel waits for `gate_acc`;
    // with `want_done` an event `acc_done` is recorded right after it (the caller destroys it).
    hipEvent_t gate_acc = nullptr;
    bool want_done = false;
    hipEvent_t acc_done = nullptr;
    hipStream_t chain = nullptr;  // run the accumulate kernel on this stream (reduction tail stays on the job's stream)
    bool turnstile = false;       // an independent caller: the accumulate kernel takes its turn behind those of other callers on this device entry
    bool quad_tail = false;       // G1: reduction tail with four lanes per point (3x shorter serial chain, ~1.4x the ALU work): for the
                                  // tail nothing else can hide -- the last MSM of a proof
};
int msm_g1_launch(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmJob* job);
int msm_g2_launch(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmJob* job);
// Split form for MSMs that share one scalar vector: prepare once, accumulate per base array (bases indexed by scalar
// index; indices below `skip_below` are ignored).
int msm_prep_need(size_t n, const zk_msm_cfg* cfg, hipStream_t st, size_t* need_prep, size_t* need_acc_g1, size_t* need_acc_g2);
// drop_zero_digits (here and below): the zero digits never enter the sort -- for scalars that are wire values (0 / 1 / small: most digits are zero); the number of
// pairs stays on the device, nothing synchronises; reserve msm_compact_need(n, sets) more in the preparing slot.  Same sums either way.
int msm_prepare_scalars(Slot* s, hipStream_t st, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, MsmPrep* out, bool drop_zero_digits = false);
int msm_prep_need_table(size_t n, const MsmTable& tab, hipStream_t st, size_t* need_prep, size_t* need_acc_g1, size_t* need_acc_g2);
int msm_prepare_scalars_table(Slot* s, hipStream_t st, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, const MsmTable& tab, MsmPrep* out, bool drop_zero_digits = false);
// a batch of up to three scalar vectors (n elements each) against ONE window table: one recoding, one accumulate launch, one sum per vector
int msm_prep_need_table_batch(size_t n, unsigned sets, const MsmTable& tab, hipStream_t st, size_t* need_prep, size_t* need_acc_g1);
int msm_prepare_scalars_table_batch(Slot* s, hipStream_t st, const void* const* d_scalars, unsigned sets, size_t n, const zk_msm_cfg* cfg, const MsmTable& tab, MsmPrep* out,
                                    bool drop_zero_digits = false);
size_t msm_compact_need(size_t n, unsigned sets);
// zk_bn254_msm_bases_batch_dev for scalars known to be mostly small (wire values): zero digits dropped before the sort
int msm_bases_batch_dev_sparse(uint64_t handle, size_t offset, const void* const* d_scalars, unsigned count, size_t n, const zk_msm_cfg* cfg, void* out);
int msm_g1_finish_batch(const MsmJob& job, XYZZ<HFp> out[3]);
void msm_prep_release(MsmPrep* R);
int msm_g1_accumulate(Slot* s, hipStream_t st, const MsmPrep& R, const void* d_pts, uint32_t skip_below, MsmJob* job);
int msm_g2_accumulate(Slot* s, hipStream_t st, const MsmPrep& R, const void* d_pts, uint32_t skip_below, MsmJob* job);
// up to three G1 MSMs over the same prepared scalars in ONE accumulate launch (jobs[0]->gate_acc gates it; the last job's acc_done owns the event)
int msm_g1_accumulate_batch(int nb, Slot* const* sl, const hipStream_t* sts, const MsmPrep& R, const void* const* d_pts, const uint32_t* skip_below,
                            MsmJob* const* jobs);
int msm_g1_finish(const MsmJob& job, XYZZ<HFp>* out);
int msm_g2_finish(const MsmJob& job, XYZZ<HFp2>* out);
// Device-pointer MSMs returning the un-normalised total (host XYZZ); they synchronise `st` before returning.
int msm_g1_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp>* out);
int msm_g2_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp2>* out);

}  // namespace zkmi
