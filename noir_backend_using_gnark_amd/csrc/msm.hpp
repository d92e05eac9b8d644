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
// Device-pointer MSMs returning the un-normalised total (host XYZZ); they synchronise `st` before returning.
int msm_g1_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp>* out);
int msm_g2_xyzz(Slot* s, hipStream_t st, const void* d_pts, const void* d_scalars, size_t n, const zk_msm_cfg* cfg, XYZZ<HFp2>* out);

}  // namespace zkmi
