"""One process per GPU: sharding of one proof across the GPUs of a node -- by POINT RANGE (default) or by digit WINDOW (table rows).

Window mode (BASELINE north_star "MSM window buckets ... shard across the 8 GPUs"): every rank holds the whole key but only the rows
w = rank + k * world of its window tables (zk_groth16_pk.flags bit 2), computeH stays block-sharded, the blocks of h are all-gathered
(`all_gather_blocks`), every rank runs the five MSMs over ALL wires restricted to its windows, and the 768-byte records are all-gathered
and finalized exactly as in range mode.  Same table memory per rank (rows/world x all points = all rows x points/world), but the scalars
must be whole on every rank (w replicated, h all-gathered: 32 B x N x (world-1)/world per rank over xGMI) and ceil(255/c) windows do not
divide evenly (13 rows over 8 ranks: 2,2,2,2,2,1,1,1) -- which is why range mode is the default.


The path shards by POINT RANGE (SURVEY.md §8e "point-range-sharded"): rank g owns bases / scalars [g*n/G, (g+1)*n/G),
runs the local Pippenger MSMs, and the only exchange is an all-gather of the un-normalised partial sums (96 limbs =
768 B per rank per proof) -- RCCL over xGMI on GPUs (`torch.distributed` backend "nccl"), gloo in the CPU tests.
Field / group addition is not an RCCL reduction op, so the "reduce" is all-gather + a local combine on every rank
(zk_bn254_g1_sum_xyzz / zk_bn254_groth16_finalize, host side, O(G)).

computeH shards too (SURVEY.md §8e, 4-step NTT): every rank owns one block of a, b, c and of h; the top log2(G) butterfly
stages of each transform run on all-to-all-transposed data (`block_exchange`, RCCL all_to_all_single over xGMI), the rest
is a block-local transform -- 10 array exchanges of 32*M*(G-1)/G bytes per rank and proof, no other traffic."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, lib, vp


def dist():
    import torch.distributed as d
    return d


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when world > 1."""
    import torch
    import torch.distributed as d
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or _force_collectives()) and not d.is_initialized():
        if backend is None:
            # ZKMI_DIST_BACKEND=gloo: several ranks may then share one GPU (tensors are staged through the host) -- used to
            # exercise the whole multi-process path on a one-GPU box; RCCL refuses two ranks on one device
            backend = os.environ.get("ZKMI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only without a launcher (the forced world of one rank)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        d.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _force_collectives() -> bool:
    """ZKMI_FORCE_COLLECTIVES=1: a world of ONE rank still goes through every collective (RCCL executes them as local copies) -- the only way to run
    the RCCL call pattern (dtypes, async work handles, stream ordering against libzkmi's kernels) on a one-GPU box; tests/test_gpu_parity.py::test_rccl_executes_the_collectives_of_the_sharded_proof_world_of_one."""
    return os.environ.get("ZKMI_FORCE_COLLECTIVES", "0") == "1"


def _no_peers(d) -> bool:
    return not (d.is_available() and d.is_initialized()) or (d.get_world_size() == 1 and not _force_collectives())


def window_rows(window_bits: int, rank: int, world: int) -> list[int]:
    """Window (table-row) sharding: the digit windows w = rank, rank + world, ... of the ceil(255 / c) windows belong to `rank`."""
    return list(range(rank, (255 + window_bits - 1) // window_bits, world))


def all_gather_blocks(x):
    """All ranks' blocks concatenated in rank order (the `coefficient exchange` of the window-sharded mode: every rank needs the whole h).
    RCCL all_gather_into_tensor under nccl; gloo gathers host copies."""
    import torch
    d = dist()
    if _no_peers(d):
        return x
    world = d.get_world_size()
    if d.get_backend() == "nccl":
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        d.all_gather_into_tensor(out, x.contiguous())
        return out
    xh = x.cpu() if x.is_cuda else x
    rows = [torch.empty_like(xh) for _ in range(world)]
    d.all_gather(rows, xh)
    return torch.cat(rows).to(x.device)


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous balanced range [lo, hi) of n units owned by `rank`."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_limbs(local: np.ndarray) -> np.ndarray:
    """All-gather a small uint64 record from every rank -> (world, len).  Device tensors under nccl (RCCL), host under gloo."""
    import torch
    d = dist()
    local = np.ascontiguousarray(local, dtype=np.uint64).reshape(-1)
    if _no_peers(d):
        return local.reshape(1, -1)
    world = d.get_world_size()
    t = torch.from_numpy(local.view(np.int64).copy())
    if d.get_backend() == "nccl":
        t = t.cuda()
        out = torch.empty((world, t.numel()), dtype=torch.int64, device=t.device)
        d.all_gather_into_tensor(out, t)
    else:
        rows = [torch.empty_like(t) for _ in range(world)]
        d.all_gather(rows, t)
        out = torch.stack(rows)
    return out.cpu().numpy().view(np.uint64)


def sharded_g1_multi_exp(d_points_local: int, d_scalars_local: int, n_local: int, config=None) -> np.ndarray:
    """Every rank passes ITS slice (device pointers); returns the affine sum over all ranks, identical on every rank."""
    from . import bn254
    part = bn254.g1_multi_exp_dev(d_points_local, d_scalars_local, n_local, config, partial=True)
    return bn254.g1_sum_partials(all_gather_limbs(part))


def sharded_g2_multi_exp(d_points_local: int, d_scalars_local: int, n_local: int, config=None) -> np.ndarray:
    from . import bn254
    part = bn254.g2_multi_exp_dev(d_points_local, d_scalars_local, n_local, config, partial=True)
    return bn254.g2_sum_partials(all_gather_limbs(part))


def groth16_msm5_local(d_a, d_b, d_b2, d_w, nw, d_k, d_wk, nk, d_z, d_h, nz) -> np.ndarray:
    """The five MSMs of this rank's slice of one proof -> 96-limb record of XYZZ partial sums."""
    out = np.zeros(96, dtype=np.uint64)
    check(lib().zk_bn254_groth16_msm5_dev(C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_b2), C.c_void_p(d_w), C.c_size_t(nw),
                                          C.c_void_p(d_k), C.c_void_p(d_wk), C.c_size_t(nk), C.c_void_p(d_z), C.c_void_p(d_h), C.c_size_t(nz),
                                          vp(out), None))
    return out


def groth16_finalize(pk, partials: np.ndarray, r, s) -> bytes:
    """Host tail of groth16.Prove from the gathered (world, 96) partial records; same bytes on every rank."""
    partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 96)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
    proof = (C.c_uint8 * 128)()
    check(lib().zk_bn254_groth16_finalize(pk.handle, vp(partials), C.c_size_t(partials.shape[0]), vp(r), vp(s), proof))
    return bytes(proof)


def _a2a_transport() -> str:
    """The transport of the NTT transposes is chosen ONCE, identically on every rank (an environment switch, read the same way by
    all of them): RCCL all_to_all_single by default, ZKMI_A2A=0 selects all_gather + select (G x the traffic).  There is no
    fall-back in the middle of a run -- ranks that disagreed on the collective would hang the job."""
    return "all_gather" if os.environ.get("ZKMI_A2A", "1") == "0" else "all_to_all"


def block_exchange(x):
    """The transpose of the sharded NTT: `x` (a torch tensor, this rank's array) is cut into `world` equal chunks, chunk r
    goes to rank r and chunk s of the result is what rank s sent.  Applying it twice restores the block.
    RCCL all_to_all_single under nccl; gloo (no all-to-all) gathers and selects -- CPU tests only."""
    import torch
    d = dist()
    if _no_peers(d):
        return x
    world, rank = d.get_world_size(), d.get_rank()
    y = torch.empty_like(x)
    if d.get_backend() == "nccl":
        if _a2a_transport() == "all_to_all":
            d.all_to_all_single(y, x)
            return y
        full = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        d.all_gather_into_tensor(full.view(-1), x.reshape(-1))
        chunk = x.numel() // world
        y.view(-1).copy_(full.view(world, world, chunk)[:, rank, :].reshape(-1))
        return y
    xh = x.cpu() if x.is_cuda else x  # gloo collectives take host tensors
    rows = [torch.empty_like(xh) for _ in range(world)]
    d.all_gather(rows, xh)
    chunk = x.numel() // world
    yh = torch.cat([rows[s_].reshape(-1)[rank * chunk:(rank + 1) * chunk] for s_ in range(world)]).reshape(x.shape)
    y.copy_(yh)
    return y


def _h_shard_phase_hip(phase, a, b, c, log_d, log_g, rank):
    """zk_bn254_groth16_h_shard_dev on torch CUDA tensors, asynchronous on torch's current stream."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    if not st:  # the null stream: libzkmi then runs on a stream of its own and synchronises, so torch's work must be complete first
        torch.cuda.current_stream().synchronize()
    check(lib().zk_bn254_groth16_h_shard_dev(C.c_int(phase), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr() if b is not None else 0),
                                             C.c_void_p(c.data_ptr() if c is not None else 0), C.c_uint32(log_d), C.c_uint32(log_g),
                                             C.c_uint32(rank), C.c_void_p(st)))


def block_exchange_async(x):
    """block_exchange with the collective left in flight: returns (y, wait) -- call wait() before touching y.  RCCL: all_to_all_single with
    async_op (the collective runs on RCCL's stream behind the work already queued on the current stream; wait() orders the current stream
    behind it, the host does not block).  Anything else: the synchronous exchange."""
    d = dist()
    if not _no_peers(d) and d.get_backend() == "nccl" and _a2a_transport() == "all_to_all":
        import torch
        y = torch.empty_like(x)
        work = d.all_to_all_single(y, x, async_op=True)
        return y, work.wait
    return block_exchange(x), (lambda: None)


def compute_h_sharded(a, b, c, log_d: int, rank: int, world: int, phase=_h_shard_phase_hip, exchange=block_exchange, pipelined=None,
                      exchange_async=None, six_transforms=None):
    """gnark computeH with a, b, c, h sharded by blocks over `world` = 2^g ranks.  a, b, c: this rank's blocks (M = 2^log_d / world
    elements of 4 x int64 each; destroyed).  Returns the array holding this rank's block of h (gnark's bit-reversed order).
    `phase` / `exchange` are injectable so that the CPU tests can run the same schedule on the oracle's arithmetic over gloo.
    pipelined (default: on under RCCL): the three arrays go through phases 0, 1 and the first third of phase 2 ONE ARRAY AHEAD of their
    transposes -- while a is in its butterfly stages the all-to-all of b (then c) is in flight on RCCL's stream; 9 of the 10 exchanges overlap
    with compute (unmeasured on hardware: this pool has one GPU per box; the schedule is covered by the gloo tests)."""
    log_g = world.bit_length() - 1
    if (1 << log_g) != world:
        raise ValueError("world size must be a power of two")
    if pipelined is None:
        d = dist()
        pipelined = bool(not _no_peers(d) and d.get_backend() == "nccl")
    if six_transforms is None:
        six_transforms = os.environ.get("ZKMI_H_SKIP_C", "1") != "0"
    if six_transforms:
        # c stays in coefficient form (exact by linearity: csrc/ntt.hip compute_h_inplace): c only goes through the first inverse transform (two transposes,
        # phases 0 and 6) and is subtracted from this rank's block of the result in phase 8 -- 9 array transposes instead of 10, two transforms of c fewer
        xa = exchange_async or (block_exchange_async if (pipelined and exchange is block_exchange) else (lambda x: (exchange(x), (lambda: None))))
        flight = [xa(v) for v in (a, b, c)]
        nxt = []
        for v, wait in flight:                       # cross stages of FFTInverse(DIF), one array ahead of the transposes
            wait()
            phase(0, v, None, None, log_d, log_g, rank)
            nxt.append(xa(v))
        (a, wa), (b, wb), (c, wc) = nxt
        ab = []
        for v, wait in ((a, wa), (b, wb)):           # blocks of a, b: rest of the inverse, scaling, block part of FFT(DIT, coset); then transposed again
            wait()
            phase(1, v, None, None, log_d, log_g, rank)
            ab.append(xa(v))
        wc()
        phase(6, c, None, None, log_d, log_g, rank)  # block of c: its coefficients -- stays here
        out = []
        for v, wait in ab:
            wait()
            phase(4, v, None, None, log_d, log_g, rank)
            out.append(v)
        a, b = out
        phase(7, a, b, None, log_d, log_g, rank)
        a = exchange(a)
        phase(8, a, None, c, log_d, log_g, rank)
        return a
    if not pipelined:
        a, b, c = exchange(a), exchange(b), exchange(c)
        phase(0, a, b, c, log_d, log_g, rank)
        a, b, c = exchange(a), exchange(b), exchange(c)
        phase(1, a, b, c, log_d, log_g, rank)
        a, b, c = exchange(a), exchange(b), exchange(c)
        phase(2, a, b, c, log_d, log_g, rank)
        a = exchange(a)
        phase(3, a, None, None, log_d, log_g, rank)
        return a
    xa = exchange_async or (block_exchange_async if exchange is block_exchange else (lambda x: (exchange(x), (lambda: None))))
    flight = [xa(v) for v in (a, b, c)]          # transposes into phase 0, all three in flight
    for ph in (0, 1, 4):
        nxt = []
        for v, wait in flight:
            wait()
            phase(ph, v, None, None, log_d, log_g, rank)
            nxt.append(xa(v) if ph != 4 else (v, None))   # the transpose into the next phase starts as soon as this array is done
        flight = nxt
    a, b, c = (v for v, _ in flight)
    phase(5, a, b, c, log_d, log_g, rank)
    a = exchange(a)
    phase(3, a, None, None, log_d, log_g, rank)
    return a


def _ntt_shard_step_hip(step, a, log_d, log_g, rank, inverse, decimation, coset):
    """zk_bn254_ntt_shard_dev on a torch CUDA tensor, asynchronous on torch's current stream."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    if not st:
        torch.cuda.current_stream().synchronize()
    check(lib().zk_bn254_ntt_shard_dev(C.c_int(step), C.c_void_p(a.data_ptr()), C.c_uint32(log_d), C.c_uint32(log_g), C.c_uint32(rank), C.c_int(int(inverse)),
                                       C.c_int(int(decimation)), C.c_int(int(coset)), C.c_void_p(st)))


def ntt_sharded(x, log_d: int, rank: int, world: int, inverse: bool = False, decimation: int = 1, coset: bool = False, step=_ntt_shard_step_hip,
                exchange=block_exchange):
    """(*Domain).FFT / FFTInverse over 2^log_d points sharded by blocks over `world` = 2^g ranks (BASELINE configs[4] on several GPUs): x = this rank's
    block of the stored order (M = 2^log_d / world rows of 4 x int64; transformed in place where no exchange intervenes).  decimation: 0 = DIT
    (bit-reversed in, natural out), 1 = DIF (natural in, bit-reversed out), gnark's numbering.  Two all-to-all transposes per transform.  `step` /
    `exchange` are injectable (CPU tests run the same schedule on the oracle's arithmetic)."""
    log_g = world.bit_length() - 1
    if (1 << log_g) != world:
        raise ValueError("world size must be a power of two")
    args = (log_d, log_g, rank, inverse, decimation, coset)
    if decimation == 1:  # DIF: cross stages first
        if coset and not inverse:
            step(2, x, *args)
        x = exchange(x)
        step(0, x, *args)
        x = exchange(x)
        step(1, x, *args)
        return x
    step(1, x, *args)    # DIT: the block transform first
    x = exchange(x)
    step(0, x, *args)
    x = exchange(x)
    if coset and inverse:
        step(2, x, *args)
    return x


def groth16_msm5_pk(pk, d_w: int, d_h: int, stream: int = 0) -> np.ndarray:
    """The five MSMs of this rank's slice against its resident key slice `pk` (window tables included) -> 96-limb record."""
    out = np.zeros(96, dtype=np.uint64)
    check(lib().zk_bn254_groth16_msm5_pk(pk.handle, C.c_void_p(d_w), C.c_void_p(d_h), vp(out), C.c_void_p(stream)))
    return out


def groth16_msm5_pk_begin(pk, d_w: int) -> int:
    """Starts the preparation of the wire scalars at once (call before compute_h_sharded); returns a session for groth16_msm5_pk_end."""
    sess = C.c_uint64(0)
    check(lib().zk_bn254_groth16_msm5_pk_begin(pk.handle, C.c_void_p(d_w), C.byref(sess)))
    return sess.value


def groth16_msm5_pk_end(session: int, d_h: int, stream: int = 0) -> np.ndarray:
    out = np.zeros(96, dtype=np.uint64)
    check(lib().zk_bn254_groth16_msm5_pk_end(C.c_uint64(session), C.c_void_p(d_h), vp(out), C.c_void_p(stream)))
    return out


def groth16_msm5_pk_abort(session: int) -> None:
    """Gives the session up (an exception between _begin and _end): drains its streams and releases the five stream slots."""
    check(lib().zk_bn254_groth16_msm5_pk_abort(C.c_uint64(session)))


def groth16_session_stream(session: int) -> int:
    """HIP stream handle of the session (wrap it with torch.cuda.ExternalStream and run compute_h_sharded under it)."""
    st = C.c_void_p(0)
    check(lib().zk_bn254_groth16_msm5_session_stream(C.c_uint64(session), C.byref(st)))
    return int(st.value or 0)
