#!/usr/bin/env python3
"""bench.py -- Groth16 prove (BN254) on synthetic R1CS data.

N = 1 (default): BASELINE.json configs[1], "Synthetic R1CS 2^20 constraints, BN254 Groth16 prove on 1xMI355X (G1 MSM + Fr NTT)".
One step = one proof from the solver output onwards (computeH = 7 NTTs, 4 G1 MSMs, 1 G2 MSM, host tail), inputs already resident
in HBM.  The same JSON line also carries, measured in the same run:
  * `prove_ms_host_inputs`  the proof through the entry point a cgo caller has (host slices: a, b, c, w cross PCIe inside the call),
  * `at_2p24`               the metric's other size (2^24 constraints on this one GPU: key with 84 GB of window tables), whose proof
                            bytes are checked against the recombination of two half-size slices run through the table-less path
                            (another window width, Horner, other task sizes) AND against the CPU oracle's bytes (`at_2p24.cpu_baseline`; --no-verify-2p24-oracle skips it).
--gpus N > 1 (one process per GPU, torchrun): ONE proof over N * 2^21 constraints, range-sharded -- N = 8 is BASELINE.json
configs[2] (2^24 constraints).  Every rank owns one block of a, b, c, w, h and of the proving key (with its window tables); computeH
is block-sharded (the top log2 N butterfly stages of each transform run on all-to-all-transposed data: 10 RCCL all_to_all_single
per proof), the five MSMs run on the rank's slice, and an all-gather of the 768-byte partial-sum record lets every rank finish.

Prints ONE JSON line on rank 0 (contract in the task statement): metric / value / unit follow BASELINE.json; `roofline` describes the
dominant kernel (hipEvent pairs recorded inside libzkmi on the stream the kernels run on, live over the timed region);
`cpu_baseline` is the CPU oracle (a restatement of gnark's algorithm, NOT the gnark binary: no Go toolchain here) timed on this
box's host cores on the SAME 2^20 proof -- whose bytes are compared with the GPU's (parity at full size).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from bench_blocks.common import (GOLDEN, HBM_PEAK_GBS, MASK, N_PUBLIC, R_FR, HAND_WRITTEN_HOT, Instance, block_roofline, dominant_by_time, oracle_proof, traffic_ratio, valu_fraction,  # noqa: E402,F401
                                 seed_at)
from bench_blocks.export_path import export_path_block, export_path_groth16_block  # noqa: E402
from bench_blocks.launch import needs_self_launch, self_launch  # noqa: E402
from bench_blocks.inner_boundary import inner_boundary_block, two_slice_recombination  # noqa: E402
from bench_blocks.micro import micro_block, micro_sharded_block, srs_block  # noqa: E402
from bench_blocks.plonk import plonk_block  # noqa: E402
from bench_blocks.single_process import single_process_main  # noqa: E402
from bench_blocks.toolchain import go_toolchain_probe  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed proofs (default 200: ~2 s of GPU work at 2^20, enough for a utilisation sampler to see)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=None, help="log2(constraints) per GPU (default: 20 on one GPU = configs[1]; 21 per GPU on several, so that 8 GPUs prove configs[2]'s 2^24)")
    ap.add_argument("--scalars", choices=["uniform", "witness"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tables", action="store_true", help="disable the precomputed window tables of the resident proving key")
    ap.add_argument("--shard", choices=["range", "windows"], default="range", help="multi-GPU decomposition of the MSMs: by point range (default: no scalar "
                    "exchange) or by digit window / table row (north_star wording: whole key on every rank, h all-gathered)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU decomposition (sharded computeH phases + msm5_pk + all-gather + finalize) even at N=1")
    ap.add_argument("--no-2p24", action="store_true", help="skip the second measured block (2^24 constraints on this GPU)")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the host-input (PCIe-inclusive) measurement")
    ap.add_argument("--no-plonk", action="store_true", help="skip the PLONK block (configs[3]: plonk.Prove at 2^22 gates, verified by the oracle's pairing verifier)")
    ap.add_argument("--plonk-log-n", type=int, default=22)
    ap.add_argument("--no-micro", action="store_true", help="skip the configs[4] block (2^26-point G1 MSM + 2^26 NTT) and the SRS-load block")
    ap.add_argument("--micro-log-n", type=int, default=26, help="log2 of the points of the sharded configs[4] block at N > 1 (total over all ranks)")
    ap.add_argument("--single-process", action="store_true", help="with --gpus N: ONE process drives N device entries through the same zk_bn254_groth16_prove call (csrc/multidev.hip: "
                    "composite key by wire range, block-sharded computeH, peer-copy transposes); no torch.distributed.  On a box with fewer GPUs the devices are listed "
                    "repeatedly (virtual entries: correctness of the path, not a scaling figure).  --log-n is the TOTAL size")
    ap.add_argument("--lib", default=None, help="measurement tooling: 'exp' binds this run to libzkmi_exp.so (the A/B switches of DESIGN.md 8), or a path to another build")
    ap.add_argument("--no-export", action="store_true", help="skip the export-path block (PlonkPreprocess -> PlonkProveWithPK -> PlonkVerifyWithVK through libgnark_backend.so at 2^19 gates)")
    ap.add_argument("--export-log-gates", type=int, default=19)
    ap.add_argument("--export-g16-log-constraints", type=int, default=20, help="size of the Groth16 export-path block (Preprocess / ProveWithPK / VerifyWithVK through libgnark_backend.so)")
    ap.add_argument("--verify-2p24-oracle", action="store_true", help="(default since round 6; kept for old command lines) check the 2^24 proof bytes against the CPU oracle")
    ap.add_argument("--no-verify-2p24-oracle", action="store_true", help="skip the CPU oracle's proof of the 2^24 instance (~40 s on the 16 CPUs a GPU box gives; --no-cpu-baseline skips it too)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no launcher around: print the torch.distributed.run command this script would start and exit")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` typed as is: become the launcher (a child torch.distributed.run with N ranks) BEFORE anything touches the GPU -- no torch.cuda
    # call, no libzkmi load above this line.  Under an existing launcher (RANK / WORLD_SIZE set) this is skipped and we are one of the ranks.
    if needs_self_launch(args):
        sys.exit(self_launch(args, os.path.abspath(__file__), sys.argv[1:]))

    import torch
    import noir_backend_using_gnark_amd as zk
    from noir_backend_using_gnark_amd import _lib, parallel as par
    if args.lib:
        _lib.use_library(os.path.join(ROOT, "noir_backend_using_gnark_amd", "csrc", "build_exp", "libzkmi_exp.so") if args.lib == "exp" else args.lib)

    if args.single_process:
        return single_process_main(args, zk, _lib)
    rank, world, local = par.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    L = _lib.lib()
    local = local % max(1, torch.cuda.device_count())  # ranks share a GPU only in the gloo dry run on a one-GPU box
    _lib.check(L.zk_init(C.c_int(local)))
    torch.cuda.set_device(local)
    _lib.require_device()

    log_n = args.log_n if args.log_n is not None else (20 if world == 1 else 21)
    log_ng = log_n + (world.bit_length() - 1)
    if (1 << (log_ng - log_n)) != world:
        raise SystemExit("--gpus must be a power of two")
    N_loc, N_g = 1 << log_n, 1 << log_ng
    witness = 1 if args.scalars == "witness" else 0
    lo, hi = par.shard_range(N_g, rank, world)  # this rank's slice of every wire-indexed / coefficient-indexed array
    assert hi - lo == N_loc
    sharded = world > 1 or args.force_sharded
    np_loc = N_PUBLIC if rank == 0 else 0  # public wires live in rank 0's slice; gnark's pk.G1.K starts at the first private wire

    t_setup = time.time()
    win = sharded and args.shard == "windows"
    if win:  # every rank: the whole key (its own table rows), the whole w, its block of a, b, c
        inst = Instance(L, _lib, zk, log_ng, 0, N_PUBLIC, witness, True, full_inputs=False, torch=torch, window_shard=(rank, world), abc_block=(lo, N_loc))
    else:
        inst = Instance(L, _lib, zk, log_n, lo, np_loc if sharded else N_PUBLIC, witness, not args.no_tables,
                        shard_full_z=(sharded and rank != world - 1), full_inputs=not sharded, torch=torch)
    pk, d_w, r, s = inst.pk, inst.d_w, inst.r, inst.s
    fallback_stream = torch.cuda.Stream(priority=-1) if sharded else None
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def step():
        if not sharded:
            return zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, d_w, r, s, n_constraints=N_g, on_device=True)
        sess = par.groth16_msm5_pk_begin(pk, d_w.ptr)  # digits / sort / task plan of w run under computeH and its exchanges
        try:
            try:
                side = torch.cuda.ExternalStream(par.groth16_session_stream(sess))  # the library's computeH stream, shared with torch / RCCL
            except Exception:  # a torch build without ExternalStream: any non-null stream works (the library then bridges with events)
                side = fallback_stream
            with torch.cuda.stream(side):
                # the prover consumes its working buffers: at N > 1 the first all-to-all already writes fresh ones, at N = 1 copy
                a, b, c = (t.clone() for t in inst.t_abc) if world == 1 else inst.t_abc
                h = par.compute_h_sharded(a, b, c, log_ng, rank, world)
                if win:
                    h = par.all_gather_blocks(h)  # the coefficient exchange of the window mode: every rank needs the whole h
                live, sess = sess, None  # _end releases the session whatever it returns
                rec = par.groth16_msm5_pk_end(live, h.data_ptr(), side.cuda_stream)
                return par.groth16_finalize(pk, par.all_gather_limbs(rec), r, s)
        finally:
            if sess is not None:  # computeH or an exchange raised between _begin and _end: give the five stream slots back
                par.groth16_msm5_pk_abort(sess)

    def barrier():
        if world > 1:
            par.dist().barrier()
        torch.cuda.synchronize()
        _lib.check(L.zk_dev_sync())

    proof = None
    for _ in range(args.warmup):
        proof = step()
    # the timed region: EXACTLY args.steps proofs, barrier + synchronize on both sides, the library's per-launch event pairs OFF (they cost ~1 % at 2^20)
    _lib.profile(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # the roofline pass: the same number of the same proofs once more with a hipEvent pair around every launch, on the stream it is launched on
    _lib.profile(True)
    _lib.profile_reset()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        proof_p = step()
    barrier()
    elapsed_profiled = time.perf_counter() - t1
    _lib.profile(False)
    prof = _lib.profile_read()
    assert proof_p == proof, "the profiled pass proved other bytes"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if par.dist().get_backend() == "nccl" else "cpu")
        par.dist().all_reduce(t, op=par.dist().ReduceOp.MAX)
        elapsed = float(t.item())
        allp = par.all_gather_limbs(np.frombuffer(proof, dtype=np.uint64))
        assert (allp == allp[0]).all(), "ranks disagree on the proof bytes"

    n_ranks_seen, rank_devices = 1, [local]
    if par.dist().is_available() and par.dist().is_initialized():
        on = "cuda" if par.dist().get_backend() == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.int64, device=on)
        par.dist().all_reduce(ones)
        n_ranks_seen = int(ones.item())
        devs = torch.zeros(world, dtype=torch.int64, device=on)
        devs[rank] = local
        par.dist().all_reduce(devs)
        rank_devices = [int(x) for x in devs.tolist()]

    g1_units = N_g + N_g + (N_g - N_PUBLIC) + (N_g - 1)
    ms_per_step = elapsed / args.steps * 1e3
    value = g1_units * args.steps / elapsed

    # ---- roofline of the dominant kernel (by total time over the timed region)
    # restricted to the hand-written hot kernels of the path; with the five MSMs of a proof running on concurrent
    # streams an event pair also sees the time a kernel spends sharing the machine, exactly as rocprofv3 does
    hot = {k: v for k, v in prof.items() if k in HAND_WRITTEN_HOT[:4]}
    name, (launches, tot_ms) = max(hot.items(), key=lambda kv: kv[1][1])
    per_launch_ms = tot_ms / launches
    if name == "msm_accumulate_g1":
        # the G1 MSMs of a proof come in launches / steps accumulate launches (A, B1, K share one: same sorted digits)
        units_per_launch, bytes_per_unit = g1_units / world / (launches / float(args.steps)), 96.0
    elif name == "msm_accumulate_g2":
        units_per_launch, bytes_per_unit = N_loc, 160.0
    else:  # an NTT pass: 64 B per element per pass
        units_per_launch, bytes_per_unit = (N_loc if sharded else N_g), 64.0
    achieved = units_per_launch * bytes_per_unit / (per_launch_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction:
    # tools/pmc_traffic.py) -- taken at the SAME per-GPU size and table setting, else null
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and not args.no_tables:
        try:
            traffic = json.load(open(tpath)).get(name, {}).get("by_log_n", {}).get(str(log_n), {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # the honest ceiling of this kernel is the VALU, not HBM (bench_blocks/common.py valu_fraction)
    valu = valu_fraction(L, _lib, name, units_per_launch, per_launch_ms, N_g if win else N_loc, bool(pk.info()["tables"]))
    roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(per_launch_ms, 4), "launches": launches, "measured_in": "a second pass of %d proofs with event pairs on (%.3f ms per proof; the timed region runs without them)"
                % (args.steps, elapsed_profiled / args.steps * 1e3), "valu": valu, "dominant_by_time": dominant_by_time(prof, args.steps),
                "note": "VALU-bound kernel (254-bit modular multiplies on 32-bit integer ALUs); see DESIGN.md for the ALU-issue fraction",
                "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
    traffic_ratio(roofline, units_per_launch, bytes_per_unit)

    # the metric's name is BASELINE.json's, verbatim; the size this run used is in config.workload / config.constraints
    baseline_metric = "Groth16 prove ms + BN254 G1 MSM scalar-muls/sec at 2^20 / 2^24 constraints"
    try:
        baseline_metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        pass
    if log_ng == 20 and world == 1:
        bcfg = "BASELINE.json configs[1]: Synthetic R1CS 2^20 constraints, BN254 Groth16 prove on 1xMI355X (G1 MSM + Fr NTT)"
    elif log_ng == 24 and world == 8:
        bcfg = "BASELINE.json configs[2]: Synthetic R1CS 2^24 constraints, G1+G2 MSM sharded across 8xMI355X via RCCL/xGMI (2^21 per GPU)"
    elif world > 1:
        bcfg = "BASELINE.json configs[2] shape at %d GPUs: 2^%d constraints, 2^%d per GPU (weak scaling towards 8 x 2^21 = 2^24)" % (world, log_ng, log_n)
    else:
        bcfg = "configs[1] workload at 2^%d" % log_ng
    out = {
        "metric": baseline_metric,
        "value": round(value, 1), "unit": "G1 scalar-muls/s (whole prove: 4 G1 MSMs + G2 MSM + 7 NTTs per step)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "prove_ms": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "groth16_prove_bn254_synthetic_r1cs_2^%d" % log_ng, "baseline_config": bcfg, "constraints": N_g, "wires": N_g,
                   "n_public": N_PUBLIC, "scalars": args.scalars, "per_gpu_constraints": N_loc, "window_tables": bool(pk.info()["tables"]),
                   "parallelism": "single GPU" if not sharded else
                   ("one proof window-sharded x%d: block-sharded computeH (all-to-all transposes), all-gather of h, MSMs over all wires on this rank's table rows "
                    "(all-gather of partial sums)" % world) if win else
                   "one proof range-sharded x%d: block-sharded computeH (all-to-all transposes) + MSMs on rank-local key slices (all-gather of partial sums)" % world},
        "roofline": roofline, "proof_sha": hashlib.sha256(proof).hexdigest()[:16], "setup_s": round(t_setup, 2),
    }
    if par.dist().is_available() and par.dist().is_initialized():
        out["config"]["collectives"] = par.dist().get_backend()  # "nccl" = RCCL carried the exchanges of this run
        out["n_ranks_seen"] = n_ranks_seen  # ranks that answered an all-reduce of ones over that backend, and the devices they ran on
        out["config"]["rank_devices"] = rank_devices

    single = rank == 0 and not sharded
    # ---- the same proof through the entry point a cgo caller has: host slices in, 128 bytes out (PCIe-inclusive; never `value`)
    if single and not args.no_host_inputs:
        ha, hb, hc, hw = (d.to_numpy(np.uint64, (N_g, 4)) for d in (inst.d_a, inst.d_b, inst.d_c, inst.d_w))
        zk.prove(pk, ha, hb, hc, hw, r, s)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            hp = zk.prove(pk, ha, hb, hc, hw, r, s)
        host_ms = (time.perf_counter() - t0) / reps * 1e3
        out["prove_ms_host_inputs"] = {"value": round(host_ms, 3), "reps": reps, "bytes_over_pcie": int(4 * N_g * 32), "proof_matches_device_inputs": bool(hp == proof),
                                       "note": "zk_bn254_groth16_prove(on_device=0): a, b, c, w are pageable host arrays uploaded inside the call"}
        del ha, hb, hc, hw

    # ---- the same proof's hot operations through the INNER boundary, call by call, with host slices (north_star's literal flow)
    if single and not args.no_host_inputs and log_n <= 22:
        out["inner_boundary_2p%d" % log_n] = inner_boundary_block(L, _lib, zk, par, inst, proof)
        if not out["inner_boundary_2p%d" % log_n]["proof_from_these_results_matches_single_call"]:
            out["parity_error"] = "inner boundary: the proof assembled from separate MultiExp calls differs from the single-call prover's"

    # ---- the same key with a witness-like wire vector (SURVEY 8d cfg2's second distribution: 50 % in {0, 1}, 25 % < 2^32, 25 % uniform): giant buckets, split tasks, folds
    if single and not args.no_host_inputs and args.scalars == "uniform":
        d_w2 = _lib.DeviceBuffer(N_g * 32)
        _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d_w2.ptr), C.c_size_t(N_g), C.c_uint64(seed_at(0xC, 5, 0)), C.c_int(1), C.c_int(1), None))
        run_w = lambda: zk.prove(pk, inst.d_a, inst.d_b, inst.d_c, d_w2, r, s, n_constraints=N_g, on_device=True)
        pw0 = run_w()
        reps = 20
        _lib.check(L.zk_dev_sync())
        t0 = time.perf_counter()
        for _ in range(reps):
            pw = run_w()
        w_ms = (time.perf_counter() - t0) / reps * 1e3
        out["prove_ms_witness_like_scalars"] = {"value": round(w_ms, 3), "reps": reps, "deterministic": bool(pw == pw0),
                                                "note": "same key and a, b, c; wire values 50 % in {0, 1}, 25 % below 2^32, 25 % uniform (bytes checked against the oracle in tests/test_gpu_parity.py at 2^14)"}
        d_w2.free()

    # ---- CPU baseline: the oracle proves the SAME instance on this box's host cores (rank 0, N=1 only)
    if single and not args.no_cpu_baseline:
        cpu_proof, cpu_s, cores = oracle_proof(inst, log_n)
        from oracle import oracle as _orc  # (the checker, after the timed region)
        out["cpu_baseline"] = {"value": round(g1_units / cpu_s, 1), "unit": out["unit"], "cores": cores, "host_cpus_shown": _orc.host_cpus(), "kind": "port",
                               "cores_note": "threads the oracle started = min(OpenMP default, cgroup CPU quota): the CPUs this process may use, not the logical CPUs the box shows",
                               "sample": "1 full proof of the same instance (2^%d constraints, same seeds) by oracle/bn254_oracle.c "
                                         "(OpenMP, window-parallel Pippenger + radix-2 FFT; plain C on unsigned __int128, no assembly -- a restatement, "
                                         "NOT a gnark figure: gnark's assembly field arithmetic would be several times quicker, so GPU / this is not a speed-up over gnark)" % log_n,
                               "prove_ms": round(cpu_s * 1e3, 1), "proof_bytes_match_gpu": bool(cpu_proof == proof)}
        if cpu_proof != proof:
            out["parity_error"] = "GPU proof bytes differ from the CPU oracle's"

    # ---- the reference's live call end to end through libgnark_backend.so (child processes; the GPU is shared with this one, which is idle meanwhile).
    # Before the blocks that hold tens of GB of host arrays (2^24, PLONK 2^22, 2^26): behind them the children's host-bandwidth-bound phases -- content keys,
    # the values' upload -- measured 1.7-3.8 x slower (profiles/rnd5_w_bench_default_line.json: warm PlonkProveWithPK 19.3 ms against 15.5-16.6 alone / in a short line)
    if single and log_n == 20 and not args.no_export:
        try:
            out["export_path"] = export_path_block(args.export_log_gates)
            if not out["export_path"]["ok"]:
                out["parity_error"] = "export path: a proof made through libgnark_backend.so does not verify"
        except Exception as e:
            out["export_path"] = {"error": str(e)[:600]}
        # ---- the metric's own proof system behind the reference's ABI at the metric's size: Preprocess -> ProveWithPK -> VerifyWithVK on a RawR1CS of 2^20 constraints
        try:
            out["export_path_groth16"] = export_path_groth16_block(args.export_g16_log_constraints)
            if not out["export_path_groth16"]["ok"]:
                out["parity_error"] = "Groth16 export path: a proof made through libgnark_backend.so does not verify"
        except Exception as e:
            out["export_path_groth16"] = {"error": str(e)[:600]}
    # ---- second measured block: the metric's other size, 2^24 constraints on this GPU (BASELINE metric "at 2^20 / 2^24 constraints")
    if single and log_n == 20 and not args.no_2p24 and not args.no_tables:
        inst.free()
        t1 = time.time()
        big = Instance(L, _lib, zk, 24, 0, N_PUBLIC, witness, True)
        setup24 = time.time() - t1
        run24 = lambda: zk.prove(big.pk, big.d_a, big.d_b, big.d_c, big.d_w, big.r, big.s, n_constraints=big.N, on_device=True)
        p24 = run24()
        _lib.check(L.zk_dev_sync())
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            p24 = run24()
        _lib.check(L.zk_dev_sync())
        ms24 = (time.perf_counter() - t0) / reps * 1e3  # timed WITHOUT the per-launch event pairs of the profile ...
        _lib.profile(True)
        _lib.profile_reset()
        for _ in range(reps):                             # ... which come from the same number of proofs run once more
            run24()
        _lib.check(L.zk_dev_sync())
        _lib.profile(False)
        prof24 = _lib.profile_read()
        blk = {"constraints": big.N, "prove_ms": round(ms24, 2), "value": round(big.g1_units() / (ms24 * 1e-3), 1), "unit": out["unit"], "steps": reps,
               "warmup": 1, "setup_s": round(setup24, 2), "proof_sha": hashlib.sha256(p24).hexdigest()[:16], "window_tables": bool(big.pk.info()["tables"]),
               "verified_by": "two-slice recombination through the table-less msm5 path + finalize (independent window width / Horner / task sizes)",
               "proof_bytes_match_recombination": bool(two_slice_recombination(big, par, _lib, L) == p24)}
        n24 = prof24.get("msm_accumulate_g1", (1, 0))[0]
        blk["roofline"] = block_roofline(prof24, reps, big.g1_units() / max(1.0, n24 / float(reps)), big.N, big.N, 24, L=L, lib=_lib, n_plan=big.N)
        if not blk["proof_bytes_match_recombination"]:
            out["parity_error"] = "2^24: single-call proof differs from the two-slice recombination"
        if not (args.no_verify_2p24_oracle or args.no_cpu_baseline):
            cpu_proof, cpu_s, cores = oracle_proof(big, 24)
            blk["cpu_baseline"] = {"value": round(big.g1_units() / cpu_s, 1), "unit": out["unit"], "cores": cores, "kind": "port", "prove_ms": round(cpu_s * 1e3, 1),
                                   "sample": "1 full proof of the same 2^24 instance by oracle/bn254_oracle.c (OpenMP; plain C, no assembly -- NOT a gnark figure)",
                                   "proof_bytes_match_gpu": bool(cpu_proof == p24)}
            if cpu_proof != p24:
                out["parity_error"] = "2^24: GPU proof bytes differ from the CPU oracle's"
        if args.scalars == "uniform":  # the same key with a witness-like wire vector (50 % in {0, 1}, 25 % below 2^32, 25 % uniform): its zero digits never enter the sort
            d_w24 = _lib.DeviceBuffer(big.N * 32)
            _lib.check(L.zk_bn254_fr_random_dev(C.c_void_p(d_w24.ptr), C.c_size_t(big.N), C.c_uint64(seed_at(0xC, 5, 0)), C.c_int(1), C.c_int(1), None))
            run_w24 = lambda: zk.prove(big.pk, big.d_a, big.d_b, big.d_c, d_w24, big.r, big.s, n_constraints=big.N, on_device=True)
            pw24 = run_w24()
            _lib.check(L.zk_dev_sync())
            t0 = time.perf_counter()
            for _ in range(reps):
                pw24b = run_w24()
            w_ms24 = (time.perf_counter() - t0) / reps * 1e3
            blk["prove_ms_witness_like_scalars"] = {"value": round(w_ms24, 2), "reps": reps, "deterministic": bool(pw24 == pw24b)}
            d_w24.free()
        out["at_2p24"] = blk
        big.free()
        inst = None
    # ---- third measured block: BASELINE configs[3], the PLONK prover at 2^22 gates
    if single and log_n == 20 and not args.no_plonk:
        if inst is not None:
            inst.free()
        out["plonk_2p%d" % args.plonk_log_n] = plonk_block(L, _lib, args.plonk_log_n, cpu_oracle=not args.no_cpu_baseline)
        if not out["plonk_2p%d" % args.plonk_log_n]["proof_verifies"]:
            out["parity_error"] = "PLONK: the proof does not verify"
        if not (out["plonk_2p%d" % args.plonk_log_n].get("cpu_baseline") or {"proof_bytes_match_gpu": True})["proof_bytes_match_gpu"]:
            out["parity_error"] = "PLONK: GPU proof bytes differ from the C oracle's"
        inst = None
    # ---- configs[4] (standalone 2^26 MSM / NTT) and the SRS load of the reference's size
    if single and log_n == 20 and not args.no_micro:
        if inst is not None:
            inst.free()
            inst = None
        out["micro_2p26"] = micro_block(L, _lib, zk, 26, cpu_legs=not args.no_cpu_baseline)
        out["srs_read_1e6"] = srs_block(_lib)
        if not (out["micro_2p26"]["equals_split_recombination"] and out["micro_2p26"]["equals_window_table_path"] and out["srs_read_1e6"]["write_of_read_is_identity"]):
            out["parity_error"] = "micro-benchmark cross-check failed"
        if not (out["micro_2p26"].get("cpu_baseline_msm", {}).get("point_matches_gpu", True) and out["micro_2p26"].get("cpu_baseline_ntt", {}).get("image_matches_gpu", True)):
            out["parity_error"] = "2^26 micro-benchmark: the GPU's result differs from the CPU oracle's"
    if rank == 0:
        out["go_toolchain"] = go_toolchain_probe()
    # ---- configs[4] on several GPUs: the 2^26-point MSM range-sharded and the 2^26-point FFT block-sharded over the ranks
    if (world > 1 or (args.force_sharded and par._force_collectives())) and not args.no_micro:
        inst.free()
        blk = micro_sharded_block(L, _lib, zk, par, torch, args.micro_log_n, rank, world)
        out["micro_2p%d_sharded" % args.micro_log_n] = blk
        if not (blk["msm_same_on_every_rank"] and blk["msm_equals_odd_split_recombination"] and blk["ntt_inverse_of_forward_is_identity"]):
            out["parity_error"] = "sharded micro-benchmark cross-check failed"
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        par.dist().barrier()
        par.dist().destroy_process_group()


if __name__ == "__main__":
    main()
