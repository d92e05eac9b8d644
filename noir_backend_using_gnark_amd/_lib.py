"""ctypes binding of libzkmi.so (include/zkmi.h).  The product path fails loudly here: a missing library raises at
load time, and every compute entry point returns ZK_ERR_NO_DEVICE (raised as ZkmiError) when no GPU is usable."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkmi.so")


def use_library(path: str) -> None:
    """Measurement tooling only (tools/ab_bench.py, bench.py --lib): bind this process to another build of the library -- libzkmi_exp.so, the build with the A/B
    switches of DESIGN.md compiled in (`make -C csrc EXPERIMENTS=1`; never shipped, never beside the package in a deployment).  Must be called before the first
    use; the package itself never looks at the environment for this (the product library reads no experiment variable: csrc/ctx.hpp)."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise RuntimeError("the library is already loaded from %s" % LIB_PATH)
    LIB_PATH = path


ZK_OK, ZK_ERR_LEN, ZK_ERR_NB_TASKS, ZK_ERR_NO_DEVICE, ZK_ERR_HIP, ZK_ERR_ARG, ZK_ERR_HANDLE, ZK_ERR_BUSY = 0, -1, -2, -3, -4, -5, -6, -7


class ZkmiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libzkmi error %d: %s" % (code, msg))
        self.code = code


class MsmCfg(C.Structure):
    _fields_ = [("nb_tasks", C.c_int), ("scalars_mont", C.c_int), ("window_bits", C.c_int), ("device_mask", C.c_int)]


class Groth16PK(C.Structure):
    _fields_ = [("log_domain", C.c_uint32), ("n_wires", C.c_size_t), ("n_public", C.c_size_t),
                ("g1_alpha", C.c_void_p), ("g1_beta", C.c_void_p), ("g1_delta", C.c_void_p),
                ("g1_a", C.c_void_p), ("g1_b", C.c_void_p), ("g1_k", C.c_void_p), ("g1_z", C.c_void_p),
                ("g2_beta", C.c_void_p), ("g2_delta", C.c_void_p), ("g2_b", C.c_void_p), ("bases_on_device", C.c_int), ("flags", C.c_int),
                ("infinity_a", C.c_void_p), ("infinity_b", C.c_void_p), ("nb_infinity_a", C.c_size_t), ("nb_infinity_b", C.c_size_t),
                ("table_window_bits", C.c_int), ("device_mask", C.c_int), ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32)]


class R1CS(C.Structure):
    _fields_ = [("n_constraints", C.c_size_t), ("n_wires", C.c_size_t), ("n_public", C.c_size_t)] + \
               [(m + k, C.c_void_p) for m in ("l", "r", "o") for k in ("_ptr", "_idx", "_val")]


class PlonkCircuit(C.Structure):
    _fields_ = [("n_public", C.c_size_t), ("n_constraints", C.c_size_t), ("n_vars", C.c_size_t),
                ("ql", C.c_void_p), ("qr", C.c_void_p), ("qo", C.c_void_p), ("qm", C.c_void_p), ("qk", C.c_void_p),
                ("xa", C.c_void_p), ("xb", C.c_void_p), ("xc", C.c_void_p), ("coeffs_on_device", C.c_int), ("reserved", C.c_int)]


class PlonkPK(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("n_public", C.c_size_t), ("n_constraints", C.c_size_t), ("n_vars", C.c_size_t)] + \
               [(k, C.c_void_p) for k in ("ql", "qr", "qm", "qo", "cqk", "lqk", "s1", "s2", "s3", "permutation", "xa", "xb", "xc",
                                          "vk_s", "vk_ql", "vk_qr", "vk_qm", "vk_qo", "vk_qk")]


class PlonkVK(C.Structure):
    _fields_ = [("size", C.c_uint64), ("n_public", C.c_uint64), ("size_inv", C.c_uint64 * 4), ("generator", C.c_uint64 * 4), ("coset_shift", C.c_uint64 * 4),
                ("s", C.c_uint64 * 24), ("ql", C.c_uint64 * 8), ("qr", C.c_uint64 * 8), ("qm", C.c_uint64 * 8), ("qo", C.c_uint64 * 8), ("qk", C.c_uint64 * 8)]


PLONK_PROOF_BYTES = 548

# every symbol include/zkmi.h declares (tests check that the library exports exactly these)
SYMBOLS = [
    "zk_device_count", "zk_init", "zk_last_error", "zk_version",
    "zk_bn254_g1_msm", "zk_bn254_g2_msm", "zk_bn254_g1_msm_dev", "zk_bn254_g2_msm_dev",
    "zk_bn254_g1_msm_partial_dev", "zk_bn254_g2_msm_partial_dev", "zk_bn254_g1_sum_xyzz", "zk_bn254_g2_sum_xyzz",
    "zk_bn254_msm_plan_info", "zk_bn254_bases_register", "zk_bn254_bases_register_dev", "zk_bn254_bases_register_cfg", "zk_bn254_bases_build_table", "zk_bn254_bases_lagrange", "zk_bn254_bases_free", "zk_bn254_msm_bases", "zk_bn254_msm_bases_batch", "zk_bn254_msm_bases_batch_dev", "zk_bn254_msm_bases_dev", "zk_bn254_scalars_register", "zk_bn254_scalars_free", "zk_bn254_msm_bases_prepared",
    "zk_bn254_ntt", "zk_bn254_ntt_dev", "zk_bn254_bit_reverse", "zk_bn254_bit_reverse_dev",
    "zk_bn254_groth16_compute_h", "zk_bn254_groth16_compute_h_dev", "zk_bn254_groth16_h_shard_dev", "zk_bn254_ntt_shard_dev",
    "zk_bn254_felts_decode_hex", "zk_bn254_felts_decode_hex_dev", "zk_bn254_felts_decode_bytes_dev", "zk_bn254_felts_encode_hex",
    "zk_bn254_groth16_pk_load", "zk_bn254_groth16_pk_free", "zk_bn254_groth16_pk_info", "zk_bn254_groth16_prove",
    "zk_bn254_groth16_pk_read", "zk_bn254_groth16_pk_write", "zk_bn254_groth16_vk_write",
    "zk_bn254_groth16_verify", "zk_bn254_plonk_verify", "zk_bn254_pairing_check",
    "zk_bn254_r1cs_load", "zk_bn254_r1cs_free", "zk_bn254_r1cs_eval_abc_dev", "zk_bn254_groth16_setup", "zk_bn254_groth16_prove_r1cs",
    "zk_bn254_groth16_msm5_dev", "zk_bn254_groth16_msm5_pk", "zk_bn254_groth16_msm5_pk_begin", "zk_bn254_groth16_msm5_pk_end", "zk_bn254_groth16_msm5_pk_abort", "zk_bn254_groth16_msm5_session_stream", "zk_bn254_groth16_finalize",
    "zk_bn254_plonk_setup", "zk_bn254_plonk_pk_load", "zk_bn254_plonk_pk_free", "zk_bn254_plonk_pk_lagrange_srs", "zk_bn254_plonk_pk_export", "zk_bn254_plonk_pk_read", "zk_bn254_plonk_pk_write", "zk_bn254_plonk_prove", "zk_bn254_plonk_synth_qk_dev",
    "zk_plonk_preprocess", "zk_plonk_prove_with_pk", "zk_bn254_plonk_pk_info", "zk_acir_to_sparse_r1cs", "zk_groth16_r1cs_from_raw",
    "zk_groth16_preprocess", "zk_groth16_prove_with_pk", "zk_groth16_prove_with_meta",
    "zk_bn254_fr_random_dev", "zk_bn254_g1_generate_dev", "zk_bn254_g2_generate_dev", "zk_bn254_fr_mul_dev", "zk_bn254_kzg_new_srs_dev", "zk_bn254_kzg_srs_read", "zk_bn254_kzg_srs_write",
    "zk_dev_alloc", "zk_dev_free", "zk_dev_h2d", "zk_dev_d2h", "zk_dev_sync",
    "zk_profile_enable", "zk_profile_reset", "zk_profile_count", "zk_profile_get", "zk_profile_host", "zk_selftest_host",
    "zk_init_devices", "zk_warm_streams", "zk_init_flags", "zk_bn254_kzg_srs_g2", "zk_device_entries", "zk_set_entry", "zk_set_default_devices", "zk_default_devices", "zk_bn254_ntt_devices",
    "zk_acir_public_witnesses", "zk_acir_lower_resident", "zk_export_cache_info", "zk_export_cache_clear", "zk_bn254_plonk_pk_bytes",
    "zk_groth16_lower_resident", "zk_groth16_key_resident", "zk_groth16_public_inputs", "zk_bn254_groth16_pk_build_tables", "zk_bn254_groth16_pk_bytes",
    "zk_export_set_new_srs_size", "zk_export_new_srs_size", "zk_warm_session_streams", "zk_warm_session_streams_background", "zk_background_wait", "zk_background_hold", "zk_background_set_yield_ms", "zk_bn254_bases_build_table_background",
]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libzkmi.so is missing at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C noir_backend_using_gnark_amd/csrc).  There is no CPU fallback for the product path." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.zk_last_error.restype = C.c_char_p
        _lib.zk_version.restype = C.c_char_p
        for name in SYMBOLS:
            fn = getattr(_lib, name)
            if name == "zk_export_new_srs_size":
                fn.restype = C.c_size_t
            elif name not in ("zk_last_error", "zk_version"):
                fn.restype = C.c_int
    return _lib


def check(rc: int) -> None:
    if rc != ZK_OK:
        raise ZkmiError(rc, (lib().zk_last_error() or b"").decode())


def device_count() -> int:
    return int(lib().zk_device_count())


def require_device() -> None:
    if device_count() <= 0:
        raise ZkmiError(ZK_ERR_NO_DEVICE, "no HIP device visible; the MI355X path has no CPU fallback")


def vp(x) -> C.c_void_p:
    """numpy array / int device pointer / None -> c_void_p"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.ctypes.data)


class DeviceBuffer:
    """A hipMalloc'd buffer owned through the C ABI (no torch needed).  `.ptr` is the raw device address."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().zk_dev_alloc(C.byref(p), C.c_size_t(nbytes)))
        self.ptr = int(p.value)
        self.nbytes = nbytes

    @classmethod
    def from_numpy(cls, a):
        import numpy as np
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 16))
        if a.nbytes:
            check(lib().zk_dev_h2d(C.c_void_p(b.ptr), vp(a), C.c_size_t(a.nbytes)))
        return b

    def to_numpy(self, dtype, shape):
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(lib().zk_dev_d2h(vp(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib().zk_dev_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def profile(enable: bool) -> None:
    check(lib().zk_profile_enable(C.c_int(1 if enable else 0)))


def profile_reset() -> None:
    check(lib().zk_profile_reset())


def profile_read() -> dict:
    """kernel name -> (launches, total_ms), measured with hipEvent pairs on the stream the kernels ran on."""
    out = {}
    n = lib().zk_profile_count()
    for i in range(n):
        name = C.create_string_buffer(128)
        launches = C.c_uint64()
        ms = C.c_double()
        check(lib().zk_profile_get(C.c_int(i), name, C.c_size_t(128), C.byref(launches), C.byref(ms)))
        out[name.value.decode()] = (int(launches.value), float(ms.value))
    return out


def split_profile(prof: dict):
    """(kernels, host sections): host-side wall-clock sections carry a dotted prefix ("plonk.round1...", "export.pk_content_key") or the "host_" prefix."""
    host = {k: v for k, v in prof.items() if "." in k or k.startswith("host_")}
    return {k: v for k, v in prof.items() if k not in host}, host
