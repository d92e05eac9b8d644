"""GPU suite: several device entries in ONE process, behind the C ABI (csrc/multidev.hip; SURVEY §8b `device_mask`, §8e).  The reference is one process
(gnark_backend_ffi/main.go:24-37 -> backend/plonk/plonk.go:53-73), so its exports can only use a second GPU if the library spreads the work itself.  This
pool has one GPU per box: the same device is listed 8 times (virtual devices) and every result must equal the single-entry bytes -- for 2, 4 and 8
entries; a box with several GPUs also runs the real-peer case.  Each scenario is a subprocess (tests/multidev_worker.py): zk_init_devices is process-wide."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run(mode):
    r = subprocess.run([sys.executable, os.path.join(HERE, "multidev_worker.py"), mode], capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def all_true(res, skip=("entries", "real_devices")):
    bad = [k for k, v in res.items() if k not in skip and v is not True]
    assert not bad, bad


def test_msm_over_host_slices_on_2_3_4_8_entries():
    """zk_bn254_g1_msm / g2_msm with zk_msm_cfg.device_mask: the slices go to the entries by range, the partial sums are combined on the host -- the
    oracle's point for every entry count, for a single-bit mask (that entry), for more entries than points; a mask naming a missing entry is refused."""
    all_true(run("msm"))


def test_composite_resident_bases_commit_with_host_and_device_scalars():
    """zk_bn254_bases_register* under a process default of 2 / 4 entries: a composite handle, ranges resident per entry; zk_bn254_msm_bases / _dev with
    offsets that cut across the ranges equal the single-entry handle's commits; an overrun is upstream's length error."""
    all_true(run("bases"))


def test_groth16_composite_key_proves_the_single_gpu_bytes_on_2_4_8_entries():
    """zk_groth16_pk.device_mask: slice keys per entry, computeH block-sharded with its nine transposes as copies between the entries, five MSMs per slice,
    768-byte records finalized on the host.  Proof bytes equal the single-entry prover's (which equal the oracle's) for device-resident and host inputs,
    for host-resident bases with fewer constraints than the domain on entries 2..5; three entries are refused."""
    res = run("groth16")
    all_true(res)
    assert res["single_equals_oracle"] is True


def test_ntt_over_host_slices_all_eight_modes_on_2_4_8_entries():
    all_true(run("ntt"))


def test_handles_carry_their_entry_and_zk_set_entry_moves_the_rest():
    res = run("entries")
    assert res.pop("handle_entry") == 3
    all_true(res)


def test_plonk_export_against_an_srs_spread_over_two_entries():
    """PlonkPreprocess / PlonkProveWithPK restated (zk_plonk_*) with the KZG SRS resident by range on two entries: every commitment is a composite commit
    (the polynomial stays on the prover's GPU, ranges of it travel to the other entry); key text and proof bytes are the golden ones."""
    all_true(run("plonk"))


def test_real_peer_devices_when_the_box_has_them():
    res = run("real_peers")
    if "skipped" in res:
        pytest.skip(res["skipped"])
    all_true(res)
