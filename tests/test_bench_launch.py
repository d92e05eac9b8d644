"""`python3 bench.py --gpus N` without a launcher: the command bench.py starts for its ranks, checked on CPU through --dry-launch.
The run itself (two ranks over gloo sharing one GPU) is tests/test_gpu_parity.py::test_two_process_sharded_proof_equals_single_process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BARE = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ZKMI_DIST_BACKEND")}


def dry(*extra, env=BARE):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-launch"] + list(extra), capture_output=True, text=True, timeout=300, env=env)
    return run


def test_bare_gpus_n_builds_the_contract_launcher_line():
    run = dry("--gpus", "8", "--steps", "7", "--warmup", "2")
    assert run.returncode == 0, run.stderr[-2000:]
    rec = json.loads(run.stdout.strip().splitlines()[-1])["dry_launch"]
    cmd = rec["command"]
    # python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]  # our own arguments, unchanged, without --dry-launch
    # this container shows no GPU: fewer devices than ranks -> the ranks are told to exchange over gloo and the record says they share devices
    assert rec["visible_gpus"] == 0 and rec["ranks_share_gpus"] is True and rec["backend_env"] == "gloo"


def test_under_a_launcher_bench_does_not_launch_again():
    from bench_blocks import launch

    class A:
        gpus, single_process = 4, False
    old = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE")}
    try:
        assert launch.needs_self_launch(A)
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "4"
        assert not launch.needs_self_launch(A)  # torchrun's environment: we are one of the ranks
        del os.environ["RANK"], os.environ["WORLD_SIZE"]
        A.gpus = 1
        assert not launch.needs_self_launch(A)
        A.gpus, A.single_process = 4, True
        assert not launch.needs_self_launch(A)  # one process driving N device entries: no ranks to start
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def test_launcher_does_not_touch_the_gpu_or_the_library():
    """The self-launch decision is taken before torch.cuda / libzkmi are imported: the dry launch must not have mapped libzkmi.so."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--dry-launch']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code\n"
            "maps = open('/proc/self/maps').read()\nassert 'libzkmi' not in maps and 'libamdhip64' not in maps, 'the launcher mapped GPU libraries'\n") % os.path.join(ROOT, "bench.py")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=BARE)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-2000:]
