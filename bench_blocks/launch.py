"""`python3 bench.py --gpus N` typed WITHOUT a launcher: start the N ranks ourselves.

The contract's N > 1 command is `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`;
when RANK / WORLD_SIZE are absent this module builds exactly that command and runs it as a CHILD process (subprocess -- never os.exec*: a process that has
touched the GPU must not be replaced, and this one must not touch it at all: no torch.cuda call, no libzkmi load happens before or inside this module).
The child's rank 0 prints the ONE JSON line; it is relayed on our stdout with the launch recorded in `config.launch`, and the child's status is ours."""
import json
import os
import socket
import subprocess
import sys


def needs_self_launch(args) -> bool:
    """--gpus N > 1, not the one-process multi-device mode, and no launcher's environment around us."""
    return args.gpus > 1 and not args.single_process and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ


def free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def visible_gpus() -> int:
    """Devices this box shows, counted in a short-lived CHILD (torch.cuda.device_count() reads the driver's list without initialising a device on this image;
    asking from a child keeps even the HIP runtime's shared objects out of the launcher's own address space)."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def launch_command(bench_path: str, argv: list[str], n: int, port: int) -> list[str]:
    """The contract's launcher line for N ranks on one node, with our own arguments passed through unchanged (minus --dry-launch)."""
    passed = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), bench_path] + passed


def child_env(n: int, gpus: int) -> dict:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))  # torchrun would otherwise pin every rank to one thread, loudly
    if gpus < n and "ZKMI_DIST_BACKEND" not in env:
        # fewer devices than ranks: RCCL refuses two ranks on one device, so the exchanges go over gloo (host-staged) and ranks share GPUs --
        # the whole multi-process path still runs (a correctness run, not a scaling figure; the line says so)
        env["ZKMI_DIST_BACKEND"] = "gloo"
    return env


def self_launch(args, bench_path: str, argv: list[str]) -> int:
    n = args.gpus
    gpus = visible_gpus()
    port = free_port()
    cmd = launch_command(bench_path, argv, n, port)
    env = child_env(n, gpus)
    record = {"by": "bench.py itself (no RANK / WORLD_SIZE in the environment)", "command": cmd, "visible_gpus": gpus,
              "backend_env": env.get("ZKMI_DIST_BACKEND") or "nccl (RCCL)", "ranks_share_gpus": gpus < n}
    if args.dry_launch:
        print(json.dumps({"dry_launch": record}))
        return 0
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    line_seen = False
    for line in p.stdout:  # relay as it comes; the bench line (rank 0's JSON) gains the launch record
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                d = json.loads(s)
                d.setdefault("config", {})["launch"] = record
                s = json.dumps(d)
                line_seen = True
            except ValueError:
                pass
            print(s, flush=True)
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = p.wait()
    if rc == 0 and not line_seen:
        print("bench.py: the launched ranks exited 0 without printing the bench line", file=sys.stderr)
        return 1
    return rc
