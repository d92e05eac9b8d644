"""Probe for a Go toolchain + module cache: the day one exists, tools/go_pin pins the oracle against the real gnark / gnark-crypto."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import ROOT, N_PUBLIC, HBM_PEAK_GBS, R_FR, seed_at


def go_toolchain_probe():
    """is there a Go toolchain (and gnark's module cache) on this box?  If so tools/go_pin checks the committed fixtures against
    the real gnark / gnark-crypto (go.mod:5,23) and the counts are reported; otherwise the oracle stays the checker ("parity unpinned")."""
    import shutil
    import subprocess
    go = shutil.which("go")
    out = {"go": go, "version": None, "module_cache": None, "go_pin": None}
    if not go:
        return out
    try:
        out["version"] = subprocess.run([go, "version"], capture_output=True, text=True, timeout=30).stdout.strip()
        cache = subprocess.run([go, "env", "GOMODCACHE"], capture_output=True, text=True, timeout=30).stdout.strip()
        have = os.path.isdir(os.path.join(cache, "github.com", "consensys")) if cache else False
        out["module_cache"] = {"path": cache, "has_consensys_modules": have}
        if have:
            r = subprocess.run([go, "run", "."], cwd=os.path.join(ROOT, "tools", "go_pin"), capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, GOFLAGS="-mod=mod", GOPROXY="off"))
            txt = r.stdout + r.stderr
            out["go_pin"] = {"rc": r.returncode, "pass": txt.count("PASS"), "fail": txt.count("FAIL"), "tail": txt[-400:]}
    except Exception as e:
        out["error"] = str(e)[:200]
    return out
