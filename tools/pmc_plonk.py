"""PLONK alone under a PMC pass: one plonk.Setup + proofs at 2^log gates and nothing else, so that every k_accumulate dispatch of the process is one of PLONK's.
    rocprofv3 --pmc FETCH_SIZE -d <dir> --output-format csv -- python3 tools/pmc_plonk.py [log_gates] [reps]
(the program itself after `--`: the profiler's preloaded library initialises the GPU before Python starts)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noir_backend_using_gnark_amd import _lib  # noqa: E402
from bench_blocks.plonk import plonk_block  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
blk = plonk_block(_lib.lib(), _lib, log_n, reps=reps)
print(json.dumps({k: blk.get(k) for k in ("prove_ms", "proof_verifies", "same_bytes_both_ways")}))
