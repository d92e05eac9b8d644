"""One-off: the Go-ABI shim with the reference's real SRS size (1,000,000 points): time of the first call (generate + save srs.hex) and of a second
process (load srs.hex).  usage: python tools/dbg/goffi_full_srs.py"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def felts_wire(values):  # fr.Vector.MarshalBinary: u32 BE count | count x 32 B BE (no oracle in tools/)
    return len(values).to_bytes(4, "big") + b"".join((v % R).to_bytes(32, "big") for v in values)


R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
e = json.load(open(os.path.join(ROOT, "tests", "golden", "plonk_golden.json")))[0]
values = [int(v, 16) for v in e["values"]]
tmp = tempfile.mkdtemp()
job = dict(what="plonk", acir=json.dumps(e["acir"]), values=felts_wire(values).hex(), values_wrong_public=felts_wire(values).hex(), random_values=felts_wire(values).hex())
f = os.path.join(tmp, "job.json")
json.dump(job, open(f, "w"))
env = dict(os.environ, XDG_CONFIG_HOME=os.path.join(tmp, "cfg"), PYTHONPATH=ROOT)
env.pop("ZKMI_TEST_NEW_SRS_SIZE", None)
os.makedirs(os.path.join(tmp, "cfg"))
out = {}
for name in ("first_process_generates", "second_process_loads"):
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "goffi_worker.py"), f], capture_output=True, text=True, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out[name] = {"wall_s": round(dt, 2), "verifies": d["verifies"]}
out["srs_hex_bytes"] = os.path.getsize(os.path.join(tmp, "cfg", "noir-lang", "srs.hex"))
print(json.dumps(out))
