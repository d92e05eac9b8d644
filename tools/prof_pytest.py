import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
import pytest
pr = cProfile.Profile()
pr.enable()
t0 = time.time()
rc = pytest.main(["tests/test_gpu_plonk.py", "-q", "-x", "-k", sys.argv[1]])
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("wall", time.time() - t0, "rc", rc)
print(s.getvalue()[:9000])
