"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/bn254_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
