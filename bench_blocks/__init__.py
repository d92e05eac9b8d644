"""The blocks of bench.py's JSON line, one module each; bench.py is the driver (argument parsing, the timed region, the line itself)."""
