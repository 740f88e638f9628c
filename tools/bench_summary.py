#!/usr/bin/env python3
"""Reads bench.py's JSON line from stdin and prints a one-line summary."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d["config"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "%.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "nnzL", c["nnz_L"], "levels", c["levels"],
      "factor_s %.1f" % c["factor_numeric_s"], {k: round(v, 3) for k, v in d["roofline"]["phases_ms_per_iter"].items()})
