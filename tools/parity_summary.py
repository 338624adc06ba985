#!/usr/bin/env python3
"""Summarise gpurun_out/parity_buckets.jsonl (written by tests/util.py:assert_spmv_close during a `pytest -m gpu` run):
per value type and row-length bucket, the rows compared and the worst excess of the device's error over the oracle's,
in units of sum_j |a_ij x_j| -- the quantity the parity bound limits to 1e-5 (f32) / 1e-12 (f64).
    python3 tools/parity_summary.py [file] > profiles/rNN_parity_buckets.txt"""
import collections
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_buckets.jsonl"
agg = collections.defaultdict(lambda: [0, -1e300])
n = 0
with open(path) as f:
    for line in f:
        rec = json.loads(line)
        n += 1
        for b, v in rec["buckets"].items():
            k = (rec["dtype"], b)
            agg[k][0] += v["rows"]
            agg[k][1] = max(agg[k][1], v["worst_excess"])
print("# %d device-vs-oracle SpMV comparisons of one `pytest tests -m gpu` run" % n)
print("# bound: |y_gpu - exact| <= |y_oracle - exact| + tol * sum|a x|, tol = 1e-5 (f32) / 1e-12 (f64); exact = f64 / 80-bit row sums")
print("# excess = (|y_gpu - exact| - |y_oracle - exact|) / sum|a x|  (<= 0: the device is at least as close as the reference's own fold)")
print("%-8s %-10s %12s %14s" % ("dtype", "row length", "rows", "worst excess"))
order = ["1-8", "9-32", "33-128", "129-512", "513-"]
for dt in ("float32", "float64"):
    for b in order:
        if (dt, b) in agg:
            print("%-8s %-10s %12d %14.3g" % (dt, b, agg[(dt, b)][0], agg[(dt, b)][1]))
