#!/usr/bin/env python3
"""HBM traffic per launch of the kernels of one SpMV variant, from rocprofv3 PMC counters (development / evidence tool).

    python3 tools/pmc_kernels.py --case uniform --variant tiled [--rows N] [--match k_t2_]

Two child passes of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit the TCC's slots together and
are never mixed with tracing), each of which also runs an element-wise kernel of KNOWN traffic (x += y on 2^26 f32, 16 B per
lane), so the counters' bytes-per-unit are calibrated in the same pass instead of assumed (gfx950: FETCH_SIZE counts a 128-B
request as 64 B, MI355X_MICROARCH.md "HBM").  Prints, per kernel, the calibrated bytes read and written per launch, next to
the algorithmic bytes of the product.  The parent never touches the GPU.
"""
import argparse
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALIB_N = 1 << 26


def make_case(case, rows, k):
    import numpy as np
    from sparsemat_amd import synth
    if case in ("powerlaw", "powerlaw32"):
        dtype = np.float64 if case == "powerlaw" else np.float32
        return synth.crs_powerlaw(synth.SEED_MATRIX, rows, rows, dtype), dtype
    if case.startswith("lap"):
        g = int(case[3:])
        return synth.crs_laplace3d(g, g, g, np.float32), np.float32
    dtype = np.float64 if case.endswith("64") else np.float32
    pat = {"banded": synth.PATTERN_BANDED, "uniform": synth.PATTERN_UNIFORM, "diag": synth.PATTERN_DIAG,
           "window": synth.PATTERN_WINDOW}[case.replace("64", "")]
    return synth.crs_fixed(synth.SEED_MATRIX, pat, rows, k, dtype), dtype


def child(args):
    import ctypes as C
    import numpy as np
    import sparsemat_amd as sm
    from sparsemat_amd import _lib, synth
    lib, check = sm.lib(), _lib.check
    check(lib.smh_set_device(0))
    m, dtype = make_case(args.case, args.rows, args.k)
    n = m.n_rows()
    xbuf, xptr = synth.gen_x(synth.SEED_X, m.n_cols(), dtype)
    ybuf = synth.DeviceBuffer(n * np.dtype(dtype).itemsize)
    for _ in range(args.launches + 1):
        m.mvp_dev(xptr, m.n_cols(), ybuf.ptr, args.variant)
    check(lib.smh_device_synchronize())
    vs = np.dtype(dtype).itemsize
    print("ALGO %d" % (m.n_non_zero_entries() * (vs + 4) + (n + 1) * 4 + n * vs + m.n_cols() * vs), flush=True)
    print("NNZ %d" % m.n_non_zero_entries(), flush=True)
    a, b = sm.DenseVec.zeros(CALIB_N, np.float32), sm.DenseVec.zeros(CALIB_N, np.float32)
    for _ in range(3):
        a.add(b)
    check(lib.smh_device_synchronize())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="uniform")
    ap.add_argument("--variant", default="tiled")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--match", default="", help="only kernels whose name contains one of these (comma separated)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    algo = nnz = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="smh_pmck_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--child",
               "--case", args.case, "--variant", args.variant, "--rows", str(args.rows), "--k", str(args.k), "--launches", str(args.launches)]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        text = r.stdout.decode(errors="replace")
        for line in text.splitlines():
            if line.startswith("ALGO "):
                algo = int(line.split()[1])
            if line.startswith("NNZ "):
                nnz = int(line.split()[1])
        if r.returncode != 0:
            print(text[-3000:])
            sys.exit("rocprofv3 --pmc %s failed: rc=%d" % (counter, r.returncode))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") == counter:
                        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void smh::", "")
                        per[name][counter].append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
    cal = [k for k in per if "k_ew" in k]
    if not cal:
        sys.exit("no calibration kernel in the counter output")
    c = per[cal[0]]
    f_fetch = (2 * CALIB_N * 4) / (sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024.0)
    f_write = (CALIB_N * 4) / (sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024.0)
    print("case %s variant %s rows %d: nnz %s, algorithmic bytes per product %s" % (args.case, args.variant, args.rows, nnz, algo))
    print("calibration on %s: FETCH_SIZE x %.4f, WRITE_SIZE x %.4f bytes per counted KiB/1024 (guide: 2.0 / 1.0)" % (cal[0][:40], f_fetch, f_write))
    tot = 0.0
    for name, d in sorted(per.items()):
        if "k_ew" in name or (args.match and not any(m in name for m in args.match.split(","))):
            continue
        fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        if not fs or not ws:
            continue
        # the first launch of the child builds plans: the per-product launches are the last `launches`
        fs, ws = fs[-args.launches:], ws[-args.launches:]
        rd, wr = f_fetch * sum(fs) / len(fs) * 1024.0, f_write * sum(ws) / len(ws) * 1024.0
        tot += rd + wr
        line = "  %-60s read %8.3f GB  written %7.3f GB  (%d launches)" % (name[:60], rd / 1e9, wr / 1e9, len(fs))
        if nnz:
            line += "  = %.2f + %.2f B per entry" % (rd / nnz, wr / nnz)
        print(line)
    if algo:
        print("  sum %.3f GB per product = %.2f x the algorithmic bytes%s" % (tot / 1e9, tot / algo, ("; %.2f B per entry" % (tot / nnz)) if nnz else ""))


if __name__ == "__main__":
    main()
