#!/usr/bin/env python3
"""bench.py -- the contract benchmark: f32 CSR SpMV, 10M rows x 32 nnz/row per GPU (BASELINE C2/C5).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the resident matrix: y = A.x with A, x, y in HBM.
  N = 1  BASELINE configs[1]: SparseMatCRS<f32,u32>, 10,000,000 rows, 32 nnz/row, SURVEY 8d's primary pattern
         to the letter (--pattern window, the default: 32 distinct columns drawn without replacement from
         [i-4096, i+4096], stored ascending; DESIGN.md "Synthetic inputs"), kernel chosen by AUTO (K1r,
         lanes 8).  `other_pattern` carries the kernel time of the stratified subset (one column per 256-wide
         stratum: what rounds 1-2 put the headline on) measured in the same process.
  N > 1  weak scaling, BASELINE configs[4] shape: block b owns rows [b*10M, (b+1)*10M) of an (N*10M)-row
         matrix with global columns (SparseMatPar), step = ONE library call (smh_par_spmv_dev): the local SpMV into the block's
         slice of y + the exchange INSIDE libsparsemat_hip.so (csrc/par.hip) that makes y usable as the next x on every GPU -- a
         window exchange runs on a second stream beside the product of the block's interior rows (`exchange_hidden_ms` = the
         same K steps with that switched off minus the headline's; results are bit-identical):
         --exchange allgather: in-place ncclAllGather of the y slices; window: grouped ncclSend/ncclRecv of
         exactly the entries each block's columns reference (this banded matrix: 4096 entries from each
         neighbour); auto (default): the window when no block receives half of the vector or more.
         Launched by torch.distributed.run: one process per GPU, the ranks meet through smh_comm_*
         (ncclCommInitRank; the 128-byte id travels through a file in /tmp), rank = block.  Launched bare
         (`python bench.py --gpus N`): ONE process owns all N blocks (ncclCommInitAll, or
         SMH_PAR_BACKEND=peer for the direct peer-read exchange).  torch is not imported either way.
`value` = algorithmic bytes moved by all GPUs / wall time of the K timed steps (barrier + device
synchronisation on both sides, max over ranks).
Algorithmic bytes per GPU and step: nnz*(4+4) + (rows+1)*4 + rows*4 [y] + x_ref*4, where x_ref is the
number of distinct x entries the block's rows can reference (rows + band width).

One JSON line on stdout (rank 0).  `roofline` is the SpMV kernel alone (HIP events around every launch, on
the launch stream): `frac` = algorithmic bytes / mean kernel time / 8 TB/s as the metric defines it, and
`frac_traffic` = the HBM bytes the launch really moved (rocprofv3 PMC, two child passes of this script:
FETCH_SIZE with the gfx950 factor CALIBRATED in the same pass on an element-wise kernel of known traffic,
+ WRITE_SIZE) / mean kernel time / 8 TB/s -- the physical fraction; the kernel streams 16-bit ring-slot
columns, so it moves fewer bytes than the CSR arrays hold.  `cold` = the same launch after a 512 MiB
flush of L2 + Infinity Cache.  Should an optional N > 1 leg (exchange self-check, all-gather leg) hang, the
line is still printed, naming the leg, and every rank exits with status 3.  `cpu_baseline` is the CPU oracle (oracle/, the reference's algorithm
restated in C: the reference is Rust and cannot be built here), ONE thread like the reference, timed on
this host in this run; `cpu_baseline_all_cores` is the same loop spread over all cores with OpenMP --
NOT reference behaviour, reported so that the GPU/CPU ratio is not inflated by the reference being serial.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_PER_GPU = 10_000_000
NNZ_PER_ROW = 32
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0      # measured float4-copy ceiling, same guide
KERNEL_SUBSTR = "k_spmv_ring"
CALIB_SUBSTR = "k_ew"       # the element-wise kernel the PMC child pass runs for calibration
CALIB_N = 1 << 26           # x += y on 2^26 f32: reads 2 x 256 MiB, writes 256 MiB, 16 B per lane
FLUSH_BYTES = 512 << 20
WATCHDOG_EXIT = 3           # exit status of every rank when an optional N > 1 leg hung (the headline line is still printed)


def algorithmic_bytes(rows, nnz, x_ref):
    return nnz * 8 + (rows + 1) * 4 + rows * 4 + x_ref * 4


def pmc_traffic(args):
    """HBM bytes per SpMV launch from rocprofv3 PMC counters, or (None, note, None).  Runs BEFORE this process touches
    the GPU: two child passes of this script (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2 -> separate passes),
    never mixed with tracing.  Each pass also runs an element-wise kernel of KNOWN traffic (x += y, 16 B per lane like the
    SpMV's streams), from which the counter's bytes-per-unit is calibrated instead of assumed."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", None
    out, calib = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="smh_pmc_", dir="/tmp")
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--child", "--steps", "3", "--warmup", "1",
                   "--rows", str(args.rows), "--pattern", args.pattern]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            vals, cal = [], []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        if KERNEL_SUBSTR in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
                        elif CALIB_SUBSTR in row.get("Kernel_Name", ""):
                            cal.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if r.returncode != 0 or not vals or not cal:
                return None, "rocprofv3 --pmc %s: rc=%d, %d + %d samples" % (counter, r.returncode, len(vals), len(cal)), None
            out[counter] = sum(vals) / len(vals)
            calib[counter] = sum(cal) / len(cal)
    except Exception as e:  # profiling is best effort; the measurement itself never depends on it
        return None, "pmc pass failed: %r" % (e,), None
    # counters are in KiB; bytes per counted KiB from the kernel of known traffic (gfx950: FETCH_SIZE tallies a 128-B
    # request as 64 B -> ~2.0; WRITE_SIZE ~1.0)
    f_fetch = (2 * CALIB_N * 4) / (calib["FETCH_SIZE"] * 1024.0)
    f_write = (CALIB_N * 4) / (calib["WRITE_SIZE"] * 1024.0)
    cal_info = {"kernel": "k_ew<Add> x += y, %d f32 (reads %d B, writes %d B)" % (CALIB_N, 2 * CALIB_N * 4, CALIB_N * 4),
                "fetch_size_kib": calib["FETCH_SIZE"], "write_size_kib": calib["WRITE_SIZE"],
                "fetch_factor_measured": f_fetch, "write_factor_measured": f_write,
                "fetch_factor_guide": 2.0, "spmv_fetch_size_kib": out["FETCH_SIZE"], "spmv_write_size_kib": out["WRITE_SIZE"]}
    if not (1.7 <= f_fetch <= 2.3 and 0.85 <= f_write <= 1.15):
        return None, "PMC calibration off: fetch x%.3f write x%.3f" % (f_fetch, f_write), cal_info
    return ((f_fetch * out["FETCH_SIZE"] + f_write * out["WRITE_SIZE"]) * 1024.0,
            "(%.3f*FETCH_SIZE + %.3f*WRITE_SIZE) KiB; factors measured in the same passes on k_ew (guide: 2.0 / 1.0)" % (f_fetch, f_write),
            cal_info)


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a
    one-GPU job a share of its cores; 256 OpenMP threads on a 16-core share only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.999)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.999)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(rows, x_host, y_gpu_host, pattern):
    """Reference algorithm on one host core (the oracle), same workload, plus the parity gate; and the same loop on all
    cores (OpenMP), which is NOT what the reference does."""
    import numpy as np
    import oracle
    from sparsemat_amd import synth
    t0 = time.time()
    off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, pattern, rows, NNZ_PER_ROW, np.float32)
    gen_s = time.time() - t0

    def med5(fn):
        times, out = [], None
        for _ in range(5):
            t0 = time.perf_counter()
            out = fn()
            times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2], out

    med, y_ref = med5(lambda: oracle.spmv(off, col, val, x_host))
    cores = usable_cores()
    med_omp, (y_omp, threads) = med5(lambda: oracle.spmv_omp(off, col, val, x_host, cores))
    assert y_omp.tobytes() == y_ref.tobytes()
    b = algorithmic_bytes(rows, len(val), rows)
    # parity gate: componentwise |dy| <= 1e-5 * sum_j |a_ij x_j|
    scale = oracle.spmv_abs(off, col, val, x_host)
    err = np.abs(y_gpu_host.astype(np.float64) - y_ref.astype(np.float64))
    worst = float((err / np.maximum(scale, 1e-300)).max())
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    one = {
        "value": b / med / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "all %d rows x %d nnz (whole N=1 workload), median of 5 passes, %.3f s per pass" % (rows, NNZ_PER_ROW, med),
        "ms_per_step": med * 1e3, "gflops": 2.0 * len(val) / med / 1e9, "host_cpu": cpu_model,
        "host_cores_available": os.cpu_count(), "generate_s": gen_s,
        "parity_max_rel_err_vs_sum_abs": worst, "parity_ok": bool(worst <= 1e-5),
    }
    allc = {
        "value": b / med_omp / 1e9, "unit": "GB/s", "cores": threads, "kind": "port-openmp",
        "note": "NOT reference behaviour: the reference's mvp is serial (its mvp_par is commented out); same per-row loop, rows "
                "spread over all host cores, bit-identical y",
        "sample": "the same workload, median of 5 passes, %.4f s per pass" % med_omp, "ms_per_step": med_omp * 1e3,
    }
    return one, allc


class Events:
    """HIP events through the C ABI (smh_event_*): recorded on the stream the kernel is launched on."""

    def __init__(self, lib, check, n):
        self.lib, self.check = lib, check
        self.ev = []
        for _ in range(n):
            a, b = C.c_void_p(), C.c_void_p()
            check(lib.smh_event_create(C.byref(a)))
            check(lib.smh_event_create(C.byref(b)))
            self.ev.append((a, b))

    def start(self, i, stream):
        self.check(self.lib.smh_event_record(self.ev[i][0], C.c_void_p(stream or 0)))

    def stop(self, i, stream):
        self.check(self.lib.smh_event_record(self.ev[i][1], C.c_void_p(stream or 0)))

    def times_ms(self):
        out = []
        for a, b in self.ev:
            ms = C.c_float()
            self.check(self.lib.smh_event_elapsed_ms(a, b, C.byref(ms)))
            out.append(ms.value)
        return out


def exchange_check(lib, check, synth, np, par, x, y, n, comm, pattern):
    """N > 1, after a step: every local block holds, next to its own slice of y, the entries it RECEIVED in the exchange.
    It can generate any rows of the synthetic matrix and x is complete on every device, so it multiplies the 64 rows on
    either side of its slice itself (bit-exact K1s kernel) and compares them with what arrived.  A missing or stale halo
    is an O(1) error; two kernels' rounding differs by ~1e-6.  The worst error over all ranks goes into the JSON line."""
    tol, worst, rows_checked = 1e-4, 0.0, 0
    try:
        for b in range(par.n_local_blocks()):
            r0, r1, dev, _ = par.block(b)
            check(lib.smh_set_device(dev))
            for ra, rb in ((max(0, r0 - 64), r0), (r1, min(n, r1 + 64))):
                if ra >= rb:
                    continue
                small = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, NNZ_PER_ROW, np.float32, ra, rb)
                out = synth.DeviceBuffer((rb - ra) * 4)
                small.mvp_dev(x.ptr(b), n, out.ptr, "stream")
                check(lib.smh_device_synchronize())
                want = out.download(np.float32, rb - ra)
                got = np.empty(rb - ra, np.float32)
                check(lib.smh_dev_download(got.ctypes.data, C.c_void_p(y.ptr(b) + ra * 4), got.nbytes))
                worst = max(worst, float(np.max(np.abs(got.astype(np.float64) - want))))
                rows_checked += rb - ra
        check(lib.smh_set_device(par.block(0)[2]))
        err = None
    except Exception as e:  # the bench line is still worth printing
        err, worst = "%s: %s" % (type(e).__name__, e), float("inf")
    if comm is not None:
        worst = comm.max(worst)
    out = {"what": "rows received in the exchange against the receiver's own product of those rows",
           "rows_per_rank": rows_checked, "max_abs_err": worst if worst != float("inf") else None, "tol": tol, "ok": bool(worst <= tol)}
    if err:
        out["error"] = err
    return out


def time_launches(lib, check, fn, stream, sync, reps, warm):
    """`reps` launches of fn, each bracketed by HIP events recorded on the launch stream, after `warm` untimed ones."""
    for _ in range(warm):
        fn()
    sync()
    ev = Events(lib, check, reps)
    for i in range(reps):
        ev.start(i, stream)
        fn()
        ev.stop(i, stream)
    sync()
    return stats(ev.times_ms())


def warm_inspectors(synth, np):
    """The FIRST use of a kernel in a process pays for loading its code object (hundreds of ms for rocPRIM's sort instantiations):
    a per-process cost, not a per-matrix one.  Every inspector whose time the line reports is run once on a small matrix first."""
    small = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_WINDOW, 300_000, NNZ_PER_ROW, np.float32)
    small.prepare_stats("auto")
    small.prepare_stats("merge")
    for dtype in (np.float32, np.float64):
        t = synth.crs_powerlaw(synth.SEED_MATRIX, 200_000, 200_000, dtype)
        t.prepare_stats("tiled")
        t.prepare_stats("merge")
    lap = synth.crs_laplace3d(64, 64, 64, np.float32)
    lap.prepare_stats("auto")
    lap.prepare_stats("merge")


def inspector_cost(mat, variant, xptr, x_len, yptr, stream, lib, check, sync, chosen_ms):
    """What the chosen kernel's set-up costs against north_star's two plain row-major variants, which need none worth the name
    (K1 = a (sub-)wavefront per row, no plan, no derived arrays; K2 = merge path, a table of 2 u32 per 2048 items): the reference has
    no set-up at all (sparsemat_crs.rs:102-110 is a slice zip), so `break_even_products` = prepare_ms / (best plain kernel - chosen
    kernel) is the number of products after which the inspectors have paid for themselves."""
    prepare_ms, derived = mat.prepare_stats(variant)
    mat.set_ring(0)
    k1 = time_launches(lib, check, lambda: mat.mvp_dev(xptr, x_len, yptr, "vector", stream=stream), stream, sync, 5, 2)["median"]
    mat.set_ring(-1)
    k2 = time_launches(lib, check, lambda: mat.mvp_dev(xptr, x_len, yptr, "merge", stream=stream), stream, sync, 5, 2)["median"]
    plain = min(k1, k2)
    gain = plain - chosen_ms
    return {"prepare_ms": prepare_ms, "derived_bytes": derived,
            "plain_row_major_ms": {"vector_k1_no_ring": k1, "merge_k2": k2},
            "break_even_products": (prepare_ms / gain) if gain > 0 else None,
            "note": "prepare_ms = create-time inspection + smh_crs_prepare (host wall, device-synchronised; the process's first-use kernel "
                    "loads were paid before, on small matrices); derived_bytes = device memory the plan holds beside the CRS arrays; the "
                    "reference has no set-up step (sparsemat_crs.rs:102-110)"}


def parity_rows(np, oracle, y_gpu, blocks, x_host, tol, exact):
    """Sampled row blocks against the oracle (which generated them itself): bit-exact, or SURVEY 8d's componentwise bound
    |dy_i| <= tol * sum_j |a_ij x_j|.  blocks: [(row_begin, off, col, val)] with offsets rebased to the block."""
    worst, rows, same = 0.0, 0, True
    for rb, off, col, val in blocks:
        nr = len(off) - 1
        y_ref = oracle.spmv(off, col, val, x_host)
        got = y_gpu[rb:rb + nr]
        rows += nr
        if exact:
            same = same and got.tobytes() == y_ref.tobytes()
        scale = oracle.spmv_abs(off, col, val, x_host)
        err = np.abs(got.astype(np.float64) - y_ref.astype(np.float64)) / np.maximum(scale, 1e-300)
        worst = max(worst, float(err.max()))
    out = {"rows_checked": rows, "blocks": [(int(b[0]), int(len(b[1]) - 1)) for b in blocks], "max_rel_err_vs_sum_abs": worst, "tol": tol,
           "ok": bool(worst <= tol and (same or not exact))}
    if exact:
        out["bit_exact"] = bool(same)
    return out


def med_of(fn, reps):
    ts, out = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], out


def extra_configs(args, lib, check, sm, synth, np, stream, sync):
    """BASELINE configs[2] (C3) and configs[3] (C4: SpMV and CG per iteration) measured in the same process as the headline, each
    with HIP-event kernel times, `roofline.frac` by SURVEY 8d's algorithmic bytes, a parity gate on sampled row blocks against the
    oracle and a one-core CPU baseline on a stated sample.  PMC traffic for these kernels is collected separately
    (profiles/r04_pmc_*; tools/pmc_kernels.py) -- `traffic` stays null here."""
    import oracle
    out = {}
    # ---- C3: f64, power-law row lengths 1..2048 (alpha 1.52), uniform columns, AUTO ----------------------------------------
    t0 = time.perf_counter()
    n3 = args.c3_rows
    m3 = synth.crs_powerlaw(synth.SEED_MATRIX, n3, n3, np.float64)
    nnz3 = m3.n_non_zero_entries()
    xb3, xp3 = synth.gen_x(synth.SEED_X, n3, np.float64)
    yb3 = synth.DeviceBuffer(n3 * 8)
    variant3, lanes3 = m3.resolved_variant()
    prep3 = m3.prepare_stats("auto")  # (build before the clock; reported below)
    k3 = time_launches(lib, check, lambda: m3.mvp_dev(xp3, n3, yb3.ptr, "auto", stream=stream), stream, sync, 20, 3)
    b3 = nnz3 * 12 + (n3 + 1) * 4 + n3 * 8 + n3 * 8
    y3 = yb3.download(np.float64, n3)
    x3 = xb3.download(np.float64, n3)
    s_rows = min(n3, 1_000_000)
    starts = sorted({0, max(0, n3 // 2 - s_rows // 2), max(0, n3 - 50_000)})
    blocks = []
    for rb in starts:
        re = min(n3, rb + (s_rows if rb == starts[len(starts) // 2] else 50_000))
        blocks.append((rb,) + tuple(oracle.gen_powerlaw(synth.SEED_MATRIX, n3, n3, np.float64, row_begin=rb, row_end=re)))
    par3 = parity_rows(np, oracle, y3, blocks, x3, 1e-12, False)
    big = max(blocks, key=lambda b: len(b[1]))
    med, _ = med_of(lambda: oracle.spmv(big[1], big[2], big[3], x3), 5)
    share = len(big[3]) / nnz3
    cost3 = inspector_cost(m3, "auto", xp3, n3, yb3.ptr, stream, lib, check, sync, k3["mean"])
    out["C3"] = {
        "workload": "f64 CSR SpMV, %d rows, power-law row lengths 1..2048 (alpha 1.52, %d entries), uniform columns, 1xMI355X (BASELINE configs[2])" % (n3, nnz3),
        "dtype": "f64", "kernel": "%s (AUTO)%s" % (variant3, " = k_t3_expand + k_t3_reduce (two passes)" if variant3 == "tiled" else ""),
        "kernel_ms": k3["mean"], "kernel_ms_median": k3["median"], "kernel_ms_min": k3["min"], "kernel_ms_max": k3["max"], "launches": k3["launches"],
        "timing": "HIP events on the launch stream around every product (both passes)",
        "algorithmic_bytes": b3,
        "roofline": {"bound": "hbm", "achieved": b3 / (k3["mean"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": b3 / (k3["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "traffic_note": "PMC bytes per product for this kernel pair: profiles/r04_pmc_k2t_powerlaw.txt (separate rocprofv3 --pmc passes)"},
        "gflops": 2.0 * nnz3 / (k3["mean"] * 1e-3) / 1e9,
        "parity": par3,
        "cpu_baseline": {"value": b3 * share / med / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "rows [%d, %d) of the same matrix (%d entries = %.1f %% of the workload, all of x), median of 5 passes, %.3f s per pass; "
                                   "bytes = the workload's algorithmic bytes x that share" % (big[0], big[0] + len(big[1]) - 1, len(big[3]), 100 * share, med),
                         "ms_per_step_extrapolated": med / share * 1e3},
        "inspector": cost3, "set_up_s": time.perf_counter() - t0,
    }
    del m3, xb3, yb3, y3, x3, blocks, big
    # ---- C4: 7-point Laplacian on a g^3 grid, f32: the SpMV and the CG iteration ---------------------------------------------
    t0 = time.perf_counter()
    g = args.c4_grid
    n4 = g * g * g
    m4 = synth.crs_laplace3d(g, g, g, np.float32)
    nnz4 = m4.n_non_zero_entries()
    xb4, xp4 = synth.gen_x(synth.SEED_X, n4, np.float32)
    yb4 = synth.DeviceBuffer(n4 * 4)
    variant4, _ = m4.resolved_variant()
    m4.prepare_stats("auto")
    k4 = time_launches(lib, check, lambda: m4.mvp_dev(xp4, n4, yb4.ptr, "auto", stream=stream), stream, sync, 20, 3)
    b4 = nnz4 * 8 + (n4 + 1) * 4 + 2 * n4 * 4
    y4 = yb4.download(np.float32, n4)
    x4 = xb4.download(np.float32, n4)
    s_rows = min(n4, 8_000_000)
    starts = sorted({0, max(0, n4 // 2 - s_rows // 2), max(0, n4 - 70_000)})
    blocks = []
    for rb in starts:
        re = min(n4, rb + (s_rows if rb == starts[len(starts) // 2] else 70_000))
        blocks.append((rb,) + tuple(oracle.laplace3d_rows(g, g, g, rb, re, np.float32)))
    par4 = parity_rows(np, oracle, y4, blocks, x4, 1e-5, variant4 == "stream")
    big = max(blocks, key=lambda b: len(b[1]))
    med, _ = med_of(lambda: oracle.spmv(big[1], big[2], big[3], x4), 5)
    share = len(big[3]) / nnz4
    cost4 = inspector_cost(m4, "auto", xp4, n4, yb4.ptr, stream, lib, check, sync, k4["mean"])
    direct = m4.stream_direct() if variant4 == "stream" else False
    n_dict = len(m4.stream_value_dict()) if direct else 0
    out["C4_spmv"] = {
        "workload": "f32 CSR SpMV, 7-point Laplacian %d^3 (%d rows, %d entries), natural ordering, 1xMI355X (BASELINE configs[3], the product)" % (g, n4, nnz4),
        "dtype": "f32", "kernel": "%s (AUTO)%s" % (variant4, (" = k_spmv_stream_xd (x staged in LDS, 16-bit stage offsets, byte row lengths" + (
            "; K1s XD-V: the matrix's %d distinct values in a dictionary, their indices in the codes' spare bits, the value array not read)" % n_dict if n_dict else ")")) if direct else ""),
        "kernel_ms": k4["mean"], "kernel_ms_median": k4["median"], "kernel_ms_min": k4["min"], "kernel_ms_max": k4["max"], "launches": k4["launches"],
        "timing": "HIP events on the launch stream around every product",
        "algorithmic_bytes": b4,
        "roofline": {"bound": "hbm", "achieved": b4 / (k4["mean"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": b4 / (k4["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "traffic_note": "PMC bytes per product: profiles/r04_pmc_k1s_xd_lap512.txt (separate rocprofv3 --pmc passes)"},
        "gflops": 2.0 * nnz4 / (k4["mean"] * 1e-3) / 1e9,
        "parity": par4,
        "cpu_baseline": {"value": b4 * share / med / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "rows [%d, %d) of the same matrix (%d entries = %.1f %% of the workload), median of 5 passes, %.3f s per pass; bytes = the "
                                   "workload's algorithmic bytes x that share" % (big[0], big[0] + len(big[1]) - 1, len(big[3]), 100 * share, med),
                         "ms_per_step_extrapolated": med / share * 1e3},
        "inspector": cost4, "set_up_s": time.perf_counter() - t0,
    }
    del y4, x4, blocks, big, xb4, yb4
    # CG (linearsolver.rs:27-61): b = A.1, x0 = 0, a fixed number of iterations (f32 never reaches 1e-12 at this size)
    t0 = time.perf_counter()
    iters = args.cg_iters
    ones = sm.DenseVec.from_vec(np.ones(n4, np.float32))
    bvec = sm.DenseVec.zeros(n4, np.float32)
    m4.mvp_dev(ones.data_ptr(), n4, bvec.data_ptr(), "auto")
    check(lib.smh_device_synchronize())
    cg = sm.ConjugateGradient(0.0, iters, check_every=iters)
    xs = sm.DenseVec.zeros(n4, np.float32)
    cg.solve(m4, bvec, xs)  # warm: plans, graph instantiation
    per_it = []
    for _ in range(3):
        xs = sm.DenseVec.zeros(n4, np.float32)
        check(lib.smh_device_synchronize())
        t1 = time.perf_counter()
        cg.solve(m4, bvec, xs)
        check(lib.smh_device_synchronize())
        per_it.append((time.perf_counter() - t1) / cg.iterations * 1e3)
    per_it.sort()
    it_ms = per_it[len(per_it) // 2]
    b_cg = b4 + 9 * n4 * 4
    # what the solver reports against what its x really leaves: r = b - A x recomputed with the library's own kernels
    ax = sm.DenseVec.zeros(n4, np.float32)
    m4.mvp_dev(xs.data_ptr(), n4, ax.data_ptr(), "auto")
    check(lib.smh_device_synchronize())
    res = bvec.clone()
    res.sub(ax)
    true_r = float(res.norm())
    err1 = float(np.abs(xs.to_numpy() - 1.0).max())
    del ax, res, ones
    # the solver against the oracle where the oracle finishes in seconds: same code path (graph, fused dot, device scalars), a
    # 64^3 grid, the reference's stop rule.  tol 0.1: well above what f32 attains on this matrix -- from ~0.03 down the reference's
    # own SEQUENTIAL f32 dots (vector.rs:50-58) stall its convergence (102 / 128 / 156 iterations for 0.03 / 0.01 / 0.001) while
    # the device's tree sums keep going (120 for 0.001), so the iteration counts only agree above that
    gs, tol = min(g, 64), 1e-1
    ns = gs ** 3
    so, sc, sv = oracle.laplace3d(gs, gs, gs, np.float32)
    sb = oracle.spmv(so, sc, sv, np.ones(ns, np.float32))
    x_ref, it_ref, _ = oracle.cg(ns, ns, so, sc, sv, sb, np.zeros(ns, np.float32), tol=tol, iter_max=2000)
    small = sm.SparseMatCRS.from_raw_parts(ns, ns, so, sc, sv)
    xsm = np.zeros(ns, np.float32)
    cgs = sm.ConjugateGradient(tol, 2000)
    cgs.solve(small, sb, xsm)
    d_small = float(np.abs(xsm.astype(np.float64) - x_ref).max())
    # one core on a grid the oracle finishes in seconds: 10 iterations of the reference's recurrence
    gc = min(g, 160)
    nc = gc ** 3
    co, cc, cv = oracle.laplace3d(gc, gc, gc, np.float32)
    cb = oracle.spmv(co, cc, cv, np.ones(nc, np.float32))
    c_iters = 10
    t1 = time.perf_counter()
    oracle.cg(nc, nc, co, cc, cv, cb, np.zeros(nc, np.float32), tol=0.0, iter_max=c_iters)
    c_it = (time.perf_counter() - t1) / c_iters
    bc = len(cv) * 8 + (nc + 1) * 4 + 2 * nc * 4 + 9 * nc * 4
    out["C4_cg"] = {
        "workload": "ConjugateGradient::solve, 7-point Laplacian %d^3 f32, b = A.1, x0 = 0, device-resident, %d iterations per solve (BASELINE configs[3])" % (g, iters),
        "dtype": "f32", "kernel": "hipGraph per iteration: %s SpMV with the p.Ap partials in its epilogue + k_sum_stage1 + k_cg_par_update (alpha; r, r.r) + k_cg_par_p (beta, stop test; x, p)" % variant4,
        "ms_per_iteration": it_ms, "ms_per_iteration_min": per_it[0], "ms_per_iteration_max": per_it[-1], "solves": len(per_it), "iterations_per_solve": cg.iterations,
        "timing": "host wall clock around smh_cg_solve_vec (device-synchronised on both sides) / iterations, median of 3 solves; per-kernel times at this size: profiles/r04_cg_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/cg_bench.py)",
        "algorithmic_bytes": b_cg, "algorithmic_bytes_reference_op_sequence": b4 + 12 * n4 * 4,
        "roofline": {"bound": "hbm", "achieved": b_cg / (it_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": b_cg / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "traffic_note": "minimum-traffic schedule of SURVEY 8d: B_spmv + 9 n sizeof(T)"},
        "parity": {"oracle_case": "%d^3 f32, tol %g, reference stop rule: iterations device %d / oracle %d, max |x - x_oracle| %.3g" % (gs, tol, cgs.iterations, it_ref, d_small),
                   "ok": bool(abs(cgs.iterations - it_ref) <= 1 and d_small <= 10 * tol),
                   "full_size": {"iterations": cg.iterations, "r_norm_reported": float(np.sqrt(cg.r_norm_squared)), "r_norm_recomputed": true_r,
                                 "max_abs_err_vs_ones": err1}},
        "cpu_baseline": {"value": bc / c_it / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "the oracle's CG (linearsolver.rs:27-61 restated) on the same stencil at %d^3 (%d rows), %d iterations, %.3f s per iteration; "
                                   "bytes = B_spmv + 9 n sizeof(T) of that grid" % (gc, nc, c_iters, c_it),
                         "ms_per_iteration_extrapolated": c_it * 1e3 * n4 / nc},
        "set_up_s": time.perf_counter() - t0,
    }
    return out



def stats(ts):
    s = sorted(ts)
    return {"mean": sum(s) / len(s), "median": s[len(s) // 2], "min": s[0], "max": s[-1], "launches": len(s)}


def spawn_ranks(n):
    """N children of this process, one per GPU, each running this script with the launcher's environment; rank 0's JSON line is
    passed through on stdout; the exit status is the worst child's.  The children are killed by PID if the job overruns."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                SMH_BENCH_LAUNCH="bench.py --gpus %d (spawned ranks)" % n)
    procs = []
    log = None
    if "--rccl-probe" in sys.argv:
        log = "/tmp/smh_rccl_probe_%d.log" % os.getpid()
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        if log and r == 0:  # RCCL's own account of the job, from rank 0 (read back by rccl_report)
            env.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=log, SMH_BENCH_RCCL_LOG_PATH=log)
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    limit = float(os.environ.get("SMH_BENCH_SPAWN_TIMEOUT_S", "1500"))
    deadline = time.time() + limit
    worst, first_death = 0, None
    # all ranks are watched together: a rank that dies leaves its peers inside a collective, so they get a short grace and are
    # then killed by PID (rank 0's one line fits the pipe's buffer; it is read once everybody has gone)
    while any(pr.poll() is None for pr in procs):
        now = time.time()
        if first_death is None and any(pr.poll() not in (None, 0) for pr in procs):
            first_death = now
        if now > deadline or (first_death is not None and now > first_death + 15):
            sys.stderr.write("bench.py: %s -- killing the remaining ranks\n" % (
                "spawned ranks did not finish within %g s" % limit if now > deadline else "a rank exited with an error"))
            worst = WATCHDOG_EXIT
            break
        time.sleep(0.1)
    for pr in procs:
        if pr.poll() is None:
            pr.kill()
            pr.wait()
        if pr.returncode:
            worst = max(worst, abs(pr.returncode) if pr.returncode > 0 else WATCHDOG_EXIT)
    out0 = procs[0].stdout.read() or b""
    sys.stdout.buffer.write(out0)
    sys.stdout.flush()
    if log:
        try:
            os.remove(log)
        except OSError:
            pass
    return worst


def run_rccl_probe(n_gpus, rows, pattern_name):
    """Rank 0 of an N > 1 job, after the measurements: the same job once more in small (2 steps, all-gather exchange) as fresh
    processes with RCCL's log switched on -- the headline never runs with logging -- and what that log says.  Bounded; a probe
    that fails or overruns costs nothing but its own field."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                            "TORCHELASTIC_RUN_ID", "SMH_BENCH_LAUNCH")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus), "--rows", str(rows), "--steps", "2", "--warmup", "1",
           "--exchange", "allgather", "--pattern", pattern_name, "--rccl-probe"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(env, SMH_BENCH_SPAWN_TIMEOUT_S="75"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=100)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "probe exit status %d, %d line(s)" % (r.returncode, len(lines))}
        return json.loads(lines[-1]).get("rccl_probe")
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def rccl_report(comm, sm):
    """What RCCL itself says about this job (N > 1 lines carry it, so the first record from a node answers 'did RCCL see N ranks,
    which RCCL, ring or direct' by itself): communicator size and device from the library, the version, and -- when the run was
    started with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,TUNING NCCL_DEBUG_FILE=... (the spawn path and the launcher path both set
    it for rank 0 unless SMH_BENCH_RCCL_LOG=0) -- the lines of RCCL's own log that name the channels, rings / trees and the
    algorithm / protocol it chose per collective."""
    out = {}
    try:
        out["version"] = sm.Comm.rccl_version()
        if comm is not None:
            seen, dev = comm.ranks_seen()
            out["ranks_seen"], out["device"] = seen, dev
    except Exception as e:
        out["error"] = "%s: %s" % (type(e).__name__, e)
    path = os.environ.get("SMH_BENCH_RCCL_LOG_PATH")
    if path and os.path.exists(path):
        keep, algo = [], []
        try:
            with open(path, errors="replace") as f:
                for line in f:
                    t = line.strip()
                    if "Algo" in t and "proto" in t:
                        if len(algo) < 12 and t not in algo:
                            algo.append(t[-200:])
                    elif any(k in t for k in ("Channel 00", "Ring 00", "Trees", "Connected all", "channels", "comm 0x", "NCCL version", "RCCL version", "P2P", "xGMI", "XGMI")):
                        if len(keep) < 24:
                            keep.append(t[-200:])
            out["log_algo_proto"] = algo
            out["log_excerpt"] = keep
        except OSError as e:
            out["log_error"] = str(e)
    return out



def rendezvous_id(rank, make_id):
    """The 128-byte communicator id from rank 0 to the other ranks of this launch through a file in /tmp.  The name is
    keyed by the launcher (parent pid + its start time) and MASTER_PORT, so concurrent or earlier launches never collide."""
    ppid = os.getppid()
    start = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            start = f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    path = "/tmp/smh_comm_id_%d_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0"))
    if rank == 0:
        uid = make_id()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid, path
    deadline = time.time() + 600
    fresh_after = time.time() - 900  # (a leftover of a crashed launch with a recycled pid would be older than this)
    while time.time() < deadline:
        try:
            if os.path.getmtime(path) >= fresh_after:
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid, path
        except OSError:
            pass
        time.sleep(0.02)
    raise RuntimeError("rank %d: no communicator id at %s after 600 s" % (rank, path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU (default: the BASELINE size)")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--exchange", default="auto", choices=["auto", "window", "halo", "allgather"],
                    help="N>1: what a step exchanges after the local SpMV (auto: only the vector entries each "
                         "block references when that is less than half of the vector, else the all-gather)")
    ap.add_argument("--pattern", default="window", choices=["window", "stratified"],
                    help="column pattern of the synthetic matrix: 'window' = SURVEY 8d's primary workload to the letter (32 distinct "
                         "columns drawn without replacement from [i-4096, i+4096], stored ascending); 'stratified' = the easier-to-plan "
                         "subset of it the rounds before measured (one column per 256-wide stratum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # rocprofv3 --pmc child pass
    ap.add_argument("--one-process", action="store_true",
                    help="bare --gpus N > 1: ONE process owns all N blocks (ncclCommInitAll / peer reads) instead of the default, N child "
                         "processes of this one, one per GPU (ncclCommInitRank) -- every block's work is then issued by one host thread")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: append BASELINE configs[2] (C3) and configs[3] (C4 SpMV, C4 CG) to the line as `configs` "
                         "(auto: when --rows is the BASELINE size)")
    ap.add_argument("--rccl-probe", action="store_true", help=argparse.SUPPRESS)  # short N > 1 run with RCCL's own log switched on (see rccl_report)
    ap.add_argument("--c3-rows", type=int, default=10_000_000, help=argparse.SUPPRESS)
    ap.add_argument("--c4-grid", type=int, default=512, help=argparse.SUPPRESS)
    ap.add_argument("--cg-iters", type=int, default=100, help=argparse.SUPPRESS)
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # torch.distributed.run (or our own spawn): one process per GPU
    if not launched and args.gpus > 1 and not args.one_process and not args.child:
        # bare `bench.py --gpus N`: N fresh children, one per GPU, started BEFORE this process makes any GPU call (it never
        # does); they meet exactly as under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE, id file keyed by this pid)
        sys.exit(spawn_ranks(args.gpus))

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to stdout when its first communicator is
    # created (and libraries may chatter), so fd 1 is pointed at stderr for the run and the JSON goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world if launched else max(1, args.gpus)
    one_process = not launched and n_gpus > 1  # bare `bench.py --gpus N`: this process owns all N blocks

    traffic, traffic_note, calibration = None, "skipped", None
    if n_gpus == 1 and not launched and not args.child and not args.no_traffic:
        traffic, traffic_note, calibration = pmc_traffic(args)  # before this process initialises the GPU

    import numpy as np
    import sparsemat_amd as sm
    from sparsemat_amd import _lib, synth
    lib, check = sm.lib(), _lib.check

    n_dev = C.c_int(0)
    check(lib.smh_device_count(C.byref(n_dev)))
    if n_dev.value == 0:
        sys.exit("bench.py needs a HIP device: sparsemat_amd has no CPU fallback")
    # SMH_BENCH_SHARE_DEVICES=1 (rehearsal on a box with fewer GPUs): block b goes to device b % visible devices, so the
    # one-process N > 1 path -- adopted blocks, distributed vectors, the exchange (PEER: blocks share a device) -- runs anyway
    share = os.environ.get("SMH_BENCH_SHARE_DEVICES") == "1"
    if one_process and n_gpus > n_dev.value and not share:
        sys.exit("bench.py --gpus %d: only %d device(s) visible" % (n_gpus, n_dev.value))

    rows = args.rows
    n = rows * n_gpus
    pattern = synth.PATTERN_WINDOW if args.pattern == "window" else synth.PATTERN_BANDED
    pattern_text = ("columns drawn without replacement from [i-4096, i+4096], ascending" if args.pattern == "window"
                    else "banded-stratified columns (one per 256-wide stratum of [i-4096, i+4096))")
    band = min(n, 256 * NNZ_PER_ROW)
    exchange_req = "window" if args.exchange == "halo" else args.exchange
    comm = par = None
    rdzv_path = None

    # SMH_BENCH_FORCE_PAR=1: take the partitioned path (communicator, smh_par_*, distributed vectors) even with one GPU --
    # how the N > 1 code is rehearsed end to end on a one-GPU box
    force_par = os.environ.get("SMH_BENCH_FORCE_PAR") == "1"
    if n_gpus == 1 and not force_par:
        check(lib.smh_set_device(local_rank if launched else 0))
        if not args.child:
            warm_inspectors(synth, np)  # (so that prepare_ms below is the matrix's own set-up, not the process's first kernel loads)
        mat = synth.crs_fixed(synth.SEED_MATRIX, pattern, n, NNZ_PER_ROW, np.float32)
        blocks = [mat]
        xbuf, xptr = synth.gen_x(synth.SEED_X, n, np.float32)
        ybuf = synth.DeviceBuffer(n * 4)
        yptr = ybuf.ptr
        s = C.c_void_p()
        check(lib.smh_stream_create(C.byref(s)))
        stream = s.value
        exchange_mode, backend = "none", "none"

        def spmv():
            mat.mvp_dev(xptr, n, yptr, args.variant, stream=stream)

        def exchange():
            pass

        def sync():
            check(lib.smh_stream_synchronize(C.c_void_p(stream)))
    else:
        if launched:
            # (SMH_BENCH_SHARE_DEVICES: ranks share the visible devices -- only meaningful with the test suite's stand-in for RCCL,
            # tests/mock_rccl; real RCCL refuses two ranks on one device)
            check(lib.smh_set_device(local_rank % n_dev.value if share else local_rank))
            uid, rdzv_path = rendezvous_id(rank, sm.Comm.unique_id)
            comm = sm.Comm(uid, world, rank)  # ncclCommInitRank: collective
            blocks = [synth.crs_fixed(synth.SEED_MATRIX, pattern, n, NNZ_PER_ROW, np.float32, rank * rows, (rank + 1) * rows)]
            par = sm.SparseMatParLocal.for_rank(comm, n, blocks[0])
        else:
            blocks = []
            for d in range(n_gpus):
                check(lib.smh_set_device(d % n_dev.value))
                blocks.append(synth.crs_fixed(synth.SEED_MATRIX, pattern, n, NNZ_PER_ROW, np.float32, d * rows, (d + 1) * rows))
            check(lib.smh_set_device(0))
            par = sm.SparseMatParLocal.adopt(blocks, n)
        mat = blocks[0]
        x, y = par.vec(), par.vec()
        par.synchronize()  # (the vectors' zero fill runs on the blocks' streams)
        for b in range(par.n_local_blocks()):  # x born on every device (no PCIe): the same seeded vector everywhere
            check(lib.smh_set_device(par.block(b)[2]))
            synth.gen_x(synth.SEED_X, n, np.float32, ptr=x.ptr(b))
        check(lib.smh_set_device(par.block(0)[2]))
        exchange_mode, max_recv = par.exchange_mode(exchange_req)
        backend = par.backend()
        stream = par.block_stream(0)

        def spmv():
            par.mvp_dev(x, y, args.variant, "none")

        skip_exchange = os.environ.get("SMH_BENCH_SKIP_EXCHANGE") == "1"  # test knob: exchange_check must then fail

        def exchange():
            if not skip_exchange:
                par.exchange(y, exchange_req)

        def step():
            # ONE call: the local products and the exchange inside the library -- a window exchange runs on a second stream per
            # block beside the product of the block's interior rows (csrc/par.hip; bit-identical to product-then-exchange)
            par.mvp_dev(x, y, args.variant, "none" if skip_exchange else exchange_req)

        def sync():
            par.synchronize()

    nnz = mat.n_non_zero_entries()
    head_prepare = mat.prepare_stats(args.variant) if par is None else None  # (the plan is built here, before any clock starts)
    variant, lanes = mat.resolved_variant()
    _, ring_frac, ring_active, _, _ = mat.ring_plan()
    bytes_gpu = algorithmic_bytes(rows, nnz, min(n, rows + band))

    def barrier():
        sync()
        if comm is not None:
            comm.barrier()  # device drained, then all ranks meet (an RCCL all-reduce on the communicator)

    if par is None:
        def step():
            spmv()

    for _ in range(args.warmup):  # exactly the body of a timed step, so lazily created state exists before the clock starts
        step()
    barrier()
    ev = Events(lib, check, args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev.start(i, stream)  # HIP events on the launch stream (block 0 of this process): N = 1: around the SpMV kernel
        step()               # N > 1: y becomes usable as the next x on every GPU inside this call
        ev.stop(i, stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)
    kernel = stats(ev.times_ms())
    step_events = None
    if par is not None:
        # N > 1: the events above bracket block 0's whole step; the SpMV kernel alone (the roofline's launch) is timed here
        step_events = kernel
        kev = Events(lib, check, args.steps)
        for i in range(args.steps):
            kev.start(i, stream)
            spmv()
            kev.stop(i, stream)
        barrier()
        kernel = stats(kev.times_ms())

    if args.rccl_probe:
        if rank == 0:
            os.write(json_fd, (json.dumps({"rccl_probe": dict(rccl_report(comm, sm), exchange=exchange_mode, backend=backend, n_gpus=n_gpus,
                                                              ms_per_step=elapsed / args.steps * 1e3)}) + "\n").encode())
        if comm is not None:
            comm.barrier()
            if rank == 0 and rdzv_path:
                try:
                    os.remove(rdzv_path)
                except OSError:
                    pass
        par.close()
        if comm is not None:
            comm.close()
        return
    if args.child:
        # the calibration kernel of the PMC pass: x += y on CALIB_N f32 (known traffic, 16 B per lane)
        a, b = sm.DenseVec.zeros(CALIB_N, np.float32), sm.DenseVec.zeros(CALIB_N, np.float32)
        for _ in range(3):
            a.add(b)
        check(lib.smh_device_synchronize())
        return

    # cold: the same launch after 512 MiB of stores flushed L2 and the Infinity Cache (N = 1 only)
    cold = cold_reads = None
    if n_gpus == 1:
        flush = synth.DeviceBuffer(FLUSH_BYTES)
        cev = Events(lib, check, 20)
        for i in range(20):
            check(lib.smh_dev_memset(C.c_void_p(flush.ptr), i & 0xFF, FLUSH_BYTES, C.c_void_p(stream)))
            cev.start(i, stream)
            spmv()
            cev.stop(i, stream)
        sync()
        cold = stats(cev.times_ms())
        # ... and after 512 MiB of READS (a dot product of the flush buffer with itself): the caches hold nothing of the product's
        # either, but no dirty lines -- the stores above leave ~290 MB of them, written back while the timed launch runs
        scratch, res = synth.DeviceBuffer(int(lib.smh_blas_dot_scratch_bytes())), synth.DeviceBuffer(8)
        rev = Events(lib, check, 21)
        for i in range(21):
            check(lib.smh_blas_dot_dev(0, C.c_void_p(flush.ptr), C.c_void_p(flush.ptr), FLUSH_BYTES // 4, C.c_void_p(res.ptr),
                                       C.c_void_p(scratch.ptr), C.c_void_p(stream)))
            rev.start(i, stream)
            spmv()
            rev.stop(i, stream)
        sync()
        cold_reads = stats(rev.times_ms()[1:])  # (the first one still meets the memsets' dirty lines)
        del flush, scratch, res

    # rides along (N = 1, plain path): the other column pattern on the same box in the same process -- the rounds before this
    # one put the headline on the stratified subset of SURVEY 8d's window matrices; both numbers stay visible
    other = None
    if n_gpus == 1 and par is None and os.environ.get("SMH_BENCH_NO_OTHER_PATTERN") != "1":
        o_name = "stratified" if args.pattern == "window" else "window"
        o_mat = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED if args.pattern == "window" else synth.PATTERN_WINDOW,
                                n, NNZ_PER_ROW, np.float32)
        for _ in range(3):
            o_mat.mvp_dev(xptr, n, yptr, args.variant, stream=stream)
        oev = Events(lib, check, 20)
        for i in range(20):
            oev.start(i, stream)
            o_mat.mvp_dev(xptr, n, yptr, args.variant, stream=stream)
            oev.stop(i, stream)
        sync()
        o = stats(oev.times_ms())
        other = {"pattern": o_name, "kernel_ms": o["mean"], "kernel_ms_median": o["median"], "kernel_ms_min": o["min"],
                 "kernel_ms_max": o["max"], "launches": o["launches"],
                 "frac": bytes_gpu / (o["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "kernel": "%s lanes=%d" % o_mat.resolved_variant()}
        del o_mat
        if not args.no_cpu_baseline:  # y of the headline matrix for the parity gate below
            spmv()
            sync()

    ms_per_step = elapsed / args.steps * 1e3
    value = bytes_gpu * n_gpus / (elapsed / args.steps) / 1e9
    kernel_ms = kernel["mean"]
    achieved = bytes_gpu / (kernel_ms * 1e-3) / 1e9
    traffic_rate = (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None
    if n_gpus == 1:
        workload = "f32 CSR SpMV, %d rows x %d nnz/row, %s, 1xMI355X" % (rows, NNZ_PER_ROW, pattern_text)
        parallelism = "single GPU"
    else:
        how = "one process per GPU (ncclCommInitRank)" if launched else "one process, %d devices (%s)" % (
            n_gpus, "ncclCommInitAll" if backend == "rccl" else "peer reads")
        workload = ("f32 CSR SpMV, %d rows (%d per GPU) x %d nnz/row, %s, row-partitioned over %d GPUs, "
                    "exchange of y per step inside libsparsemat_hip.so: %s over %s, %s" % (n, rows, NNZ_PER_ROW, pattern_text, n_gpus, exchange_mode, backend, how))
        parallelism = "rows/%d + %s exchange (%s)" % (n_gpus, exchange_mode, backend)
    result = {
        "metric": "csr_spmv_effective_hbm_GBps", "value": value, "unit": "GB/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": workload, "pattern": args.pattern, "rows_per_gpu": rows, "nnz_per_gpu": nnz, "index": "u32",
            "kernel": "%s lanes=%d ring=%s (ring rows %.3f)" % (variant, lanes, ring_active, ring_frac),
            "parallelism": parallelism, "exchange": exchange_mode, "exchange_backend": backend,
            "launch": (os.environ.get("SMH_BENCH_LAUNCH") or "torch.distributed.run, one process per GPU") if launched else ("one process" if n_gpus > 1 else "single process"),
        },
        "gflops": 2.0 * nnz * n_gpus / (elapsed / args.steps) / 1e9,
        # per step: the local SpMV kernel (HIP events, block 0 of rank 0) and what the rest of the step costs (the exchange
        # and launch gaps; max over ranks is in ms_per_step)
        "spmv_kernel_ms": kernel_ms, "step_minus_kernel_ms": ms_per_step - kernel_ms,
        "pct_of_hbm_peak": 100.0 * value / (HBM_PEAK_GBPS * n_gpus),
        "algorithmic_bytes_per_gpu_step": bytes_gpu,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            # the PHYSICAL fraction: HBM bytes the launch really moved (PMC) / kernel time / peak.  `frac` above is the
            # metric's definition (algorithmic CSR bytes / time); it is larger because ring phases stream a 16-bit column
            # array (6 instead of 8 bytes per f32 entry leave HBM).  Lead with frac_traffic when judging the kernel.
            "frac_traffic": (traffic_rate / HBM_PEAK_GBPS) if traffic else None,
            "traffic_rate": traffic_rate,
            "traffic_over_algorithmic": (traffic / bytes_gpu) if traffic else None,
            "traffic_frac_of_measured_copy": (traffic_rate / HBM_COPY_GBPS) if traffic else None,
            "traffic_note": traffic_note, "traffic_calibration": calibration,
            "kernel": KERNEL_SUBSTR if ring_active else variant,
            "kernel_ms": kernel_ms, "kernel_ms_median": kernel["median"], "kernel_ms_min": kernel["min"],
            "kernel_ms_max": kernel["max"], "launches": kernel["launches"],
            "frac_median": bytes_gpu / (kernel["median"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "algorithmic_bytes": bytes_gpu,
            "cold": None if cold is None else {
                "flush": "%d MiB of stores before every launch (L2 + Infinity Cache)" % (FLUSH_BYTES >> 20),
                "kernel_ms_median": cold["median"], "kernel_ms_min": cold["min"], "kernel_ms_max": cold["max"],
                "launches": cold["launches"], "frac": bytes_gpu / (cold["median"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_traffic": (traffic / (cold["median"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            },
            "cold_after_reads": None if cold_reads is None else {
                "flush": "%d MiB of READS before every launch (nothing of the product in L2 / Infinity Cache, no dirty lines to write back)" % (FLUSH_BYTES >> 20),
                "kernel_ms_median": cold_reads["median"], "kernel_ms_min": cold_reads["min"], "kernel_ms_max": cold_reads["max"],
                "launches": cold_reads["launches"], "frac": bytes_gpu / (cold_reads["median"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_traffic": (traffic / (cold_reads["median"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            },
            "layout": "f32 values + 16-bit ring-slot columns for LDS-ring phases (u32 columns kept for the rest)",
        },
    }
    if other is not None:
        result["other_pattern"] = other
    if par is not None:
        # what RCCL itself reports: ncclCommCount of this rank's communicator (one process per GPU); the one-process mode's
        # communicators (ncclCommInitAll) live inside the library's blocks and are not asked
        result["rccl"] = rccl_report(comm, sm) if backend == "rccl" else {"note": "exchange backend '%s': no RCCL call in this run" % backend}
        result["ranks_seen"] = result["rccl"].get("ranks_seen") if comm is not None else None
    if head_prepare is not None and not args.child:
        result["inspector"] = inspector_cost(mat, args.variant, xptr, n, yptr, stream, lib, check, sync, kernel_ms)
        if not args.no_cpu_baseline:  # (the plain kernels wrote y: the headline matrix's product again for the parity gate)
            spmv()
            sync()
    if step_events is not None:
        result["step_ms_block0_events"] = step_events["mean"]
    # The optional legs of an N > 1 run come AFTER the headline is complete, under a watchdog: should one of them hang (it is
    # the first time this code meets a real multi-GPU node), every rank gives up after 180 s (SMH_BENCH_WATCHDOG_S), rank 0 prints the
    # line without them (naming the leg that hung) and every rank exits with status 3: the measured headline is never lost to an
    # extra, and a hung collective never looks like success.
    watchdog = None
    if par is not None and n_gpus > 1:
        import threading

        patience = float(os.environ.get("SMH_BENCH_WATCHDOG_S", "180"))

        leg = {"name": "exchange_check"}

        def give_up():
            sys.stderr.write("bench.py: rank %d: optional leg '%s' did not finish within %g s -- exit status %d\n" % (
                rank, leg["name"], patience, WATCHDOG_EXIT))
            if rank == 0:
                result["cpu_baseline"] = None
                result["optional_legs"] = "gave up after %g s in '%s' (a GPU / collective hang: exit status %d)" % (
                    patience, leg["name"], WATCHDOG_EXIT)
                result["hung_leg"] = leg["name"]
                os.write(json_fd, (json.dumps(result) + "\n").encode())
            os._exit(WATCHDOG_EXIT)
        watchdog = threading.Timer(patience, give_up)
        watchdog.daemon = True
        watchdog.start()
    xcheck = allgather_leg = None
    if par is not None and n_gpus > 1:
        if os.environ.get("SMH_BENCH_HANG_IN_LEGS") == "1":  # test knob: what the watchdog is for
            time.sleep(10 ** 6)
        xcheck = exchange_check(lib, check, synth, np, par, x, y, n, comm, pattern)  # (collective: every rank calls it)
        # what running the exchange beside the interior rows' product hides: the same K steps once more with the overlap off
        # (product, THEN exchange on the same stream) -- bit-identical results, only the schedule differs
        if exchange_mode == "window" and not skip_exchange:
            leg["name"] = "no_overlap_leg"
            par.set_overlap(False)
            for _ in range(min(args.warmup, 2) + 1):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el = time.perf_counter() - t0
            if comm is not None:
                el = comm.max(el)
            par.set_overlap(True)
            result["ms_per_step_no_overlap"] = el / args.steps * 1e3
            result["exchange_hidden_ms"] = el / args.steps * 1e3 - ms_per_step
            result["interior_rows_block0"] = list(par.interior(0, args.variant))
        # BASELINE configs[4] names the all-gather of the dense vector: when AUTO chose the cheaper window exchange, the same
        # K steps are timed once more with the in-place all-gather and reported beside the headline (never instead of it)
        if exchange_mode != "allgather" and os.environ.get("SMH_BENCH_NO_ALLGATHER_LEG") != "1":
            leg["name"] = "allgather_leg"
            try:
                for _ in range(min(args.warmup, 2) + 1):
                    spmv()
                    par.exchange(y, "allgather")
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    spmv()
                    par.exchange(y, "allgather")
                barrier()
                el = time.perf_counter() - t0
                if comm is not None:
                    el = comm.max(el)
                allgather_leg = {"exchange": "allgather", "ms_per_step": el / args.steps * 1e3,
                                 "value": bytes_gpu * n_gpus / (el / args.steps) / 1e9, "unit": "GB/s",
                                 "received_bytes_per_gpu_step": (n - rows) * 4,
                                 "exchange_check": exchange_check(lib, check, synth, np, par, x, y, n, comm, pattern)}
            except Exception as e:
                allgather_leg = {"exchange": "allgather", "error": "%s: %s" % (type(e).__name__, e)}

    if watchdog is not None:
        watchdog.cancel()
    if xcheck is not None:
        result["exchange_check"] = xcheck
    if allgather_leg is not None:
        result["allgather_leg"] = allgather_leg
    # (decided identically on every rank: rank 0 runs the probe, the others wait for it on the host)
    probe_wanted = bool(n_gpus > 1 and launched and backend == "rccl" and rdzv_path and os.environ.get("SMH_BENCH_RCCL_PROBE", "0" if share else "1") == "1")
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            if par is None:
                x_host, y_host = xbuf.download(np.float32, n), ybuf.download(np.float32, n)
            else:  # (SMH_BENCH_FORCE_PAR: the partitioned path with one block)
                x_host, y_host = x.download_block(0), y.download()
            one, allc = cpu_baseline(rows, x_host, y_host, pattern)
            result["cpu_baseline"], result["cpu_baseline_all_cores"] = one, allc
        else:
            result["cpu_baseline"] = None
        want_configs = args.configs == "on" or (args.configs == "auto" and rows == ROWS_PER_GPU)
        if n_gpus == 1 and par is None and want_configs:
            # the other single-GPU BASELINE configs, after the headline is complete; the headline's buffers go first
            del mat, blocks, xbuf, ybuf
            try:
                result["configs"] = extra_configs(args, lib, check, sm, synth, np, stream, sync)
            except Exception as e:
                result["configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if probe_wanted:
            result.setdefault("rccl", {})["probe"] = run_rccl_probe(n_gpus, rows, args.pattern)
            try:  # (the other ranks wait for this file on the HOST: nobody sits inside a collective while the probe's processes use the GPUs)
                with open(rdzv_path + ".probe_done", "w") as f:
                    f.write("done")
            except OSError:
                pass
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    elif probe_wanted and rdzv_path:
        deadline = time.time() + 130  # (the probe is bounded to 100 s)
        while time.time() < deadline and not os.path.exists(rdzv_path + ".probe_done"):
            time.sleep(0.05)
    if comm is not None:
        comm.barrier()
        if rank == 0 and rdzv_path and probe_wanted:
            try:
                os.remove(rdzv_path + ".probe_done")
            except OSError:
                pass
        if rank == 0 and rdzv_path:
            try:
                os.remove(rdzv_path)
            except OSError:
                pass
    if par is not None:
        par.close()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
