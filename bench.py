#!/usr/bin/env python3
"""bench.py -- the contract benchmark: f32 CSR SpMV, 10M rows x 32 nnz/row per GPU (BASELINE C2/C5).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the resident matrix: y = A.x with A, x, y in HBM.
  N = 1  BASELINE configs[1]: SparseMatCRS<f32,u32>, 10,000,000 rows, 32 nnz/row, banded-stratified
         columns (DESIGN.md "Synthetic inputs"), kernel chosen by SMH_SPMV_AUTO (K1r, lanes 8).
  N > 1  weak scaling, BASELINE configs[4] shape: rank r owns rows [r*10M, (r+1)*10M) of an
         (N*10M)-row matrix with global columns (SparseMatPar), step = local SpMV + ONE RCCL exchange
         that makes y usable as the next x on every rank.  --exchange allgather: all-gather of the y
         slices (every rank gets the full vector); halo: every rank gets exactly the entries its block
         references (for this banded matrix: 4096 entries from each neighbour, one grouped launch of
         point-to-point sends / receives on slices of the vector itself);
         auto (default) picks halo when that is less than half of the vector.
`value` = algorithmic bytes moved by all ranks / wall time of the K timed steps (max over ranks).
Algorithmic bytes per rank and step: nnz*(4+4) + (rows+1)*4 + rows*4 [y] + x_ref*4, where x_ref is
the number of distinct x entries the rank's rows can reference (rows + band width).

One JSON line on stdout (rank 0).  `roofline` is the SpMV kernel alone (HIP events around every
launch, on the launch stream); `traffic` comes from two rocprofv3 --pmc child passes of this same
script (FETCH_SIZE x2 per the gfx950 correction, + WRITE_SIZE, KiB units); `cpu_baseline` is the
CPU oracle (oracle/, the reference's algorithm restated in C: the reference is Rust and cannot be
built here), one thread, timed on this host in this run.
"""
import argparse
import csv
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


def algorithmic_bytes(rows, nnz, x_ref):
    return nnz * 8 + (rows + 1) * 4 + rows * 4 + x_ref * 4


def pmc_traffic(args):
    """HBM bytes per SpMV launch from rocprofv3 PMC counters, or None.  Runs BEFORE this process
    touches the GPU: two child passes of this script (TCC has 4 slots: FETCH_SIZE takes 3,
    WRITE_SIZE 2 -> separate passes)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="smh_pmc_", dir="/tmp")
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--child", "--steps", "3", "--warmup", "1",
                   "--rows", str(args.rows)]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if KERNEL_SUBSTR in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            vals.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if r.returncode != 0 or not vals:
                return None, "rocprofv3 --pmc %s: rc=%d, %d samples" % (counter, r.returncode, len(vals))
            out[counter] = sum(vals) / len(vals)
    except Exception as e:  # profiling is best effort; the measurement itself never depends on it
        return None, "pmc pass failed: %r" % (e,)
    # units KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, "2*FETCH_SIZE+WRITE_SIZE (KiB), gfx950 correction"


def cpu_baseline(rows, x_host, y_gpu_host):
    """Reference algorithm on one host core (the oracle), same workload, plus the parity gate."""
    import numpy as np
    import oracle
    from sparsemat_amd import synth
    t0 = time.time()
    off, col, val = oracle.gen_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, rows, NNZ_PER_ROW, np.float32)
    gen_s = time.time() - t0
    times = []
    y_ref = None
    for _ in range(5):
        t0 = time.perf_counter()
        y_ref = oracle.spmv(off, col, val, x_host)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
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
    return {
        "value": b / med / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "all %d rows x %d nnz (whole N=1 workload), median of 5 passes, %.3f s per pass" % (rows, NNZ_PER_ROW, med),
        "ms_per_step": med * 1e3, "gflops": 2.0 * len(val) / med / 1e9, "host_cpu": cpu_model,
        "host_cores_available": os.cpu_count(), "generate_s": gen_s,
        "parity_max_rel_err_vs_sum_abs": worst, "parity_ok": bool(worst <= 1e-5),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU (default: the BASELINE size)")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo", "allgather"],
                    help="N>1: what a step exchanges after the local SpMV (auto: only the vector entries each "
                         "rank's block references when that is less than half of the vector, else the all-gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # rocprofv3 --pmc child pass
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to stdout when its first communicator is
    # created (and libraries may chatter), so fd 1 is pointed at stderr for the run and the JSON goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    traffic, traffic_note = None, "skipped"
    if world == 1 and "RANK" not in os.environ and not args.child and not args.no_traffic:
        traffic, traffic_note = pmc_traffic(args)  # before this process initialises the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    import sparsemat_amd as sm
    from sparsemat_amd import synth
    from sparsemat_amd.sparsemat_par import HipBlock, SparseMatPar

    torch.cuda.set_device(local_rank)
    sm._lib.check(sm.lib().smh_set_device(local_rank))
    # SMH_BENCH_FORCE_DIST=1: exercise the RCCL path (process group + in-place all-gather) even with one rank
    force_dist = os.environ.get("SMH_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rows = args.rows
    n = rows * world
    begin, end = rank * rows, (rank + 1) * rows
    mat = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, n, NNZ_PER_ROW, np.float32, begin, end)
    nnz = mat.n_non_zero_entries()
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    synth.gen_x(synth.SEED_X, n, np.float32, ptr=x.data_ptr())
    y = torch.zeros(n, dtype=torch.float32, device="cuda")
    par = SparseMatPar(world, n, n, rank, HipBlock(mat, args.variant))
    exchange_mode = par.setup_window_exchange(x, args.exchange) if use_dist else "none"
    variant, lanes = mat.resolved_variant()
    _, ring_frac, ring_active, _, _ = mat.ring_plan()
    band = min(n, 256 * NNZ_PER_ROW)
    bytes_rank = algorithmic_bytes(rows, nnz, min(n, rows + band))

    y_local = y[begin:end]

    def step():
        # exactly the body of a timed step: local SpMV into this rank's slice of y, then (N > 1) the exchange that
        # makes y usable as the next x -- so the warm-up also creates whatever the first exchange sets up lazily
        par.mvp_local(x, y_local)
        if use_dist:
            par.exchange_window(y)

    stream = torch.cuda.current_stream()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    # HIP events around every SpMV kernel launch (same stream as the launch)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        par.mvp_local(x, y_local)
        ev[i][1].record(stream)
        if use_dist:
            par.exchange_window(y)  # y becomes usable as the next x on this rank
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.child:
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = bytes_rank * world / (elapsed / args.steps) / 1e9
    achieved = bytes_rank / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "csr_spmv_effective_hbm_GBps", "value": value, "unit": "GB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": ("f32 CSR SpMV, %d rows x %d nnz/row, banded-stratified columns, 1xMI355X" % (rows, NNZ_PER_ROW))
            if world == 1 else
            ("f32 CSR SpMV, %d rows (%d per GPU) x %d nnz/row, banded-stratified columns, row-partitioned over %d GPUs, "
             "RCCL exchange of y per step: %s" % (n, rows, NNZ_PER_ROW, world, exchange_mode)),
            "rows_per_gpu": rows, "nnz_per_gpu": nnz, "index": "u32", "kernel": "%s lanes=%d ring=%s (ring rows %.3f)" % (variant, lanes, ring_active, ring_frac),
            "parallelism": "rows/%d + %s exchange (RCCL)" % (world, exchange_mode) if world > 1 else "single GPU",
            "exchange": exchange_mode,
        },
        "gflops": 2.0 * nnz * world / (elapsed / args.steps) / 1e9,
        # per step: the local SpMV kernel (HIP events, this rank) and what the rest of the step costs (the exchange and
        # launch gaps; max over ranks is in ms_per_step)
        "spmv_kernel_ms": kernel_ms, "step_minus_kernel_ms": ms_per_step - kernel_ms,
        "pct_of_hbm_peak": 100.0 * value / (HBM_PEAK_GBPS * world),
        "algorithmic_bytes_per_gpu_step": bytes_rank,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_note": traffic_note, "kernel": KERNEL_SUBSTR if ring_active else variant,
            "kernel_ms": kernel_ms, "algorithmic_bytes": bytes_rank, "frac_of_measured_copy": achieved / HBM_COPY_GBPS,
            # `achieved` is the ALGORITHMIC bytes of the CSR SpMV over the kernel time, as the metric is
            # defined.  The kernel moves fewer bytes than that: ring phases stream a 16-bit column array (the low
            # halves of the u32 columns, built once), so the HBM traffic per launch (`traffic`, PMC) is ~2.04 GB for
            # the 2.68 GB algorithmic figure -- the rate of real HBM traffic is `traffic_rate` (GB/s).
            "layout": "f32 values + 16-bit ring-slot columns for LDS-ring phases (u32 columns kept for the rest)",
            "traffic_rate": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
            "traffic_frac_of_peak": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(rows, x.cpu().numpy(), y.cpu().numpy())
        else:
            result["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
