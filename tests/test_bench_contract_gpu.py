"""bench.py as the driver runs it, at small sizes: ONE JSON line on stdout with the contract's keys, `roofline` and `cpu_baseline`
at N = 1; the N > 1 code path (one process, blocks sharing the device) with its exchange self-check -- which must fail when
the exchange is skipped; and the one-process-per-GPU launch with a lone rank under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def run_bench(args, env=None, launcher=None, expect_rc=0):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=600)
    assert r.returncode == expect_rc, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]  # exactly one line on stdout
    return json.loads(lines[0])


def test_single_gpu_line(gpu):
    d = run_bench(["--rows", "1000000", "--steps", "7", "--warmup", "2", "--no-traffic", "--configs", "on", "--c3-rows", "600000", "--c4-grid", "96",
                   "--cg-iters", "20"])
    for k in CONTRACT:
        assert k in d, k
    assert (d["n_gpus"], d["steps"], d["warmup"], d["dtype"], d["scaling"], d["vs_baseline"]) == (1, 7, 2, "f32", "weak", None)
    assert d["metric"] == "csr_spmv_effective_hbm_GBps" and d["unit"] == "GB/s" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # the headline sits on SURVEY 8d's literal pattern; the stratified subset rides along
    assert d["config"]["pattern"] == "window" and "without replacement from [i-4096, i+4096]" in d["config"]["workload"]
    assert d["other_pattern"]["pattern"] == "stratified" and d["other_pattern"]["launches"] == 20 and d["other_pattern"]["kernel_ms"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches"] == 7 and r["kernel_ms_min"] <= r["kernel_ms_median"] <= r["kernel_ms_max"] and r["traffic"] is None
    assert r["cold"]["launches"] == 20
    assert r["cold_after_reads"]["launches"] == 20 and r["cold_after_reads"]["kernel_ms_median"] > 0
    assert abs(d["value"] - d["algorithmic_bytes_per_gpu_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["parity_ok"] is True and c["value"] > 0
    assert d["cpu_baseline_all_cores"]["cores"] >= 1
    # what the inspectors cost, against north_star's two plain row-major kernels
    i = d["inspector"]
    assert i["prepare_ms"] > 0 and i["derived_bytes"] >= 2 * 32 * 1000000 and set(i["plain_row_major_ms"]) == {"vector_k1_no_ring", "merge_k2"}
    # the other single-GPU BASELINE configs ride along, each with its roofline, parity gate and one-core baseline
    c = d["configs"]
    assert set(c) == {"C3", "C4_spmv", "C4_cg"}, c
    for name in ("C3", "C4_spmv", "C4_cg"):
        e = c[name]
        assert e["roofline"]["bound"] == "hbm" and e["roofline"]["peak"] == 8000.0 and 0 < e["roofline"]["frac"] < 1.2, (name, e["roofline"])
        assert e["parity"]["ok"] is True, (name, e["parity"])
        assert e["cpu_baseline"]["cores"] == 1 and e["cpu_baseline"]["kind"] == "port" and e["cpu_baseline"]["value"] > 0 and "sample" in e["cpu_baseline"]
    assert c["C3"]["dtype"] == "f64" and c["C3"]["parity"]["tol"] == 1e-12 and c["C3"]["launches"] == 20 and c["C3"]["inspector"]["prepare_ms"] > 0
    assert c["C4_spmv"]["parity"]["bit_exact"] is True and c["C4_spmv"]["algorithmic_bytes"] > 0
    assert c["C4_cg"]["iterations_per_solve"] == 20 and c["C4_cg"]["algorithmic_bytes"] == c["C4_spmv"]["algorithmic_bytes"] + 9 * 96 ** 3 * 4


def test_n_gpus_rehearsal_checks_its_exchange(gpu):
    env = {"SMH_BENCH_SHARE_DEVICES": "1"}
    d = run_bench(["--gpus", "4", "--one-process", "--rows", "500000", "--steps", "5", "--warmup", "2"], env)
    assert d["n_gpus"] == 4 and d["config"]["exchange"] == "window" and d["config"]["exchange_backend"] == "peer"
    assert d["roofline"]["traffic"] is None and d.get("cpu_baseline") is None
    x = d["exchange_check"]
    assert x["ok"] is True and x["rows_per_rank"] == 2 * 64 + 2 * 128 and x["max_abs_err"] <= x["tol"]
    # the window exchange ran beside the interior rows' product; the same steps with that switched off are reported next to it
    assert d["ms_per_step_no_overlap"] > 0 and "exchange_hidden_ms" in d and d["step_ms_block0_events"] > 0
    assert 0 == d["interior_rows_block0"][0] < d["interior_rows_block0"][1] < 500000  # (block 0: only its last rows are referenced)
    a = d["allgather_leg"]  # configs[4]'s literal exchange, timed beside the headline
    assert a["exchange"] == "allgather" and a["exchange_check"]["ok"] is True and a["ms_per_step"] > 0
    assert a["received_bytes_per_gpu_step"] == 3 * 500000 * 4
    # whole-job value: all blocks' bytes over the step time
    assert abs(d["value"] - 4 * d["algorithmic_bytes_per_gpu_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    d = run_bench(["--gpus", "3", "--one-process", "--rows", "400000", "--steps", "3", "--warmup", "1", "--exchange", "allgather"], env)
    assert d["config"]["exchange"] == "allgather" and d["exchange_check"]["ok"] is True and "allgather_leg" not in d
    bad = run_bench(["--gpus", "4", "--one-process", "--rows", "500000", "--steps", "3", "--warmup", "1"], dict(env, SMH_BENCH_SKIP_EXCHANGE="1"))
    assert bad["exchange_check"]["ok"] is False and bad["exchange_check"]["max_abs_err"] > 1e-2


def test_bare_gpus_n_spawns_one_process_per_gpu(gpu):
    """`bench.py --gpus 2` bare: the default is now fresh child processes, one per GPU, started before the parent touches the
    GPU (ranks meet through smh_comm_*, as under torch.distributed.run) -- here sharing the one device through the test suite's
    stand-in for RCCL.  The line says how it was launched and what the communicator reports; with the probe switched on, rank 0
    runs the job once more in small as fresh processes (the other ranks wait on the HOST meanwhile) and attaches what it said.
    (Two ranks: with this pytest process the probe brings the box to five processes on the GPU; the limit is six.)"""
    from util import build_mock_rccl
    mock = build_mock_rccl()
    d = run_bench(["--gpus", "2", "--rows", "300000", "--steps", "3", "--warmup", "1"],
                  {"LD_PRELOAD": mock, "SMH_BENCH_SHARE_DEVICES": "1", "SMH_BENCH_RCCL_PROBE": "1"})
    assert d["n_gpus"] == 2 and d["config"]["exchange_backend"] == "rccl" and "spawned ranks" in d["config"]["launch"]
    assert d["ranks_seen"] == 2 and d["rccl"]["version"] == 0  # (0: the stand-in's version)
    assert d["exchange_check"]["ok"] is True
    pr = d["rccl"]["probe"]
    assert pr.get("n_gpus") == 2 and pr.get("exchange") == "allgather" and pr.get("ranks_seen") == 2, pr


def test_one_rank_under_the_launcher(gpu):
    """python -m torch.distributed.run --nproc-per-node 1: the rank-per-GPU path (communicator from a file rendezvous,
    smh_par_create_rank, RCCL calls with a lone rank)."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29631"]
    d = run_bench(["--gpus", "1", "--rows", "500000", "--steps", "4", "--warmup", "1", "--no-traffic", "--no-cpu-baseline"],
                  {"SMH_BENCH_FORCE_PAR": "1", "SMH_PAR_EXCHANGE_SINGLE": "1"}, launcher)
    assert d["n_gpus"] == 1 and d["config"]["launch"].startswith("torch.distributed.run") and d["config"]["exchange_backend"] == "rccl"
    assert d["ranks_seen"] == 1 and d["rccl"]["version"] > 20000  # real RCCL, a lone rank


def test_a_hanging_optional_leg_does_not_cost_the_headline(gpu):
    """The extras of an N > 1 line (exchange self-check, all-gather leg) run after the headline under a watchdog: with a leg that
    never returns, the line still comes out, marked with the leg that hung -- and the process exits with status 3, not 0: a hung
    collective on a process that owns the GPU must not look like a successful run."""
    env = {"SMH_BENCH_SHARE_DEVICES": "1", "SMH_BENCH_HANG_IN_LEGS": "1", "SMH_BENCH_WATCHDOG_S": "3"}
    d = run_bench(["--gpus", "2", "--one-process", "--rows", "300000", "--steps", "3", "--warmup", "1"], env, expect_rc=3)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "gave up after 3 s" in d["optional_legs"] and d["hung_leg"] == "exchange_check"
    assert "exchange_check" not in d and "allgather_leg" not in d
