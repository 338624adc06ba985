"""Host-side logic of bench.py that needs no GPU: the spawner of one process per GPU (its environment, the pass-through of rank 0's
line, what happens when a rank dies or overruns), the reading of RCCL's own log, the sampled-rows parity gate (against the oracle)."""
import json
import os
import sys
import textwrap
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing it touches no GPU: main() is guarded)
import oracle  # noqa: E402


def _spawn(tmp_path, monkeypatch, capfdbinary, body, n=3, env=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    monkeypatch.setattr(sys, "argv", [str(script)])
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    t0 = time.time()
    rc = bench.spawn_ranks(n)
    out = capfdbinary.readouterr().out
    return rc, out, time.time() - t0


def test_spawned_ranks_get_the_launcher_environment_and_rank_0_speaks(tmp_path, monkeypatch, capfdbinary):
    rc, out, _ = _spawn(tmp_path, monkeypatch, capfdbinary, """
        import json, os
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SMH_BENCH_LAUNCH")
        print(json.dumps({k: os.environ.get(k) for k in keys}))   # (every rank prints; only rank 0's stdout is passed on)
        """)
    assert rc == 0
    lines = out.decode().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert (d["RANK"], d["LOCAL_RANK"], d["WORLD_SIZE"], d["LOCAL_WORLD_SIZE"], d["MASTER_ADDR"]) == ("0", "0", "3", "3", "127.0.0.1")
    assert int(d["MASTER_PORT"]) > 0 and "spawned ranks" in d["SMH_BENCH_LAUNCH"]


def test_a_dead_rank_takes_the_job_down_with_its_status(tmp_path, monkeypatch, capfdbinary):
    rc, out, took = _spawn(tmp_path, monkeypatch, capfdbinary, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)          # dies at once
        time.sleep(600)          # its peers would sit in a collective for ever
        """, env={"SMH_BENCH_SPAWN_TIMEOUT_S": "300"})
    assert rc == 7 and took < 60 and out == b""   # the peers were killed by PID after the grace, well before the limit


def test_ranks_that_overrun_are_killed(tmp_path, monkeypatch, capfdbinary):
    rc, out, took = _spawn(tmp_path, monkeypatch, capfdbinary, """
        import time
        time.sleep(600)
        """, n=2, env={"SMH_BENCH_SPAWN_TIMEOUT_S": "2"})
    assert rc == bench.WATCHDOG_EXIT and took < 30


class _FakeComm:
    def ranks_seen(self):
        return 8, 3


class _FakeSm:
    class Comm:
        @staticmethod
        def rccl_version():
            return 22203


def test_rccl_report_keeps_the_lines_that_answer_ring_or_direct(tmp_path, monkeypatch):
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "host:1:1 [0] NCCL INFO NCCL version 2.22.3+hip7.2",
        "host:1:1 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId c000 - Init START",
        "host:1:1 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7",
        "host:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1",
        "host:1:1 [0] NCCL INFO Connected all rings",
        "host:1:1 [0] NCCL INFO something irrelevant",
        "host:1:1 [0] NCCL INFO AllGather: 40000000 Bytes -> Algo 1 proto 2 time 1234.5",
        "host:1:1 [0] NCCL INFO AllGather: 40000000 Bytes -> Algo 1 proto 2 time 1234.5",
      ]) + "\n")
    monkeypatch.setenv("SMH_BENCH_RCCL_LOG_PATH", str(log))
    r = bench.rccl_report(_FakeComm(), _FakeSm)
    assert (r["version"], r["ranks_seen"], r["device"]) == (22203, 8, 3)
    assert r["log_algo_proto"] == ["host:1:1 [0] NCCL INFO AllGather: 40000000 Bytes -> Algo 1 proto 2 time 1234.5"]  # (once)
    assert any("Channel 00/16" in ln for ln in r["log_excerpt"]) and any("Connected all rings" in ln for ln in r["log_excerpt"])
    assert not any("irrelevant" in ln for ln in r["log_excerpt"])
    monkeypatch.delenv("SMH_BENCH_RCCL_LOG_PATH")
    assert "log_excerpt" not in bench.rccl_report(None, _FakeSm)


def test_sampled_rows_parity_gate_catches_a_wrong_row_and_passes_the_oracle_itself():
    g = 20
    off, col, val = oracle.laplace3d(g, g, g, np.float32)
    n = g ** 3
    x = oracle.gen_x(0x5EED0002, n, np.float32)
    y = oracle.spmv(off, col, val, x)
    blocks = [(rb,) + tuple(oracle.laplace3d_rows(g, g, g, rb, re, np.float32)) for rb, re in ((0, 500), (3000, 4200), (n - 300, n))]
    ok = bench.parity_rows(np, oracle, y, blocks, x, 1e-5, True)
    assert ok["ok"] and ok["bit_exact"] and ok["rows_checked"] == 500 + 1200 + 300 and ok["max_rel_err_vs_sum_abs"] == 0.0
    bad = y.copy()
    bad[3100] += np.float32(1e-2)
    r = bench.parity_rows(np, oracle, bad, blocks, x, 1e-5, True)
    assert not r["ok"] and not r["bit_exact"] and r["max_rel_err_vs_sum_abs"] > 1e-5
    # tolerance form: a last-bit difference passes the bound but is reported as not bit-exact when exactness is asked for
    ulp = y.copy()
    ulp[3100] = np.nextafter(ulp[3100], np.float32(np.inf))
    assert bench.parity_rows(np, oracle, ulp, blocks, x, 1e-5, False)["ok"] and not bench.parity_rows(np, oracle, ulp, blocks, x, 1e-5, True)["ok"]


def test_laplace3d_rows_is_the_full_generator_cut(tmp_path):
    for dtype in (np.float32, np.float64):
        off, col, val = oracle.laplace3d(11, 7, 5, dtype)
        for rb, re in ((0, 385), (13, 300), (384, 385), (100, 100)):
            o2, c2, v2 = oracle.laplace3d_rows(11, 7, 5, rb, re, dtype)
            lo, hi = int(off[rb]), int(off[re])
            assert np.array_equal(o2, off[rb:re + 1] - lo) and np.array_equal(c2, col[lo:hi]) and np.array_equal(v2, val[lo:hi])
