"""The device-memory caching layer (csrc/pool.hip): freed blocks of 1 MiB and more stay with the library and are handed out
again; smh_pool_trim returns them to the runtime; results do not depend on it."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import sparsemat_amd as sm
from sparsemat_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats():
    kept, live = C.c_size_t(), C.c_size_t()
    sm._lib.check(sm.lib().smh_pool_stats(C.byref(kept), C.byref(live)))
    return kept.value, live.value


def test_freed_blocks_are_kept_and_reused_and_trimmed(gpu):
    sm._lib.check(sm.lib().smh_pool_trim())
    kept0, live0 = stats()
    assert kept0 == 0
    a = synth.DeviceBuffer(5 * 2 ** 20 + 123)      # rounded up to 6 MiB
    assert stats() == (0, live0 + 6 * 2 ** 20)
    p = a.ptr
    del a
    assert stats() == (6 * 2 ** 20, live0)
    b = synth.DeviceBuffer(5 * 2 ** 20)            # fits the kept block (at most a quarter larger than asked)
    assert b.ptr == p and stats() == (0, live0 + 6 * 2 ** 20)
    c = synth.DeviceBuffer(2 * 2 ** 20)
    del b
    d = synth.DeviceBuffer(2 * 2 ** 20)            # 6 MiB is more than a quarter too large for 2 MiB: a fresh block
    assert d.ptr != p and stats()[0] == 6 * 2 ** 20
    small = synth.DeviceBuffer(4096)               # below 1 MiB: the runtime's own allocator, not tracked
    before = stats()
    del small
    assert stats() == before
    del c, d
    assert stats()[0] == 10 * 2 ** 20
    sm._lib.check(sm.lib().smh_pool_trim())
    assert stats() == (0, live0)


SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import sparsemat_amd as sm
from sparsemat_amd import synth
a = synth.crs_fixed(synth.SEED_MATRIX, synth.PATTERN_BANDED, 300_000, 32, np.float32)
outs = []
for _ in range(3):
    t = a.transpose()
    off, col, val = t.raw_parts()
    outs.append((off.tobytes(), col.tobytes(), val.tobytes()))
    del t
assert outs[0] == outs[1] == outs[2]
import hashlib
print(hashlib.sha256(b"".join(outs[0])).hexdigest())
"""


def test_results_do_not_depend_on_the_layer(gpu):
    """The same transposition three times in a process (scratch re-used from the second call on) and in a process with the
    layer switched off: identical bytes."""
    digests = []
    for cap in ("17179869184", "0"):
        env = dict(os.environ, SMH_POOL_MAX_BYTES=cap)
        r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1] and len(digests[0]) == 64
