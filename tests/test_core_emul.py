"""The product's device functions (smallvcm_amd/csrc/vcm_core.h), compiled for
the host and driven serially (tests/host_emul), must equal the oracle bit for
bit: framebuffer, random-number tape, merge records and workload counters."""
import numpy as np
import pytest

from emul_lib import Emul
from oracle_lib import Oracle
from smallvcm_amd.renderer import cornell_scene

CASES = [(sid, algo, 48, 1, 0, 10) for sid in range(4) for algo in range(5)] + [
    (1, 4, 96, 2, 0, 10), (3, 4, 64, 2, 2, 6), (1, 4, 32, 1, 0, 1), (1, 4, 32, 1, 0, 2), (0, 4, 32, 2, 5, 5),
    (2, 2, 40, 1, 0, 3)] + [(sid, algo, 40, 3, 0, 10) for sid in range(4) for algo in (5, 6)] + [(1, 5, 32, 1, 3, 4)]


@pytest.mark.parametrize("sid,algo,res,nit,mn,mx", CASES)
def test_device_functions_equal_oracle(sid, algo, res, nit, mn, mx):
    sc = cornell_scene(sid, res, res)
    o, e = Oracle(sc, algo, threads=4), Emul(sc, algo)
    for it in range(nit):
        o.run_iteration(it, mn, mx)
        e.run_iteration(it, mn, mx)
    assert np.array_equal(o.framebuffer(), e.framebuffer())
    for a, b in zip(o.counts(), e.counts()):
        assert np.array_equal(a, b)
    assert np.array_equal(o.records().view(np.uint32), e.records().view(np.uint32))
    so, se = o.stats(), e.stats()
    for k in se:
        assert so[k] == se[k], k
