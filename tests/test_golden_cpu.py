"""The oracle must reproduce the committed golden fixtures (tests/golden/*.npz) bit-exactly."""
import hashlib

import numpy as np
import pytest

from golden_util import golden_cases, load_golden
from oracle.np2_oracle import Oracle


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_golden(name):
    pu, yaks, opts, exp_b, exp_p, digests = load_golden(name)
    o = Oracle(yaks)
    o.set_trace(True)
    b, p = o.polish(pu, opts)
    assert np.array_equal(b, exp_b) and np.array_equal(p, exp_p)
    for d in digests:
        ps, st, h = d.split(":")
        t = o.trace(int(ps), st)
        got = "-" if t is None else hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]
        assert got == h, (ps, st)
