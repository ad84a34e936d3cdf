"""Stage-by-stage comparison of the HIP path against the oracle on tests/golden/ref_bundle (run on a GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from nextpolish2_amd import Opts, Polisher  # noqa: E402
from nextpolish2_amd import io as np2io  # noqa: E402
from nextpolish2_amd.bamio import records_to_arrays  # noqa: E402
from oracle import np2_oracle as orc  # noqa: E402
from stage_diff import compare  # noqa: E402
from test_ref_bundle import bundle  # noqa: E402

name, ref, recs, yaks = bundle()
arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
pu = orc.front_end(ref, arr, cig, asc, asc_off, np2io.FrontOpts())
o = orc.Oracle(yaks)
o.set_trace(True)
ob, op = o.polish(pu, Opts())
print("oracle", len(ob), o.stats())
g = Polisher(yaks)
g.set_trace(True)
gb, gp = g.polish(pu, Opts())
ok = compare(o, g, 2, print)
print("FINAL identical:", np.array_equal(ob, gb) and np.array_equal(op, gp))
