import glob
import os

import numpy as np

from nextpolish2_amd import Opts
from nextpolish2_amd._types import Pileup, Yak

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    pu = Pileup(z["ref"], z["reads"], z["nibbles"], name=name)
    yaks = [Yak(int(k), z[f"yak{i}_words"], z[f"yak{i}_off"]) for i, k in enumerate(z["ks"])]
    opts = Opts(model="ref" if bool(z["model_ref"][0]) else "len", use_all_reads=bool(z["use_all_reads"][0]))
    return pu, yaks, opts, z["out_bases"], z["out_pos"], [str(x) for x in z["stage_digests"]]
