"""The C-ABI library loads and exports every symbol include/np2.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from nextpolish2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "np2.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(np2_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    assert set(syms) == set(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = api.lib()
    for s in header_symbols():
        assert hasattr(L, s), s


def test_struct_layouts_match_header():
    from nextpolish2_amd._types import np2_opts_t, np2_read_t, np2_yak_t
    assert C.sizeof(np2_read_t) == 24 and np2_read_t.nib_off.offset == 8 and np2_read_t.n_cols.offset == 16
    assert C.sizeof(np2_yak_t) == 32 and np2_yak_t.words.offset == 16
    assert C.sizeof(np2_opts_t) == 16 and np2_opts_t.max_indel_len.offset == 4 and np2_opts_t.model_ref.offset == 12


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under nextpolish2_amd/ may reference it
    pkg = os.path.join(ROOT, "nextpolish2_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".sh")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "np2_oracle" not in txt and "np2o_" not in txt and "oracle/" not in txt.replace("no oracle", ""), f


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from nextpolish2_amd import Polisher
    from nextpolish2_amd._types import Yak
    y = Yak(21, np.zeros(0, np.uint64), np.zeros(1025, np.uint64))
    with pytest.raises(api.Np2Error) as e:
        Polisher([y])
    assert e.value.code == -2  # NP2_E_DEVICE
