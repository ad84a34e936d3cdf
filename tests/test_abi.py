"""The C-ABI library loads and exports every symbol include/np2.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from nextpolish2_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="np2.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(np2_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    assert set(syms) == set(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = api.lib()
    for s in header_symbols() + header_symbols("np2_io.h"):
        assert hasattr(L, s), s


def test_io_header_declares_expected_entry_points():
    assert set(header_symbols("np2_io.h")) == set(api.IO_ABI_SYMBOLS)


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


def test_product_phasing_vote_agrees_with_the_oracle_on_random_signed_graphs():
    """Two independent restatements of louvain.rs + hashbrown iteration order (oracle/hashbrown_emul.hpp vs
    csrc/np2_phase_host.hpp, the latter with edge-driven aggregation) must pick the same losing reads, including on
    tie-heavy graphs where only the emulated bucket order decides."""
    import numpy as np
    from nextpolish2_amd.api import phase_vote
    from oracle import np2_oracle as orc
    rng = np.random.default_rng(12)
    for trial in range(120):
        n = int(rng.integers(3, 70))
        ids = rng.choice(np.arange(1, 400), size=n, replace=False)
        pairs = {}
        m = int(rng.integers(n, 4 * n))
        for _ in range(m):
            a, b = rng.choice(ids, size=2, replace=False)
            a, b = int(min(a, b)), int(max(a, b))
            # clustered signs: same "haplotype" (id parity) mostly positive, so conflicts are genuine; small integer
            # weights make ties between communities frequent
            same = (a % 2) == (b % 2)
            w = float(rng.integers(1, 3)) * (1.0 if (same or rng.random() < 0.1) else -1.0)
            if rng.random() < 0.05:
                w = -3.0
            pairs[(a, b)] = pairs.get((a, b), 0.0) + w
        plist = [(a, b, w) for (a, b), w in pairs.items()]
        # the oracle applies insert_data(a, b) then insert_data(b, a) per pair in this order
        edges, keys, seen = [], [], set()
        for a, b, w in plist:
            edges.append((a, b, w))
            edges.append((b, a, w))
            for k in (a, b):
                if k not in seen:
                    seen.add(k)
                    keys.append(k)
        ref = None
        if trial % 3 == 0:
            ref = {int(k): float(rng.choice([-1.0, 1.0, 2.0])) for k in rng.choice(ids, size=max(1, n // 3), replace=False)}
        try:
            exp = orc.phase_communities(edges, ref)
        except orc.RefPanic:
            with pytest.raises(api.Np2Error):
                phase_vote(keys, plist, ref)
            continue
        assert phase_vote(keys, plist, ref) == exp, trial
