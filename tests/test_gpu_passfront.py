"""The fused pass front (csrc/np2_passfront.hip: records of a contig tile -> the tile's piece of the raw consensus in one
kernel) against the oracle and against the unfused kernels it replaces, including the tiles it hands to the big variant and
the passes it hands back to the unfused kernels.  The limits of the LDS variants are lowered through test hooks
(NP2_PF_CAP, NP2_PF_CAP_BIG, NP2_PF_HALO, NP2_PF_COV_MAX: read when a context is created) so that small inputs take those branches."""
import numpy as np
import pytest

from nextpolish2_amd import BatchPolisher, Opts, Polisher
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from test_gpu_parity import check_all_stages

pytestmark = pytest.mark.gpu


def _polish(yaks, pu, opts=None):
    g = Polisher(yaks)
    b, p = g.polish(pu, opts or Opts())
    return b, p, g.timings()


def test_unfused_front_still_matches_the_oracle_at_every_stage(monkeypatch):
    """NP2_FRONT_UNFUSED=1: k_tile_count / k_tile_write / k_dp_bt_* / k_cns_* — the general path the fused front falls back
    to — stays under the stage-by-stage comparison."""
    monkeypatch.setenv("NP2_FRONT_UNFUSED", "1")
    s = Synth(50000, depth=30, seed=32, diploid=True, read_len_mean=8000.0, read_len_sd=1500.0)
    check_all_stages(s.pileup, [s.yak(21), s.yak(31)], Opts())
    s = Synth(30000, depth=8, seed=41, diploid=True, read_err_rate=0.02, read_len_mean=4000.0, read_len_sd=800.0)
    check_all_stages(s.pileup, [s.yak(21)], Opts())


@pytest.mark.parametrize("seed,depth,err,diploid", [(901, 30, 0.002, True), (902, 12, 0.02, True), (903, 60, 0.01, False),
                                                     (904, 3, 0.03, True)])
def test_fused_equals_unfused_equals_oracle(monkeypatch, seed, depth, err, diploid):
    s = Synth(70000, depth=depth, seed=seed, diploid=diploid, read_err_rate=err, read_len_mean=6000.0, read_len_sd=1500.0)
    yaks = [s.yak(21), s.yak(31)]
    ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
    fb, fp, ft = _polish(yaks, s.pileup)
    assert np.array_equal(fb, ob) and np.array_equal(fp, op)
    # (60x at 1 % errors fills tiles beyond the middle variant: one pass is handed back before the big one joins the launches)
    assert ft.get("front_redo", 0) <= (1 if depth >= 60 else 0)
    monkeypatch.setenv("NP2_FRONT_UNFUSED", "1")
    ub, up, _ = _polish(yaks, s.pileup)
    assert np.array_equal(ub, ob) and np.array_equal(up, op)


@pytest.mark.parametrize("cap,halo", [(48, 16), (200, 0), (960, 16)])
def test_tiles_that_do_not_fit_go_to_the_big_variant(monkeypatch, cap, halo):
    """Lowered limits: a tile with more than `cap` records, or whose last run is still open `halo` positions into the next
    tile, is listed and redone by k_pf_tile_big (3584 records, a whole tile of halo) — no pass is handed back."""
    monkeypatch.setenv("NP2_PF_CAP", str(cap))
    monkeypatch.setenv("NP2_PF_HALO", str(halo))
    monkeypatch.setenv("NP2_PF_BIG", "1")  # (the big variant in the launch sequence from the start: see the next test)
    for seed, depth, err in ((911, 30, 0.004), (912, 15, 0.008)):
        s = Synth(60000, depth=depth, seed=seed, diploid=True, read_err_rate=err, read_len_mean=5000.0, read_len_sd=1200.0)
        yaks = [s.yak(21)]
        ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
        fb, fp, ft = _polish(yaks, s.pileup)
        assert np.array_equal(fb, ob) and np.array_equal(fp, op)
        assert "front_redo" not in ft
    # ... and with every stage traced
    s = Synth(40000, depth=25, seed=913, diploid=True, read_err_rate=0.01, read_len_mean=5000.0, read_len_sd=1200.0)
    check_all_stages(s.pileup, [s.yak(21), s.yak(31)], Opts())


def test_the_big_variant_joins_the_launch_sequence_once_a_contig_has_needed_it(monkeypatch):
    """k_pf_tile_big (a whole tile of halo) is two launches per step that an ordinary pileup never needs: a context launches
    it from the first pass on in which the middle variant could not hold a tile — that pass is redone by the unfused
    kernels, the later ones are fused again."""
    monkeypatch.setenv("NP2_PF_HALO", "0")  # (no halo: a run that reaches the end of its tile is open — for the middle variant too)
    s = Synth(60000, depth=20, seed=914, diploid=True, read_err_rate=0.006, read_len_mean=5000.0, read_len_sd=1200.0)
    yaks = [s.yak(21)]
    ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
    g = Polisher(yaks)
    b, p = g.polish(s.pileup, Opts())
    assert np.array_equal(b, ob) and np.array_equal(p, op)
    assert g.timings().get("front_redo", 0) == 1  # the first pass that needed it; the second pass of the same call was fused
    b, p = g.polish(s.pileup, Opts())
    assert np.array_equal(b, ob) and np.array_equal(p, op)
    assert "front_redo" not in g.timings()


@pytest.mark.parametrize("hook", [{"NP2_PF_CAP": "48", "NP2_PF_CAP_BIG": "48"}, {"NP2_PF_COV_MAX": "20"}])
def test_a_pass_the_fused_front_cannot_hold_is_redone_unfused(monkeypatch, hook):
    for k, v in hook.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("NP2_PF_BIG", "1")
    s = Synth(60000, depth=30, seed=921, diploid=True, read_len_mean=7000.0, read_len_sd=1500.0)
    yaks = [s.yak(21), s.yak(31)]
    ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
    fb, fp, ft = _polish(yaks, s.pileup)
    assert np.array_equal(fb, ob) and np.array_equal(fp, op)
    assert ft.get("front_redo", 0) >= 1
    # the batch driver takes the same detour (a diverging contig costs batching, never correctness)
    g = Polisher(yaks)
    c = g.upload(s.pileup)
    bp = BatchPolisher(g, 2)
    for bb, pp in bp.polish([c, c], Opts(), want_pos=True):
        assert np.array_equal(bb, ob) and np.array_equal(pp, op)
    bp.close()


def test_contig_ends_and_short_contigs():
    """Runs that reach the contig end (no clean position closes them: the best end node is picked on chip, its sign checked
    by the host against the total of the gains), contigs shorter than a tile, contigs that end a few positions into a
    tile (the last tile lives in its neighbour's halo)."""
    for L in (300, 1000, 1023, 1024, 1025, 1030, 1090, 2047, 2050, 3100):
        for seed in (0, 1, 2):
            s = Synth(L, depth=12, seed=9300 + 7 * L + seed, diploid=bool(seed & 1), read_err_rate=0.03,
                      read_len_mean=min(900.0, L * 0.8), read_len_sd=100.0, read_len_min=max(50, L // 4))
            yaks = [s.yak(21)]
            try:
                ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
            except orc.RefPanic:  # (the same refusal on both sides)
                with pytest.raises(Np2Error):
                    _polish(yaks, s.pileup)
                continue
            fb, fp, _ = _polish(yaks, s.pileup)
            assert np.array_equal(fb, ob) and np.array_equal(fp, op), (L, seed)
