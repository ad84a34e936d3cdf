"""The hand-derived known answers of tests/test_oracle_pinning.py (Cartesian recheck of chained regions, the seed rules
of fill_seed_lqseqs), asked of the HIP path through the C ABI: the same pileups, the same expectations written out in
those tests — not a comparison with the oracle."""
import pytest

import test_oracle_pinning as tp
from nextpolish2_amd import Polisher

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [tp.test_three_chained_regions_cartesian_order_and_last_writer_wins,
                                  tp.test_lone_winner_is_promoted_only_when_the_other_reads_all_differ,
                                  tp.test_long_indel_seed_is_refused_beyond_max_indel_len])
def test_hand_derived_final_pass_cases_on_the_hip_path(monkeypatch, case):
    monkeypatch.setattr(tp.orc, "Oracle", lambda yaks: Polisher(yaks, device=0))
    case()
