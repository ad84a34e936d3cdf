"""The hand-derived known answers of tests/test_oracle_pinning.py (Cartesian recheck of chained regions, the seed rules
of fill_seed_lqseqs), asked of the HIP path through the C ABI: the same pileups, the same expectations written out in
those tests — not a comparison with the oracle."""
import numpy as np
import pytest

import test_oracle_pinning as tp
import test_oracle_pinning2 as tp2
import test_oracle_pinning3 as tp3
from nextpolish2_amd import Polisher

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [tp.test_three_chained_regions_cartesian_order_and_last_writer_wins,
                                  tp.test_lone_winner_is_promoted_only_when_the_other_reads_all_differ,
                                  tp.test_long_indel_seed_is_refused_beyond_max_indel_len])
def test_hand_derived_final_pass_cases_on_the_hip_path(monkeypatch, case):
    monkeypatch.setattr(tp.orc, "Oracle", lambda yaks: Polisher(yaks, device=0))
    case()


@pytest.mark.parametrize("case", [tp2.test_lq_region_pad_extend_and_delayed_close,
                                  tp2.test_decode_limit_invalid_kmer_and_columns_before_the_start,
                                  tp2.test_homopolymer_length_difference_is_not_a_heterozygous_marker,
                                  tp2.test_three_disagreements_override_the_summed_pair_weight])
def test_second_set_of_hand_derived_cases_on_the_hip_path(monkeypatch, case):
    """LQ close / pad / extend, decode limit + start filter, is_valid_snp, dif <= -3: the expectations written out in
    tests/test_oracle_pinning2.py, asked of the HIP path (traces through np2_trace_get)."""
    monkeypatch.setattr(tp2.orc, "Oracle", lambda yaks: Polisher(yaks, device=0))
    case()


@pytest.mark.parametrize("case", [tp3.test_equal_paths_the_later_node_wins_unless_it_starts_with_a_gap,
                                  tp3.test_at_the_contig_end_the_last_node_of_maximal_score_wins,
                                  tp3.test_an_insertion_carried_by_half_of_the_rows_is_taken,
                                  tp3.test_reads_that_start_at_position_one_move_the_start_of_the_consensus_position_two_does_not,
                                  tp3.test_a_region_keeps_its_first_sixty_candidates_in_read_order,
                                  tp3.test_a_long_candidate_scores_the_rarest_of_its_kmers,
                                  tp3.test_two_alleles_of_equal_support_the_first_in_read_order_seeds,
                                  tp3.test_a_marker_needs_min_c_reads_and_singletons_take_no_part,
                                  tp3.test_the_recheck_prefers_a_valid_contig_string_over_the_majority,
                                  tp3.test_two_valid_strings_go_on_to_the_next_table,
                                  tp3.test_with_all_reads_the_model_decides_which_haplotype_goes])
def test_dp_tie_breaks_on_the_hip_path(monkeypatch, case):
    """The consensus DP's tie rules (main.rs:1664, 1676), expectations written out in tests/test_oracle_pinning3.py."""
    monkeypatch.setattr(tp3.orc, "Oracle", lambda yaks: Polisher(yaks, device=0))
    case()


def test_clip_filter_first_range_on_the_product_front_end():
    """filter_alignseqs_by_clip's first range (main.rs:531-574) through np2_contig_from_records: the hand-derived flags of
    tests/test_oracle_pinning2.py::test_clip_filter_first_range_is_the_whole_contig."""
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import records_to_arrays
    ref, recs = tp2.clip_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pol = Polisher([])
    c = np2io.contig_from_records(pol, ref.encode(), arr, cig, seq4, np2io.FrontOpts())
    got = np2io.export_contig(pol, c, np.frombuffer(ref.encode(), dtype=np.uint8))
    L = len(ref)
    assert got.reads["aln_t_s"].tolist() == [0, 10, 400, 1000, 3000, L - 1620]
    assert got.reads["aln_t_e"].tolist() == [L - 1, 1609, 1999, 2599, 4599, L - 21]
    assert (got.reads["flags"] & 1).tolist() == [0, 0, 0, 1, 1, 0]


def test_read_admission_boundaries_and_trim_on_the_product_front_end():
    """main.rs:1758-1771 and trim(8) through np2_contig_from_records: the expectations of
    tests/test_oracle_pinning3.py::test_read_admission_boundaries_and_trim."""
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import records_to_arrays
    ref, recs, want = tp3.admission_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pol = Polisher([])
    c = np2io.contig_from_records(pol, ref.encode(), arr, cig, seq4, np2io.FrontOpts())
    got = np2io.export_contig(pol, c, np.frombuffer(ref.encode(), dtype=np.uint8))
    assert list(zip(got.reads["aln_t_s"].tolist(), got.reads["aln_t_e"].tolist())) == want
    assert (got.reads["flags"] & 1).tolist() == [0, 0, 0, 1, 0, 0, 0]


def test_cigar_operations_and_clipping_on_the_product_front_end():
    """fill_with_cigar / is_clip (main.rs:386-440, 1796-1797) through np2_contig_from_records (k_columnarise): the strings
    and flags written out in tests/test_oracle_pinning3.py::test_cigar_operations_and_what_counts_as_clipped."""
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import records_to_arrays
    ref, recs, want = tp3.cigar_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pol = Polisher([])
    c = np2io.contig_from_records(pol, ref.encode(), arr, cig, seq4, np2io.FrontOpts())
    tp3.check_cigar_case(np2io.export_contig(pol, c, np.frombuffer(ref.encode(), dtype=np.uint8)), ref, want)
