"""Hand-traced SwissTable order vectors that BOTH hashbrown order models must reproduce: the
oracle's (oracle/hashbrown_emul.hpp) and the product's (csrc/np2_phase_host.hpp, behind np2_phase_vote).

Assumed crate versions (no Cargo.lock in the reference): fxhash 0.2.1, hashbrown 0.12.x as shipped in std of Rust
1.64-1.68.  The vectors below were worked out on paper from the published algorithm, not by running either model:

  FxHash of a u32 key k (one write_u32 from state 0): h = k * 0x517cc1b727220a95 mod 2^64
     k: 1 -> ...0a95   2 -> ...152a   3 -> ...1fbf   4 -> ...2a54   5 -> ...34e9   6 -> ...3f7e   7 -> ...4a13   8 -> ...54a8
  bucket = h & (buckets - 1); buckets 4 -> capacity 3, 8 -> 7, 16 -> 14; iteration = ascending bucket index.

  (1) insert 1, 2, 3 into an empty map: first insert reserves 1 -> 4 buckets; h & 3 = 1, 2, 3 -> order [1, 2, 3].
  (2) ... then insert 4: its slot (h & 3 = 0) is EMPTY and no growth is left -> resize to capacity_to_buckets(4) = 8;
      old items re-inserted in old index order (1, 2, 3): h & 7 = 5, 2, 7; then 4 -> 4.  Order by index: [2, 4, 1, 3].
  (3) insert 1..8: as (2) up to 7 items in 8 buckets (1->5, 2->2, 3->7, 4->4, 5->1, 6->6, 7->3); 8 finds EMPTY slot 0 with
      no growth left -> resize to capacity_to_buckets(8) = next_pow2(8 * 8 / 7) = 16; re-insert in old index order
      (5, 2, 7, 4, 1, 6, 3): h & 15 = 9, 10, 3, 4, 5, 14, 15; then 8 -> 8.  Order by index: [7, 4, 1, 8, 5, 2, 6, 3].
  (4) insert 1, 2, 3; remove 2; insert 4; insert 5: whether the freed slot is EMPTY (growth returned) or DELETED, the
      table is full when 5 (or 4) arrives and is resized to 8 buckets: 4 -> 4, 1 -> 5, 3 -> 7, 5 -> 1.  Order [5, 4, 1, 3].
  (5) entry().or_insert on a vacant key reserves first (rustc_entry): entry 1, 2, 3 gives the 4-bucket layout of (1);
      entry 4 must reserve(1) with no growth left -> 8 buckets BEFORE inserting -> same layout as (2): [2, 4, 1, 3].
"""
import pytest

from nextpolish2_amd.api import swiss_order as product_order
from oracle.np2_oracle import swiss_order as oracle_order

INS, REM, ENT = 0, 1, 2
VECTORS = [
    ([(INS, 1), (INS, 2), (INS, 3)], [1, 2, 3]),
    ([(INS, 1), (INS, 2), (INS, 3), (INS, 4)], [2, 4, 1, 3]),
    ([(INS, k) for k in range(1, 9)], [7, 4, 1, 8, 5, 2, 6, 3]),
    ([(INS, 1), (INS, 2), (INS, 3), (REM, 2), (INS, 4), (INS, 5)], [5, 4, 1, 3]),
    ([(ENT, 1), (ENT, 2), (ENT, 3), (ENT, 4)], [2, 4, 1, 3]),
    ([(INS, 3), (INS, 3), (REM, 7), (INS, 1)], [1, 3]),
]


@pytest.mark.parametrize("script,expected", VECTORS)
def test_both_order_models_reproduce_the_hand_traced_vectors(script, expected):
    assert oracle_order(script) == expected
    assert product_order(script) == expected


def test_order_models_agree_on_long_random_scripts():
    import numpy as np
    rng = np.random.default_rng(12)
    for _ in range(50):
        script = [(int(rng.choice([INS, INS, REM, ENT])), int(rng.integers(0, 300))) for _ in range(int(rng.integers(10, 600)))]
        assert oracle_order(script) == product_order(script)
