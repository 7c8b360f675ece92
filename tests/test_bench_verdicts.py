"""bench.py's exit code follows its own verdicts (no GPU needed: the function that reads the line)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_clean_line_has_no_failed_checks():
    line = {"verified": True, "grid_16384": {"verified": True, "recompute": {"verified": True}, "exchange": {"verified": None}},
            "float_modes": {"strict": {"end_to_end_vs_oracle": {"bit_equal": True}, "stages_within_1e-5": True},
                            "fast": {"end_to_end_vs_oracle": {"bit_equal": False}, "stages_within_1e-5": True},
                            "relaxed": {"end_to_end_vs_oracle": {"bit_equal": False}, "stages_within_1e-5": False}}}
    assert bench.failed_checks(line) == []          # relaxed leaving the band is documented, reported, not a failure


def test_every_kind_of_wrong_plane_is_named():
    assert bench.failed_checks({"verified": False}) == ["verified"]
    assert bench.failed_checks({"verified": True, "grid_16384": {"recompute": {"verified": False}}}) == \
        ["grid_16384/recompute/verified"]
    assert bench.failed_checks({"float_modes": {"fast": {"stages_within_1e-5": False}}}) == \
        ["float_modes/fast/stages_within_1e-5"]
    assert bench.failed_checks({"float_modes": {"strict": {"end_to_end_vs_oracle": {"bit_equal": False}}}}) == \
        ["float_modes/strict/end_to_end_vs_oracle/bit_equal"]
