"""The committed evidence set (profiles/) must belong to the tree: profiles/check.py -- the device sources unchanged since the
set was measured, every rocprofv3 kernel average in agreement with the bench lines of the same set."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def test_evidence_set_matches_the_tree():
    import check
    problems = check.check(ROOT)
    assert not problems, "\n".join(problems)
