"""SURVEY 8(f) row 2 on the GPU: the timing half of the reference's impl/compare.py (:171-219) as tools/compare.py
reproduces it.  Every operation the table times on the MI355X is checked against the oracle in the same run (the
oracle's outputs for the scalars it is timed on must equal the first rows of the GPU batch), and the table has the
reference's six Curve4Q rows."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare  # noqa: E402


def test_timing_table_rows_are_checked_against_the_oracle():
    lines = []
    rows = compare.timing_table(n=1 << 12, cpu_ops=6, out=lines.append)
    assert [r[0] for r in rows] == ["MUL_windowed(m,P)", "MUL_windowed(m,G,table)", "MUL_endo(m,P)", "MUL_endo(m,G,table)",
                                    "DH_windowed(m,G)", "DH_endo(m,G)"]             # compare.py:175-204, Curve4Q half
    assert all(checked == 6 for _, _, _, checked in rows)
    assert all(cpu > 0 and gpu > 0 and cpu / gpu > 100 for _, cpu, gpu, _ in rows)    # a batch of 2^12 already beats one core by far
    assert len(lines) == 1 + len(rows) and lines[0].startswith("operation")
    endo = dict((r[0], r[1]) for r in rows)
    assert endo["MUL_endo(m,P)"] < endo["MUL_windowed(m,P)"]                          # the reference's own ordering
