"""Static audit of the constant-time kernels' gfx950 machine code (tools/ct_isa_audit.py; no GPU: hipcc cross-compiles).

draft-ladd-cfrg-4q.md:753-758 asks that no memory address and no branch depend on secret data.  In every ladder-step loop of
the kernels built with CT = true, each memory instruction's address must come from loop-invariant registers by address
arithmetic only, and each branch condition from scalars only.  The same audit run over the default-mode kernels -- where a
digit of the scalar IS a table address, as in the reference (curve4q.py:232, :440) -- must object: that is the control that the
audit can see a digit-dependent address at all.  (The dynamic counterpart is profiles/r03_ct_invariance.txt.)"""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import ct_isa_audit  # noqa: E402


def test_constant_time_kernels_have_no_digit_dependent_address_or_branch(tmp_path):
    rows, problems = ct_isa_audit.audit_units(ct_isa_audit.UNITS, str(tmp_path))
    assert not problems, problems[:5]
    kinds = {name for _, name, loops, _ in rows if loops}
    assert len(kinds) >= 34                                     # 4 fused + 18 pair- and quad-lane + 6 LDS ladders + 2 combs + the small-batch comb (two / four lanes) + the mixed-batch queue and tail kernels
    assert sum(loops for _, _, loops, _ in rows) >= 36
    assert sum(1 for _, name, loops, _ in rows if "comb_quad_kernel" in name and loops == 1) == 2
    # two and four lanes per element x (variable and fixed base x MUL / DH x endo / windowed + the mixed MUL_endo kernel): one ladder loop each
    assert sum(1 for _, name, loops, _ in rows if "pair_kernel" in name and "ELi2ELb" in name and loops == 1) == 9
    assert sum(1 for _, name, loops, _ in rows if "pair_kernel" in name and "ELi4ELb" in name and loops == 1) == 9
    assert any("mixed_ct_tail_kernel" in name and loops == 2 for _, name, loops, _ in rows)     # shared table in LDS / per-lane table in memory
    assert any("mixed_queue_kernel" in name and loops == 2 for _, name, loops, _ in rows)      # both kinds of work item


def test_the_audit_objects_to_the_default_kernels(tmp_path):
    rows, problems = ct_isa_audit.audit_units(["fourq_chain.hip"], str(tmp_path))
    flagged = {p.split(":")[0] for p in problems}
    ladders = {name for _, name, loops, _ in rows if loops and "ladder_kernel" in name}
    assert ladders and ladders <= flagged                      # every default ladder gathers by digit, and the audit says so
