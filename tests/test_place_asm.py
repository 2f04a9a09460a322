"""The build's code placement pass (tools/asmgen/place_asm.py) on hand-made gfx950 assembly: what it must do (every 8-byte instruction on an
8-byte boundary, by re-encoding or by one s_nop in front of a long run) and the one thing it must never do (put anything between an
s_getpc_b64 and the s_add_u32 / s_addc_u32 that follow it: their literals are computed for exactly that spacing, and a call through a
shifted address lands four bytes in front of its target).  Needs the ROCm assembler, no GPU."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))
import place_asm                                   # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(place_asm.LLVM, "clang")), reason="needs the ROCm assembler")

HEAD = "\t.text\n\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"\n"
CALLEE = "\t.p2align 8\n\t.type g,@function\ng:\n\ts_setpc_b64 s[30:31]\n.Lfunc_end0:\n\t.size g, .Lfunc_end0-g\n"


def func(name, body, k=1):
    return "\t.globl %s\n\t.p2align 8\n\t.type %s,@function\n%s:\n%s.Lfunc_end%d:\n\t.size %s, .Lfunc_end%d-%s\n" % (name, name, name, body, k, name, k, name)


def place(tmp_path, text):
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    stats = place_asm.place_file(str(src), str(dst))
    obj = tmp_path / "out.o"
    place_asm.assemble(str(dst), str(obj))
    return stats, dst.read_text(), place_asm.disassemble(str(obj))


MADS = "".join("\tv_mad_u64_u32 v[%d:%d], vcc, v0, v1, v[%d:%d]\n" % (2 * i + 2, 2 * i + 3, 2 * i + 2, 2 * i + 3) for i in range(6))


def test_a_call_sequence_is_left_alone(tmp_path):
    # s_getpc_b64 at offset 0: its s_add_u32 (8 bytes) starts at 4 mod 8 with a run of 8-byte instructions behind it -- the case in which an
    # earlier version of the pass put an s_nop between the two
    body = "\ts_getpc_b64 s[0:1]\n\ts_add_u32 s0, s0, g@rel32@lo+4\n\ts_addc_u32 s1, s1, g@rel32@hi+12\n" + MADS + "\ts_swappc_b64 s[30:31], s[0:1]\n\ts_endpgm\n"
    stats, text, funcs = place(tmp_path, HEAD + CALLEE + func("f", body))
    seq = [m for m, _ in funcs["f"]]
    k = seq.index("s_getpc_b64")
    assert seq[k + 1:k + 3] == ["s_add_u32", "s_addc_u32"]
    assert "s_nop" in seq[k + 3:]                  # the run of multiply-adds behind the sequence still gets its s_nop, after the s_addc_u32
    off = 0
    for mn, size in funcs["f"]:
        if mn.startswith("v_mad"):
            assert off % 8 == 0
        off += size


def test_a_shifted_call_sequence_is_refused(tmp_path):
    body = "\ts_getpc_b64 s[0:1]\n\ts_nop 0\n\ts_add_u32 s0, s0, g@rel32@lo+4\n\ts_addc_u32 s1, s1, g@rel32@hi+12\n\ts_swappc_b64 s[30:31], s[0:1]\n\ts_endpgm\n"
    src = tmp_path / "bad.s"
    src.write_text(HEAD + CALLEE + func("f", body))
    with pytest.raises(RuntimeError, match="s_getpc_b64 followed by"):
        place_asm.place_file(str(src), str(tmp_path / "bad_out.s"))


def test_misplaced_wide_instructions_are_fixed_by_reencoding_first(tmp_path):
    body = "\tv_add_u32_e32 v0, v1, v2\n" + MADS + "\tv_add_u32_e32 v3, v1, v2\n\tv_mad_u64_u32 v[20:21], vcc, v0, v1, v[20:21]\n\ts_endpgm\n"
    stats, text, funcs = place(tmp_path, HEAD + func("f", body, k=0))
    assert stats["misaligned_after"] == 0 and stats["promoted"] == 2 and stats["nops"] == 0
    assert [m for m, _ in funcs["f"]].count("v_add_u32_e64") == 2
    assert len(funcs["f"]) == 10                   # no instruction added


def test_the_shipped_library_keeps_every_call_sequence_intact(tmp_path):
    """the same check on the code objects of the library the tests load"""
    from fourq_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    import importlib.util
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(os.path.dirname(place_asm.__file__), "code_objects.py"))
    co = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(co)
    objs = co.code_objects(_lib.LIB_PATH, str(tmp_path))                 # compressed offload bundles since round 6: unpacked here
    assert len(objs) == 4                          # one code object per translation unit
    for o in objs:
        assert open(o, "rb").read(4) == b"\x7fELF"
        place_asm.check_pc_relative(o)


def test_the_shipped_build_was_placed():
    """a unit whose placement failed is built without the pass (correct, about a percent slower, fourq_amd/build.py): the build says so on
    stderr and in fourq_amd/code_placement.json -- which must not say it of the library the tests load"""
    import json
    from fourq_amd import build
    if not os.path.exists(build.PLACEMENT_PATH):
        pytest.skip("library not built here")
    report = json.load(open(build.PLACEMENT_PATH))
    assert sorted(report) == sorted(build.SOURCES)
    for unit, r in report.items():
        assert r["placed"], unit
        assert r["at_4_mod_8_before"] > 0.3 * r["wide_instructions"]                 # what hipcc leaves behind (39-45 %)
        assert r["at_4_mod_8_after"] < 0.08 * r["wide_instructions"], (unit, r)      # 1-5 %: short runs behind control flow


# ---- the register-ownership check of the build (fourq_amd/build.py check_register_ranges) -------------------------------
def _listing(tmp_path, body, next_free, accum=None, descriptor=True):
    text = HEAD + "\t.globl _Z1kv\n_Z1kv:\n" + body + "\ts_endpgm\n"
    if descriptor:
        text += "\t.amdhsa_kernel _Z1kv\n\t\t.amdhsa_next_free_vgpr %d\n" % next_free
        if accum is not None:
            text += "\t\t.amdhsa_accum_offset %d\n" % accum
        text += "\t.end_amdhsa_kernel\n"
    path = tmp_path / "k.s"
    path.write_text(text)
    return str(path)


def test_register_check_passes_a_kernel_that_owns_what_it_names(tmp_path):
    from fourq_amd import build
    build.check_register_ranges(_listing(tmp_path, "\tv_add_u32_e32 v255, v0, v1\n", 256, 256))
    build.check_register_ranges(_listing(tmp_path, "\tv_mad_u64_u32 v[174:175], vcc, v0, v1, v[2:3]\n", 216, 176))      # 176 arch VGPRs + 40 AGPRs


def test_register_check_refuses_a_register_past_the_allocation(tmp_path):
    from fourq_amd import build
    with pytest.raises(build.RegisterOwnershipError, match="names v255 but owns 128"):
        build.check_register_ranges(_listing(tmp_path, "\tv_add_u32_e32 v255, v0, v1\n", 128, 128))


def test_register_check_counts_arch_vgprs_not_the_unified_total(tmp_path):
    """On gfx950 next_free_vgpr is VGPRs + AGPRs: a kernel with accum_offset 176 and 80 AGPRs has next_free_vgpr 256, and a body naming
    v236 would alias its AGPRs while staying below that total."""
    from fourq_amd import build
    with pytest.raises(build.RegisterOwnershipError, match="names v237 but owns 176 arch VGPRs"):
        build.check_register_ranges(_listing(tmp_path, "\tv_mad_u64_u32 v[236:237], vcc, v0, v1, v[2:3]\n", 256, 176))


def test_register_check_refuses_a_listing_it_cannot_read(tmp_path):
    from fourq_amd import build
    with pytest.raises(build.RegisterOwnershipError, match="no .amdhsa_next_free_vgpr"):
        build.check_register_ranges(_listing(tmp_path, "\tv_add_u32_e32 v3, v0, v1\n", 0, descriptor=False))


def test_the_placement_fallback_does_not_swallow_a_register_finding(monkeypatch, tmp_path):
    """one() in build_library() catches RuntimeError from the placement detour and rebuilds without it; the register finding is a
    different exception type, raised by both paths, so it can never end as a 'code placement failed' log line and a shipped kernel."""
    from fourq_amd import build
    assert not issubclass(build.RegisterOwnershipError, RuntimeError)
    calls = []

    def fake_run(cmd, verbose=False):
        calls.append(cmd)
        if "-S" in cmd:
            out = cmd[cmd.index("-o") + 1]
            with open(out, "w") as fh:
                fh.write(open(_listing(tmp_path, "\tv_add_u32_e32 v255, v0, v1\n", 128, 128)).read())
        return ""
    monkeypatch.setattr(build, "_run", fake_run)
    for place in (True, False):
        with pytest.raises(build.RegisterOwnershipError):
            build.compile_unit("fourq_amd.hip", str(tmp_path / "x.o"), ["-O3"], place=place, placement={})
    assert not os.path.exists(str(tmp_path / "x.check.s"))


# ---- placed == unplaced, by construction ------------------------------------------------------------------------
BODY = ("\ts_getpc_b64 s[4:5]\n\ts_add_u32 s4, s4, g@rel32@lo+4\n\ts_addc_u32 s5, s5, g@rel32@hi+12\n"
        "\tv_add_u32_e32 v1, v2, v3\n\tv_add_co_u32_e32 v1, vcc, v2, v3\n" + MADS +          # 8-byte run at 4 mod 8 behind an _e32: re-encoded
        "\ts_cbranch_scc1 .LBB0_1\n" + MADS + "\tv_cndmask_b32_e32 v1, v2, v3, vcc\n"                 # ... behind a branch: one s_nop inserted
        "\t.p2align 3\n.LBB0_1:\n\tv_cmp_lt_u32_e32 vcc, v1, v2\n" + MADS +                          # the assembler pads here, differently in the two objects
        "\ts_cbranch_vccnz .LBB0_1\n\ts_nop 0\n\ts_endpgm\n")


def _objs(tmp_path, plain_text, placed_text):
    a, b = tmp_path / "a.s", tmp_path / "b.s"
    a.write_text(plain_text)
    b.write_text(placed_text)
    place_asm.assemble(str(a), str(tmp_path / "a.o"))
    place_asm.assemble(str(b), str(tmp_path / "b.o"))
    return str(tmp_path / "a.o"), str(tmp_path / "b.o"), plain_text, placed_text


def test_placed_object_is_the_plain_one_plus_padding(tmp_path):
    text = HEAD + func("f", BODY) + CALLEE
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    stats = place_asm.place_file(str(src), str(dst))
    assert stats["promoted"] >= 1 and stats["nops"] >= 1 and stats["instructions_compared"] >= 25        # the pass did both things it can do, and it was checked
    a, b, ta, tb = _objs(tmp_path, text, dst.read_text())
    assert place_asm.check_equivalent(a, b, ta, tb) == stats["instructions_compared"]
    # forward and backward branches over re-encoded / inserted code still name the same instruction: offsets differ, targets do not
    plain, placed = place_asm.listing(a)["f"], place_asm.listing(b)["f"]
    assert [i[2] for i in plain if i[1].startswith("s_cbranch")] != [i[2] for i in placed if i[1].startswith("s_cbranch")]


@pytest.mark.parametrize("damage", ["swap", "operand", "drop", "branch", "real-nop"])
def test_a_corrupted_rewrite_is_caught(tmp_path, damage):
    """What the check exists for: a rewrite that is NOT the input plus padding -- two instructions swapped, an operand changed, an instruction
    or a compiler-emitted hazard nop dropped, a branch sent elsewhere -- must fail it."""
    text = HEAD + func("f", BODY) + CALLEE
    src, dst = tmp_path / "in.s", tmp_path / "out.s"
    src.write_text(text)
    place_asm.place_file(str(src), str(dst))
    lines = dst.read_text().splitlines(keepends=True)
    first_mad = next(i for i, ln in enumerate(lines) if "v_mad_u64_u32 v[2:3]" in ln)
    if damage == "swap":
        lines[first_mad], lines[first_mad + 1] = lines[first_mad + 1], lines[first_mad]
    elif damage == "operand":
        lines[first_mad] = lines[first_mad].replace("v0, v1", "v1, v0")
    elif damage == "drop":
        del lines[first_mad]
    elif damage == "branch":
        k = next(i for i, ln in enumerate(lines) if "s_cbranch_scc1" in ln)
        lines[k] = "\ts_cbranch_scc1 .Lelsewhere\n"
        lines.insert(first_mad + 2, ".Lelsewhere:\n")
    else:
        k = max(i for i, ln in enumerate(lines) if ln.strip() == "s_nop 0" and "s_endpgm" in lines[i + 1])     # the compiler's own nop in front of s_endpgm
        del lines[k]
    a, b, ta, tb = _objs(tmp_path, text, "".join(lines))
    with pytest.raises(RuntimeError, match="diverge|differs"):
        place_asm.check_equivalent(a, b, ta, tb)


def test_the_shipped_build_compared_every_unit_with_its_unplaced_object():
    """build_library() runs the comparison on every translation unit (place_file); the numbers it saw are in code_placement.json."""
    import json
    from fourq_amd import build
    rec = json.load(open(build.PLACEMENT_PATH))
    assert set(rec) == set(build.SOURCES)
    for unit, r in rec.items():
        assert r["placed"] is True and r["instructions_compared_with_the_unplaced_object"] > 40000, unit
