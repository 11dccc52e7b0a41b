"""The long-K kernel unrolls its multiply step against tables (bitmm_fp4_stream.hip.h: st_exp_by / st_shift_by) that tools/stream_schedule.py
generates: the header must hold what the generator prints, and the generator's order must be executable - every operand expanded before the
MFMA that reads it (a whole gap ahead where the order allows one expansion a gap), bit 3's expansion of a fragment behind its four in-place
shifts, the shifts behind its bit-2 expansion."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("stream_schedule", os.path.join(ROOT, "tools", "stream_schedule.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _header_tables(name):
    src = open(os.path.join(ROOT, "qgtc_ppopp22_amd", "csrc", "bitmm_fp4_stream.hip.h")).read()
    body = src[src.index(f"constexpr int {name}(int n)"):]
    body = body[:body.index("\n}\n")]
    rows = re.findall(r"constexpr int t\[\] = \{([0-9, ]+)\}", body)
    conds = re.findall(r"RF == (\d) && CF == (\d)", body)
    keys = [(int(a), int(b)) for a, b in conds] + [(4, 2)]          # (the last branch is the else: 4 x 2)
    assert len(rows) == len(keys) == 4
    return {k: [int(v) for v in r.split(",")] for k, r in zip(keys, rows)}


@pytest.mark.parametrize("RF,CF", [(2, 1), (2, 2), (4, 1), (4, 2)])
def test_header_tables_are_the_generators(RF, CF):
    gen = _generator()
    mf, ops, EP, EN, exp_by, shift_by, todo, gap_of = gen.schedule(RF, CF)
    assert _header_tables("st_exp_by")[(RF, CF)] == exp_by
    assert _header_tables("st_shift_by")[(RF, CF)] == shift_by
    MN = 4 * RF * CF
    assert len(exp_by) == len(shift_by) == MN and EP == 3 and EN == 4 * (RF + CF)
    assert exp_by == sorted(exp_by) and shift_by == sorted(shift_by)
    assert exp_by[-1] == EN + EP and shift_by[-1] == 4 * (RF + CF)       # every own expansion, the next step's first three, every shift


@pytest.mark.parametrize("RF,CF", [(2, 1), (2, 2), (4, 1), (4, 2)])
def test_the_order_is_executable(RF, CF):
    """Replay the kernel's emission (bitmm_fp4_stream.hip.h::step): MFMA n, then the expansions up to exp_by[n], then the shifts up to
    shift_by[n] - operands must exist when their MFMA issues, in the register slot the MFMA reads, with the right bit of the right words."""
    gen = _generator()
    mf, ops, EP, EN, exp_by, shift_by, todo, gap_of = gen.schedule(RF, CF)
    per_bit, OPB, MN = RF * CF, RF + CF, 4 * RF * CF
    order = [o for (o, m) in ops]                                          # operands of a bit in need order
    assert order[0] == ("a", 0) and order[1] == ("b", 0)                   # (the kernel's decode: o == 0 -> a0, 1 .. CF -> b, then a1 ..)
    assert order == [("a", 0)] + [("b", j) for j in range(CF)] + [("a", i) for i in range(1, RF)]
    A, B = {}, {}                                                          # slot -> (step, bit, fragment)
    shifted = {o: 0 for o in order}                                        # dwords of the fragment shifted so far

    def expand(e, step):
        s, o = divmod(e, OPB)
        kind, f = order[o]
        if step == 0:
            assert shifted[(kind, f)] == (4 if s == 3 else 0), (e, "bit 3 reads the shifted words, bits 0-2 the unshifted ones")
        (A if kind == "a" else B)[((s * RF + f) & 3) if kind == "a" else (s & 1, f)] = (step, s, f)

    for e in range(EP):
        expand(e, 0)
    made, done = EP, 0
    for n in range(MN):
        s, q = divmod(n, per_bit)
        i, j = mf[q]
        assert A.get((s * RF + i) & 3) == (0, s, i), (n, "the A operand of MFMA n is not there")
        assert B.get((s & 1, j)) == (0, s, j), (n, "the B operand of MFMA n is not there")
        while made < exp_by[n]:
            expand(made % EN, made // EN)
            made += 1
        while done < shift_by[n]:
            shifted[order[done >> 2]] += 1
            done += 1
    # what the step leaves for the next one: its first three operands, bit 0
    assert A.get(0) == (1, 0, 0) and B.get((0, 0)) == (1, 0, 0)
    if CF == 2:
        assert B.get((0, 1)) == (1, 0, 1)
    else:
        assert A.get(1) == (1, 0, 1)
