"""Round 6: the pinned instruction order of k_bitmm_fp4_stream's multiply step (bitmm_fp4_stream.hip.h, the ST_EXP_BY / ST_SHIFT_BY tables).

A step is 4 RF CF MFMAs: bits s = 0..3 of every nibble, and per bit the RF x CF fragment pairs in snake order (row i: j up, row i + 1: j down),
so every MFMA needs at most one operand that the one before it did not. Behind MFMA n ('gap n') the wave issues VALU work for LATER MFMAs:
  * expansions (four ANDs: one fragment's words masked to bit s), in the order the MFMAs need them - per bit a0, b0, (b1,) a1, (a2, a3) - each
    at least one whole gap ahead of its MFMA and as late as that allows, at most one a gap; the last gaps make the NEXT step's first operands;
  * shifts (bit 3 of a nibble is E2M1's sign, so it is multiplied as bit 0 of word >> 3): one dword each, IN PLACE, after the fragment's bit-2
    expansion and before its bit-3 one, filled into the gaps up to six VALU operations a gap (8 issue cycles an MFMA + 4 a VALU operation = 32,
    what the MFMA runs).
Prints, per (RF, CF), the cumulative counts the kernel unrolls against.  python tools/stream_schedule.py"""


def order(RF, CF):
    """MFMAs of one bit: (i, j) in snake order; operands of one bit in need order with the MFMA (within the bit) that needs each first"""
    mf = []
    for i in range(RF):
        js = range(CF) if i % 2 == 0 else range(CF - 1, -1, -1)
        mf += [(i, j) for j in js]
    ops, seen = [], set()
    for m, (i, j) in enumerate(mf):
        for o in (("a", i), ("b", j)):
            if o not in seen:
                seen.add(o)
                ops.append((o, m))
    return mf, ops


def schedule(RF, CF):
    mf, ops = order(RF, CF)
    per_bit, MN = RF * CF, 4 * RF * CF
    exps = []  # (operand, bit, step-relative MFMA index that needs it); the next step's first operands included
    for s in range(4):
        exps += [(o, s, s * per_bit + m) for (o, m) in ops]
    EN = len(exps)
    first = [(o, 0, MN + m) for (o, m) in ops if m <= 1]          # the next step's operands of MFMAs 0 and 1
    EP = len(first)
    todo = exps[EP:] + first                                        # (this step's first EP were made by the step before)
    def assign(cap):   # as late as possible, at most cap a gap, a whole gap ahead of the MFMA, in need order
        gap_of, load, latest = {}, [0] * MN, MN - 1
        for k in range(len(todo) - 1, -1, -1):
            g = min(todo[k][2] - 2, latest)
            while g >= 0 and load[g] >= cap:
                g -= 1
            if g < 0:
                return None
            gap_of[k], latest = g, g
            load[g] += 1
        return gap_of
    gap_of = assign(1) or assign(2) or assign(3)
    exp_by = [EP + sum(1 for k in gap_of if gap_of[k] <= n) for n in range(MN)]
    # shifts: fragments in the order of their bit-3 expansions; window (gap of the bit-2 expansion, gap of the bit-3 expansion)
    idx = {(e[0], e[1]): k for k, e in enumerate(todo) if e[2] < MN or True}
    frags = [o for (o, m) in ops]
    win = []
    for o in frags:
        k2 = next(k for k, e in enumerate(todo) if e[0] == o and e[1] == 2 and e[2] < MN)
        k3 = next(k for k, e in enumerate(todo) if e[0] == o and e[1] == 3 and e[2] < MN)
        win.append((gap_of[k2], gap_of[k3]))                       # shifts in gaps >= lo (behind the expansion there), < hi
    shift_by, done = [], 0
    rem = [4] * len(frags)
    for n in range(MN):
        room = 6 - 4 * sum(1 for k in gap_of if gap_of[k] == n)
        for f, (lo, hi) in enumerate(win):
            # urgent first: everything whose window closes at n + 1
            pass
        for f, (lo, hi) in enumerate(win):
            if rem[f] and n >= lo:
                must = rem[f] if n + 1 >= hi else 0
                take = max(min(rem[f], max(room, 0)), must)
                if sum(rem[:f]):                                    # in order: a fragment's shifts only when the earlier fragments are done
                    take = 0 if not must else take
                rem[f] -= take
                room -= take
                done += take
        shift_by.append(done)
    assert done == 4 * len(frags), (RF, CF, done)
    # checks: every bit-3 expansion behind its fragment's four shifts, every shift behind the bit-2 expansion
    for f, (lo, hi) in enumerate(win):
        assert shift_by[hi - 1] >= 4 * (f + 1) and (lo == 0 or shift_by[lo - 1] <= 4 * f), (RF, CF, f, lo, hi, shift_by)
    return mf, ops, EP, EN, exp_by, shift_by, todo, gap_of


if __name__ == "__main__":
    for RF, CF in ((2, 1), (2, 2), (4, 1), (4, 2)):
        mf, ops, EP, EN, exp_by, shift_by, todo, gap_of = schedule(RF, CF)
        print(f"// RF = {RF}, CF = {CF}: MFMAs of a bit {mf}; operands of a bit {[o[0] + str(o[1]) + '@' + str(m) for (o, m) in ops]}; EP = {EP}, EN = {EN}")
        print(f"//   VALU operations per gap: {[4 * sum(1 for k in gap_of if gap_of[k] == n) + shift_by[n] - (shift_by[n - 1] if n else 0) for n in range(len(exp_by))]}")
        print(f"{{{', '.join(map(str, exp_by))}}},")
        print(f"{{{', '.join(map(str, shift_by))}}},")
