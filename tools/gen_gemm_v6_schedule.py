#!/usr/bin/env python3
"""Emit the compute waves' instruction stream of csrc/gemm_v6.hip (persistent 128x256 GEMM with LOADER waves: the four compute
waves issue no vector-memory instruction inside the K-loop).  A K-tile is 64 MFMAs; MF(S, I, J) = acc[I][J] += W-fragment I x
A-fragment J of 32-wide k-step S (MFZ: the same with C = 0).  Side instructions, at most one per MFMA gap:

  RA(1, j) / RW(1, i)   ds_read_b128 of the k-step-1 fragments of THIS K-tile (ring stage g % 3)
  SYNC()                s_waitcnt lgkmcnt(0) + the K-tile's ONE barrier: every compute wave holds all fragments of K-tile g (the
                        loader waves refill its stage behind it) and the loader waves have seen K-tile g + 1 land
  RA(0, j) / RW(0, i)   ds_read_b128 of the k-step-0 fragments of K-tile g + 1 (stage (g + 1) % 3)
  NEXT()                s_waitcnt lgkmcnt(0), ring stage advances

Variants: A = first K-tile of an output tile (C = 0), B = steady state, L = last K-tile of an output tile (no reads for g + 1:
the epilogue follows and the next tile re-reads its first fragments behind it -- they would cost 48 registers across the epilogue
of a kernel that has 256 per wave).  Edit the tables, run the script: it rewrites the GENERATED block of csrc/gemm_v6.hip.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "gemm_v6.hip")

SYNC_AT = 27


def schedule(variant):
    side = {m: [] for m in range(64)}
    for j in range(4):
        side[j].append(f"RA(1, {j});")
    for i in range(8):
        side[4 + 2 * i].append(f"RW(1, {i});")
    side[SYNC_AT].append("SYNC();")
    if variant != "L":
        reads = [f"RW(0, {i});" for i in range(5)] + [f"RA(0, {j});" for j in range(4)] + [f"RW(0, {i});" for i in range(5, 8)]
        for n, r in enumerate(reads):
            side[SYNC_AT + 2 + 2 * n].append(r)
    side[63].append("NEXT();")
    # a fragment register may be re-loaded only behind the last MFMA that reads it (slot of MF(s, i, j) = 32 s + 4 i + j)
    for m in range(64):
        for ins in side[m]:
            mt = re.match(r"R([AW])\((\d), (\d)\);", ins)
            if mt and mt.group(2) == "0":
                idx = int(mt.group(3))
                last = 28 + idx if mt.group(1) == "A" else 4 * idx + 3
                assert m > last, (ins, m, last)
    return side


def emit():
    out = []
    for v in "ABL":
        side = schedule(v)
        out.append(f"        {'if' if v == 'A' else '} else if'} constexpr (V == '{v}') {{")
        for m in range(64):
            s, rest = divmod(m, 32)
            i, j = divmod(rest, 4)
            mf = "MFZ" if (v == "A" and s == 0) else "MF"
            out.append(f"            {mf}({s}, {i}, {j});" + ("  " + " ".join(side[m]) if side[m] else ""))
    out.append("        }")
    return "\n".join(out)


def main():
    src = open(PATH).read()
    new, n = re.subn(r"(// GENERATED-BEGIN[^\n]*\n).*?([ \t]*// GENERATED-END)",
                     lambda m: m.group(1) + emit() + "\n" + m.group(2), src, flags=re.S)
    assert n == 1, "GENERATED markers not found"
    if "--check" in sys.argv:
        if new != src:
            raise SystemExit(f"{PATH}: the GENERATED block is out of date (run this script without --check)")
        print("up to date", PATH)
        return
    open(PATH, "w").write(new)
    print("rewrote", PATH)


if __name__ == "__main__":
    main()
