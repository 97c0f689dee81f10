#!/usr/bin/env python3
"""Emit the hand-placed instruction streams of csrc/gemm_v4.hip (persistent 256x256 GEMM, one wave per SIMD, 128x128 per
wave): the four K-tile variants of one output tile.  A K-tile is 128 MFMAs (v_mfma_f32_16x16x32_bf16, 16 cycles each);
MF(S, I, J) = acc[I][J] += W-fragment I x A-fragment J of 32-wide k-step S (MFZ: the same with C = 0, first K-tile).
Between them go, at most one per MFMA gap, the "side" instructions:

  RA(1, j) / RW(1, i)   ds_read_b128 of the k-step-1 fragments of THIS K-tile
  RA(0, j) / RW(0, i)   ds_read_b128 of the k-step-0 fragments of K-tile t+1
  DA(q) / DW(q)         LDS-DMA pieces of K-tile t+2 into the stage this K-tile is leaving (variants A, B)
  PA(q) / PW(q)         LDS-DMA pieces of K-tile 0 of the NEXT OUTPUT TILE into this K-tile's stage (variant C)
  QA(q) / QW(q)         ... of its K-tile 1 (variant D)
  barriers: B1 = every wave holds all A fragments of this K-tile -> the A half of its stage may be re-filled,
            B2 = ... all W fragments, B3 = K-tile t+1 has landed for everybody, BP = every wave is done with LDS.

Variants:  A = first K-tile of an output tile (C = 0), B = steady state, C = last but one (no K-tile t+2 left; starts the
next output tile's prefetch), D = last (drain; rest of the prefetch).  Edit the tables, run the script: it rewrites the
block between the GENERATED markers of csrc/gemm_v4.hip.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "gemm_v4.hip")


def schedule(variant):
    side = {m: [] for m in range(128)}
    for j in range(8):                                   # phase A: k-step-1 A fragments of this K-tile
        side[j].append(f"RA(1, {j});")
    if variant in "AB":
        side[15].append("WAIT_LGKM0(); BAR();")          # B1
        for q in range(8):                               # phase B: DMA of A(t+2) + k-step-1 W fragments
            side[16 + 4 * q].append(f"DA({q});")
            side[18 + 4 * q].append(f"RW(1, {q});")
        side[55].append("WAIT_LGKM0(); BAR();")          # B2
        for q in range(5):                               # phase C: DMA of W(t+2)
            side[56 + 4 * q].append(f"DW({q});")
        side[75].append("WAIT_VM(13); BAR(); FLIP0();")  # B3: all but this K-tile's own 13 pieces have landed
        for s, r in zip(range(76, 124, 3), [f"RA(0, {j});" for j in range(8)] + [f"RW(0, {i});" for i in range(8)]):
            side[s].append(r)
        for q, m in zip(range(5, 8), (78, 90, 102)):
            side[m].append(f"DW({q});")
        side[127].append("WAIT_LGKM0(); FLIP1();")
    elif variant == "C":
        side[15].append("WAIT_LGKM0();")
        for q in range(8):
            side[18 + 4 * q].append(f"RW(1, {q});")
        side[55].append("WAIT_LGKM0();")
        side[75].append("WAIT_VM(0); BAR(); FLIP0();")   # B3: the last K-tile has landed; this stage is free
        for s, r in zip(range(76, 124, 3), [f"RA(0, {j});" for j in range(8)] + [f"RW(0, {i});" for i in range(8)]):
            side[s].append(r)
        for s, r in zip(range(77, 125, 3), [f"PA({q});" for q in range(8)] + [f"PW({q});" for q in range(8)]):
            side[s].append(r)
        side[127].append("WAIT_LGKM0(); FLIP1();")
    else:                                                # D
        side[15].append("WAIT_LGKM0();")
        for q in range(8):
            side[18 + 4 * q].append(f"RW(1, {q});")
        side[55].append("WAIT_LGKM0(); BAR();")          # BP: nobody reads LDS any more
        for s, r in zip(range(57, 105, 3), [f"QA({q});" for q in range(8)] + [f"QW({q});" for q in range(8)]):
            side[s].append(r)
        side[127].append("FLIP0(); FLIP1();")
    return side


def emit():
    out = []
    for v in "ABCD":
        side = schedule(v)
        out.append(f"        {'if' if v == 'A' else '} else if'} constexpr (V == '{v}') {{")
        for m in range(128):
            s, rest = divmod(m, 64)
            i, j = divmod(rest, 8)
            mf = "MFZ" if (v == "A" and s == 0) else "MF"
            out.append(f"            {mf}({s}, {i}, {j});" + ("  " + " ".join(side[m]) if side[m] else ""))
    out.append("        }")
    return "\n".join(out)


def main():
    src = open(PATH).read()
    new, n = re.subn(r"(// GENERATED-BEGIN[^\n]*\n).*?([ \t]*// GENERATED-END)",
                     lambda m: m.group(1) + emit() + "\n" + m.group(2), src, flags=re.S)
    assert n == 1, "GENERATED markers not found"
    if "--check" in sys.argv:                       # the committed kernel source must be what the tables generate
        if new != src:
            raise SystemExit(f"{PATH}: the GENERATED block is out of date (run this script without --check)")
        print("up to date", PATH)
        return
    open(PATH, "w").write(new)
    print("rewrote", PATH)


if __name__ == "__main__":
    main()
