#!/usr/bin/env python3
"""Emit the hand-placed instruction stream of one K-tile of gemm_v3.hip (one wave per SIMD, 128 x 128 per wave).

A K-tile is 128 MFMAs (v_mfma_f32_16x16x32_bf16, 16 cycles each): MF(S, I, J) = acc[I][J] += W-fragment I x
A-fragment J of 32-wide k-step S.  Between them go, at most one per MFMA gap, the tile's "side" instructions:

  RA(1, j) / RW(1, i)   ds_read_b128 of the k-step-1 fragments of THIS tile            (8 + 8)
  DA(q) / DW(q)         LDS-DMA pieces of tile t+2 into the stage this tile is leaving  (8 + 8)
  RA(0, j) / RW(0, i)   ds_read_b128 of the k-step-0 fragments of tile t+1              (8 + 8)
  three waits + barriers: B1 = every wave has read all of A(t)  -> A half of the stage is free for DMA,
                          B2 = ... all of W(t)                   -> W half is free,
                          B3 = tile t+1 has landed for everybody -> its fragments may be read.

The table below says after which MFMA (0..127) each side instruction is issued.  Edit it, run this script, paste nothing:
the script rewrites the block between the GENERATED markers of csrc/gemm_v3.hip.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "gemm_v3.hip")


def schedule():
    side = {m: [] for m in range(128)}
    # phase A: k-step-1 A fragments of this tile under the first MFMAs, then barrier 1
    for j in range(8):
        side[j].append(f"RA(1, {j});")
    side[15].append("WAIT_LGKM0(); BAR1();")
    # phase B: DMA of A(t+2) interleaved with the k-step-1 W fragments, then barrier 2
    for q in range(8):
        side[16 + 4 * q].append(f"DA({q});")
        side[18 + 4 * q].append(f"RW(1, {q});")
    side[55].append("WAIT_LGKM0(); BAR2();")
    # phase C: DMA of W(t+2); k-step 1 starts at MFMA 64
    for q in range(5):
        side[56 + 4 * q].append(f"DW({q});")
    side[75].append("WAIT_VM_NEXT_TILE(); BAR3(); FLIP0();")
    # phase D: fragments of tile t+1 (k-step 0) + the last DMA pieces
    rd = [f"RA(0, {j});" for j in range(8)] + [f"RW(0, {i});" for i in range(8)]
    slots = list(range(76, 124, 3))            # 16 reads
    for s, r in zip(slots, rd):
        side[s].append(r)
    for q, m in zip(range(5, 8), (78, 90, 102)):
        side[m].append(f"DW({q});")
    side[127].append("WAIT_LGKM0(); FLIP1();")
    return side


def emit():
    side = schedule()
    lines = []
    for m in range(128):
        s, rest = divmod(m, 64)
        i, j = divmod(rest, 8)
        lines.append(f"        MF({s}, {i}, {j});" + ("  " + " ".join(side[m]) if side[m] else ""))
    return "\n".join(lines)


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
