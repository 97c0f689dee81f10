"""Emits the K-tile body of csrc/gemm_fp8_v4.hip between its GENERATED markers: 64 MFMAs in the order phase 0 = W blocks 0..3
x A blocks 0..7, phase 1 = W blocks 4..7 x A blocks 0..7 (A-major), with the barriers, the fragment re-reads and the 16 LDS-DMA
pieces of a K-tile at the positions of the placement table below (a piece "at n" is issued right behind MFMA n).
usage: python tools/gen_gemm_fp8_schedule.py [--check]"""
import os, sys

HIP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bind_your_avatar_implementation_amd", "csrc",
                   "gemm_fp8_v4.hip")
# placement tables: MFMA index behind which piece k is issued (pieces 0..7: A, 8..15: W).  BYA_F8_PLACE selects one at build
# time (A/B runs: tools/fp8_gemm_zeros_probe.py); PLACE_DEFAULT is what ships.
# entry: (MFMA behind which barrier B1 sits, positions)
PLACEMENTS = {
    0: (3, list(range(4, 20))),                                    # one behind each of MFMAs 4..19 (the first version: 412 us on QKV)
    3: (3, list(range(3, 64, 4))[:16]),                            # every 4th, through the whole K-tile (388 us)
    4: (3, list(range(4, 30, 3)) + list(range(33, 54, 3))),        # 9 in phase 0, 7 in the first two thirds of phase 1 (369-383 us)
    7: (5, list(range(6, 31, 3)) + list(range(33, 52, 3))),        # the same with B1 two MFMAs later (366-379 us)
    8: (7, list(range(8, 31, 3)) + list(range(33, 55, 3)), True),  # W(4..7) reads spread over MFMAs 0..3, B1 behind MFMA 7
    9: (9, list(range(10, 31, 3)) + list(range(32, 57, 3)), True), # ... B1 behind MFMA 9 (368 us): SHIPPED
    10: (11, list(range(12, 31, 3)) + list(range(32, 59, 3))[:9], True),   # ... B1 behind MFMA 11
    # measured and dropped (profiles/history/r3_fp8_gemm_probe.json): every 2nd MFMA (385), every 3rd (386), two
    # behind every 4th (396), B1 four MFMAs later (370), nine in phase 0 + seven densely at the start of phase 1 (367)
}
PLACE_DEFAULT = 9
B2_AFTER = 30


def body(entry):
    B1_AFTER, pos = entry[0], entry[1]
    spread = len(entry) > 2 and entry[2]
    assert len(pos) == 16 and all(B1_AFTER <= n < 64 for n in pos) and pos == sorted(pos), entry
    out = []
    emit = out.append
    if spread:
        emit("            RWF(4, cWl, cWh);")
    else:
        emit("            RWF(4, cWl, cWh); RWF(5, cWl, cWh); RWF(6, cWl, cWh); RWF(7, cWl, cWh);")
    for n in range(64):
        phase, j, i = n >> 5, (n >> 2) & 7, (n & 3) + 4 * (n >> 5)
        line = f"            MF8({i}, {j});"
        if spread and n < 3:
            line += f" RWF({5 + n}, cWl, cWh);"
        if n == B1_AFTER:
            line += " B1();"
        for k, at in enumerate(pos):
            if at == n:
                line += f" PIECE({k & 7}, {'true' if k >= 8 else 'false'});"
        if n == B2_AFTER:
            line += f" B2({sum(1 for at in pos if at <= B2_AFTER)});"
        if n == 31:
            line += " REREAD_W();"
        if phase == 1 and (n & 3) == 3:
            line += f" REREAD_A({j});"
        emit(line)
    return out


def generated():
    lines = ["            // GENERATED-BEGIN (tools/gen_gemm_fp8_schedule.py)"]
    first = True
    for key, pos in PLACEMENTS.items():
        lines.append(f"#{'if' if first else 'elif'} BYA_F8_PLACE == {key}")
        lines += body(pos)
        first = False
    lines.append("#else")
    lines.append('#error "BYA_F8_PLACE: unknown placement"')
    lines.append("#endif")
    lines.append("            // GENERATED-END")
    return lines


def main():
    src = open(HIP).read().split("\n")
    a = next(i for i, l in enumerate(src) if "GENERATED-BEGIN" in l)
    b = next(i for i, l in enumerate(src) if "GENERATED-END" in l)
    new = src[:a] + generated() + src[b + 1:]
    if "--check" in sys.argv:
        if new != src:
            sys.exit("gemm_fp8_v4.hip: GENERATED block is stale (run tools/gen_gemm_fp8_schedule.py)")
        want = f"#define BYA_F8_PLACE {PLACE_DEFAULT}"
        if not any(l.strip().startswith(want) for l in src):
            sys.exit(f"gemm_fp8_v4.hip: default placement is not {PLACE_DEFAULT}")
        print("ok")
        return
    open(HIP, "w").write("\n".join(new))


if __name__ == "__main__":
    main()
