#!/usr/bin/env python3
"""Emit the hand-placed instruction stream of csrc/attn_w4.hip: the joint self-attention at head_dim 64 with the
static-bound softmax, ONE wave per SIMD, FOUR 32-row query blocks per wave (128 rows per wave, 512 per workgroup).

One 64-key tile is 64 MFMAs (v_mfma_f32_32x32x16_bf16, 32 cycles each) in four PERIODS b = 0..3 of two groups:

  group A  QK_b      8 MFMAs: S_b[u] = K[u] . Q_b^T, key half u = 0, 1, four 16-wide d-steps s each (u-major)
  group B  PV_{b-1}  8 MFMAs: O_{b-1}[d] += V^T[d] . P_{b-1}^T over the four 16-key steps ks, d-half d = 0, 1
           (b = 0: block 3 of the PREVIOUS tile)

and the softmax of block b (32 scores per lane: v_exp_f32, v_add_f32 into the row sum, v_cvt_pk_bf16_f32 into the P
fragment) rides in the MFMA gaps of the two groups BEHIND QK_b: its u = 0 half under PV_{b-1}, its u = 1 half under
QK_{b+1}.  So every MFMA gap carries 2 exp + 2 add + 1 convert = 28 issue cycles + the MFMA's own 8: the kernel is bound
by the vector issue port at ~36.5 cycles per MFMA (MI355X_MICROARCH.md, cycle constants) -- what a hand-placed stream
buys is that nothing ELSE is exposed: S is double-buffered (S[b & 1]), P likewise, the K fragments of tile t + 1 are
read under PV_2, the V fragments of tile t under QK_1, one workgroup barrier per tile sits at the head of PV_2.

Side instructions:
  ds_read_b64_tr_b16  V fragment (d, ks), half h of tile t                  period 1 group A (V registers free after PV_3)
  vmcnt(4) s_barrier  tile t + 1 has landed for everybody, tile t's stage free   period 3 group B, gap 0
  ds_read_b128        K fragment (u, s) of tile t + 1                        period 3 group B (K registers free after QK_3)
  LDS-DMA             pieces K0 K1 V0 V1 of tile t + 3                       period 3 group B

ONE asm statement per group (8 MFMAs and everything between them).  hipcc pads BETWEEN asm statements that hand registers
to each other -- it counts an asm statement as zero wait states, so with one statement per instruction it put an s_nop
behind nearly every MFMA (49 per tile, 4 issue cycles each); inside a statement nothing is padded, and the hazards inside
are kept by construction: an exp result is consumed >= 3 instructions later, a P word >= 2 MFMAs after its convert, a
score >= 4 MFMAs after the MFMA that finished it.  Every output is early-clobber: an LDS read lands asynchronously, it
must not share a register with anything the statement still reads.

Variants: 'L' = the loop body (one tile); 'M' = the same stream for the LAST tile of a piece, with the scores of keys that do
not exist (rows past Skv: zeros in LDS) set to -inf between the statements -- MASKPAD(half, first key of the half), compiler
code: the u = 0 half of block b behind QK_b (its last MFMA is four MFMAs old by then), the u = 1 half behind PV_{b-1} (eight
MFMAs later), each before the softmax that reads it; 'T' = the tail behind the last tile (the u = 1 half of block 3's softmax,
then PV_3).  Edit the tables, run the script: it rewrites the block between the GENERATED markers of csrc/attn_w4.hip;
--check verifies the committed source is what the tables generate (tests/test_abi_cpu.py).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "attn_w4.hip")
RB = 128                                                   # bytes per K / V row in LDS
# Ablation builds (tools/attn_w4_ablate.py; NEVER in the product): bit mask of work to leave out of the stream -- results
# become meaningless, only the time is read.  1: v_exp -> v_mov, 2: no row-sum adds, 4: no converts, 8: no LDS reads / DMA /
# rendezvous inside the loop, 16: Q fragments in VGPRs instead of AGPRs, 32: no VALU at all; finer: 128: no s_barrier,
# 256: no LDS-DMA, 512: no ds_reads, 1024: no lgkmcnt waits, 2048: no vmcnt wait, 4096: no K reads, 8192: no V reads,
# 16384: all V reads in one burst behind the first MFMA of QK_1, 32768: V fragments by eight ds_read_b128 instead of sixteen
# ds_read_b64_tr_b16 (what a V^T layout in memory would allow; addresses meaningless here); 65536 (results VALID): the four
# LDS-DMA pieces issued behind their gap's softmax instructions instead of behind an s_nop.
ABLATE = 0


class Stmt:
    """One asm statement under construction: operands are C++ expressions, deduplicated, outputs numbered first."""

    def __init__(self):
        self.outs, self.ins, self.lines, self.pre, self.post = [], [], [], [], []

    def out(self, expr, cons="=&v"):
        for k, (e, c) in enumerate(self.outs):
            if e == expr:
                return ("o", k)
        self.outs.append((expr, cons))
        return ("o", len(self.outs) - 1)

    def inp(self, expr, cons="v"):
        for k, (e, c) in enumerate(self.outs):            # a value produced earlier in this statement
            if e == expr:
                return ("o", k)
        for k, (e, c) in enumerate(self.ins):
            if e == expr:
                return ("i", k)
        self.ins.append((expr, cons))
        return ("i", len(self.ins) - 1)

    def add(self, fmt, *ops):
        self.lines.append((fmt, ops))

    def render(self, indent):
        n_out = len(self.outs)
        num = lambda r: r[1] if r[0] == "o" else n_out + r[1]
        text = "\\n\\t".join(fmt.format(*[f"%{num(r)}" for r in ops]) for fmt, ops in self.lines)
        o = ", ".join(f'"{c}"({e})' for e, c in self.outs)
        i = ", ".join(f'"{c}"({e})' for e, c in self.ins)
        body = " ".join(self.pre) + f' asm volatile("{text}" : {o} : {i} : "memory"); ' + " ".join(self.post)
        return indent + "{ " + body.strip() + " }"


def softmax_half(st, gap, sb, u, blk):
    """VALU work of MFMA gap `gap` (0..7) for the 16 scores sacc[sb][u][0..15] of block blk.
    Pair g = elements 2g, 2g + 1: its exps in gap g, its adds / convert one gap behind (the last pair's in gap 7 too).
    (Carrying the last pair into the next group's first gap -- five VALU instructions in EVERY gap instead of 2 ... 8 --
    measured the same within noise: the stream is not limited by the uneven gaps.)"""
    def t(i):
        return st.out(f"t{u}_{i}")                          # exp2 of score i: a scratch register of this statement

    def fin(g):
        e = 2 * g
        ks, w = 2 * u + e // 8, (e % 8) // 2
        ps0, ps1 = st.out(f"psum[{blk}][0]", "+v"), st.out(f"psum[{blk}][1]", "+v")
        word = st.out(f"w{ks}_{w}")
        return [("v_add_f32 {0}, {0}, {1}", (ps0, t(e))), ("v_add_f32 {0}, {0}, {1}", (ps1, t(e + 1))),
                ("v_cvt_pk_bf16_f32 {0}, {1}, {2}", (word, t(e), t(e + 1)))]
    ex = [("v_mov_b32 {0}, {1}" if ABLATE & 1 else "v_exp_f32 {0}, {1}", (t(2 * gap + j), st.inp(f"sacc[{sb}][{u}][{2 * gap + j}]")))
          for j in range(2)]
    prev = fin(gap - 1) if gap > 0 else []
    seq = [ex[0]] + prev[:1] + [ex[1]] + prev[1:]           # exp, add, exp, add, cvt: transcendental / plain alternate
    if gap == 7:
        seq += fin(7)
    for fmt, ops in seq:
        if (ABLATE & 32) or (ABLATE & 2 and fmt.startswith("v_add")):
            continue
        if ABLATE & 4 and fmt.startswith("v_cvt"):
            fmt = "v_mov_b32 {0}, 0"                         # (keeps the P word defined; a constant move)
            ops = ops[:1]
        st.add(fmt, *ops)


def close_softmax(st, u, pb):
    """C++ around the statement: scratch declarations, P words back into their fragments."""
    for i in range(16):
        st.pre.append(f"float t{u}_{i};")
    for ks in (2 * u, 2 * u + 1):
        for w in range(4):
            st.pre.append(f"uint32_t w{ks}_{w};")
            st.post.append(f"pf[{pb}][{ks}][{w}] = w{ks}_{w};")


def group_a(variant, b):
    """QK_b (loop) with the u = 1 half of block b-1's softmax; tail: that half alone."""
    st = Stmt()
    sb = b & 1
    prev_blk, prev_pb, prev_sb = (b - 1) % 4, (b - 1) & 1, (1 - sb if variant == "L" else 1)
    reads = [(d, ks, h) for ks in range(4) for d in range(2) for h in range(2)]     # V fragments: 3 3 3 3 2 2 per gap
    cut = [0, 3, 6, 9, 12, 14, 16, 16, 16]
    for g in range(8):
        u, s = divmod(g, 4)
        if variant == "L":
            acc = st.out(f"sacc[{sb}][{u}]")
            k, q = st.inp(f"kf[{u}][{s}]"), st.inp(f"qf[{b}][{s}]", "v" if ABLATE & 16 else "a")
            st.add("v_mfma_f32_32x32x16_bf16 {0}, {1}, {2}, " + ("0" if s == 0 else "{0}"), acc, k, q)
            if b == 1 and ABLATE & 32768:
                for j in ((2 * g, 2 * g + 1) if g < 4 else ()):
                    ks, d = divmod(j, 2)
                    # the K-fragment address pattern (a conflict-free b128 A-operand read of a [64][128 B] tile), on the V tile
                    st.add("ds_read_b128 {0}, {1} offset:" + str(d * 32 * RB + 64 * RB), st.out(f"vq[{d}][{ks}]"), st.inp(f"kaddr[{ks}]"))
            elif b == 1 and not ABLATE & (8 | 512 | 8192):
                for d, ks, h in (reads if g == 0 else []) if ABLATE & 16384 else reads[cut[g]:cut[g + 1]]:
                    st.add("ds_read_b64_tr_b16 {0}, {1} offset:" + str((ks * 16 + h * 8) * RB),
                           st.out(f"vh[{d}][{ks}][{h}]"), st.inp(f"vaddr[{d}]"))
        softmax_half(st, g, prev_sb, 1, prev_blk)
    if variant == "L" and b == 1 and not ABLATE & (8 | 512 | 1024):
        st.add("s_waitcnt lgkmcnt(0)")
    close_softmax(st, 1, prev_pb)
    return st


def group_b(variant, b):
    """PV_{b-1} with the u = 0 half of block b's softmax (loop only); period 3: rendezvous, K reads, DMA."""
    st = Stmt()
    sb, pb = b & 1, b & 1
    pv_blk, pv_pb = (b - 1) % 4, (b - 1) & 1
    for ks in range(4):
        for d in range(2):
            if ABLATE & 32768:
                st.pre.append(f"const u32x4 vv{d}_{ks} = vq[{d}][{ks}];")
            else:
                st.pre.append(f"const u32x4 vv{d}_{ks} = {{vh[{d}][{ks}][0][0], vh[{d}][{ks}][0][1], vh[{d}][{ks}][1][0], vh[{d}][{ks}][1][1]}};")
    for g in range(8):
        ks, d = divmod(g, 2)
        acc = st.out(f"oacc[{pv_blk}][{d}]", "+a")
        st.add("v_mfma_f32_32x32x16_bf16 {0}, {1}, {2}, {0}", acc, st.inp(f"vv{d}_{ks}"), st.inp(f"pf[{pv_pb}][{ks}]"))
        if variant == "L" and b == 3 and not ABLATE & 8:
            if g == 0:
                if not ABLATE & 2048:
                    st.add("s_waitcnt vmcnt(4)")
                if not ABLATE & 128:
                    st.add("s_barrier")
            if g < 4 and not ABLATE & (512 | 4096):
                for j in (2 * g, 2 * g + 1):
                    u, s = divmod(j, 4)
                    st.add("ds_read_b128 {0}, {1} offset:" + str(u * 32 * RB), st.out(f"kf[{u}][{s}]"), st.inp(f"kaddr[{s}]"))
            elif g >= 4 and not ABLATE & 256:
                q = g - 4                                   # pieces K0 K1 V0 V1 of tile t + 3
                st.add("s_mov_b32 m0, {0}", st.inp(f"dma_dst{q}", "s"))
                if ABLATE & 65536:
                    # (experiment, results stay valid) the DMA behind the gap's softmax instructions: they cover the m0 hazard
                    softmax_half(st, g, sb, 0, b)
                else:
                    st.add("s_nop 0")
                st.add("buffer_load_dwordx4 {0}, {1}, {2} offen lds", st.inp(f"dvo[{q}]"),
                       st.inp("rsK" if q < 2 else "rsV", "s"), st.inp("soffK" if q < 2 else "soffV", "s"))
                if ABLATE & 65536:
                    continue
        if variant == "L":
            softmax_half(st, g, sb, 0, b)
    if variant == "L" and b == 3 and not ABLATE & 8:
        if not ABLATE & (512 | 1024):
            st.add("s_waitcnt lgkmcnt(0)")
        for q in range(4):
            st.pre.append(f"const uint32_t dma_dst{q} = dma_dst + {(q & 1) * 1024} + {(q >> 1)} * TILE_BYTES;")
    if variant == "L":
        close_softmax(st, 0, pb)
    return st


def emit():
    lines = []
    for v in "LMT":
        lines.append(f"        {'if' if v == 'L' else '} else if'} constexpr (VAR == '{v}') {{")
        for b in (range(4) if v in "LM" else [4]):
            lines.append(f"            // period {b}: " + (f"QK_{b} | PV_{(b - 1) % 4}" if v in "LM" else "rest of block 3's softmax | PV_3"))
            lines.append(group_a("L" if v == "M" else v, b).render("            "))
            if v == "M":
                lines.append(f"            MASKPAD(sacc[{b & 1}][0], 0);")
            lines.append(group_b("L" if v == "M" else v, b).render("            "))
            if v == "M":
                lines.append(f"            MASKPAD(sacc[{b & 1}][1], 32);")
    lines.append("        }")
    return "\n".join(lines)


def main():
    global ABLATE
    if "--ablate" in sys.argv:                              # tools/attn_w4_ablate.py: write a side copy, never the product file
        ABLATE = int(sys.argv[sys.argv.index("--ablate") + 1])
        out = sys.argv[sys.argv.index("--out") + 1]
        src = open(PATH).read()
        new, n = re.subn(r"(// GENERATED-BEGIN[^\n]*\n).*?([ \t]*// GENERATED-END)",
                         lambda m: m.group(1) + emit() + "\n" + m.group(2), src, flags=re.S)
        assert n == 1
        open(out, "w").write(new)
        return
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
