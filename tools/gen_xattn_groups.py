"""Prints the four asm walk macros of csrc/xattn_fused.hip (XG_WQ / XG_WO / XG_K / XG_V): the step sequence of a fragment group —
D reads in flight, one MFMA per fragment with an exact lgkmcnt wait, the freed register refilled at once, and the eight LDS-DMA pieces
of the group three ahead spread between the MFMAs (one piece per four MFMAs).  The macro text in the .hip file is this script's output
(python tools/gen_xattn_groups.py); edit the schedule here."""
D = 6                                  # fragment reads in flight


def fmt(name, seq):
    lines, cur = [], "    "
    for s in seq:
        if len(cur) + len(s) + 1 > 122:
            lines.append(cur.rstrip()); cur = "    "
        cur += s + " "
    lines.append(cur.rstrip())
    return f"#define {name} \\\n" + " \\\n".join(lines) + "\n"


def group(n, base, cb, dma, first=False, unswapped=False, tail_nop=False):
    """n fragments at slot offsets (base + f) KiB; cb(f) -> (accumulator, B operand); dma: {step: piece} issued behind that step's MFMA;
    first: the group opens its accumulators — the first MFMA on each takes the inline constant 0 as C (no zeroing, no live zero tuple)."""
    out = [f"XRD(t{f}, {(base + f) * 1024})" for f in range(D)]
    seen = set()
    for f in range(n):
        c, b = cb(f)
        t = f"t{f % D}"
        z = "0" if (first and c not in seen) else ""
        seen.add(c)
        if c.startswith("A"):                       # named accumulator tile
            lo = int(c[1:])
            if f + D < n:
                out.append(f"XSA({t}, {lo}, {lo + 15}, {b}, {D - 1}, {(base + f + D) * 1024})")
            else:
                out.append(f"XTA({t}, {lo}, {lo + 15}, {b}, {min(D - 1, n - 1 - f)})")
        elif f + D < n:
            out.append(f"XS{'U' if unswapped else ''}{z}({t}, {c}, {b}, {D - 1}, {(base + f + D) * 1024})")
        else:
            out.append(f"XT{'U' if unswapped else ''}{z}({t}, {c}, {b}, {min(D - 1, n - 1 - f)})")
        if f in dma:
            p = dma[f]
            out.append(f"XD({(p & ~3) * 1024}, {(p & 3) * 1024})")       # the instruction offset moves BOTH the global and the LDS address
            if p == 3:
                out.append("XDADV")
    if tail_nop:
        out.append("XNOP")
    return out


if __name__ == "__main__":
    d32 = {4 * i + 2: i for i in range(8)}                      # pieces 0..7 behind MFMAs 2, 6, ..., 30
    dk = {1: 0, 4: 1, 7: 2, 10: 3}                              # K part: pieces 0..3
    dv = {1: 4, 4: 5, 7: 6, 10: 7}                              # V part: pieces 4..7
    txt = fmt("XG_WQ_FIRST", group(32, 0, lambda f: (f"q{f & 1}", f"b{f >> 1}"), d32, first=True)) + "\n"
    txt += fmt("XG_WQ", group(32, 0, lambda f: (f"q{f & 1}", f"b{f >> 1}"), d32, tail_nop=True)) + "\n"
    # temporal sub-layer: V = Xn . Wv^T with the operands the other way round (A = the token fragments in registers, B = the weight fragment):
    # D[token][channel], lane = channel — the A-operand layout of V^T in O^T = V^T P^T, no transpose
    txt += fmt("XG_WV_FIRST", group(32, 0, lambda f: (f"q{f & 1}", f"b{f >> 1}"), d32, first=True, unswapped=True)) + "\n"
    txt += fmt("XG_WV", group(32, 0, lambda f: (f"q{f & 1}", f"b{f >> 1}"), d32, unswapped=True, tail_nop=True)) + "\n"
    # W_out: the 256 accumulators of the tile live in a[0:255] BY NAME (tile nt = a[16 nt : 16 nt + 15]); group j covers tiles 8 j .. 8 j + 7
    for j in range(2):
        txt += fmt(f"XG_WO{j}", group(32, 0, lambda f: (f"A{16 * (8 * j + 2 * (f >> 3) + (f & 1))}", f"b{(f >> 1) & 3}"), d32)) + "\n"
    txt += fmt("XG_K", group(12, 0, lambda f: (f"c{f % 3}", f"b{f // 3}"), dk, first=True, tail_nop=True)) + "\n"
    txt += fmt("XG_V", group(12, 12, lambda f: (f"c{f & 1}", f"b{f >> 1}"), dv, first=True, tail_nop=True)) + "\n"
    # feed-forward kernel, W_down over ONE 32-wide k slice (half a chunk: two k-steps) onto all 16 accumulator tiles: fragment
    # f = (channel tile 2 (f >> 2) + (f & 1), k-step (f >> 1) & 1)
    txt += fmt("XG_WD32", group(32, 0, lambda f: (f"A{16 * (2 * (f >> 2) + (f & 1))}", f"b{(f >> 1) & 1}"), d32)) + "\n"
    # d = 512 attention (csrc/attn512x.hip): S^T = K . Q^T over k = 512 — the 32 K fragments of a 32-key tile as two statements of 16 (an asm
    # statement takes at most 30 operands), even / odd k-steps on two accumulators (no MFMA waits for its predecessor's result)
    dh = {3: 0, 7: 1, 11: 2, 15: 3}
    txt += fmt("XG_SA", group(16, 0, lambda f: (f"c{f & 1}", f"b{f}"), dh, first=True)) + "\n"
    txt += fmt("XG_SB", group(16, 16, lambda f: (f"c{f & 1}", f"b{f}"), {k: v + 4 for k, v in dh.items()}, tail_nop=True))
    print(txt, end="")
