"""Discrete-event check of the 8-phase protocol of conv_gemm256p_kernel (tools/next/conv_gemm_8phase.patch): two wave groups,
barriers as global rendezvous, LDS-DMA pairs with random (in-order) completion, counted vmcnt waits.  Verifies for every ds_read:
RAW — both groups' parts of the pair that carries (K-tile, region) have landed AND the reader passed a barrier after the issuing
groups' waits; WAR — no group has ISSUED the pair that overwrites the region for K-tile+2 before this read completed."""
import random, sys

PAIRS = ["W02", "X01", "X23", "W13"]           # j = 0..3

def program(grp, nk):
    ops = []
    def dma(T, j): ops.append(("dma", T, j))
    # prologue
    for j in range(4): dma(0, [0,1,2,3][j])
    if nk > 1:
        dma(1, 0); dma(1, 1); ops.append(("wait", 3))     # vmcnt(6) = 3 pairs
    else:
        ops.append(("wait", 1))                            # vmcnt(2)
    ops.append(("bar",))
    if grp == 1: ops.append(("bar",))
    for t in range(nk):
        more1, more2 = t + 1 < nk, t + 2 < nk
        # phase 1
        ops.append(("read", t, "W02")); ops.append(("read", t, "X01")); ops.append(("read", t, "X23"))   # X: a wave reads ITS piece; both pairs cover all waves
        if more1: dma(t + 1, 2)
        ops.append(("bar",)); ops.append(("retire",)); ops.append(("mfma",)); ops.append(("bar",))
        # phase 2
        ops.append(("read", t, "X01")); ops.append(("read", t, "X23"))
        if more1: dma(t + 1, 3)
        ops.append(("wait", 4 if more1 else 0))
        ops.append(("bar",)); ops.append(("retire",)); ops.append(("mfma",)); ops.append(("bar",))
        # phase 3
        ops.append(("read", t, "W13"))
        if more2: dma(t + 2, 0)
        ops.append(("bar",)); ops.append(("retire",)); ops.append(("mfma",)); ops.append(("bar",))
        # phase 4
        if more2:
            dma(t + 2, 1); ops.append(("wait", 3))
        elif more1:
            ops.append(("wait", 1))
        ops.append(("bar",)); ops.append(("mfma",)); ops.append(("bar",))
    if grp == 0: ops.append(("bar",))
    return ops

def simulate(nk, seed, lat_lo, lat_hi):
    rng = random.Random(seed)
    progs = [program(0, nk), program(1, nk)]
    pc = [0, 0]; now = [0.0, 0.0]
    issued = [[], []]          # per group: list of (T, j, issue_time, done_time)
    pending_reads = [[], []]   # reads issued, not yet retired: (T, region, issue_time)
    read_log = []              # (grp, T, region, issue_time, retire_time)
    bar_wait = [None, None]
    def outstanding(g, t):
        return sum(1 for (_, _, _, d) in issued[g] if d > t)
    while pc[0] < len(progs[0]) or pc[1] < len(progs[1]):
        progressed = False
        for g in (0, 1):
            while pc[g] < len(progs[g]) and bar_wait[g] is None:
                op = progs[g][pc[g]]
                if op[0] == "dma":
                    last_done = issued[g][-1][3] if issued[g] else 0.0
                    done = max(now[g] + rng.uniform(lat_lo, lat_hi), last_done)       # in-order completion
                    issued[g].append((op[1], op[2], now[g], done)); now[g] += 1
                elif op[0] == "wait":
                    n = op[1]
                    # advance time until outstanding <= n
                    dones = sorted(d for (_, _, _, d) in issued[g] if d > now[g])
                    if len(dones) > n:
                        now[g] = dones[len(dones) - n - 1]
                elif op[0] == "read":
                    pending_reads[g].append((op[1], op[2], now[g])); now[g] += 1
                elif op[0] == "retire":
                    now[g] += rng.uniform(5, 40)          # LDS latency
                    for (T, R, ti) in pending_reads[g]: read_log.append((g, T, R, ti, now[g]))
                    pending_reads[g] = []
                elif op[0] == "mfma":
                    now[g] += rng.uniform(200, 300)
                elif op[0] == "bar":
                    bar_wait[g] = now[g]
                pc[g] += 1; progressed = True
        if bar_wait[0] is not None and bar_wait[1] is not None:
            tt = max(bar_wait) + 5
            now = [tt, tt]; bar_wait = [None, None]; progressed = True
        elif not progressed:
            return "DEADLOCK pc=%s/%s %s/%s" % (pc[0], len(progs[0]), pc[1], len(progs[1]))
    # checks
    errs = []
    for (g, T, R, ti, tr) in read_log:
        j = PAIRS.index(R)
        for g2 in (0, 1):
            land = [d for (T2, j2, _, d) in issued[g2] if T2 == T and j2 == j]
            if not land: errs.append(f"no DMA for K-tile {T} {R} grp {g2}"); continue
            if land[0] > ti: errs.append(f"RAW: grp {g} reads K-tile {T} {R} at {ti:.0f}, grp {g2} part lands {land[0]:.0f}")
            over = [i for (T2, j2, i, _) in issued[g2] if T2 == T + 2 and j2 == j]
            if over and over[0] < tr: errs.append(f"WAR: grp {g} read of K-tile {T} {R} retires {tr:.0f}, grp {g2} restages at {over[0]:.0f}")
    # leftovers: all DMAs complete by the end
    end = max(now)
    for g2 in (0, 1):
        if any(d > end for (_, _, _, d) in issued[g2]): errs.append("DMA outstanding at the end")
    return errs

bad = 0
for nk in (1, 2, 3, 4, 5, 8, 9, 18):
    for seed in range(200):
        for (lo, hi) in ((100, 400), (1000, 3000), (10, 20), (3000, 9000)):
            r = simulate(nk, seed, lo, hi)
            if r:
                bad += 1
                if bad <= 8: print(nk, seed, lo, hi, r if isinstance(r, str) else r[:3])
print("violations:", bad)
