#!/usr/bin/env python3
"""Where a wave of the particle kernels spends its life.  Needs a build with -DSMAC_PHASE_CLOCK=1 (1 workgroup in 16 files s_memtime at its phase
boundaries; python3 -m softmac_amd.build -DSMAC_PHASE_CLOCK=1 --out=libsoftmac_hip_phase.so) and a dump written by the library when the handle is
destroyed (SMAC_PHASE_DUMP=file).  Prints, per kernel, the mean s_memtime ticks between consecutive markers and their share of the wave's life.
   SMAC_LIB=.../libsoftmac_hip_phase.so SMAC_PHASE_DUMP=/tmp/p.txt python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f64 --no-cloth --no-env-loop
   python3 tools/phase_clock.py /tmp/p.txt"""
import sys

NAMES = {
    "k_p2g_g2p_grad": [(0, "start"), (1, "zero scatter tile, stage 2 gather tiles, x[f-1]; barrier"), (2, "24+9 rows in, constitutive_fwd (SVD), park in LDS"),
                       (3, "p2g.grad 27-node gather"), (4, "x/v rows out, constitutive adjoint"), (5, "F_tmp adjoint (C, F re-read), 18 rows out"),
                       (6, "tile_scale (wave max, barrier)"), (7, "g2p.grad 27-node gather + 81 ds_add, x.grad out"), (8, "barrier"), (9, "tile -> slab")],
    "k_p2g": [(16, "start (tile zeroed)"), (17, "24 rows in, SVD, stress, F' out, band test"), (18, "tile_scale (barrier)"), (19, "108 ds_add"), (20, "barrier"),
              (21, "tile -> slab")],
    "k_p2g_g2p_grad, every wave": [(10, "entry"), (11, "chunk descriptor in")],
    "k_p2g, every wave": [(22, "entry"), (23, "primitive states -> LDS, barrier, chunk descriptor in")],
    "k_g2p": [(24, "start"), (25, "x in, tile staged; barrier"), (26, "27-node gather, 15 rows out")],
    # round 5: the contact adjoint (workgroups that hold hits; the markers inside the hit loop fire in its first trip)
    "k_contact_grad": [(32, "entry"), (33, "hit count, first hit's block and own hit record in (asked for together); barrier"),
                       (34, "tile zeroed; position, x.grad's old value and the flush nodes' {m,p} asked for"),
                       (35, "position in, 27 node records (grid_v_out.grad, {m,p}) gathered, group reductions"), (36, "forward replay + dual chains of the primitives in range"),
                       (37, "barrier; node scatter into the LDS tile, x.grad out"), (38, "barrier"), (39, "late global atomics, tile flushed through grid_op's node adjoint"),
                       (40, "primitive-state adjoints out")],
}


def main(path):
    M = 1 << 64
    acc = {}
    hist = {}
    slow = {}
    for line in open(path):
        m, a, c = (int(v) for v in line.split())
        if m == -1:                                # "-1 kernel*64+bin count": every wave's entry-to-exit time of the two contact kernels, bins of 1024 ticks
            hist[a] = hist.get(a, 0) + c
            continue
        if m == -2:                                # "-2 record*8+word value": the slowest waves of k_contact_grad
            slow[a] = c
            continue
        s, n = acc.get(m, (0, 0))
        acc[m] = ((s + a) % M, n + c)
    for kern, marks in NAMES.items():
        n0 = acc.get(marks[0][0], (0, 0))[1]
        if not n0:
            continue
        print(f"{kern}: {n0} sampled waves")
        total = 0
        rows = []
        for (ma, _), (mb, label) in zip(marks, marks[1:]):
            (sa, na), (sb, nb) = acc[ma], acc[mb]
            note = "" if na == nb == n0 else f"   (hits {na} -> {nb})"
            d = ((sb - sa) % M) / max(nb, 1) if na == nb else float("nan")
            rows.append((label, d, note))
            total += d if d == d else 0
        for label, d, note in rows:
            print(f"   {d:10.0f} ticks  {100 * d / total:5.1f} %   {label}{note}")
        print(f"   {total:10.0f} ticks  per wave")
    for k, kern in enumerate(("k_contact_hits", "k_contact_grad")):
        bins = {b - 64 * k: c for b, c in hist.items() if 64 * k <= b < 64 * (k + 1)}
        if bins:
            n = sum(bins.values())
            print(f"{kern}: entry-to-exit time of all {n} waves (a launch lasts as long as its slowest wave), k ticks: share of the waves")
            print("   " + "  ".join(f"{b}{'+' if b == 63 else ''}k: {100 * c / n:.1f} %" for b, c in sorted(bins.items())))
    slow_report(slow)


def slow_report(slow):
    n = (max(slow) + 1) // 8 if slow else 0
    if n:
        print(f"k_contact_grad: {n} waves over 36,000 ticks (the first 256 recorded): ticks entry-to-exit | workgroup wave | band masks of the wave's two hits | hit count | "
              "ticks from entry to: loads in, nodes gathered, chains done, tile scattered + barrier, flushed")
    for r in range(n):
        w = [slow.get(8 * r + q, 0) for q in range(8)]
        print(f"   {w[0]:6d} | wg {w[1] & 0xffff:4d} wave {(w[1] >> 16) & 0xff} | {(w[1] >> 24) & 0xff:#04x} {(w[1] >> 32) & 0xff:#04x} | {w[2]:5d} | " + " ".join(f"{v:6d}" for v in w[3:8]))


if __name__ == "__main__":
    main(sys.argv[1])

