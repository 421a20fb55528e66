"""The idle gaps of a rocprofv3 kernel trace inside its densest window: for every gap above a threshold, the kernel that
ended before it and the one that started after it (which host code the GPU was waiting for).
usage: python scripts/gpu_gaps.py <results.db> [window_ms] [min_gap_us]"""
import bisect
import collections
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
    min_gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 8e3
    # (hipBLASLt GEMMs are bench.py's clock warm-up spin, not part of any step: they would win the "densest window" contest)
    rows = sorted(r for r in cur.execute("select start, end, name from kernels") if not r[2].startswith("Cijk_"))
    merged = []  # [start, end, last kernel name, first kernel name]
    for s, e, n in rows:
        if merged and s <= merged[-1][1]:
            if e > merged[-1][1]:
                merged[-1][1], merged[-1][2] = e, n
        else:
            merged.append([s, e, n, n])
    starts = [m[0] for m in merged]
    cum = [0]
    for m in merged:
        cum.append(cum[-1] + m[1] - m[0])
    best = None
    for i, m in enumerate(merged):
        k = bisect.bisect_right(starts, m[0] + win) - 1
        span = merged[k][1] - m[0]
        if span >= 0.8 * win:
            f = (cum[k + 1] - cum[i]) / span
            if best is None or f > best[0]:
                best = (f, i, k)
    f, i, k = best
    span = merged[k][1] - merged[i][0]
    print(f"window {span/1e6:.2f} ms, busy {f*100:.1f} %, idle {(1-f)*span/1e3:.0f} us in {k-i} gaps")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for j in range(i, k):
        gap = merged[j + 1][0] - merged[j][1]
        if gap >= min_gap:
            key = (merged[j][2][:70], merged[j + 1][3][:70])
            agg[key][0] += 1
            agg[key][1] += gap
    tot = sum(v[1] for v in agg.values())
    print(f"gaps >= {min_gap/1e3:.0f} us: {sum(v[0] for v in agg.values())} gaps, {tot/1e3:.0f} us")
    for (a, b), (n, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{n:4d} x {g/n/1e3:7.1f} us   after {a}\n                     before {b}")


if __name__ == "__main__":
    main()
