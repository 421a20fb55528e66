"""Idle time between consecutive kernels of the c3 training step, from a rocprofv3 --kernel-trace sqlite file:
where does the step's wall time go that is NOT kernel time (launch gaps, host syncs, dependency bubbles)?

    python scripts/step_gaps.py <results.db> [marker-substring]

A step = the window between two consecutive launches of the marker kernel (default: the field MLP's backward, one per
training step).  Prints, averaged over the steady windows: wall, busy (sum of kernel durations), idle, the histogram of the
gaps and the largest gaps with the kernels on either side."""
import collections
import sqlite3
import sys


def rows_of(con):
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for cand in ["kernels"] + [n for n in names if "kernel" in n.lower()]:
        if cand not in names:
            continue
        cols = [r[1] for r in cur.execute(f"pragma table_info('{cand}')")]
        if "start" in cols and "end" in cols and ("name" in cols or "kernel_name" in cols):
            nm = "name" if "name" in cols else "kernel_name"
            return list(cur.execute(f"select {nm}, start, end from '{cand}' order by start"))
    raise SystemExit(f"no kernel table with start/end/name in {names}")


def short(n, k=70):
    n = n.replace("void ", "").replace("nrhip::", "").replace("(anonymous namespace)::", "")
    return n if len(n) <= k else n[:k] + "..."


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "mlp_chain_bwd_wg_kernel<48"
    rows = rows_of(sqlite3.connect(db))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 4:
        raise SystemExit(f"marker {marker!r}: {len(marks)} launches")
    wins = [(marks[i], marks[i + 1]) for i in range(1, len(marks) - 1)]
    n_k = collections.Counter(b - a for a, b in wins).most_common(1)[0][0]
    wins = [(a, b) for a, b in wins if b - a == n_k]  # the steady step: same kernel sequence
    print(f"{len(wins)} steady windows of {n_k} kernels (marker {marker!r})")
    wall = busy = 0.0
    gaps = [0.0] * n_k
    for a, b in wins:
        wall += (rows[b][1] - rows[a][1]) / 1e3
        for k in range(n_k):
            r, nx = rows[a + k], rows[a + k + 1]
            busy += (r[2] - r[1]) / 1e3
            gaps[k] += max(0.0, (nx[1] - r[2]) / 1e3)
    n = len(wins)
    wall, busy, gaps = wall / n, busy / n, [g / n for g in gaps]
    print(f"per step: wall {wall:.1f} us, kernel time {busy:.1f} us, idle {sum(gaps):.1f} us ({100 * sum(gaps) / wall:.1f} %)")
    hist = collections.Counter()
    for g in gaps:
        hist["<1" if g < 1 else "1-2" if g < 2 else "2-4" if g < 4 else "4-8" if g < 8 else "8-20" if g < 20 else ">=20"] += 1
    tot = {k: sum(g for g in gaps if (k == "<1" and g < 1) or (k == "1-2" and 1 <= g < 2) or (k == "2-4" and 2 <= g < 4) or
                  (k == "4-8" and 4 <= g < 8) or (k == "8-20" and 8 <= g < 20) or (k == ">=20" and g >= 20)) for k in hist}
    for k in ("<1", "1-2", "2-4", "4-8", "8-20", ">=20"):
        if k in hist:
            print(f"  gaps {k:>5s} us: {hist[k]:4d} boundaries, {tot[k]:7.1f} us")
    a0 = wins[0][0]
    print("largest gaps (us, mean over the windows): after -> before")
    for k in sorted(range(n_k), key=lambda k: -gaps[k])[:25]:
        print(f"  {gaps[k]:7.1f}  {short(rows[a0 + k][0])}  ->  {short(rows[a0 + k + 1][0])}")


if __name__ == "__main__":
    main()
