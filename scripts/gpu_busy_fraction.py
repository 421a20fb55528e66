"""GPU busy fraction of a rocprofv3 kernel trace: the union of the kernels' [start, end] intervals against the span between
the first and the last of them, over the densest window of N seconds (so that warm-up / set-up gaps do not count).
usage: python scripts/gpu_busy_fraction.py <results.db> [window_ms]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tables if "kernel_dispatch" in x and "rocpd" in x]
    name = t[0] if t else [x for x in tables if "kernel" in x][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({name})")]
    s_col = "start" if "start" in cols else [c for c in cols if "start" in c][0]
    e_col = "end" if "end" in cols else [c for c in cols if "end" in c][0]
    iv = sorted((s, e) for s, e in cur.execute(f"select {s_col},{e_col} from {name}") if e > s)
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 50e6  # ns
    # densest window: slide over starts
    best = None
    j = 0
    busy_prefix = [0]
    merged = []
    for s, e in iv:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [m[0] for m in merged]
    cum = [0]
    for s, e in merged:
        cum.append(cum[-1] + (e - s))
    for i, (s, _) in enumerate(merged):
        k = bisect.bisect_right(starts, s + win) - 1
        busy = cum[k + 1] - cum[i]
        span = merged[k][1] - s
        if span >= 0.8 * win and (best is None or busy / span > best[0]):
            best = (busy / span, s, span, k - i + 1)
    print(f"{len(iv)} kernels, {len(merged)} busy intervals; densest {win/1e6:.0f} ms window: busy {best[0]*100:.1f} % of {best[2]/1e6:.2f} ms "
          f"({best[3]} intervals, mean gap {(1-best[0])*best[2]/max(best[3]-1,1)/1e3:.2f} us)")


if __name__ == "__main__":
    main()
