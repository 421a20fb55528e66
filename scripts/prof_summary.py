"""Summarise rocprofv3 sqlite outputs (gpurun_out/...) into small text files for profiles/.
usage: python scripts/prof_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def short(name, n=90):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n] + "..."


for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"== {path}")
    print("-- kernel-trace stats (top kernels): calls, total_us, avg_us, pct")
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels limit 40"):
        print(f"{calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}%  {short(name)}")
    grp = {"nrhip": [0, 0.0], "other": [0, 0.0]}
    others = []
    for name, calls, total in cur.execute("select name,total_calls,total_duration from top_kernels"):
        # kernels whose templates the tool leaves mangled (_ZN5nrhip...) are this library's as well
        k = "nrhip" if ("nrhip::" in name or "_ZN5nrhip" in name) else "other"
        grp[k][0] += calls
        grp[k][1] += total
        if k == "other":
            others.append((total, calls, name))
    tot = grp["nrhip"][1] + grp["other"][1]
    if tot > 0:
        print("-- by origin: hand-written kernels %d launches %.1f us (%.1f%%) | torch/rocm library kernels %d launches "
              "%.1f us (%.1f%%)" % (grp["nrhip"][0], grp["nrhip"][1], 100 * grp["nrhip"][1] / tot, grp["other"][0],
                                    grp["other"][1], 100 * grp["other"][1] / tot))
        for total, calls, name in sorted(others, reverse=True)[:12]:
            print(f"   other: {calls:6d} {total:10.1f} us  {short(name, 110)}")
    rows = list(cur.execute(
        "select substr(kernel_name,1,200), counter_name, count(*), avg(value), min(value), max(value), avg(duration)"
        " from counters_collection group by kernel_name, counter_name order by avg(value)*count(*) desc limit 10"))
    if rows:
        print("-- PMC per dispatch: counter, n, avg, min, max, avg_duration_ns")
        for name, cn, n, av, mn, mx, dur in rows:
            print(f"{cn:12s} n={n:4d} avg={av:14.2f} min={mn:14.2f} max={mx:14.2f} dur_ns={dur:10.0f}  {short(name, 70)}")
