"""per-kernel mean / median / max of every counter in rocprofv3 sqlite outputs: python scripts/pmc_report.py <kernel-substr> <db>...
(the median is the figure to quote when a run mixes launch sizes of one kernel, e.g. bench.py --config c3's parity check)"""
import sqlite3, statistics, sys
sub = sys.argv[1]
for path in sys.argv[2:]:
    cur = sqlite3.connect(path).cursor()
    vals = {}
    for cn, v in cur.execute("select counter_name, value from counters_collection where kernel_name like ?", (f"%{sub}%",)):
        vals.setdefault(cn, []).append(v)
    for cn, v in sorted(vals.items()):
        print(f"{cn:32s} n={len(v):3d} avg={sum(v) / len(v):18.1f} median={statistics.median(v):18.1f} max={max(v):18.1f}")
