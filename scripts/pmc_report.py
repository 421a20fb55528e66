"""per-kernel mean of every counter in rocprofv3 sqlite outputs: python scripts/pmc_report.py <kernel-substr> <db>..."""
import sqlite3, sys
sub = sys.argv[1]
for path in sys.argv[2:]:
    cur = sqlite3.connect(path).cursor()
    for cn, n, av in cur.execute("select counter_name, count(*), avg(value) from counters_collection "
                                 "where kernel_name like ? group by counter_name", (f"%{sub}%",)):
        print(f"{cn:32s} n={n:3d} avg={av:18.1f}")
