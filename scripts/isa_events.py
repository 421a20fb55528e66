#!/usr/bin/env python3
"""usage: isa_events.py <file.hip> <mangled-name-substring>  -- compile with -save-temps into /tmp and print the order of
vector loads (GL) / stores (GS) / scalar loads (SL) / vmcnt waits (W[n]) / lgkmcnt waits (K[n]) / MFMAs (M) / branches
of one kernel, run-length compressed: shows whether prefetched loads really stay in flight across the MFMA phase."""
import os, re, subprocess, sys, tempfile
src, key = os.path.abspath(sys.argv[1]), sys.argv[2]
d = tempfile.mkdtemp(prefix="isa_")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                "-ffp-contract=off", "-Wno-unused-result", "-save-temps", "-c", src, "-o", "/dev/null"], cwd=d,
               stderr=subprocess.DEVNULL, check=True)
s = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
i = m.start(); j = s.index("s_endpgm", i)
ev = []
for l in s[i:j].split("\n"):
    l = l.strip()
    mm = re.match(r"([a-z_0-9]+)", l)
    if not mm:
        if l.startswith(".LBB"): ev.append("|" + l.split(":")[0][4:])
        continue
    op = mm.group(1)
    if op.startswith(("global_load", "buffer_load")): ev.append("GL")
    elif op.startswith(("global_store", "buffer_store")): ev.append("GS")
    elif op.startswith("scratch_"): ev.append("SCR")
    elif op.startswith("v_mfma"): ev.append("M")
    elif op == "s_waitcnt":
        a = re.search(r"vmcnt\((\d+)\)", l); b = re.search(r"lgkmcnt\((\d+)\)", l)
        if a: ev.append("W[%s]" % a.group(1))
        if b and os.environ.get("LGKM"): ev.append("K[%s]" % b.group(1))
    elif op.startswith("s_cbranch") or op == "s_branch": ev.append("->" + l.split()[-1][4:])
    elif op.startswith("s_load"): ev.append("SL")
out, prev, cnt = [], None, 0
for e in ev + [None]:
    if e == prev: cnt += 1
    else:
        if prev: out.append(prev + ("x%d" % cnt if cnt > 1 else ""))
        prev, cnt = e, 1
print(m.group(1)); print(" ".join(out))
