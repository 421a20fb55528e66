"""profiles/traffic_<name>.json from the text of the PMC passes (scripts/final_measure.sh: `== <pass>: <counter> (<kernel>)`
headers followed by scripts/pmc_report.py lines), stamped with the sha1 of the kernel sources of THIS tree so that bench.py
can tell a measured roofline.traffic from a stale one.  usage: python scripts/traffic_from_pmc.py <pmc text> <round tag> [name,name,...]
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB (median per launch): MI355X_MICROARCH.md, HBM section -- FETCH_SIZE tallies
gfx950's 128-byte fabric requests at 64 B, WRITE_SIZE is taken as reported."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# traffic name -> (fetch pass, write pass, kernel description, sources, bench command)
ENTRIES = {
    "render_kernel": ("c1_fetch", "c1_write", "nrhip::render_kernel<16,2,64,fp32,composite> (bench.py, BASELINE config[1])",
                      ["render.hip", "common.h", "rayorder.h"], "bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-variants"),
    "proposal_sampler": ("c2_fetch", "c2_write", "nrhip::proposal_sampler_kernel<false,6,false> (bench.py --config c2)",
                         ["sampler.hip", "common.h"], "bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline"),
    "field_fwd_train": ("c3_fetch", "c3_write", "nrhip::render_kernel<8,4,32,fp32,train> (fused training forward, bench.py --config c3)",
                        ["render.hip", "common.h", "rayorder.h"], "bench.py --config c3 --steps 3 --warmup 1 --no-rgb-decoder"),
    "render_actors_fp16": ("c4_fetch", "c4_write", "nrhip::render_kernel<8,4,32,fp16,composite,ACT> (the rays with candidate actors, "
                           "bench.py --config c4)", ["render.hip", "common.h", "rayorder.h", "actors.hip"],
                           "bench.py --config c4 --steps 5 --warmup 2 --train-steps 0"),
    "proposal_sampler_actors": ("c4s_fetch", "c4s_write", "nrhip::proposal_sampler_kernel<false,6,ACT> (bench.py --config c4)",
                                ["sampler.hip", "common.h"], "bench.py --config c4 --steps 5 --warmup 2 --train-steps 0"),
    "field_fwd_train_ovr_fp16": ("c4t_fetch", "c4t_write", "nrhip::render_kernel<8,4,32,fp16,train,OVR> (fused training forward with row "
                                 "overrides, bench.py --config c4)", ["render.hip", "common.h", "rayorder.h"],
                                 "bench.py --config c4 --steps 2 --warmup 1 --train-steps 36"),
}


def sha1(rel):
    return hashlib.sha1(open(os.path.join(ROOT, "neurad_studio_amd", "csrc", rel), "rb").read()).hexdigest()


def main():
    text, tag = open(sys.argv[1]).read(), sys.argv[2]
    passes, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"== (\S+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+).*median=\s*([0-9.]+)", line)
        if m and cur:
            passes[cur] = (m.group(1), int(m.group(2)), float(m.group(3)))
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None  # optional: restrict to these traffic names
    for name, (pf, pw, kernel, srcs, cmd) in ENTRIES.items():
        if only is not None and name not in only:
            continue
        if pf not in passes or pw not in passes:
            print("missing passes for", name)
            continue
        fkb, wkb = passes[pf][2], passes[pw][2]
        rec = {"round": tag, "kernel": kernel,
               "command": cmd + "  under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, --kernel-trace only; median per launch)",
               "launches_per_pass": [passes[pf][1], passes[pw][1]], "fetch_size_kb_per_launch_raw": fkb, "write_size_kb_per_launch_raw": wkb,
               "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE tallies 128-B fabric requests at 64 B on gfx950 -> x2 on the read "
                             "side; WRITE_SIZE as reported",
               "hbm_bytes_per_launch": int((2 * fkb + wkb) * 1024), "source_sha1": {s: sha1(s) for s in srcs}}
        path = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
        if os.path.exists(path):  # keep what a re-measurement does not change, and say what the figure was before
            old = json.load(open(path))
            for k in ("algorithmic_bytes_per_launch", "note"):
                if k in old:
                    rec[k] = old[k]
            if "hbm_bytes_per_launch" in old:
                rec["previous"] = {"round": old.get("round"), "hbm_bytes_per_launch": old["hbm_bytes_per_launch"]}
        json.dump(rec, open(path, "w"), indent=1)
        print(name, rec["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
