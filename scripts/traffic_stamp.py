"""Stamp profiles/traffic_<name>.json with the sha1 of the kernel sources its PMC passes were taken on, so that bench.py can
tell a measured `roofline.traffic` from a stale one:  python scripts/traffic_stamp.py <name> <source> [<source>...]
(sources relative to neurad_studio_amd/csrc).  Run it right after re-measuring (scripts/profile_render.sh / pmc_pass.sh)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha1(rel):
    return hashlib.sha1(open(os.path.join(ROOT, "neurad_studio_amd", "csrc", rel), "rb").read()).hexdigest()


if __name__ == "__main__":
    name, srcs = sys.argv[1], sys.argv[2:]
    path = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    d = json.load(open(path))
    d["source_sha1"] = {s: sha1(s) for s in srcs}
    json.dump(d, open(path, "w"), indent=1)
    print(path, d["source_sha1"])
