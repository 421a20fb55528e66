"""Build ``oracle/_ref/``: the reference's own Python package as SOURCELESS BYTECODE, so that it travels to the GPU box.

TEST INFRASTRUCTURE -- nothing under ``neurad_studio_amd/`` imports it.

``/root/reference`` exists only in the build container.  A C reference would be compiled from its sources where they lie
into ``oracle/_ref/*.so``; the reference here is pure Python (zero native code), so the same recipe is its byte-compiler:
every ``nerfstudio/**/*.py`` is compiled *from where it lies* with ``py_compile`` into ``oracle/_ref/nerfstudio/**/*.pyc``
(legacy sourceless layout: ``module.pyc`` next to where ``module.py`` would be, unchecked-hash invalidation, so the
importer never looks for a source file).  No reference source text is copied: ``oracle/_ref/`` holds code objects only,
is listed in ``.gitignore`` (out of history) and not in ``.gpurunignore`` (ships with the lease like ``oracle/_build/``).
Both sides run the same image (CPython 3.10), so the bytecode's magic number matches.

What uses it: ``oracle/ref_import.py`` falls back to ``oracle/_ref`` when ``/root/reference`` is absent, which lets
``tests/test_gpu_reference_plugin.py`` run the reference's NeuRADModel(implementation="torch") on the GPU box's CPU next to
the ``neurad-hip`` plugin on the MI355X, and ``bench.py`` time the reference's torch field evaluation on that box's cores
(``cpu_baseline.kind = "reference"``).

    python oracle/make_ref.py            # (also run by __graft_entry__.build() when /root/reference is present)
"""
from __future__ import annotations

import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.environ.get("NEURAD_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(HERE, "_ref")
PACKAGE = "nerfstudio"
# sub-packages the hot path never imports (viewer front ends, dataset download/processing scripts): left out of the build
SKIP_DIRS = ("viewer_legacy/app", "scripts/datasets", "scripts/docs", "process_data")


def _stamp(files) -> dict:
    newest = max(os.path.getmtime(f) for f in files)
    return {"python": list(sys.version_info[:3]), "magic": importlib.util.MAGIC_NUMBER.hex(), "n_modules": len(files),
            "newest_source_mtime": newest, "source_root": SRC_ROOT}


def build(force: bool = False) -> str:
    src_pkg = os.path.join(SRC_ROOT, PACKAGE)
    if not os.path.isdir(src_pkg):
        raise RuntimeError(f"reference tree not found at {SRC_ROOT}")
    files = []
    for d, dirs, names in os.walk(src_pkg):
        rel = os.path.relpath(d, src_pkg)
        if any(rel == s or rel.startswith(s + "/") for s in SKIP_DIRS):
            dirs[:] = []
            continue
        dirs[:] = [x for x in dirs if x != "__pycache__"]
        files += [os.path.join(d, n) for n in names if n.endswith(".py")]
    stamp = _stamp(files)
    stamp_path = os.path.join(OUT, "STAMP.json")
    if not force and os.path.exists(stamp_path):
        try:
            if json.load(open(stamp_path)) == stamp:
                return OUT
        except Exception:
            pass
    shutil.rmtree(OUT, ignore_errors=True)
    for f in files:
        rel = os.path.relpath(f, SRC_ROOT)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: what tracebacks name (the reference path, for the maintainer reading a failure)
        py_compile.compile(f, cfile=dst, dfile=os.path.join("<reference>", rel), doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    with open(stamp_path, "w") as fh:
        json.dump(stamp, fh)
    return OUT


if __name__ == "__main__":
    out = build(force="--force" in sys.argv)
    n = sum(len([x for x in fs if x.endswith(".pyc")]) for _, _, fs in os.walk(out))
    print(f"{out}: {n} modules (bytecode only)")
