"""oracle/make_ref.py -- the recipe that takes the (pure-Python) reference to the GPU box as sourceless bytecode -- builds a
tree that holds NO source text and from which the reference's hot-path modules import and run (build container only: needs
/root/reference).  What runs on it on the GPU box: tests/test_gpu_reference_plugin.py and bench.py's cpu_baseline."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_ref  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(make_ref.SRC_ROOT, make_ref.PACKAGE)),
                                reason="reference sources not present (the GPU box uses the prebuilt oracle/_ref)")


def test_bytecode_tree_has_no_sources_and_runs_the_reference_field():
    out = make_ref.build()
    names = [f for _, _, fs in os.walk(out) for f in fs]
    assert any(n.endswith(".pyc") for n in names) and len(names) > 100
    assert not [n for n in names if n.endswith((".py", ".pyi", ".cu", ".cpp", ".md"))], "bytecode only: no source files"
    assert os.path.exists(os.path.join(out, "nerfstudio", "fields", "neurad_field.pyc"))
    # listed in .gitignore (out of history), NOT in .gpurunignore (ships with the lease)
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read()
    gi = os.path.join(ROOT, ".gpurunignore")
    assert not os.path.exists(gi) or "oracle/_ref" not in open(gi).read()
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(ROOT, "oracle")!r})
        import ref_import
        assert ref_import.reference_kind() == "bytecode" and ref_import.reference_available()
        ref_import.install()
        import torch
        import nerfstudio.fields.neurad_field as f
        assert f.__file__.endswith("neurad_field.pyc") and {out!r} in f.__file__
        from nerfstudio.cameras.rays import RayBundle
        from nerfstudio.field_components.neurad_encoding import NeuRADHashEncodingConfig, StaticSettings
        from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig
        from nerfstudio.model_components.ray_samplers import PowerSampler
        cfg = f.NeuRADFieldConfig(grid=NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=8)))
        fld = f.NeuRADField(cfg, actors=DynamicActors(DynamicActorsConfig(), trajectories=[]), static_scale=100.0,
                            implementation="torch").eval()
        R = 5
        rb = RayBundle(origins=torch.zeros(R, 3), directions=torch.nn.functional.normalize(torch.randn(R, 3), dim=-1),
                       pixel_area=torch.full((R, 1), 1e-6), nears=torch.zeros(R, 1), fars=torch.full((R, 1), 50.0),
                       times=torch.zeros(R, 1))
        out = fld(PowerSampler(num_samples=7).eval()(rb))
        assert all(torch.isfinite(v).all() for v in out.values())
        print("REF-BYTECODE-OK")
        """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, NEURAD_REFERENCE_ROOT="/nonexistent", PYTHONDONTWRITEBYTECODE="1"))
    assert "REF-BYTECODE-OK" in r.stdout, r.stderr[-3000:]
