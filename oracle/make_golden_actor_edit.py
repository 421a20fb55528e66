"""Golden vectors for the EVAL-time actor edit (DynamicActors.edit_boxes2world, model_components/dynamic_actors.py:181-249:
the viewer's sliders and the actor-shift FID evaluation of pipelines/ad_pipeline.py:476-480), produced by the reference
itself: the field / rays of oracle/make_golden_actors.py (its inputs are read back from tests/golden/field_actors.npz) with
``actors.actor_editing`` set.  Run in the build container only:  python oracle/make_golden_actor_edit.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ref_import

ref_import.install()
import synth  # noqa: E402
from make_golden import T, save, set_linear  # noqa: E402
from make_golden_actors import trajectories  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RaySamples  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402

# (lateral, longitudinal, height, rotation, index): a shift of every actor, a yaw of one, both on an index past the last actor
# (clamped, dynamic_actors.py:192), and a height-only edit (ignored, :182-187)
EDITS = [(1.5, -2.0, 0.3, 0.0, -1.0), (0.0, 0.0, 0.0, 0.4, 1.0), (0.8, 0.0, 0.0, -0.3, 7.0), (0.0, 0.0, 0.5, 0.0, -1.0)]


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "field_actors.npz"))
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11),
                                    actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=9, use_4d_hashgrid=False))
    fld = NeuRADField(NeuRADFieldConfig(grid=grid), actors=actors, static_scale=100.0, implementation="torch").eval()
    actors.eval()
    fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5))
    for i, gr in enumerate(fld.hashgrid.actor_grids):
        gr.hash_table.data = T(synth.hash_table(4 * 2**9, 4, seed=400 + i, scale=0.7))
    for k, l in enumerate(fld.mlp_geo.layers):
        set_linear(l, 200 + 10 * k)
    for k, l in enumerate(fld.mlp_feature.layers):
        set_linear(l, 300 + 10 * k)
    R, S = g["starts"].shape
    fr = Frustums(origins=T(g["o"])[:, None].expand(R, S, 3), directions=T(g["d"])[:, None].expand(R, S, 3),
                  starts=T(g["starts"])[..., None], ends=T(g["ends"])[..., None],
                  pixel_area=T(g["area"])[:, None, None].expand(R, S, 1))
    rs = RaySamples(frustums=fr, times=T(g["times"])[:, None, None].expand(R, S, 1))
    out = {}
    with torch.no_grad():
        base = fld(rs)
        assert torch.equal(base[FieldHeadNames.ALPHA][..., 0], T(g["alpha"])), "not the field of field_actors.npz"
        for e, (lat, lon, hgt, rot, idx) in enumerate(EDITS):
            actors.actor_editing.update(lateral=lat, longitudinal=lon, height=hgt, rotation=rot, index=idx)
            o = fld(rs)
            b2w, _ = actors.get_boxes2world(T(g["times"]), flatten=False)
            hit = fld.hashgrid._split_static_vs_actors(rs.frustums.get_fast_isotropic_gaussian(1), rs.times,
                                                       rs.frustums.directions)[0]
            out.update({f"e{e}_b2w": b2w, f"e{e}_alpha": o[FieldHeadNames.ALPHA][..., 0],
                        f"e{e}_hit_ray": hit[0], f"e{e}_hit_sample": hit[1], f"e{e}_hit_actor": hit[2]})
            if e < 2:  # (the features of two edits are enough to pin the box-frame positions / directions; keeps the file small)
                out[f"e{e}_feature"] = o[FieldHeadNames.FEATURE]
            print(e, "hits", hit[0].shape[0], "alpha changed on", int((o[FieldHeadNames.ALPHA] != base[FieldHeadNames.ALPHA]).sum()))
    save("field_actors_edit", edits=np.array(EDITS, np.float32), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
