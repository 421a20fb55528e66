"""Evidence for one "not supported" in this package: NeuRADFieldConfig.num_multisamples > 1 together with dynamic actors.

The REFERENCE cannot run that configuration either.  With M probes per frustum the hit samples' positions are [P, M, 3]
(neurad_encoding.py:197) and go to transform_points_pairwise together with ONE box transform per pair, [P, 1, 4, 4] (:198);
that helper flattens both into a torch.bmm (cameras/lidars.py:559), which needs equal batch sizes: P * M points against P
rotations raises for every M > 1 as soon as one sample lies inside a box.  (No method config sets num_multisamples != 1,
fields/neurad_field.py:67.)  NeuRADField.__init__ of this package therefore raises NotImplementedError up front instead of
failing in the first batch that meets an actor.

Build container only (imports /root/reference):  python oracle/check_multisample_actors_reference.py
"""
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
from make_golden import T  # noqa: E402
from make_golden_actors import trajectories  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RaySamples  # noqa: E402
from nerfstudio.field_components.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402
from nerfstudio.model_components.dynamic_actors import DynamicActors, DynamicActorsConfig  # noqa: E402


def run(M):
    g = np.load(os.path.join(ROOT, "tests", "golden", "field_actors.npz"))
    actors = DynamicActors(DynamicActorsConfig(), trajectories=trajectories())
    grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11),
                                    actor=ActorSettings(flip_prob=0.25, log2_hashmap_size=9, use_4d_hashgrid=False))
    fld = NeuRADField(NeuRADFieldConfig(grid=grid, num_multisamples=M), actors=actors, static_scale=100.0,
                      implementation="torch").eval()
    actors.eval()
    R, S = g["starts"].shape
    fr = Frustums(origins=T(g["o"])[:, None].expand(R, S, 3), directions=T(g["d"])[:, None].expand(R, S, 3),
                  starts=T(g["starts"])[..., None], ends=T(g["ends"])[..., None],
                  pixel_area=T(g["area"])[:, None, None].expand(R, S, 1))
    rs = RaySamples(frustums=fr, times=T(g["times"])[:, None, None].expand(R, S, 1))
    with torch.no_grad():
        return fld(rs)


if __name__ == "__main__":
    torch.manual_seed(0)
    run(1)
    print("num_multisamples = 1 with actors: reference runs")
    for M in (2, 3):
        try:
            run(M)
        except Exception as e:  # noqa: BLE001
            msg = str(e).strip().splitlines()
            print(f"num_multisamples = {M} with actors: reference raises {type(e).__name__}:", msg[0][:200])
            assert "bmm" in str(e) or "batch2" in str(e)
        else:
            raise SystemExit(f"the reference ran with num_multisamples = {M}: the premise of this note is wrong")
