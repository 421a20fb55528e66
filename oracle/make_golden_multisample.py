"""Golden vectors for NeuRADFieldConfig.num_multisamples > 1 (fields/neurad_field.py:67,134; cameras/rays.py:109-124;
neurad_encoding.py:297-304), produced by the reference itself: the SDF field of oracle/make_golden.py:golden_field (same
weights, same rays -- read back from tests/golden/field_sdf.npz) evaluated with 3 multisamples, forward and the gradients of
a seeded linear functional.  Run in the build container only:  python oracle/make_golden_multisample.py"""
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
from make_golden import T, no_actors, save, set_linear  # noqa: E402
from nerfstudio.cameras.rays import Frustums, RaySamples  # noqa: E402
from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: E402
from nerfstudio.field_components.neurad_encoding import ActorSettings, NeuRADHashEncodingConfig, StaticSettings  # noqa: E402
from nerfstudio.fields.neurad_field import NeuRADField, NeuRADFieldConfig  # noqa: E402

M = 3


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "field_sdf.npz"))
    grid = NeuRADHashEncodingConfig(static=StaticSettings(log2_hashmap_size=11), require_actor_grad=True,
                                    actor=ActorSettings(flip_prob=0.25))
    fld = NeuRADField(NeuRADFieldConfig(grid=grid, use_sdf=True, num_multisamples=M), actors=no_actors(), static_scale=100.0,
                      implementation="torch").eval()
    fld.hashgrid.static_grid.hash_table.data = T(synth.hash_table(8 * 2**11, 4, seed=51, scale=0.5))
    for k, l in enumerate(fld.mlp_geo.layers):
        set_linear(l, 200 + 10 * k)
    for k, l in enumerate(fld.mlp_feature.layers):
        set_linear(l, 300 + 10 * k)
    R, S = g["starts"].shape
    fr = Frustums(origins=T(g["o"])[:, None].expand(R, S, 3), directions=T(g["d"])[:, None].expand(R, S, 3),
                  starts=T(g["starts"])[..., None], ends=T(g["ends"])[..., None],
                  pixel_area=T(g["area"])[:, None, None].expand(R, S, 1))
    rs = RaySamples(frustums=fr, times=torch.zeros(R, S, 1))
    gauss = rs.frustums.get_fast_isotropic_gaussian(M)
    with torch.no_grad():
        enc, _ = fld.hashgrid(gauss, rs.times, None)
    out = fld(rs)
    gf, ga = T(g["g_feature"]), T(g["g_head"])[..., None]
    ((out[FieldHeadNames.FEATURE] * gf).sum() + (out[FieldHeadNames.ALPHA] * ga).sum()).backward()
    tg = fld.hashgrid.static_grid.hash_table.grad
    nz = tg.abs().sum(-1) > 0
    assert not torch.equal(out[FieldHeadNames.ALPHA][..., 0].detach(), T(g["alpha"])), "multisampling changed nothing?"
    save("field_multisample", num_multisamples=np.int32(M), gmean=gauss.mean, gstd=gauss.std[..., 0], enc=enc,
         feature=out[FieldHeadNames.FEATURE], sdf=out[FieldHeadNames.SDF][..., 0], alpha=out[FieldHeadNames.ALPHA][..., 0],
         tg_idx=nz.nonzero()[:, 0], tg_val=tg[nz], geo_dw0=fld.mlp_geo.layers[0].weight.grad)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
