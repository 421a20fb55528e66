"""Checkpoint compatibility: the HIP-backed modules expose the reference's state_dict names."""
import torch


def test_state_dict_names_match_reference_layout():
    from neurad_studio_amd.models.neurad import NeuRADHotPath, NeuRADHotPathConfig

    c = NeuRADHotPathConfig()
    for g in (c.field.grid, c.sampling.proposal_field_1.grid, c.sampling.proposal_field_2.grid):
        g.static.log2_hashmap_size = 8
    m = NeuRADHotPath(c, static_scale=100.0, num_sensors=7, duration=8.0)
    sd = m.state_dict()
    expect = {
        "field.hashgrid.static_grid.hash_table": (8 * 2**8, 4), "field.hashgrid.static_grid.scalings": (8,),
        "field.mlp_geo.layers.0.weight": (32, 32), "field.mlp_geo.layers.1.weight": (33, 32),
        "field.mlp_geo.layers.1.bias": (33,), "field.mlp_feature.layers.0.weight": (32, 48),
        "field.mlp_feature.layers.2.weight": (32, 32), "field.sdf_to_density.beta": (1,),
        "field.sdf_to_density.beta_min": (), "proposal_fields.0.hashgrid.static_grid.hash_table": (6 * 2**8, 1),
        "proposal_fields.1.density_decoder.weight": (1, 6), "appearance_embedding.weight": (7 * 8, 16),
    }
    for k, shape in expect.items():
        assert k in sd and tuple(sd[k].shape) == shape, k
    groups = m.get_param_groups()
    assert len(groups["hashgrids"]) == 3 and any(p is m.field.sdf_to_density.beta for p in groups["fields"])
    assert torch.equal(sd["field.hashgrid.static_grid.scalings"],
                       torch.tensor([32., 70., 156., 344., 760., 1680., 3709., 8191.]))
