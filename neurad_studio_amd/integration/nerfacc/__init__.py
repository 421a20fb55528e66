"""``import nerfacc`` -> the dense-mode functions neurad-studio calls (models/neurad.py:716-723,734;
model_components/renderers.py:88,130,133,345,404,407,455,486; ray_samplers.py:527-540) on the HIP kernels."""
from neurad_studio_amd.shims.nerfacc import (OccGridEstimator, accumulate_along_rays,  # noqa: F401
                                             render_weight_from_alpha, render_weight_from_density)

__version__ = "0.5.2"  # the version neurad-studio pins (pyproject.toml:36)
__all__ = ["OccGridEstimator", "accumulate_along_rays", "render_weight_from_alpha", "render_weight_from_density"]
