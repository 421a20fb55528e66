"""FieldHeadNames: the reference's own enum whenever nerfstudio is importable (so that the dictionaries the fields
return are keyed exactly like ``outputs[FieldHeadNames.ALPHA]`` in nerfstudio/models/neurad.py:712), else a copy of
the members the hot path uses (nerfstudio/field_components/field_heads.py:29-46)."""
from enum import Enum

try:
    from nerfstudio.field_components.field_heads import FieldHeadNames  # noqa: F401
except Exception:  # nerfstudio not installed (e.g. the GPU test box): same member names / values

    class FieldHeadNames(Enum):
        RGB = "rgb"
        DENSITY = "density"
        NORMALS = "normals"
        SDF = "sdf"
        ALPHA = "alpha"
        GRADIENT = "gradient"
        FEATURE = "feature"
