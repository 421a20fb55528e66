"""FieldHeadNames used on the hot path (mirror of nerfstudio/field_components/field_heads.py:29-46)."""
from enum import Enum


class FieldHeadNames(Enum):
    DENSITY = "density"
    NORMALS = "normals"
    SDF = "sdf"
    ALPHA = "alpha"
    FEATURE = "feature"
