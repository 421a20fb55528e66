"""``import tinycudann`` -> HIP-backed modules with the call forms neurad-studio uses (encodings.py:134-137,370-373,
780-783; mlp.py:109-113,251-268).  Numerics follow the reference's torch branch (SURVEY §8a-H1'); run the reference
with ``use_4d_hashgrid=False`` (3-D per-actor grids), as its own CPU launch config does (.vscode/launch.json:89-91)."""
from neurad_studio_amd.shims.tinycudann import Encoding, Network, NetworkWithInputEncoding  # noqa: F401

__all__ = ["Encoding", "Network", "NetworkWithInputEncoding"]
