"""Import the read-only reference (neurad-studio) torch path in THIS container only.

TEST INFRASTRUCTURE -- never imported by the product path.  Used by
``oracle/make_golden.py`` to generate the fixtures under ``tests/golden/`` and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, e.g.
on the GPU box).

The reference needs a handful of pure-import dependencies that are not installed
here: they are stubbed.  ``tinycudann`` must stay absent so that
``nerfstudio.utils.external.TCNN_EXISTS`` is False and every component takes its
``implementation="torch"`` branch -- that branch is the parity target.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types
from unittest import mock

_HERE = os.path.dirname(os.path.abspath(__file__))
_SOURCE_ROOT = os.environ.get("NEURAD_REFERENCE_ROOT", "/root/reference")
# the reference byte-compiled by oracle/make_ref.py (sourceless .pyc, gitignored, ships with the gpurun lease): what the
# GPU box has in place of /root/reference
_BYTECODE_ROOT = os.path.join(_HERE, "_ref")


def _pick_root() -> str:
    if os.path.isdir(os.path.join(_SOURCE_ROOT, "nerfstudio")):
        return _SOURCE_ROOT
    if os.path.exists(os.path.join(_BYTECODE_ROOT, "nerfstudio", "__init__.pyc")):
        return _BYTECODE_ROOT
    return _SOURCE_ROOT


REFERENCE_ROOT = _pick_root()


def reference_available() -> bool:
    return (os.path.isdir(os.path.join(REFERENCE_ROOT, "nerfstudio"))
            and any(os.path.exists(os.path.join(REFERENCE_ROOT, "nerfstudio", "__init__" + e)) for e in (".py", ".pyc")))


def reference_kind() -> str:
    """"source" (the build container's /root/reference) or "bytecode" (oracle/_ref on the GPU box)"""
    return "bytecode" if REFERENCE_ROOT == _BYTECODE_ROOT else "source"


class _Anno:
    """jaxtyping-style annotation stub: Float[Tensor, "..."] -> object."""

    def __class_getitem__(cls, item):
        return object


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Magic(types.ModuleType):
    """Module whose every attribute is a MagicMock (for import-only deps)."""

    def __init__(self, name):
        super().__init__(name)
        self.__spec__ = importlib.machinery.ModuleSpec(name, None)
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        full = f"{self.__name__}.{item}"
        if full in sys.modules:  # `import viser.infra; viser.infra.Message`: the submodule, not a fresh mock
            v = sys.modules[full]
        elif item[:1].isupper() and not item.isupper():
            # CamelCase -> a real (empty) class: the reference subclasses such names (dataclass configs, nn.Modules of
            # absent packages), which a MagicMock instance cannot stand in for
            v = type(item, (_StubClass,), {"__module__": self.__name__})
        else:
            v = mock.MagicMock(name=f"{self.__name__}.{item}")
        setattr(self, item, v)
        return v


class _StubMeta(type):
    def __getattr__(cls, item):  # class-level constants of absent packages (enum members, ...)
        if item.startswith("__"):
            raise AttributeError(item)
        v = mock.MagicMock(name=f"{cls.__name__}.{item}")
        setattr(cls, item, v)
        return v


class _StubClass(metaclass=_StubMeta):
    """base of every stubbed class: accepts any constructor call, any attribute is a MagicMock"""

    def __init__(self, *args, **kwargs):
        pass

    def __init_subclass__(cls, **kwargs):
        pass

    def __class_getitem__(cls, item):
        return cls

    def __call__(self, *args, **kwargs):  # metric objects of absent packages (torchmetrics' PSNR in get_metrics_dict)
        return mock.MagicMock(name=f"{type(self).__name__}()")

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        v = mock.MagicMock(name=f"{type(self).__name__}.{item}")
        object.__setattr__(self, item, v)
        return v


class _StubFinder:
    """meta-path finder: any submodule of a stubbed top-level package (av2.utils.io, nuscenes.utils.data_classes, ...)
    resolves to another all-MagicMock module, so deep import chains of dataset devkits do not have to be listed."""

    def __init__(self):
        self.tops = set()

    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in self.tops and "." in fullname:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return _Magic(spec.name)

    def exec_module(self, module):
        pass


_FINDER = _StubFinder()


def install(allow_tcnn: bool = False) -> None:
    """Make ``import nerfstudio...`` work from the read-only reference tree (``allow_tcnn``: the shim test wants the
    ``tinycudann`` import-name package PRESENT; everything else needs it absent so that the torch branches run)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    assert allow_tcnn or "tinycudann" not in sys.modules, "tinycudann must stay absent for the torch oracle"
    if "jaxtyping" not in sys.modules:
        _stub("jaxtyping", **{k: _Anno for k in ("Float", "Int", "Shaped", "Bool", "UInt8", "Num", "Integer")})
    for name in (
        "viser", "viser.transforms", "viser.theme", "viser.infra",
        "torch.utils.tensorboard", "cv2", "nerfacc", "torchmetrics", "torchmetrics.functional",
        "torchmetrics.image", "torchmetrics.image.lpip", "torchvision", "torchvision.models",
        "torchvision.transforms", "tyro", "tyro.conf", "tyro.extras", "wandb", "comet_ml", "mediapy", "open3d",
        "pyquaternion", "trimesh", "plotly", "plotly.graph_objects", "imageio", "gsplat", "gsplat.strategy",
        "pytorch_msssim", "lpips", "pymeshlab", "xatlas", "splines", "splines.quaternion", "msgpack_numpy",
        "nuscenes", "av2", "zod", "pandaset", "pathos", "appdirs", "gdown", "ninja", "h5py", "rawpy", "newrawpy",
        "pyarrow", "tensorboard", "timm", "kornia", "opencv", "torchtyping", "typeguard", "awscli", "scikit-image",
        "skimage", "skimage.metrics", "cryptography", "nodeenv", "protobuf", "ipywidgets", "jupyterlab", "matplotlib",
        "matplotlib.pyplot", "matplotlib.cm", "matplotlib.colors", "PIL.ImageFont",
    ):
        if name in sys.modules:
            continue
        try:
            if "." not in name:
                __import__(name)
                continue
        except Exception:
            pass
        sys.modules[name] = _Magic(name)
        _FINDER.tops.add(name.split(".")[0])
    if _FINDER not in sys.meta_path:
        sys.meta_path.append(_FINDER)
    if reference_kind() == "bytecode":
        # TorchScript needs source text; the bytecode build has none.  The two scripted helpers of the reference
        # (cameras/camera_utils.py:943,1032, fisheye projection) then run as the eager functions they decorate --
        # same arithmetic.  Scoped to the reference's import: restored right after.
        import torch

        _script = torch.jit.script
        torch.jit.script = lambda fn=None, *a, **k: fn
        try:
            if REFERENCE_ROOT not in sys.path:
                sys.path.insert(0, REFERENCE_ROOT)
            import nerfstudio.cameras.camera_utils  # noqa: F401
        finally:
            torch.jit.script = _script
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
