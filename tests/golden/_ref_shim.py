"""Import shim for the read-only reference at /root/reference (build container only).

The reference package cannot be imported normally here (missing yacs/pooch/...), so the
hot-path modules are imported by registering empty namespace packages whose __path__
points into the reference tree and stubbing third-party names that are imported at module
top level but never called on the hot path.  Nothing from the
reference is copied; this file only makes `import biapy.models.resunet` work *in this
container* so golden vectors can be generated.  It is never used on the GPU box.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("BIAPY_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "biapy", "models"))


def install():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    for name, rel in [
        ("biapy", "biapy"),
        ("biapy.models", "biapy/models"),
        ("biapy.data", "biapy/data"),
        ("biapy.utils", "biapy/utils"),
        ("biapy.engine", "biapy/engine"),
    ]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, rel)]
            sys.modules[name] = m

    import torch.nn as nn

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Identity(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    try:
        import torchvision  # noqa: F401
    except Exception:
        stub("torchvision")
        stub("torchvision.ops")
        stub("torchvision.ops.stochastic_depth", StochasticDepth=_Identity)
        stub("torchvision.ops.misc", Permute=_Identity)
        tv = sys.modules["torchvision"]
        tv.transforms = stub("torchvision.transforms")                       # biapy/engine/metrics.py:21-22 (perceptual losses only)
        stub("torchvision.models", vgg16=None, VGG16_Weights=None)
    for mod, attrs in [
        ("h5py", dict(File=object, Dataset=object, Group=object)),
        ("zarr", dict(Array=object, Group=object)),
        ("tensorboardX", dict(SummaryWriter=object)),
        ("yacs", {}),
        ("yacs.config", dict(CfgNode=type("CfgNode", (dict,), {}))),
        ("torchmetrics", dict(JaccardIndex=object)),                           # biapy/engine/metrics.py:16-18: metric classes the loss
        ("torchmetrics.image", dict(StructuralSimilarityIndexMeasure=object)),  # classes pinned here never touch
        ("pytorch_msssim", dict(SSIM=object)),
    ]:
        try:
            importlib.import_module(mod)
        except Exception:
            stub(mod, **attrs)


def load(modname):
    install()
    return importlib.import_module(modname)


class _Soft(types.ModuleType):
    """Stand-in for a third-party module that a reference file imports at its top but never calls on the pinned path: every attribute
    is a callable that also works as a decorator (``@numba.njit(...)``)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)

        def deco(*a, **kw):
            if len(a) == 1 and callable(a[0]) and not kw:
                return a[0]
            return lambda f: f
        return deco


def load_post_processing():
    """biapy/data/post_processing/post_processing.py (ensemble_predictions, _pad_for_orientations, _reduce_orientations).  Its top-level
    imports of cv2, fill_voids, scikit-image, numba, ... are not installed here and are not touched by the TTA routine: soft stand-ins."""
    install()
    for n in ["cv2", "fill_voids", "skimage", "skimage.morphology", "skimage.segmentation", "skimage.filters", "skimage.measure", "skimage.exposure",
              "skimage.feature", "skimage.io", "skimage.transform", "skimage.util", "skimage.draw", "skimage.color", "imagecodecs", "tifffile",
              "nibabel", "numba", "edt", "pooch", "xarray", "bioimageio", "bioimageio.core", "bioimageio.spec", "PIL", "PIL.Image",
              "PIL.ImageEnhance", "imageio", "matplotlib", "matplotlib.pyplot", "matplotlib.transforms", "pydot", "torchinfo"]:
        try:
            importlib.import_module(n)
        except Exception:
            m = _Soft(n)
            m.__path__ = []
            sys.modules[n] = m
    for pk in ["biapy.data.post_processing", "biapy.data.generators"]:
        if pk not in sys.modules:
            m = types.ModuleType(pk)
            m.__path__ = [os.path.join(REF, *pk.split("."))]
            sys.modules[pk] = m
    return importlib.import_module("biapy.data.post_processing.post_processing")
