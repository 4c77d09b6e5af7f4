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


class _StubCalled(RuntimeError):
    pass


class _Deco:
    """What calling a stand-in with arguments returns: usable ONLY as a decorator (``@numba.njit(cache=True)`` -> ``_Deco`` -> applied to the
    function).  Anything else a reference path could do with the result of a real third-party call - index it, iterate it, do arithmetic on it,
    call it with data - raises, so a fixture can never be generated through a stubbed function by accident."""

    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **kw):
        if len(a) == 1 and not kw and (isinstance(a[0], (types.FunctionType, type)) or (callable(a[0]) and hasattr(a[0], "__name__"))):
            return a[0]
        raise _StubCalled(f"{self._name}: a stubbed third-party function was CALLED with data on a path that is being pinned - install the package or "
                          "keep this path out of the fixture")

    def _no(self, *a, **kw):
        raise _StubCalled(f"{self._name}: the result of a stubbed third-party call was used as data")

    __getitem__ = __iter__ = __len__ = __add__ = __radd__ = __mul__ = __rmul__ = __sub__ = __rsub__ = __truediv__ = __array__ = __bool__ = __float__ = __int__ = _no


class _Soft(types.ModuleType):
    """Stand-in for a third-party module that a reference file imports at its top but never calls on the pinned path: every attribute
    is a class (usable in annotations ``A | B`` and as a base class) whose call works ONLY as a decorator application (``@numba.njit`` on a
    function returns the function; ``@numba.njit(...)`` returns a ``_Deco`` that does the same); any other use of the result raises."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        modname = self.__name__

        class _SoftObj:
            def __new__(cls, *a, **kw):
                if len(a) == 1 and not kw and isinstance(a[0], (types.FunctionType, type)):
                    return a[0]
                return _Deco(f"{modname}.{k}")
        _SoftObj.__name__ = k
        setattr(self, k, _SoftObj)
        return _SoftObj


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


# ---- the whole reference package, importable ------------------------------------------------------------------------------------
_SOFT_PREFIXES = ("cv2", "fill_voids", "skimage", "imagecodecs", "tifffile", "nibabel", "numba", "edt", "pooch", "xarray", "bioimageio", "imageio",
                  "pydot", "torchinfo", "yacs", "tensorboardX", "timm", "fastremap", "cc3d", "h5py", "zarr", "numcodecs", "torchvision", "xxhash",
                  "torchmetrics", "pytorch_msssim", "marshmallow", "monai", "ptflops", "thop", "fvcore")


def load_full_reference():
    """``import biapy`` for real (build container only): a meta-path finder placed AFTER the normal ones hands out ``_Soft`` stand-ins for
    the third-party packages of the list above that are not installed, so every reference module imports; what the pinned paths actually
    execute (NumPy, PyTorch, SciPy, the reference's own code) is the real thing.  Used to drive ``Base_Workflow.process_test_sample`` -
    the sliding-window harness itself - for tests/golden/harness_golden.npz."""
    import importlib.abc
    import importlib.machinery

    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    for k in [k for k in sys.modules if k == "biapy" or k.startswith("biapy.")]:     # namespace stand-ins of install(): out
        del sys.modules[k]

    class _Loader(importlib.abc.Loader):
        def create_module(self, spec):
            m = _Soft(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass

    class _Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in _SOFT_PREFIXES:
                return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
            return None

    if not any(type(f).__name__ == "_Finder" for f in sys.meta_path):
        sys.meta_path.append(_Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return importlib.import_module("biapy.engine.base_workflow")
