import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def tiling_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "tiling_golden.npz"))


@pytest.fixture(scope="session")
def resunet_aniso_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_aniso_golden.npz"))


@pytest.fixture(scope="session")
def tta_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "tta_golden.npz"))


@pytest.fixture(scope="session")
def head_acts_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "head_acts_golden.npz"))


@pytest.fixture(scope="session")
def harness_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "harness_golden.npz"))


@pytest.fixture(scope="session")
def harness_tail_golden():
    """process_test_sample past the blended prediction (reflect-to-complete-shape crop, class arg-max) from the reference itself:
    tests/golden/make_golden.py harness_tail."""
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "harness_tail_golden.npz"))


@pytest.fixture(scope="session")
def tta_spec_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "tta_spec_golden.npz"))


@pytest.fixture(scope="session")
def tta_ensemble_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "tta_ensemble_golden.npz"))


@pytest.fixture(scope="session")
def prepost_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "prepost_golden.npz"))


@pytest.fixture(scope="session")
def tiling2d_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "tiling2d_golden.npz"))


@pytest.fixture(scope="session")
def resunet_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_golden.npz"))


@pytest.fixture(scope="session")
def unet_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "unet_golden.npz"))


@pytest.fixture(scope="session")
def resunet_dropout_golden():
    """Reference ResUNet in training mode with drop_values > 0, the masks torch drew captured (make_golden.py resunet_dropout)."""
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_dropout_golden.npz"))


@pytest.fixture(scope="session")
def resunet_activations_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_activations_golden.npz"))


@pytest.fixture(scope="session")
def resunet_class_head_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_class_head_golden.npz"))


@pytest.fixture(scope="session")
def resunet_variants_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_variants_golden.npz"))


@pytest.fixture(scope="session")
def chunked_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "chunked_golden.npz"))


@pytest.fixture(scope="session")
def rcan_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "rcan_golden.npz"))


@pytest.fixture(scope="session")
def resunetpp_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunetpp_golden.npz"))


@pytest.fixture(scope="session")
def resunet_sr_golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resunet_sr_golden.npz"))
