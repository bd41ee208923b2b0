import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "diag: runs against the DIAGNOSTIC build of the library (libdsdtm_amd_diag.so: fault injection, "
                                       "A/B switches); everything else runs against the release library")
    # torch (device memory / streams for the *_device entry points) bundles its own HIP runtime.
    # On a GPU box let it initialise before anything loads libdsdtm_amd.so (even at collection
    # time), so that both bind to the same libamdhip64 inside this process.
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The suite needs the in-tree HIP library and the CPU oracle; build them when missing or stale
    (no-ops otherwise). hipcc cross-compiles gfx950 without a GPU."""
    from dsdtm_amd.csrc import build as hip_build
    hip_build.build_all(verbose=False)      # the release library (the product) and the diagnostic one (tests marked `diag`)
    from tests import oracle_lib
    oracle_lib.load()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx():
    from dsdtm_amd import capi
    return capi.default_context(0)


@pytest.fixture(scope="session")
def gpu_ctx_diag():
    """A context of the DIAGNOSTIC library (dsdtm_debug_* entries, capi.debug_options): tests marked `diag`."""
    from dsdtm_amd import capi
    return capi.default_context(0, diag=True)


@pytest.fixture(params=["release", "diag"])
def gpu_ctx_each(request):
    """Tests whose body is a parity test with ONE step that needs a diagnostic switch: run once against the release library
    (without that step) and once against the diagnostic one (`ctx.diag` says which)."""
    from dsdtm_amd import capi
    return capi.default_context(0, diag=request.param == "diag")


_SCENES = {}


def cached_scene(**kw):
    """Synthetic scenes are deterministic in their arguments; cache them per session."""
    from dsdtm_amd import synth
    key = tuple(sorted((k, (tuple(v) if hasattr(v, "__len__") else v)) for k, v in kw.items()))
    if key not in _SCENES:
        _SCENES[key] = synth.make_scene(**kw)
    return _SCENES[key]
