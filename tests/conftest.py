import importlib.util
import os
import sys

import pytest

# Some tests hand torch device buffers to the C ABI (the gather payload).  torch's wheel bundles its own HIP runtime: it has to be
# the first one loaded in the process -- the product library then binds to it -- or a later torch.cuda initialisation finds "no HIP
# GPUs" beside the runtime the product library already brought up (seen when a test selection starts with a torch-free test).
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001 -- the CPU-only suite does not need it
    torch = None

# The library reads its diagnostic environment switches once per handle (ilqr_hip_create); many tests switch kernel families around
# calls on ONE handle (the `env` blocks): this opt-in restores the re-read at the top of every C-ABI call for the test process.
os.environ.setdefault("ILQR_ENV_PER_CALL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The product package directory is `mpc-ilqr-mujoco_amd/` (hyphenated, as the layout contract
    names it); import it under the module name mpc_ilqr_mujoco_amd."""
    name = "mpc_ilqr_mujoco_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "mpc-ilqr-mujoco_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()
