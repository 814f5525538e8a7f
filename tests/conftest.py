import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch bundles its own libamdhip64 with the SONAME of /opt/rocm's; whichever is loaded first
# serves both torch and libvrhip.  Load torch's first, whatever test file is collected first, so
# that tests which hand torch tensors to the renderer see one HIP runtime.
import torch  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
