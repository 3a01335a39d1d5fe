import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

# torch ships its own copy of the HIP runtime. Imported before libptamd.so is loaded, that copy serves both and
# the process holds one runtime; the other way round it holds two, and torch then finds no GPU. Tests that hand
# torch-owned device buffers to the C-ABI need the first order, whichever test file runs first.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
