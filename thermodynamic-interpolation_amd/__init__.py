"""MI355X-native sampler hot path for thermodynamic-interpolation.

The directory name is fixed by the build contract and is not a Python identifier; import it with
``importlib.import_module("thermodynamic-interpolation_amd")`` (tests/conftest.py and __graft_entry__.py do).
Importing the package never loads the HIP library; the first engine does, and fails loudly if it is missing.
"""
from . import weights, synthetic, build, _lib, engine, thermo, distributed, data, drivers  # noqa: F401
