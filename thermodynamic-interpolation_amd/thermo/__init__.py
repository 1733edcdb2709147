"""Host-side mirror of the reference `thermo/` sampler + model API (names, arguments, return shapes) over libti_hip.so."""
from . import adw, ambient, latent  # noqa: F401
