"""The C-ABI library loads on a machine without a GPU and exports every symbol include/ti_hip.h declares
(no compute calls here).  Also: argument validation that returns before any device work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def lib():
    ti = pkg()
    ti.build.build()                    # hipcc cross-compiles gfx950 without a GPU
    return ti._lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ti_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ti_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    ti = pkg()
    names = declared_symbols()
    assert sorted(ti._lib.ABI_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ti_version() == 5          # TI_ABI_VERSION 5: v4 (TI_TEMPLATE_PAIR) + ti_painn_debug_poison


def test_rollout_rows_matches_oracle_definition(lib):
    from oracle import oracle
    for n_step in (1, 2, 5, 11, 1001):
        for k in (0, 1, 3, 4, 10, 2000):
            assert lib.ti_rollout_rows(n_step, k) == oracle.rollout_rows(n_step, k)


def test_no_cpu_fallback_without_gpu(lib):
    """Product path must fail loudly when no device is usable -- it never routes through the oracle."""
    ti = pkg()
    if lib.ti_device_count() > 0:
        pytest.skip("a GPU is present")
    syn, W = ti.synthetic, ti.weights
    flat = W.flatten_state_dict(syn.painn_state_dict(0, 32, 1), W.painn_param_spec(0, 32, 1))
    with pytest.raises(ti._lib.TiError) as ei:
        ti.engine.PainnEngine(0, 32, 1, 3, *syn.fully_connected_template(3), np.arange(3), flat)
    assert "no HIP device" in str(ei.value) or "HIP" in str(ei.value)
    with pytest.raises(ti._lib.TiError):
        ti.engine.AdwEngine(32, 2, W.flatten_state_dict(syn.adw_state_dict(32, 2), W.adw_param_spec(32, 2), dtype=np.float64))


def test_create_argument_validation(lib):
    """Checks that run before the device is touched return TI_E_ARG / TI_E_UNSUPPORTED with a message."""
    ti = pkg()
    d = ti._lib.PainnDesc(0, 48, 2, 25, 3, 6, 100.0, 10.0, 10.0, 650.0, 700.0, 0)    # F = 48 is not supported
    w = np.zeros(4, np.float32)
    z = np.zeros(6, np.int32)
    h = lib.ti_painn_create(C.byref(d), ti._lib.fptr(w), 4, ti._lib.iptr(z), ti._lib.iptr(z), ti._lib.iptr(z), ti._lib.iptr(z), 0)
    assert not h and "n_features" in lib.ti_last_error().decode()
    d.n_features = 32
    d.n_atoms = 40
    h = lib.ti_painn_create(C.byref(d), ti._lib.fptr(w), 4, ti._lib.iptr(z), ti._lib.iptr(z), ti._lib.iptr(z), ti._lib.iptr(z), 0)
    assert not h and "n_atoms" in lib.ti_last_error().decode()
    assert lib.ti_painn_drift(None, None, C.c_float(0), None, 1, None, 0) == ti._lib.TI_E_ARG
    assert lib.ti_adw_rollout(None, None, None, None, None, 1, None, None) == ti._lib.TI_E_ARG


def test_graft_entry_build_runs():
    """The driver's "does it build" check: build() compiles (or finds up to date) every HIP object, loads the library, checks the
    ABI version and builds the oracle."""
    import __graft_entry__ as entry
    entry.build()
