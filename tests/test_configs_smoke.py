"""Every BASELINE configuration's train / sample / checkpoint path at its real shapes (small batch): autoencoder training on
3 x 16384, iCT step on the paper UNet, latent EDM training with the frozen encoder inside, latent deterministic and stochastic
sampling + decode, checkpoint round trip with EMA.  Shape-dependent limits of the dedicated kernels (e.g. the 16-channel stem of
the latent UNet) only show up here; numerics are pinned by the parity tests (tests/test_bench_config_parity.py holds these
configurations against the oracle); this walk asserts that every value it produces is finite, every parameter of the iCT step got a
gradient, shapes, and that an EMA checkpoint loads back bit for bit."""

import os
import runpy

import pytest

pytestmark = pytest.mark.gpu


def test_all_baseline_configs_run_end_to_end():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "tools", "exercise_configs.py"), run_name="__main__")
