"""CPU oracle for the tqdne 1-D EDM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``tqdne_amd`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

The oracle is a from-scratch, functional (state-dict driven) restatement in
PyTorch fp32 on the CPU of the reference's algorithm (the reference is itself
pure PyTorch, so ATen CPU kernels are the reference arithmetic):

* ``oracle.unet``        <- tqdne/unet.py, tqdne/blocks.py, tqdne/nn.py
* ``oracle.edm``         <- tqdne/edm.py
* ``oracle.consistency`` <- tqdne/consistency_model.py
* ``oracle.autoencoder`` <- tqdne/autoencoder.py (+ Encoder/Decoder of blocks.py)

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported in the build
container by ``tools/make_goldens.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py``).  ``tqdne/diffusion.py`` (DDPM) cannot be
imported even in the reference's own environment (needs ``diffusers``, absent
from its lockfile): parity unpinned for that file, nothing is restated for it.
"""
