"""MI355X-native SMPL-X fitting engine (hot path of smplifyx/fit_single_frame.py).

Import as ``smplifyx_amd`` (see ``smplifyx_amd/__init__.py`` at the repo root).

Layout
------
csrc/            HIP kernels (gfx950) + the C-ABI shared library ``libsfx.so``
_capi.py         ctypes binding of include/sfx.h
layout.py        canonical per-frame parameter block / optimiser variable lists
synthetic.py     seeded SMPL-X-shaped model + synthetic frames (numpy only)
engine.py        batched fitting engine (the MI355X-native entry point)
smplx.py camera.py prior.py fitting.py optimizers/ fit_single_frame.py utils.py
cmd_parser.py    drop-in mirror of the reference's call surface for this path
dist.py          one-process-per-GPU frame sharding + RCCL gather
"""
__version__ = "0.1.0"
