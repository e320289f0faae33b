"""enarf-gan_amd: MI355X-native (gfx950) per-ray renderer of ENARF-GAN.

Scope is the hot path of SURVEY.md §8 only: tri-plane sampling, SMPL-bone inverse transform,
styled density/colour MLP and alpha compositing, as hand-written HIP kernels behind a C-ABI
shared library (`csrc/`, `include/enarf_hip.h`), with a thin Python host layer that mirrors the
reference's operator / model interface for this path.

Imported as `enarf_gan_amd` (see the alias stub next to this directory).
"""
__version__ = "0.1.0"
