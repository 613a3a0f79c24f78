"""``model.e2fgvi_hq`` drop-in: MI355X InpaintGenerator for arbitrary resolutions (H % 60 == 0, W % 108 == 0).

Inference only: the reference module also defines ``Discriminator`` and ``spectral_norm`` for
core/trainer.py; they are exported here as stubs that explain themselves instead of failing with AttributeError."""
from e2fgvi_amd.generator import InpaintGeneratorHQ as InpaintGenerator  # noqa: F401
from e2fgvi_amd.generator import Discriminator, spectral_norm  # noqa: F401
