"""``model.e2fgvi_hq`` drop-in: MI355X InpaintGenerator for arbitrary resolutions."""
from e2fgvi_amd.generator import InpaintGeneratorHQ as InpaintGenerator  # noqa: F401
