"""``model.e2fgvi`` drop-in: MI355X InpaintGenerator for 432x240 clips."""
from e2fgvi_amd.generator import InpaintGenerator  # noqa: F401
