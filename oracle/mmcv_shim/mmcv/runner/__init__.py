"""mmcv.runner stand-in (test infrastructure).  flow_comp.py:72 downloads SPyNet weights by
URL; there is no network here, so URLs are skipped and only local files are loaded."""
import os
import torch


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    if isinstance(filename, str) and os.path.isfile(filename):
        sd = torch.load(filename, map_location=map_location)
        sd = sd.get("state_dict", sd)
        model.load_state_dict(sd, strict=strict)
        return sd
    return None
