"""Minimal stand-in for mmcv-full 1.4.8 (not installable here: no network, no wheel).

TEST INFRASTRUCTURE ONLY.  It exists so that the reference's own Python
(/root/reference/model/*.py, imported read-only, never copied) can be executed on CPU
inside this container to pin the oracle and to generate golden vectors.  It provides
exactly the symbols the reference imports (feat_prop.py:7-8, flow_comp.py:7-8):
ConvModule, constant_init, load_checkpoint, ModulatedDeformConv2d,
modulated_deform_conv2d (CPU restatement of mmcv's published kernel semantics,
see oracle/dcn.py -- parity for that op is anchored on the reference's call site
feat_prop.py:55-58 and on cross-checks against torch conv/grid_sample, because the
mmcv source is not under /root/reference).
"""
__version__ = "1.4.8-shim"
