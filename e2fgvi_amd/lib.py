"""ctypes binding of libe2fgvi_hip.so -- the C ABI declared in include/e2fgvi_hip.h.

There is deliberately NO fallback: if the HIP library is missing the import of the product path
fails loudly (build it with ``python -m e2fgvi_amd.build`` or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E2FGVI_LIB") or os.path.join(_HERE, "csrc", "libe2fgvi_hip.so")    # E2FGVI_LIB: A/B builds

ABI_VERSION = 8        # the struct layouts / symbols below; csrc/error.hip e2fgvi_abi_version() must agree (checked in load())
MAX_SRC = 4
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_DCNPOST = 0, 1, 2, 3, 4
DT_F32, DT_BF16 = 0, 1

_fp = C.c_void_p   # device pointers travel as plain addresses


class ConvDesc(C.Structure):
    _fields_ = [
        ("src", _fp * MAX_SRC), ("src_ld", C.c_int32 * MAX_SRC), ("src_coff", C.c_int32 * MAX_SRC),
        ("src_cpg", C.c_int32 * MAX_SRC), ("nsrc", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("groups", C.c_int32), ("Cout", C.c_int32), ("bk", C.c_int32),
        ("wpacked", _fp), ("bias", _fp), ("residual", _fp), ("res_ld", C.c_int32), ("res_coff", C.c_int32),
        ("dst", _fp), ("dst_ld", C.c_int32), ("dst_coff", C.c_int32), ("dst_nchw", C.c_int32),
        ("act", C.c_int32), ("slope", C.c_float), ("tile", C.c_int32),
    ]


class ConvXDesc(C.Structure):
    _fields_ = [
        ("src", _fp * MAX_SRC), ("src_ld", C.c_int32 * MAX_SRC), ("src_coff", C.c_int32 * MAX_SRC),
        ("src_cpg", C.c_int32 * MAX_SRC), ("nsrc", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("groups", C.c_int32), ("Cout", C.c_int32),
        ("wpacked", _fp), ("bias", _fp), ("residual", _fp), ("res_ld", C.c_int32), ("res_coff", C.c_int32),
        ("res_dtype", C.c_int32),
        ("dst", _fp), ("dst_ld", C.c_int32), ("dst_coff", C.c_int32), ("dst_dtype", C.c_int32),
        ("dst2", _fp), ("dst2_ld", C.c_int32), ("dst2_coff", C.c_int32),
        ("act", C.c_int32), ("slope", C.c_float), ("tile", C.c_int32), ("dst_nchw", C.c_int32), ("tap_packed", C.c_int32),
        # ABI version 6: explicit output grid / output scatter / broadcast residual (SoftComp in gather form)
        ("out_grid", C.c_int32), ("pad_left", C.c_int32),
        ("out_sy", C.c_int32), ("out_sx", C.c_int32), ("out_py", C.c_int32), ("out_px", C.c_int32),
        ("out_H", C.c_int32), ("out_W", C.c_int32), ("res_bcast", C.c_int32),
        # ABI version 8: output channels from dst2_split_from on as three exact bf16 planes in dst2 (the qkv Linear -> attention_x3)
        ("dst2_split_from", C.c_int32), ("dst2_plane_stride", C.c_int64),
    ]


class MdcnDesc(C.Structure):
    _fields_ = [
        ("src", _fp * 2), ("src_ld", C.c_int32 * 2), ("src_c", C.c_int32 * 2), ("nsrc", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("deform_groups", C.c_int32), ("Cout", C.c_int32),
        ("offset", _fp), ("off_ld", C.c_int32), ("mask", _fp), ("mask_ld", C.c_int32),
        ("flows", _fp), ("max_residue", C.c_float),
        ("wpacked", _fp), ("bias", _fp),
        ("dst", _fp), ("dst_ld", C.c_int32), ("dst_coff", C.c_int32), ("tile", C.c_int32), ("dst_dtype", C.c_int32),
        ("mfma_dtype", C.c_int32), ("src_dtype", C.c_int32), ("src_planar", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/e2fgvi_hip.h declares
_i32, _i64, _f = C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "e2fgvi_last_error": (C.c_char_p, []),
    "e2fgvi_abi_version": (C.c_int, []),
    "e2fgvi_nhwc_to_planar16": (C.c_int, [_fp, _fp, _i64, _i32, _fp]),
    "e2fgvi_conv2d_nhwc": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "e2fgvi_conv2d_nhwc_nopk": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "e2fgvi_packed_conv_weight_size": (_i64, [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32]),
    "e2fgvi_pack_conv_weight": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _fp]),
    "e2fgvi_conv2d_bf16x": (C.c_int, [C.POINTER(ConvXDesc), _fp]),
    "e2fgvi_packed_conv_weight_bf16x_size": (_i64, [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "e2fgvi_pack_conv_weight_bf16x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32), _fp]),
    "e2fgvi_packed_conv_weight_bf16x_taps_size": (_i64, [_i32, _i32, _i32, _i32]),
    "e2fgvi_pack_conv_weight_bf16x_taps": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_packed_conv_weight_f32x_taps_size": (_i64, [_i32, _i32, _i32, _i32]),
    "e2fgvi_pack_conv_weight_f32x_taps": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_conv2d_f32x": (C.c_int, [C.POINTER(ConvXDesc), _fp]),
    "e2fgvi_packed_conv_weight_f32x_size": (_i64, [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "e2fgvi_pack_conv_weight_f32x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32), _fp]),
    "e2fgvi_conv2d_f32x3": (C.c_int, [C.POINTER(ConvXDesc), _fp]),
    "e2fgvi_packed_conv_weight_f32x3_size": (_i64, [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "e2fgvi_pack_conv_weight_f32x3": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32), _fp]),
    "e2fgvi_packed_conv_weight_f32x3_taps_size": (_i64, [_i32, _i32, _i32, _i32]),
    "e2fgvi_pack_conv_weight_f32x3_taps": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_packed_winograd_weight_size": (_i64, [_i32, _i32, _i32, C.POINTER(_i32)]),
    "e2fgvi_pack_winograd_weight": (C.c_int, [_fp, _fp, _i32, _i32, _i32, C.POINTER(_i32), _fp]),
    "e2fgvi_conv3x3_winograd": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "e2fgvi_packed_winograd_weight_x3_size": (_i64, [_i32, _i32, _i32, C.POINTER(_i32)]),
    "e2fgvi_pack_winograd_weight_x3": (C.c_int, [_fp, _fp, _i32, _i32, _i32, C.POINTER(_i32), _fp]),
    "e2fgvi_conv3x3_winograd_x3": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "e2fgvi_packed_winograd4_weight_size": (_i64, [_i32, _i32, _i32, C.POINTER(_i32), _i32]),
    "e2fgvi_pack_winograd4_weight": (C.c_int, [_fp, _fp, _i32, _i32, _i32, C.POINTER(_i32), _i32, _fp]),
    "e2fgvi_conv3x3_winograd4": (C.c_int, [C.POINTER(ConvDesc), _i32, _fp]),
    "e2fgvi_packed_tail_weight_size": (_i64, [_i32, _i32]),
    "e2fgvi_pack_tail_weight": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _fp]),
    "e2fgvi_conv3x3_tail": (C.c_int, [_fp, _i32, _i32, _fp, _fp, _fp, _i32, _i32, _i32, _i32, C.c_float, _fp]),
    "e2fgvi_mdcn_nhwc": (C.c_int, [C.POINTER(MdcnDesc), _fp]),
    "e2fgvi_packed_dcn_weight_size": (_i64, [_i32, _i32, _i32, _i32]),
    "e2fgvi_pack_dcn_weight": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_pack_dcn_weight_bf16": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_pack_dcn_weight_x3": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_focal_attention": (C.c_int, [_fp, _fp, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_split3_kv": (C.c_int, [_fp, _fp, _i64, _fp]),
    "e2fgvi_focal_attention_x3": (C.c_int, [_fp, _fp, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_nchw_to_nhwc": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _f, _f, _fp]),
    "e2fgvi_nhwc_to_nchw": (C.c_int, [_fp, _i32, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_resize_bilinear": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp, _fp, _fp]),
    "e2fgvi_avgpool2_nhwc": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_spynet_level_input": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp]),
    "e2fgvi_prop_cond": (C.c_int, [_fp, _i32, _fp, _i32, _fp, _fp, _i64, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_layernorm": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, _fp]),
    "e2fgvi_window_pool": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_fold": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_unfold_gelu": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_mask_prepare": (C.c_int, [_fp, _i32, _i32, _i32, _fp, _fp, _fp, _i32, _i32, _i32, _fp]),
    "e2fgvi_masked_clip": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _fp, _i32, _i32, _fp]),
    "e2fgvi_composite": (C.c_int, [_fp, _fp, _fp, _i32, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_float_to_u8": (C.c_int, [_fp, _fp, _i64, _fp]),
    "e2fgvi_pred_to_u8": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_psnr_ssim_workspace": (_i64, [_i32, _i32, _i32]),
    "e2fgvi_psnr_ssim": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _fp, _fp, _fp]),
    "e2fgvi_softcomp_fold": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_focal_attention_bf16": (C.c_int, [_fp, _fp, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_focal_attention_bf16_variant": (C.c_int, [C.c_int]),
    "e2fgvi_nchw_to_nhwc_x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _fp]),
    "e2fgvi_resize_bilinear_bf16": (C.c_int, [_fp, _i32, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_prop_cond_x": (C.c_int, [_fp, _i32, _fp, _i32, _fp, _fp, _i64, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_prop_cond_xs": (C.c_int, [_fp, _i32, _fp, _i32, _i32, _fp, _fp, _i64, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_spynet_level_input_x": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp]),
    "e2fgvi_layernorm_x": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i64, _i32, _fp]),
    "e2fgvi_window_pool_x": (C.c_int, [_fp, _i32, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_fold_x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_unfold_gelu_x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_fold_gelu_x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_ffn_unfold_x": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_softcomp_fold_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _fp]),
    "e2fgvi_cast": (C.c_int, [_fp, _i32, _fp, _i32, _i64, _fp]),
}

_lib = None

# Launch tracing (tools/layer_table.py, bench.py's FLOP accounting): while TRACE is a list, every C-ABI call that takes a
# stream is bracketed by two events on the current stream and recorded together with the metadata the operator layer
# announced for it (layer name, shape, algorithmic / issued MACs).  None = off: one attribute test per call.
TRACE = None
NEXT_META = None


X3_PIPE = 6 * 157.3 / 2500.0       # six bf16 MACs per split-operand product, in fp32-MFMA pipe time (bf16 MAC = 157.3 / 2500 of an fp32 MAC)


def useful_macs(kernel, macs, issued):
    """The MACs of `issued` that are not tile / channel padding: what the kernel's ALGORITHM must put on the matrix pipe for the
    layer's direct-convolution `macs` (Winograd F(2x2,3x3): 16 of 36, F(2x4): 24 of 72 ... per output block; split operands:
    six bf16 terms in fp32-pipe equivalents), never more than what was issued."""
    import re
    k = str(kernel)
    f = 1.0
    m = re.match(r"conv_wino4<F\((\d)x4\)", k)
    if m:
        fy = int(m.group(1))
        f = (fy + 2) * 6 / (fy * 4 * 9.0)
    elif k.startswith("conv_wino"):
        f = 16 / 36.0
    if "x3" in k:
        f *= X3_PIPE
    return min(int(issued), int(macs * f))


def annotate(**meta):
    """metadata for the next traced launch (ignored when tracing is off)"""
    global NEXT_META
    if TRACE is not None:
        if "macs" in meta and "issued" in meta and "useful" not in meta:
            meta["useful"] = useful_macs(meta.get("kernel", ""), meta["macs"], meta["issued"])
        NEXT_META = meta


def _traced(name, fn):
    def call(*args):
        global NEXT_META
        if TRACE is None:
            return fn(*args)
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        TRACE.append({"symbol": name, "e0": e0, "e1": e1, "meta": NEXT_META})
        NEXT_META = None
        return rc
    call.__name__ = name
    return call


class HipLibraryMissing(ImportError):
    pass


def load():
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "libe2fgvi_hip.so not found at %s -- the MI355X kernels are mandatory (no CPU / eager "
            "fallback exists).  Build them: python -m e2fgvi_amd.build" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
        if args and args[-1] is _fp and res is C.c_int:      # asynchronous launch on a stream
            setattr(lib, name, _traced(name, fn))
    have = int(lib.e2fgvi_abi_version())
    if have != ABI_VERSION:
        # a stale build (or an E2FGVI_LIB A/B library of another round) would read the descriptors with another layout
        raise HipLibraryMissing("%s implements ABI version %d, this binding is version %d: rebuild it (python -m e2fgvi_amd.build)"
                                % (LIB_PATH, have, ABI_VERSION))
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def library_key():
    """sha256[:16] of the shared library that is (or would be) loaded: PMC / rocprof summaries under profiles/ carry the key of the
    library they measured, and bench.py quotes a traffic figure only when it belongs to the library it has just timed"""
    import hashlib
    try:
        with open(LIB_PATH, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def check(rc, what):
    if rc != 0:
        msg = load().e2fgvi_last_error()
        raise HipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
