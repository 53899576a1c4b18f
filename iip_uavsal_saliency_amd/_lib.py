"""ctypes binding of libuavsal_hip.so (C ABI: include/uavsal_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
launch fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UAVSAL_HIP_LIB") or os.path.join(PKG, "libuavsal_hip.so")   # override: A/B builds

PREC = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3}
ACT_NONE, ACT_RELU6, ACT_SIGMOID = 0, 1, 2
EPI_AFFINE, EPI_TWA, EPI_LSTM = 0, 1, 2
SK_TICKET_BASE = 4096      # words of stream-K flags in front of the K-split ticket counters (csrc/conv_gemm_common.h)

_ERR = {-1: "UAVSAL_EINVAL (null pointer / non-positive size)",
        -2: "UAVSAL_EALIGN (channel count / ld / pointer not 16-byte aligned)",
        -3: "UAVSAL_ESHAPE (shape not supported by the kernel)",
        -4: "UAVSAL_ESTATE (plan used in the wrong state)",
        -5: "UAVSAL_EDEVICE (a kernel reported a device-side error; the run's outputs are invalid)"}
ERR_STREAMK = 1
DW_KERNEL = {1: "dw3x3_kernel<1, 4, 4>", 2: "dw3x3_kernel<1, 2, 2>", 3: "dw3x3_kernel<2, 2, 2>", 4: "dw3x3_dilated_kernel",
             16: "dw3x3_map_lds_kernel<16, 256>", 32: "dw3x3_map_lds_kernel<32, 256>", 64: "dw3x3_map_lds_kernel<64, 256>",
             528: "dw3x3_map_lds_kernel<16, 512>", 544: "dw3x3_map_lds_kernel<32, 512>", 576: "dw3x3_map_lds_kernel<64, 512>",
             1040: "dw3x3_map_lds_kernel<16, 1024>", 1056: "dw3x3_map_lds_kernel<32, 1024>", 1088: "dw3x3_map_lds_kernel<64, 1024>",
             2048: "dw3x3_rowclass_kernel<256>"}

_f = C.c_void_p   # device pointers travel as integers


class ConvDesc(C.Structure):
    _fields_ = [("a", _f), ("lda", C.c_int32), ("a_img_stride", C.c_int64),
                ("w", _f), ("scale", _f), ("bias", _f),
                ("out", _f), ("ldc", C.c_int32), ("o_img_stride", C.c_int64),
                ("res", _f), ("ldr", C.c_int32), ("r_img_stride", C.c_int64),
                ("aux", _f), ("ldx", C.c_int32), ("x_img_stride", C.c_int64),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("taps", C.c_int32),
                ("prec", C.c_int32), ("act", C.c_int32), ("epi", C.c_int32), ("tile", C.c_int32),
                ("out2", _f), ("ld2", C.c_int32),
                ("dw_w9c", _f), ("dw_scale", _f), ("dw_bias", _f),
                ("dw_stride", C.c_int32), ("dw_Hin", C.c_int32), ("dw_Win", C.c_int32),
                ("sk_ws", _f), ("sk_ws_bytes", C.c_int64),
                ("a_split", _f), ("ldas", C.c_int32),
                ("out_split", _f), ("ldos", C.c_int32),
                ("err", _f), ("sk_spin_limit", C.c_int32), ("sk_debug_drop", C.c_int32),
                ("w_group_stride", C.c_int64), ("n_group", C.c_int32), ("a_group_off", C.c_int32)]


class DwDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("w9c", _f), ("scale", _f), ("bias", _f),
                ("out", _f), ("ldo", C.c_int32),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("stride", C.c_int32), ("dilation", C.c_int32), ("act", C.c_int32),
                ("out_split", _f), ("ldos", C.c_int32), ("dil_group_c", C.c_int32), ("dil_groups", C.c_int32 * 4)]


class DwDotDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("w9c", _f), ("scale", _f), ("bias", _f),
                ("w2", _f), ("scale2", _f), ("bias2", _f), ("out", _f), ("ldo", C.c_int32),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("act", C.c_int32)]


class StemDesc(C.Structure):
    _fields_ = [("inp", _f), ("in_u8", _f), ("w", _f), ("scale", _f), ("bias", _f),
                ("out", _f), ("ldo", C.c_int32),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("mean", C.c_float * 3), ("stdv", C.c_float * 3)]


class BilinearDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32),
                ("out", _f), ("ldo", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("n_out", C.c_int32), ("C", C.c_int32), ("src_mod", C.c_int32), ("src_div", C.c_int32),
                ("out_split", _f), ("ldos", C.c_int32)]


class TdiffDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("out", _f), ("ldo", C.c_int32),
                ("n_img", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("seq_len", C.c_int32)]


class TsumDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("out", _f), ("ldo", C.c_int32),
                ("n_groups", C.c_int32), ("T", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32)]


class LayoutDesc(C.Structure):
    _fields_ = [("inp", _f), ("out", _f), ("n_img", C.c_int32), ("C", C.c_int32), ("HW", C.c_int32),
                ("ld", C.c_int32), ("to_nhwc", C.c_int32), ("Cpad", C.c_int32)]


class PostDesc(C.Structure):
    _fields_ = [("inp", _f), ("out", _f), ("scratch", _f), ("n_img", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class GuardDesc(C.Structure):
    _fields_ = [("err", _f), ("host_err", _f), ("buf", _f * 3), ("n", C.c_int64 * 3)]


class CopyDesc(C.Structure):
    _fields_ = [("inp", _f), ("out", _f), ("in_pitch", C.c_int64), ("out_pitch", C.c_int64),
                ("row_floats", C.c_int64), ("rows", C.c_int32)]


class FillDesc(C.Structure):
    _fields_ = [("out", _f), ("n", C.c_int64), ("bits", C.c_uint32)]


class FusedIrDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("w1", _f), ("scale1", _f), ("bias1", _f),
                ("wd", _f), ("scale_d", _f), ("bias_d", _f), ("w2", _f), ("scale2", _f), ("bias2", _f),
                ("res", _f), ("ldr", C.c_int32), ("out", _f), ("ldo", C.c_int32),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("hidden", C.c_int32), ("Cout", C.c_int32), ("stride", C.c_int32), ("tile", C.c_int32)]


class WinoDesc(C.Structure):
    _fields_ = [("inp", _f), ("ldi", C.c_int32), ("in_img_stride", C.c_int64),
                ("out", _f), ("ldo", C.c_int32), ("out_img_stride", C.c_int64),
                ("n_img", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("Mp", C.c_int64), ("R", C.c_int32),
                ("scale", _f), ("bias", _f), ("act", C.c_int32), ("epi", C.c_int32),
                ("res", _f), ("ldr", C.c_int32), ("res_img_stride", C.c_int64),
                ("aux", _f), ("ldx", C.c_int32), ("aux_img_stride", C.c_int64),
                ("hprev", _f), ("ldh", C.c_int32), ("h_img_stride", C.c_int64),
                ("n_seg", C.c_int32), ("seg_in", _f * 3), ("seg_ld", C.c_int32 * 3), ("seg_c", C.c_int32 * 3),
                ("seg_H", C.c_int32 * 3), ("seg_W", C.c_int32 * 3)]


DESC_TYPES = [ConvDesc, DwDesc, StemDesc, BilinearDesc, TdiffDesc, TsumDesc, LayoutDesc, PostDesc, GuardDesc, CopyDesc,
              FusedIrDesc, WinoDesc, DwDotDesc, FillDesc]

# every symbol include/uavsal_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("uavsal_conv_gemm", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("uavsal_conv_tile", C.c_int, [C.POINTER(ConvDesc)]),
    ("uavsal_conv_uses_split", C.c_int, [C.POINTER(ConvDesc)]),
    ("uavsal_conv_dwproj", C.c_int, [C.POINTER(ConvDesc)]),
    ("uavsal_streamk_workspace_bytes", C.c_longlong, []),
    ("uavsal_conv_streamk_grid", C.c_int, [C.POINTER(ConvDesc)]),
    ("uavsal_dw3x3", C.c_int, [C.POINTER(DwDesc), C.c_void_p]),
    ("uavsal_dw3x3_dot", C.c_int, [C.POINTER(DwDotDesc), C.c_void_p]),
    ("uavsal_dw_variant", C.c_int, [C.POINTER(DwDesc)]),
    ("uavsal_stem_conv", C.c_int, [C.POINTER(StemDesc), C.c_void_p]),
    ("uavsal_bilinear_ac", C.c_int, [C.POINTER(BilinearDesc), C.c_void_p]),
    ("uavsal_tdiff", C.c_int, [C.POINTER(TdiffDesc), C.c_void_p]),
    ("uavsal_tsum", C.c_int, [C.POINTER(TsumDesc), C.c_void_p]),
    ("uavsal_layout", C.c_int, [C.POINTER(LayoutDesc), C.c_void_p]),
    ("uavsal_postprocess", C.c_int, [C.POINTER(PostDesc), C.c_void_p]),
    ("uavsal_guard", C.c_int, [C.POINTER(GuardDesc), C.c_void_p]),
    ("uavsal_copy_rows", C.c_int, [C.POINTER(CopyDesc), C.c_void_p]),
    ("uavsal_fill", C.c_int, [C.POINTER(FillDesc), C.c_void_p]),
    ("uavsal_plan_add_fill", C.c_int, [C.c_void_p, C.POINTER(FillDesc)]),
    ("uavsal_fused_ir", C.c_int, [C.POINTER(FusedIrDesc), C.c_void_p]),
    ("uavsal_fused_ir_supported", C.c_int, [C.POINTER(FusedIrDesc)]),
    ("uavsal_plan_add_fused_ir", C.c_int, [C.c_void_p, C.POINTER(FusedIrDesc)]),
    ("uavsal_wino_input", C.c_int, [C.POINTER(WinoDesc), C.c_void_p]),
    ("uavsal_wino_output", C.c_int, [C.POINTER(WinoDesc), C.c_void_p]),
    ("uavsal_plan_add_wino_input", C.c_int, [C.c_void_p, C.POINTER(WinoDesc)]),
    ("uavsal_plan_add_wino_output", C.c_int, [C.c_void_p, C.POINTER(WinoDesc)]),
    ("uavsal_plan_add_copy", C.c_int, [C.c_void_p, C.POINTER(CopyDesc)]),
    ("uavsal_plan_create", C.c_void_p, []),
    ("uavsal_plan_destroy", None, [C.c_void_p]),
    ("uavsal_plan_add_conv", C.c_int, [C.c_void_p, C.POINTER(ConvDesc)]),
    ("uavsal_plan_add_dw", C.c_int, [C.c_void_p, C.POINTER(DwDesc)]),
    ("uavsal_plan_add_dw_dot", C.c_int, [C.c_void_p, C.POINTER(DwDotDesc)]),
    ("uavsal_plan_add_stem", C.c_int, [C.c_void_p, C.POINTER(StemDesc)]),
    ("uavsal_plan_add_bilinear", C.c_int, [C.c_void_p, C.POINTER(BilinearDesc)]),
    ("uavsal_plan_add_tdiff", C.c_int, [C.c_void_p, C.POINTER(TdiffDesc)]),
    ("uavsal_plan_add_tsum", C.c_int, [C.c_void_p, C.POINTER(TsumDesc)]),
    ("uavsal_plan_add_layout", C.c_int, [C.c_void_p, C.POINTER(LayoutDesc)]),
    ("uavsal_plan_error_word", C.c_void_p, [C.c_void_p]),
    ("uavsal_plan_add_guard", C.c_int, [C.c_void_p, _f, C.c_int64, _f, C.c_int64, _f, C.c_int64]),
    ("uavsal_plan_status", C.c_int, [C.c_void_p, C.c_int]),
    ("uavsal_plan_patch_ptr", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("uavsal_plan_set_lane", C.c_int, [C.c_void_p, C.c_int]),
    ("uavsal_plan_add_fork", C.c_int, [C.c_void_p, C.c_int]),
    ("uavsal_plan_add_join", C.c_int, [C.c_void_p, C.c_int]),
    ("uavsal_plan_enable_lanes", C.c_int, [C.c_void_p, C.c_int]),
    ("uavsal_plan_size", C.c_int, [C.c_void_p]),
    ("uavsal_plan_run", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("uavsal_plan_graph_build", C.c_int, [C.c_void_p, C.c_void_p]),
    ("uavsal_plan_graph_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("uavsal_plan_time", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    ("uavsal_abi_version", C.c_int, []),
    ("uavsal_sizeof_desc", C.c_int, [C.c_int]),
    ("uavsal_build_info", C.c_char_p, []),
]

_lib = None


def load():
    """Load (once) and type the shared library.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libuavsal_hip.so is not built (%s). Run `python -m iip_uavsal_saliency_amd.build` "
            "(needs hipcc). There is no CPU fallback for the UAVSal HIP path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.uavsal_abi_version() != 20:
        raise RuntimeError("libuavsal_hip.so ABI version mismatch")
    for i, t in enumerate(DESC_TYPES):
        if lib.uavsal_sizeof_desc(i) != C.sizeof(t):
            raise RuntimeError("descriptor %s: ctypes size %d != C size %d" % (
                t.__name__, C.sizeof(t), lib.uavsal_sizeof_desc(i)))
    _lib = lib
    return lib


def check(code: int, what: str = "uavsal call"):
    if code == 0:
        return
    if code == -5:
        raise RuntimeError("%s: %s" % (what, _ERR[-5]))
    if code < 0:
        raise RuntimeError("%s rejected its arguments: %s" % (what, _ERR.get(code, str(code))))
    raise RuntimeError("%s failed with hipError_t %d" % (what, code))
