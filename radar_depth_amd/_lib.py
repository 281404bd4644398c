"""ctypes binding of libradardepth_hip.so (the C ABI declared in include/radar_depth_hip.h).

The product path has NO fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (RD_LIB_PATH: another build of the same ABI -- A/B runs of two kernel versions in one GPU job, tools/ab_lib.sh)
LIB_PATH = os.environ.get("RD_LIB_PATH") or os.path.join(HERE, "lib", "libradardepth_hip.so")

RD_MAX_TAPS = 25
RD_MAX_PHASES = 4
ACT_NONE, ACT_RELU, ACT_LEAKY02 = 0, 1, 2


class RdPhase(C.Structure):
    _fields_ = [("n_taps", C.c_int32), ("out_off_h", C.c_int32), ("out_off_w", C.c_int32),
                ("lh", C.c_int32), ("lw", C.c_int32),
                ("dh_min", C.c_int32), ("dh_max", C.c_int32), ("dw_min", C.c_int32), ("dw_max", C.c_int32),
                ("tile_begin", C.c_int32),
                ("dh", C.c_int8 * RD_MAX_TAPS), ("dw", C.c_int8 * RD_MAX_TAPS), ("widx", C.c_int16 * RD_MAX_TAPS)]


class RdConvDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32), ("ldi", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("ldo", C.c_int32),
                ("in_stride", C.c_int32), ("out_stride", C.c_int32), ("n_phases", C.c_int32),
                ("phase", RdPhase * RD_MAX_PHASES)]


class RdReduceJob(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("tmp", C.c_void_p), ("grad", C.c_void_p), ("E", C.c_int64),
                ("n_splits", C.c_int32), ("J", C.c_int32), ("S", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("O", C.c_int32), ("I", C.c_int32), ("co_off", C.c_int32), ("accumulate", C.c_int32),
                ("first_block1", C.c_int32), ("n_blocks1", C.c_int32), ("first_block2", C.c_int32), ("n_blocks2", C.c_int32),
                ("pad_", C.c_int32)]


class RadarDepthHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library once; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RadarDepthHipError(
                "libradardepth_hip.so not built (%s). Run `python -m radar_depth_amd.build`; "
                "there is no CPU fallback for the HIP path." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.rd_last_error.restype = C.c_char_p
        for name in ("rd_wgrad_workspace_floats", "rd_stem_wgrad_workspace_floats", "rd_smooth_workspace_floats",
                     "rd_head_conv_bwd_workspace_floats", "rd_gconv_workspace_floats", "rd_wgrad_bf16_workspace_floats",
                     "rd_wgrad_split_workspace_floats"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_int64
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RadarDepthHipError("%s failed (%d): %s" % (what, rc, lib().rd_last_error().decode()))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None as c_void_p."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
