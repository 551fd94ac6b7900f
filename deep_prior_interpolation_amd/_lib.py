"""ctypes binding of libdpi_hip.so (C ABI declared in include/dpi_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, importing
this module (or the first GPU op) raises.  torch is used only to obtain device pointers and the
current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdpi_hip.so")


class DpiError(RuntimeError):
    pass


ABI_VERSION = 404      # dpi_set_option replaces the ten dpi_set_* tuning exports (404); include/dpi_hip.h: dpi_conv_desc starts with its own size (300); dpi_conv_fwd_ws / dpi_conv_bwd_data_ws (301); `io` + the *_io entry points (400); dpi_pack_* (401); dpi_pack_forget (402); dpi_join_bwd (403)

# dpi_conv_desc.io bits / the `io` masks of the *_io entry points (bf16 storage of activations, BASELINE configs[4])
IO_X_BF16, IO_Y_BF16, IO_DY_BF16, IO_DX_BF16 = 1, 2, 4, 8
STORE_FWD_BF16, STORE_GRAD_BF16 = 1, 2


class ConvDesc(C.Structure):
    """dpi_conv_desc.  ConvDesc(Cin, Cout, D, H, W, k, kd, stride, precision, io) fills the leading `size` field itself."""
    _fields_ = [("size", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("k", C.c_int), ("kd", C.c_int), ("stride", C.c_int), ("precision", C.c_int), ("io", C.c_int)]

    def __init__(self, Cin=0, Cout=0, D=1, H=1, W=1, k=1, kd=1, stride=1, precision=0, io=0):
        super().__init__(C.sizeof(ConvDesc), int(Cin), int(Cout), int(D), int(H), int(W), int(k), int(kd), int(stride), int(precision), int(io))


class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_Z = C.c_size_t
_F = C.c_float
_U64 = C.c_uint64
_DESC = C.POINTER(ConvDesc)
_U = C.c_uint

# name -> (restype, argtypes).  Kept in one table so tests can check every symbol of the header is exported.
SIGNATURES = {
    "dpi_last_error": (C.c_char_p, []),
    "dpi_version": (_I, []),
    "dpi_conv_desc_size": (_I, []),
    "dpi_device_info": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_Z), C.c_char_p, _I]),
    "dpi_profile_marker": (_I, [_I, _P]),
    "dpi_conv_fwd_stat_blocks": (_I, [_DESC]),
    "dpi_conv_fwd": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "dpi_conv_bwd_data": (_I, [_DESC, _P, _P, _P, _I, _P]),
    "dpi_conv_fwd_ws_floats": (_Z, [_DESC]),
    "dpi_conv_bwd_data_ws_floats": (_Z, [_DESC]),
    "dpi_conv_fwd_ws": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "dpi_conv_bwd_data_ws": (_I, [_DESC, _P, _P, _P, _I, _P, _Z, _P]),
    "dpi_conv_bwd_data_dual": (_I, [_DESC, _P, _P, _DESC, _P, _P, _P, _I, _P, _Z, _P]),
    "dpi_conv_bwd_weight_ws_floats": (_Z, [_DESC]),
    "dpi_conv_bwd_weight": (_I, [_DESC, _P, _P, _P, _P, _P, _Z, _P]),
    "dpi_stat_blocks": (_I, [_I, _Z]),
    "dpi_channel_stats": (_I, [_P, _P, _I, _Z, _P, _P]),
    "dpi_bn_finalize": (_I, [_P, _I, _I, _Z, _P, _P, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P]),
    "dpi_chain_apply": (_I, [_P, _P, _I, _Z, _P, _P]),
    "dpi_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _I, _Z, _P, _P]),
    "dpi_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _Z, _P, _P, _P, _P]),
    "dpi_bn_bwd_apply_fork": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _Z, _P, _P, _P,
                                   _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "dpi_bn_bwd_apply_dual": (_I, [_P, _I, _I, _Z] + [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P] * 2 + [_I, _I, _P, _P, _P, _F, _P, _P]),
    "dpi_chain_add_stats": (_I, [_P, _P, _P, _P, _I, _Z, _F, _P, _P, _P]),
    "dpi_lrelu_bwd": (_I, [_P, _P, _F, _Z, _P, _P]),
    "dpi_act_fwd": (_I, [_P, _Z, _I, _P, _P]),
    "dpi_act_bwd": (_I, [_P, _P, _Z, _I, _P, _P]),
    "dpi_add": (_I, [_P, _P, _Z, _P, _P]),
    "dpi_channel_sum": (_I, [_P, _I, _Z, _P, _P, _P]),
    "dpi_upsample2x_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dpi_upsample2x_bwd_ws_floats": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "dpi_upsample2x_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dpi_crop_copy": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dpi_crop_copy_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dpi_maxpool2x2_fwd": (_I, [_P, _I, _I, _I, _P, _P]),
    "dpi_maxpool2x2_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "dpi_deconv4x4s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dpi_deconv4x4s2_bwd_data": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "dpi_deconv4x4s2_bwd_weight": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "dpi_loss_ws_doubles": (_Z, [_Z]),
    "dpi_masked_loss": (_I, [_P, _P, _P, _Z, _I, _F, _P, _P, _P, _P]),
    "dpi_adam_multi": (_I, [_P, _P, _I, _P, C.c_double, C.c_double, C.c_double, _P, _P]),
    "dpi_loop_control": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, C.c_double, C.c_double, _I, C.c_double, C.c_double, _I, C.c_double, _P]),
    "dpi_copy_if": (_I, [_P, _P, _P, _Z, _P]),
    "dpi_noise_add": (_I, [_P, _Z, _F, _U64, _P, _P, _P]),
    "dpi_fill_normal": (_I, [_P, _Z, _F, _F, _U64, _U64, _P]),
    "dpi_fir_axis0": (_I, [_P, _P, _I, _I, _I, _Z, _P, _P]),
    "dpi_axpy": (_I, [_F, _P, _Z, _P, _P]),
    "dpi_diff_axis": (_I, [_P, _Z, _I, _Z, _I, _F, _I, _P, _P]),
    "dpi_hale2d": (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _P, _P]),
    "dpi_structure_tensor": (_I, [_P, _Z, _I, _I, _F, _F, _P, _P, _P, _P]),
    "dpi_dips": (_I, [_P, _P, _P, _Z, _P, _P, _P]),
    "dpi_max_ws_floats": (_Z, [_Z]),
    "dpi_scaled_max": (_I, [_P, _Z, _F, _P, _P, _P]),
    "dpi_threshold": (_I, [_P, _Z, _P, _P, _P]),
    "dpi_pocs_project": (_I, [_P, _P, _P, _Z, _P, _P]),
    "dpi_overlap_add": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "dpi_overlap_normalize": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    # ABI 400: the same calls with the storage types of their activation / gradient tensors (`io` before the stream)
    "dpi_channel_stats_io": (_I, [_P, _P, _I, _Z, _P, _U, _P]),
    "dpi_chain_apply_io": (_I, [_P, _P, _I, _Z, _P, _U, _P]),
    "dpi_chain_add_stats_io": (_I, [_P, _P, _P, _P, _I, _Z, _F, _P, _P, _U, _P]),
    "dpi_bn_bwd_reduce_io": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _I, _Z, _P, _U, _P]),
    "dpi_bn_bwd_apply_io": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _Z, _P, _P, _P, _U, _P]),
    "dpi_bn_bwd_apply_fork_io": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _Z, _P, _P, _P,
                                      _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _U, _P]),
    "dpi_bn_bwd_apply_dual_io": (_I, [_P, _I, _I, _Z] + [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P] * 2 + [_I, _I, _P, _P, _P, _F, _P, _U, _P]),
    "dpi_upsample2x_fwd_io": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _U, _P]),
    "dpi_upsample2x_bwd_io": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _U, _P]),
    "dpi_noise_add_io": (_I, [_P, _Z, _F, _U64, _P, _P, _U, _P]),
    "dpi_noise_add_regen_io": (_I, [_Z, _F, _U64, _U64, _F, _U64, _P, _P, _U, _P]),
    # ABI 401: the packed-weight scratch of the bf16 stencil kernel
    "dpi_set_option": (_I, [C.c_char_p, _I]),              # ABI 404: the one switchboard for tests / tools (never called by the product path)
    "dpi_pack_scratch_bytes": (_Z, []),
    "dpi_pack_release": (_I, []),
    "dpi_pack_forget": (_I, [_P]),
    "dpi_pack_slot_count": (_Z, []),
    # ABI 403: BatchNorm backward of a residual join in two passes
    "dpi_join_bwd_ws_doubles": (_Z, [_I, _Z]),
    "dpi_join_bwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _Z,  _P, _P, _P, _P, _P, _F,  _P, _P, _P, _P, _P, _F,  _P, _P,  _I, _I, _P, _P, _P, _F,
                          _P, _P, _P, _P, _P, _P, _P, _U, _P]),
    "dpi_chain_add_apply": (_I, [_P, _P, _P, _P, _P, _I, _Z, _P, _U, _P]),
}

_lib = None


def load():
    """Load the library once; raises DpiError when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DpiError("libdpi_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # torch first: libdpi_hip.so must bind to the HIP runtime torch already loaded (its bundled libamdhip64).  Loaded the
    # other way round the process ends up with two runtimes and launches fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DpiError("libdpi_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.dpi_version() < ABI_VERSION or lib.dpi_conv_desc_size() != C.sizeof(ConvDesc):
        raise DpiError("libdpi_hip.so is ABI version %d with a %d-byte dpi_conv_desc; this binding needs >= %d and %d bytes (stale build? "
                       "re-run __graft_entry__.build())" % (lib.dpi_version(), lib.dpi_conv_desc_size(), ABI_VERSION, C.sizeof(ConvDesc)))
    # tests / tools: L.set_option("bf16_debug", 8) — raises on an unknown key or a value out of range
    lib.set_option = lambda key, value: check(lib.dpi_set_option(key.encode(), int(value)), "dpi_set_option(%s)" % key)
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().dpi_last_error()
        raise DpiError("%s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
