"""ctypes binding of libmmvae_hip.so (include/mmvae_hip.h).  No CPU fallback: if the library is missing or
no gfx950 device is usable, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmvae_hip.so")


class MMVAEError(RuntimeError):
    pass


class StepIO(C.Structure):
    """mmvae_mm_step_io"""
    _fields_ = [
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("step_counter", C.c_void_p),
        ("image", C.c_void_p), ("text", C.c_void_p), ("eps", C.c_void_p),
        ("enc_mask1", C.c_void_p), ("enc_mask2", C.c_void_p), ("gru_keep", C.c_void_p),
        ("enc_dropout", C.c_int), ("gru_dropout", C.c_int),
        ("force_tokens", C.c_void_p),
        ("kl_lambda", C.c_float),
        ("lambda_xy", C.c_float * 3), ("lambda_yx", C.c_float * 3),
        ("seed", C.c_ulonglong),
        ("sums", C.c_void_p), ("recon_image", C.c_void_p), ("recon_text", C.c_void_p),
        ("mu", C.c_void_p), ("logvar", C.c_void_p), ("tokens", C.c_void_p),
        ("pass_skip", C.c_int * 3),
        ("defer_unpack", C.c_int),
        ("pack_first", C.c_int),
        ("dp_split", C.c_int),
        ("early_adam", C.c_void_p),
    ]


class EarlyAdam(C.Structure):
    """struct mmvae_early_adam"""
    _fields_ = [
        ("m", C.c_void_p), ("v", C.c_void_p), ("state", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("grad_scale", C.c_float),
        ("gmap", C.c_void_p), ("ran", C.POINTER(C.c_int)),
    ]


class MnistStepIO(C.Structure):
    """mmvae_mnist_step_io"""
    _fields_ = [
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("step_counter", C.c_void_p),
        ("image", C.c_void_p), ("label", C.c_void_p), ("eps", C.c_void_p),
        ("lambda_xy", C.c_float * 3), ("lambda_yx", C.c_float * 3),
        ("kl_coef", C.c_float),
        ("seed", C.c_ulonglong),
        ("sums", C.c_void_p), ("recon_image", C.c_void_p), ("recon_text", C.c_void_p),
        ("mu", C.c_void_p), ("logvar", C.c_void_p),
        ("pass_skip", C.c_int * 3),
    ]


class CocoStepIO(C.Structure):
    """mmvae_coco_step_io"""
    _fields_ = [
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("step_counter", C.c_void_p),
        ("image", C.c_void_p), ("text", C.c_void_p), ("sos", C.c_void_p), ("eps", C.c_void_p),
        ("enc_mask1", C.c_void_p), ("enc_mask2", C.c_void_p), ("gru_keep", C.c_void_p),
        ("enc_dropout", C.c_int), ("gru_dropout", C.c_int),
        ("kl_lambda", C.c_float),
        ("lambda_xy", C.c_float * 3), ("lambda_yx", C.c_float * 3),
        ("seed", C.c_ulonglong),
        ("sums", C.c_void_p), ("recon_image", C.c_void_p), ("recon_text", C.c_void_p),
        ("mu", C.c_void_p), ("logvar", C.c_void_p),
        ("pass_skip", C.c_int * 3),
        ("defer_unpack", C.c_int),
        ("pack_first", C.c_int),
        ("optimizer_state", C.c_void_p),
    ]


class CelebaStepIO(C.Structure):
    """mmvae_celeba_step_io"""
    _fields_ = [
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("step_counter", C.c_void_p),
        ("image", C.c_void_p), ("attrs", C.c_void_p), ("eps", C.c_void_p),
        ("enc_mask", C.c_void_p), ("enc_dropout", C.c_int),
        ("kl_lambda", C.c_float),
        ("lambda_x", C.c_float * 3), ("lambda_y", C.c_float * 3),
        ("seed", C.c_ulonglong),
        ("sums", C.c_void_p), ("recon_image", C.c_void_p), ("recon_attrs", C.c_void_p),
        ("mu", C.c_void_p), ("logvar", C.c_void_p),
        ("pass_skip", C.c_int * 3),
        ("defer_unpack", C.c_int),
    ]


_P, _I, _F, _LL, _SZ, _ULL, _U = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t, C.c_ulonglong, C.c_uint

# name -> (restype, argtypes); restype int means "status code, raise on != 0"
SIGNATURES = {
    "mmvae_init": (_I, [_I]),
    "mmvae_last_error": (C.c_char_p, []),
    "mmvae_version": (C.c_char_p, []),
    "mmvae_set_stream_policy": (_I, [_I]),
    "mmvae_comm_unique_id": (_I, [_P]),
    "mmvae_comm_init": (_I, [C.POINTER(C.c_void_p), _I, _I, _P]),
    "mmvae_allreduce_grads": (_I, [_P, _P, C.c_size_t, _P]),
    "mmvae_comm_world": (_I, [_P]),
    "mmvae_comm_destroy": (_I, [_P]),
    "mmvae_mm_create": (_P, [_I, _I]),
    "mmvae_mm_destroy": (None, [_P]),
    "mmvae_mm_param_count": (_LL, [_P]),
    "mmvae_mm_num_params": (_I, [_P]),
    "mmvae_mm_param_info": (_I, [_P, _I, C.c_char_p, C.POINTER(_I), C.POINTER(_I), C.POINTER(_LL)]),
    "mmvae_mm_bn_floats": (_LL, [_P]),
    "mmvae_mm_num_bn": (_I, [_P]),
    "mmvae_mm_bn_info": (_I, [_P, _I, C.c_char_p, C.POINTER(_I), C.POINTER(_LL)]),
    "mmvae_mm_packed_elems": (_LL, [_P]),
    "mmvae_mm_packed_vec_elems": (_LL, [_P]),
    "mmvae_mm_gpk_elems": (_LL, [_P]),
    "mmvae_mm_gpk_vec_elems": (_LL, [_P]),
    "mmvae_mm_desc_bytes": (_SZ, [_P, _I]),
    "mmvae_mm_desc_copy": (_I, [_P, _I, _P]),
    "mmvae_mm_workspace_bytes": (_SZ, [_P]),
    "mmvae_mm_module_workspace_bytes": (_SZ, [_P]),
    "mmvae_mm_bind": (_I, [_P] * 11),
    "mmvae_mm_pack_weights": (_I, [_P, _P]),
    "mmvae_mm_grad_map": (_I, [_P, _P, _P]),
    "mmvae_mm_step": (_I, [_P, C.POINTER(StepIO), _I, _I, _P]),
    "mmvae_mm_wait_early_grads": (_I, [_P, _P]),
    "mmvae_mm_image_encoder_fwd": (_I, [_P, _P, _SZ, _P, _P, _P, _I, _P, _P]),
    "mmvae_mm_image_encoder_bwd": (_I, [_P, _P, _SZ, _P, _P, _P, _P]),
    "mmvae_mm_image_decoder_fwd": (_I, [_P, _P, _SZ, _P, _I, _P, _P]),
    "mmvae_mm_image_decoder_bwd": (_I, [_P, _P, _SZ, _P, _P, _P, _P]),
    "mmvae_mm_text_encoder_fwd": (_I, [_P, _P, _SZ, _P, _P, _P]),
    "mmvae_mm_text_encoder_bwd": (_I, [_P, _P, _SZ, _P, _P, _P]),
    "mmvae_mm_text_decoder_fwd": (_I, [_P, _P, _SZ, _P, _I, _P, _P, _P, _P, _P]),
    "mmvae_mm_text_decoder_bwd": (_I, [_P, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mmvae_mm_bench_layer": (_I, [_P, _P, _SZ, C.c_char_p, _I, _P]),
    "mmvae_mm_layer_flops": (C.c_double, [_P, C.c_char_p]),
    "mmvae_mm_layer_algo_flops": (C.c_double, [_P, C.c_char_p]),
    "mmvae_mm_layer_algo_bytes": (C.c_double, [_P, C.c_char_p]),
    "mmvae_debug_flops": (C.c_double, [_I]),
    "mmvae_debug_set": (_I, [C.c_char_p, _I]),
    "mmvae_mm_debug_offset": (_LL, [_P, C.c_char_p]),
    "mmvae_poe_fwd": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "mmvae_poe_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "mmvae_reparam_fwd": (_I, [_P, _P, _P, _I, _P, _P]),
    "mmvae_reparam_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "mmvae_kl_fwd": (_I, [_P, _P, _I, _P, _P]),
    "mmvae_kl_bwd": (_I, [_P, _P, _I, _F, _P, _P, _P, _P]),
    "mmvae_bce_fwd": (_I, [_P, _P, _LL, _P, _P]),
    "mmvae_bce_bwd": (_I, [_P, _P, _LL, _F, _P, _P, _P]),
    "mmvae_nll_fwd": (_I, [_P, _P, _I, _I, _P, _P]),
    "mmvae_nll_bwd": (_I, [_P, _I, _I, _F, _P, _P, _P]),
    "mmvae_normal": (_I, [_P, _LL, _ULL, _P, _U, _P]),
    "mmvae_keep_mask": (_I, [_P, _LL, _F, _ULL, _P, _U, _P]),
    "mmvae_mse_fwd": (_I, [_P, _P, _LL, _P, _P]),
    "mmvae_mse_bwd": (_I, [_P, _P, _LL, _F, _P, _P, _P]),
    "mmvae_u8_to_f32": (_I, [_P, _LL, _F, _P, _P]),
    "mmvae_u8_to_f32_after": (_I, [_P, _LL, _F, _P, _P, _P]),
    "mmvae_h2d_stage": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P]),
    "mmvae_gather_rows_u8_f32": (_I, [_P, _P, _LL, _LL, _F, _P, _P]),
    "mmvae_stream_wait_event": (_I, [_P, _P]),
    "mmvae_event_create": (_I, [C.POINTER(C.c_void_p)]),
    "mmvae_event_destroy": (_I, [_P]),
    "mmvae_event_record": (_I, [_P, _P]),
    "mmvae_event_synchronize": (_I, [_P]),
    "mmvae_stream_create": (_I, [_P]),
    "mmvae_stream_destroy": (_I, [_P]),
    "mmvae_gather_rows": (_I, [_P, _P, _LL, _LL, _P, _P]),
    "mmvae_step_status": (_I, [_P, _P]),
    "mmvae_step_losses": (_I, [_P, _P, _P, _P, _P, _P]),
    "mmvae_debug_probe": (_I, [_I]),
    "mmvae_debug_probe_read": (_I, [_P, _LL]),
    "mmvae_adam_step": (_I, [_P, _P, _P, _P, _LL, _P, _F, _F, _F, _F, _F, _P]),
    "mmvae_adam_step_packed": (_I, [_P, _P, _P, _P, _LL, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "mmvae_adam_step_packed_ranges": (_I, [_P, _P, _P, _P, _LL, C.POINTER(_LL), _I, _I, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "mmvae_mm_early_ranges": (_I, [_P, C.POINTER(_LL), _I]),
}


def _plan_api(pfx):
    """The plan/query/bind/pack entry points every model family exports under its own prefix."""
    return {
        "mmvae_%s_create" % pfx: (_P, [_I, _I]),
        "mmvae_%s_destroy" % pfx: (None, [_P]),
        "mmvae_%s_param_count" % pfx: (_LL, [_P]),
        "mmvae_%s_num_params" % pfx: (_I, [_P]),
        "mmvae_%s_param_info" % pfx: (_I, [_P, _I, C.c_char_p, C.POINTER(_I), C.POINTER(_I), C.POINTER(_LL)]),
        "mmvae_%s_bn_floats" % pfx: (_LL, [_P]),
        "mmvae_%s_num_bn" % pfx: (_I, [_P]),
        "mmvae_%s_bn_info" % pfx: (_I, [_P, _I, C.c_char_p, C.POINTER(_I), C.POINTER(_LL)]),
        "mmvae_%s_packed_elems" % pfx: (_LL, [_P]),
        "mmvae_%s_packed_vec_elems" % pfx: (_LL, [_P]),
        "mmvae_%s_gpk_elems" % pfx: (_LL, [_P]),
        "mmvae_%s_gpk_vec_elems" % pfx: (_LL, [_P]),
        "mmvae_%s_desc_bytes" % pfx: (_SZ, [_P, _I]),
        "mmvae_%s_desc_copy" % pfx: (_I, [_P, _I, _P]),
        "mmvae_%s_workspace_bytes" % pfx: (_SZ, [_P]),
        "mmvae_%s_module_workspace_bytes" % pfx: (_SZ, [_P]),
        "mmvae_%s_bind" % pfx: (_I, [_P] * 11),
        "mmvae_%s_pack_weights" % pfx: (_I, [_P, _P]),
        "mmvae_%s_grad_map" % pfx: (_I, [_P, _P, _P]),
    }


SIGNATURES.update(_plan_api("mnist"))
SIGNATURES["mmvae_mnist_create_p"] = (_P, [_I, _I, _I])
SIGNATURES["mmvae_mnist_precision"] = (_I, [_P])
SIGNATURES["mmvae_mnist_step"] = (_I, [_P, C.POINTER(MnistStepIO), _I, _I, _P])
for _m in ("image_encoder", "image_decoder", "text_encoder", "text_decoder"):
    SIGNATURES["mmvae_mnist_%s_fwd" % _m] = (_I, [_P, _P, _SZ, _P, _I, _P, _P])
SIGNATURES["mmvae_mnist_image_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P])
SIGNATURES["mmvae_mnist_image_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES["mmvae_mnist_text_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P])
SIGNATURES["mmvae_mnist_text_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES.update(_plan_api("celeba"))
SIGNATURES["mmvae_celeba_step"] = (_I, [_P, C.POINTER(CelebaStepIO), _I, _I, _P])
SIGNATURES["mmvae_celeba_image_encoder_fwd"] = (_I, [_P, _P, _SZ, _P, _P, _I, _P, _P])
SIGNATURES["mmvae_celeba_image_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P])
SIGNATURES["mmvae_celeba_image_decoder_fwd"] = (_I, [_P, _P, _SZ, _P, _I, _P, _P])
SIGNATURES["mmvae_celeba_image_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES["mmvae_celeba_attrs_encoder_fwd"] = (_I, [_P, _P, _SZ, _P, _I, _P, _P])
SIGNATURES["mmvae_celeba_attrs_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P])
SIGNATURES["mmvae_celeba_attrs_decoder_fwd"] = (_I, [_P, _P, _SZ, _P, _I, _P, _P])
SIGNATURES["mmvae_celeba_attrs_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES.update(_plan_api("coco"))
SIGNATURES["mmvae_coco_create_t"] = (_P, [_I, _I, _I])
SIGNATURES["mmvae_coco_steps"] = (_I, [_P])
SIGNATURES["mmvae_coco_step"] = (_I, [_P, C.POINTER(CocoStepIO), _I, _I, _P])
SIGNATURES["mmvae_coco_image_encoder_fwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _I, _P, _P])
SIGNATURES["mmvae_coco_image_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES["mmvae_coco_image_decoder_fwd"] = (_I, [_P, _P, _SZ, _P, _I, _P, _P])
SIGNATURES["mmvae_coco_image_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P])
SIGNATURES["mmvae_coco_text_encoder_fwd"] = (_I, [_P, _P, _SZ, _P, _P, _P])
SIGNATURES["mmvae_coco_text_encoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P])
SIGNATURES["mmvae_coco_text_decoder_fwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _I, _P, _P])
SIGNATURES["mmvae_coco_text_decoder_bwd"] = (_I, [_P, _P, _SZ, _P, _P, _P, _P, _P, _P, _P])
_STATUS = {n for n, (r, _) in SIGNATURES.items() if r is _I and not n.endswith(("_num_params", "_num_bn", "_precision", "_coco_steps", "_comm_world", "_probe_read", "_early_ranges"))}

_lib = None
_inited = set()


def load():
    """dlopen the library and declare every prototype (works without a GPU)."""
    global _lib
    if _lib is None:
        # PyTorch ships its own libamdhip64: it must be mapped first so that this library's NEEDED entry binds to the
        # same HIP runtime (two runtimes in one process see no devices)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MMVAEError("%s not found: run `python multimodal-vae_amd/build.py` (hipcc, gfx950). "
                             "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _STATUS and rc != 0:
        raise MMVAEError("%s failed (%d): %s" % (name, rc, lib.mmvae_last_error().decode()))
    return rc


def init_device(index):
    if index not in _inited:
        call("mmvae_init", int(index))
        _inited.add(index)


class OwnedEvent:
    """One HIP event created by the library (mmvae_event_create: no timing, no system-scope fence), with the three methods a
    host-side pipeline needs.  ``torch.cuda.Event()`` performs a system-scope release when it completes -- a cache write-back per
    record on the compute stream (include/mmvae_hip.h)."""

    def __init__(self):
        h = C.c_void_p()
        call("mmvae_event_create", C.byref(h))
        self.handle = h.value
        self._h = C.c_void_p(h.value)

    def record(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        call("mmvae_event_record", self._h, C.c_void_p(s.cuda_stream))

    def synchronize(self):
        call("mmvae_event_synchronize", self._h)

    def __del__(self):
        try:
            if self.handle:
                call("mmvae_event_destroy", self._h)
        except Exception:
            pass


class OwnedStream:
    """One HIP stream created by the library (mmvae_stream_create), usable wherever torch takes a stream.  NOT from torch's pool:
    the first ``torch.cuda.Stream()`` of a process creates 32 pool streams, more than the hardware queues the HIP runtime maps
    streams onto, and every kernel of the process slows down (include/mmvae_hip.h)."""

    def __init__(self, device):
        import torch
        h = C.c_void_p()
        with torch.cuda.device(device):
            call("mmvae_stream_create", C.byref(h))
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=device)

    def __del__(self):
        try:
            if self.handle:
                load().mmvae_stream_destroy(C.c_void_p(self.handle))
        except Exception:
            pass


def ptr(t):
    """data pointer of a tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())
