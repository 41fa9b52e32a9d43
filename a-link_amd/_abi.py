"""ctypes binding of libalink_hip.so (include/alink_hip.h) — the only way this package computes.

There is NO fallback: if the library is missing, or a call fails, an exception is raised.  The
prototypes below are a 1:1 transcription of include/alink_hip.h; tests/test_abi.py checks that the
built library exports every symbol the header declares.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libalink_hip.so")

DT_BF16, DT_F16, DT_F32, DT_F16X2 = 0, 1, 2, 3
LAYOUT_NHWC_F32, LAYOUT_NCHW_F32, LAYOUT_NHWC_U8 = 0, 1, 2
SCORE_UNCERTAINTY, SCORE_MARGIN, SCORE_ENTROPY, SCORE_DISPARITY = 0, 1, 2, 3


class AlinkError(RuntimeError):
    pass


class IRCfg(C.Structure):
    _fields_ = [("units", C.c_int * 4), ("widths", C.c_int * 5), ("height", C.c_int), ("width", C.c_int),
                ("emb", C.c_int), ("dtype", C.c_int), ("bn_eps", C.c_float)]


_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
_u64 = C.c_uint64
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes)
PROTOTYPES = {
    "alink_last_error": (C.c_char_p, []),
    "alink_init": (_i, [_i]),
    "alink_version": (_i, []),
    "alink_backbone_create": (_vp, [C.POINTER(IRCfg)]),
    "alink_backbone_destroy": (None, [_vp]),
    "alink_backbone_load": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "alink_backbone_num_tensors": (_i, [_vp]),
    "alink_backbone_tensor_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_sz)]),
    "alink_backbone_finalize": (_i, [_vp]),
    "alink_backbone_set_streams": (_i, [_vp, _i]),
    "alink_backbone_workspace_bytes": (_sz, [_vp, _i]),
    "alink_embed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "alink_backbone_enable_grad": (_i, [_vp]),
    "alink_backbone_set_small_batch_split": (_i, [_vp, _i]),
    "alink_backbone_calibrate": (_i, [_vp, _vp, _i, _i, _vp, _sz, _i, _vp]),
    "alink_backbone_range_flag": (_i, [_vp, _i]),
    "alink_backbone_device": (_i, [_vp]),
    "alink_backbone_set_products": (_i, [_vp, _i]),
    "alink_backbone_num_scales": (_i, [_vp]),
    "alink_backbone_get_scales": (_i, [_vp, _vp, _i]),
    "alink_backbone_set_scales": (_i, [_vp, _vp, _i]),
    "alink_conv_nhwc_x2": (_i, [_vp] * 6 + [_i] * 14 + [_vp]),
    "alink_backbone_grad_workspace_bytes": (_sz, [_vp, _i]),
    "alink_embed_cached": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "alink_embed_input_grad": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "alink_embed_profile": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "alink_conv_nhwc": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "alink_resnet50_create": (_vp, [_i, _i, _i, _f]),
    "alink_resnet50_destroy": (None, [_vp]),
    "alink_resnet50_num_tensors": (_i, [_vp]),
    "alink_resnet50_tensor_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_sz)]),
    "alink_resnet50_load": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "alink_resnet50_finalize": (_i, [_vp]),
    "alink_resnet50_workspace_bytes": (_sz, [_vp, _i]),
    "alink_resnet50_embed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "alink_resnet50_calibrate": (_i, [_vp, _vp, _i, _i, _vp, _sz, _i, _vp]),
    "alink_resnet50_range_flag": (_i, [_vp, _i]),
    "alink_resnet50_num_scales": (_i, [_vp]),
    "alink_resnet50_get_scales": (_i, [_vp, _vp, _i]),
    "alink_resnet50_set_scales": (_i, [_vp, _vp, _i]),
    "alink_resnet50_profile": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp, _vp, _vp, C.POINTER(_i)]),
    "alink_resnet50_op_name": (C.c_char_p, [_vp, _i]),
    "alink_vgg16_create": (_vp, [_i, _i, _i]),
    "alink_vgg16_destroy": (None, [_vp]),
    "alink_vgg16_num_tensors": (_i, [_vp]),
    "alink_vgg16_tensor_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_sz)]),
    "alink_vgg16_feature_size": (_i, [_vp]),
    "alink_vgg16_load": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "alink_vgg16_finalize": (_i, [_vp]),
    "alink_vgg16_workspace_bytes": (_sz, [_vp, _i]),
    "alink_vgg16_embed": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "alink_head_create": (_vp, [_i, _i, _i, _f, _f, _f]),
    "alink_head_create_ex": (_vp, [_i, _i, _i, _i, _f, _f, _f]),
    "alink_head_destroy": (None, [_vp]),
    "alink_head_num_params": (_sz, [_vp]),
    "alink_head_set_params": (_i, [_vp, _vp, _sz]),
    "alink_head_get_params": (_i, [_vp, _vp, _sz]),
    "alink_head_reset_optimizer": (_i, [_vp]),
    "alink_head_set_compute_dtype": (_i, [_vp, _i]),
    "alink_head_get_compute_dtype": (_i, [_vp]),
    "alink_head_set_lr": (_i, [_vp, _f]),
    "alink_head_get_lr": (_f, [_vp]),
    "alink_head_params_dev": (_vp, [_vp]),
    "alink_head_grads_dev": (_vp, [_vp]),
    "alink_head_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "alink_committee_forward": (_i, [C.POINTER(_vp), _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "alink_committee_forward_multi": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i64, _vp, _vp]),
    "alink_pair_scores_matrix": (_i, [C.POINTER(_vp), _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "alink_roc_counts": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "alink_head_train_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "alink_head_apply_update": (_i, [_vp, _vp]),
    "alink_head_apply_update_with": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "alink_head_set_graph": (_i, [_vp, _i]),
    "alink_head_eval": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "alink_head_custom_train_steps": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "alink_head_input_grads": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "alink_head_input_grads_relu": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "alink_head_train_step_input_grads": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "alink_smallres_create": (_vp, [_i, _i, _i, _f, _f, _f]),
    "alink_smallres_destroy": (None, [_vp]),
    "alink_smallres_num_params": (_sz, [_vp]),
    "alink_smallres_set_params": (_i, [_vp, _vp, _sz]),
    "alink_smallres_get_params": (_i, [_vp, _vp, _sz]),
    "alink_smallres_set_lr": (_i, [_vp, _f]),
    "alink_smallres_grads_dev": (_vp, [_vp]),
    "alink_smallres_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "alink_smallres_train_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _f, _i, _vp, _vp]),
    "alink_smallres_train_step_drawn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_uint64, _f, _i, _vp, _vp]),
    "alink_smallres_train_on_batch_host": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_uint64, _vp, _vp]),
    "alink_smallres_apply_update": (_i, [_vp, _vp]),
    "alink_smallres_set_graph": (_i, [_vp, _i]),
    "alink_smallres_eval": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "alink_smallres_mask_sizes": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "alink_noise_gaussian": (_i, [_vp, _vp, _i64, _f, _f, _u64, _u64, _vp]),
    "alink_noise_speckle": (_i, [_vp, _vp, _i64, _f, _u64, _u64, _vp]),
    "alink_noise_uniform": (_i, [_vp, _vp, _i64, _f, _f, _u64, _u64, _vp]),
    "alink_keep_masks": (_i, [_vp, _i64, _f, _u64, _vp]),
    "alink_keep_masks_at": (_i, [_vp, _i64, _f, _u64, _u64, _vp]),
    "alink_noise_saltpepper": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _u64, _u64, _vp]),
    "alink_noise_poisson_scratch_bytes": (_sz, [_i, _i64]),
    "alink_noise_poisson": (_i, [_vp, _vp, _i, _i64, _u64, _u64, _vp, _sz, _vp, _vp]),
    "alink_perlin_nodes": (_i, [_i, C.POINTER(_i)]),
    "alink_perlin_vectors": (_i, [_i, _i, _u64, _u64, _vp, _vp]),
    "alink_noise_perlin": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_i), _vp, _vp]),
    "alink_resize_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "alink_pgd_step": (_i, [_vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _vp]),
    "alink_perturb_images": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "alink_perturb_images_multi": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "alink_arcface_margin_workspace_bytes": (_sz, [_i, _i, _i]),
    "alink_arcface_margin_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "alink_contrastive_loss": (_i, [_vp, _vp, _vp, _i64, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "alink_score": (_i, [_i, _vp, _vp, _i, _i64, _i, _vp, _vp]),
    "alink_topk_scratch_bytes": (_sz, [_i64, _i]),
    "alink_topk": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None
_inited_devices = set()


def load():
    """dlopen the library and install prototypes.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: it ships its own libamdhip64, and the library has to bind to
    # THAT runtime instance (we are handed torch's streams and device pointers).  Loading ours first
    # pulls in the system runtime as a second copy, which then sees no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AlinkError(
            "libalink_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C a-link_amd/csrc`).  This package has no non-HIP fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().alink_last_error()
        raise AlinkError("%s failed (%d): %s" % (what or "alink call", rc, msg.decode() if msg else "?"))


def resolve_device(device):
    """None -> the process's current torch device (one process per GPU: torch.cuda.set_device(LOCAL_RANK))."""
    if device is None:
        import torch
        return int(torch.cuda.current_device())
    return int(device)


def init(device=0):
    """alink_init on `device` (once per device).  Raises if no GPU is visible."""
    device = resolve_device(device)
    lib = load()
    if device not in _inited_devices:
        check(lib.alink_init(int(device)), "alink_init")
        _inited_devices.add(device)
    return lib


def current_stream(device=None):
    """torch's current stream ON `device` (None: the current device) — a handle's launches go to a stream of
    the handle's own device, whatever device is current in the caller (include/alink_hip.h, device rule)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(device):
    """Context manager making `device` current: handles record the device that is current at their create call."""
    import torch
    return torch.cuda.device(int(device))


def ptr(t):
    """device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
