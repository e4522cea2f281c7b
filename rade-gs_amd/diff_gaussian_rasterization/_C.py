"""`_C` -- binding of libradegs_hip.so (C ABI: include/radegs.h) with the call surface of the
reference's pybind module (DGR/ext.cpp:15-19):

    rasterize_gaussians(...)            DGR/rasterize_points.h:18-42  / rasterize_points.cu:36-133
    rasterize_gaussians_backward(...)   DGR/rasterize_points.h:43-76  / rasterize_points.cu:136-246
    mark_visible(...)                   DGR/rasterize_points.h:78-81  / rasterize_points.cu:248-267
    integrate_gaussians_to_points(...)  DGR/rasterize_points.h:83-110 / rasterize_points.cu:269-388

Same positional arguments, same return tuples (note the forward's output order differs from the
Python operator's 8-tuple, exactly as upstream).  torch is used for device memory and the current
HIP stream only; no torch type crosses the C boundary.

There is NO CPU path and no fallback: a missing library or a non-GPU tensor raises.
"""
import ctypes
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RADEGS_LIB") or os.path.join(_HERE, "libradegs_hip.so")   # RADEGS_LIB: A/B runs of two builds

_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_F = ctypes.POINTER(ctypes.c_float)


class RadegsFwdArgs(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int), ("D", ctypes.c_int), ("M", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p),
                ("colors_precomp", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p),
                ("rotations", ctypes.c_void_p), ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
                ("projmatrix", ctypes.c_void_p), ("cam_pos", ctypes.c_void_p),
                ("scale_modifier", ctypes.c_float), ("tan_fovx", ctypes.c_float), ("tan_fovy", ctypes.c_float),
                ("kernel_size", ctypes.c_float),
                ("prefiltered", ctypes.c_int), ("require_coord", ctypes.c_int), ("require_depth", ctypes.c_int), ("debug", ctypes.c_int),
                ("out_color", ctypes.c_void_p), ("out_coord", ctypes.c_void_p), ("out_mcoord", ctypes.c_void_p),
                ("out_depth", ctypes.c_void_p), ("out_mdepth", ctypes.c_void_p), ("out_alpha", ctypes.c_void_p),
                ("out_normal", ctypes.c_void_p), ("radii", ctypes.c_void_p)]


class RadegsBwdArgs(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_size_t), ("P", ctypes.c_int), ("D", ctypes.c_int), ("M", ctypes.c_int), ("R", ctypes.c_int), ("width", ctypes.c_int),
                ("height", ctypes.c_int),
                ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p),
                ("colors_precomp", ctypes.c_void_p), ("alphas", ctypes.c_void_p), ("scales", ctypes.c_void_p),
                ("rotations", ctypes.c_void_p), ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
                ("projmatrix", ctypes.c_void_p), ("cam_pos", ctypes.c_void_p),
                ("scale_modifier", ctypes.c_float), ("tan_fovx", ctypes.c_float), ("tan_fovy", ctypes.c_float),
                ("kernel_size", ctypes.c_float),
                ("radii", ctypes.c_void_p), ("normalmap", ctypes.c_void_p), ("geom_buffer", ctypes.c_void_p),
                ("binning_buffer", ctypes.c_void_p), ("image_buffer", ctypes.c_void_p),
                ("dL_dpix", ctypes.c_void_p), ("dL_dpix_coord", ctypes.c_void_p), ("dL_dpix_mcoord", ctypes.c_void_p),
                ("dL_dpix_depth", ctypes.c_void_p), ("dL_dpix_mdepth", ctypes.c_void_p), ("dL_dalphas", ctypes.c_void_p),
                ("dL_dpix_normal", ctypes.c_void_p),
                ("dL_dmean2D", ctypes.c_void_p), ("dL_dcolor", ctypes.c_void_p), ("dL_dopacity", ctypes.c_void_p),
                ("dL_dmean3D", ctypes.c_void_p), ("dL_dcov3D", ctypes.c_void_p), ("dL_dsh", ctypes.c_void_p),
                ("dL_dscale", ctypes.c_void_p), ("dL_drot", ctypes.c_void_p),
                ("require_coord", ctypes.c_int), ("require_depth", ctypes.c_int), ("debug", ctypes.c_int),
                ("dL_drgb_clamped", ctypes.c_void_p), ("opacity_grad_intended", ctypes.c_int),
                ("drgb_ready", ctypes.c_void_p), ("drgb_ready_user", ctypes.c_void_p),
                ("grad_chunks", ctypes.c_int), ("grads_ready", ctypes.c_void_p), ("grads_ready_user", ctypes.c_void_p),
                ("keep_sums", ctypes.c_int), ("acc_reuse", ctypes.c_int)]


class RadegsIntegrateArgs(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int), ("D", ctypes.c_int), ("M", ctypes.c_int), ("PN", ctypes.c_int), ("width", ctypes.c_int),
                ("height", ctypes.c_int),
                ("background", ctypes.c_void_p), ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p),
                ("colors_precomp", ctypes.c_void_p), ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p),
                ("rotations", ctypes.c_void_p), ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p),
                ("projmatrix", ctypes.c_void_p), ("cam_pos", ctypes.c_void_p), ("points3D", ctypes.c_void_p),
                ("scale_modifier", ctypes.c_float), ("tan_fovx", ctypes.c_float), ("tan_fovy", ctypes.c_float),
                ("kernel_size", ctypes.c_float), ("debug", ctypes.c_int),
                ("out_color", ctypes.c_void_p), ("out_alpha_integrated", ctypes.c_void_p), ("out_color_integrated", ctypes.c_void_p),
                ("out_coordinate2d", ctypes.c_void_p), ("out_sdf", ctypes.c_void_p), ("radii", ctypes.c_void_p)]


# every symbol include/radegs.h declares
EXPORTED_SYMBOLS = ("radegs_forward", "radegs_backward", "radegs_backward_from_sums", "radegs_mark_visible", "radegs_integrate", "radegs_sh_grad_from_views", "radegs_geometry_bytes", "radegs_image_bytes",
                    "radegs_binning_bytes", "radegs_debug_export", "radegs_forget_image", "radegs_last_error", "radegs_version", "radegs_profile_enable",
                    "radegs_profile_select", "radegs_profile_stride", "radegs_binning_stats", "radegs_reload_env", "radegs_last_forward_used_streams", "radegs_profile_num_stages", "radegs_profile_stage_name", "radegs_profile_collect",
                    # fused pre/post steps (bound in graphics_utils.py / gaussian_model_ops.py)
                    "radegs_normals_forward", "radegs_normals_backward", "radegs_normal_loss_scratch_bytes",
                    "radegs_normal_loss_forward", "radegs_normal_loss_backward", "radegs_normals_last_error",
                    "radegs_filter3d_forward", "radegs_filter3d_backward", "radegs_compute_filter3d",
                    "radegs_photometric_scratch_bytes",
                    "radegs_photometric_forward", "radegs_photometric_backward", "radegs_adam_step", "radegs_knn_scratch_bytes",
                    "radegs_knn_mean_dist2")

_lib = None
# test hook: when True, the per-Gaussian accumulation scratch of the last backward is kept in LAST_ACC
KEEP_ACC = False
# False (default): the backward the reference executes, including its argument slip in the opacity-compensation gradient
# (include/radegs.h::RadegsBwdArgs.opacity_grad_intended).  True (or RADEGS_OPACITY_GRAD=intended in the environment): the derivative
# the reference's formulas intend.  Differs only for kernel_size > 0.
OPACITY_GRAD_INTENDED = os.environ.get("RADEGS_OPACITY_GRAD", "").lower() == "intended"
LAST_ACC = None
LAST_POINT_STATE = None
# Optional allocator for the 8 gradient tensors of the backward: callable(name, shape, dtype, device) -> tensor
# or None.  A data-parallel caller points it at slices of ONE flat bucket so that the gradient all-reduce needs no
# gather copy (rade-gs_amd/view_parallel.GradBucket).  Default: plain torch.empty.
# Installed PER DEVICE with set_grad_allocator(device, fn): autograd runs every device's backward on its own worker thread,
# so one process driving several GPUs must not share a single hook.  The module attribute GRAD_ALLOCATOR is the fallback for
# devices without their own entry (one process per GPU, the deployment this library is built for, needs nothing else).
GRAD_ALLOCATOR = None
_GRAD_ALLOCATORS = {}
_HOOK_LOCK = threading.Lock()


def set_grad_allocator(device, fn):
    """Install (fn) or remove (None) the gradient allocator of one device; see GRAD_ALLOCATOR."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    with _HOOK_LOCK:
        if fn is None:
            _GRAD_ALLOCATORS.pop(idx, None)
        else:
            _GRAD_ALLOCATORS[idx] = fn


def _grad_allocator_for(dev):
    with _HOOK_LOCK:
        return _GRAD_ALLOCATORS.get(dev.index, GRAD_ALLOCATOR)


# The allocator may also (a) return a (P,3) tensor for the extra name "dL_drgb_clamped" -- the backward then fills it with
# dL/dRGB (clamp mask applied) -- and (b) return SKIP_GRAD for "dL_dsh": the (P,M,3) SH gradient is then not written and
# comes back as None (it is basis(dir) x dL_drgb_clamped; view_parallel.FactoredGradExchange rebuilds the batch sum).
SKIP_GRAD = object()
# (c) if the allocator OBJECT has a method `drgb_ready()` (the allocator is then a bound method of it), the backward calls it on the
# host as soon as the kernel that writes dL_drgb_clamped is queued -- before the per-Gaussian backward -- so that an all-gather of
# those rows can run under that kernel (include/radegs.h::RadegsBwdArgs.drgb_ready).
_READY_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
_GRADS_READY_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)


def library():
    """Load libradegs_hip.so (built in-tree by rade-gs_amd/build.py).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build it with `python rade-gs_amd/build.py` "
                               "(there is no CPU or PyTorch fallback for this operator)")
        L = ctypes.CDLL(_LIB_PATH)
        L.radegs_forward.restype = ctypes.c_int
        L.radegs_forward.argtypes = [ctypes.POINTER(RadegsFwdArgs), _ALLOC_FN, ctypes.c_void_p, _ALLOC_FN, ctypes.c_void_p, _ALLOC_FN,
                                     ctypes.c_void_p, ctypes.c_void_p]
        L.radegs_backward.restype = ctypes.c_int
        L.radegs_backward.argtypes = [ctypes.POINTER(RadegsBwdArgs), _ALLOC_FN, ctypes.c_void_p, ctypes.c_void_p]
        L.radegs_backward_from_sums.restype = ctypes.c_int
        L.radegs_backward_from_sums.argtypes = [ctypes.POINTER(RadegsBwdArgs), ctypes.c_void_p, ctypes.c_void_p]
        L.radegs_integrate.restype = ctypes.c_int
        L.radegs_integrate.argtypes = [ctypes.POINTER(RadegsIntegrateArgs)] + [_ALLOC_FN, ctypes.c_void_p] * 4 + [ctypes.c_void_p]
        L.radegs_sh_grad_from_views.restype = ctypes.c_int
        L.radegs_sh_grad_from_views.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        L.radegs_mark_visible.restype = ctypes.c_int
        L.radegs_mark_visible.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.radegs_geometry_bytes.restype = ctypes.c_size_t
        L.radegs_geometry_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.radegs_image_bytes.restype = ctypes.c_size_t
        L.radegs_image_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.radegs_binning_bytes.restype = ctypes.c_size_t
        L.radegs_binning_bytes.argtypes = [ctypes.c_int]
        L.radegs_debug_export.restype = ctypes.c_longlong
        L.radegs_debug_export.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.radegs_forget_image.restype = None
        L.radegs_forget_image.argtypes = [ctypes.c_void_p]
        L.radegs_last_error.restype = ctypes.c_char_p
        L.radegs_version.restype = ctypes.c_char_p
        L.radegs_binning_stats.restype = None
        L.radegs_binning_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        L.radegs_reload_env.restype = None
        L.radegs_reload_env.argtypes = []
        L.radegs_last_forward_used_streams.restype = ctypes.c_int
        L.radegs_last_forward_used_streams.argtypes = []
        L.radegs_profile_enable.restype = None
        L.radegs_profile_enable.argtypes = [ctypes.c_int]
        L.radegs_profile_select.restype = None
        L.radegs_profile_select.argtypes = [ctypes.c_int]
        L.radegs_profile_stride.restype = None
        L.radegs_profile_stride.argtypes = [ctypes.c_int]
        L.radegs_profile_num_stages.restype = ctypes.c_int
        L.radegs_profile_stage_name.restype = ctypes.c_char_p
        L.radegs_profile_stage_name.argtypes = [ctypes.c_int]
        L.radegs_profile_collect.restype = ctypes.c_int
        L.radegs_profile_collect.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        # whatever failed, the cached accumulation scratches are no longer known to be zero (RADEGS_ERR_STATE in particular is reported
        # one call late: the backward it belongs to has poisoned its scratch with NaN): the next backward starts from a fresh one
        _ACC_SCRATCH.clear()
        raise RuntimeError(f"{what} failed ({rc}): {library().radegs_last_error().decode()}")
    return rc


def _require_gpu(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"diff_gaussian_rasterization (MI355X build): `{name}` must be a GPU tensor -- "
                           "this operator has no CPU implementation")


def _f32(t, name):
    """contiguous float32 view of an input; empty tensor == 'not provided' == NULL (reference convention)."""
    if t is None or t.numel() == 0:
        return None
    _require_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"`{name}` must be float32")
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _forget_image(holder):
    """The native side remembers which image-state buffers hold entry streams by ADDRESS (the backward replays them only for a
    buffer whose forward wrote them).  An address must leave that set when its buffer moves or dies, or a later, unrelated buffer
    landing on it would be taken for a stream image: called on resize and from the tensor's finaliser."""
    try:
        if holder[0] and _lib is not None:
            _lib.radegs_forget_image(ctypes.c_void_p(holder[0]))
    except Exception:   # interpreter shutdown: the library may already be gone; nothing left to protect then
        pass
    holder[0] = 0


# Test knob (tests/test_gpu_poison.py): every buffer the native side is about to fill -- the produced maps, radii, the three state
# buffers -- is first overwritten with 0xFF bytes / NaN, so a kernel that reads something this call has not written, or leaves a
# pixel unwritten, shows up in the results instead of hiding behind whatever the allocator's recycled memory happened to hold.
_POISON = os.environ.get("RADEGS_DEBUG_POISON", "0") == "1"
# RADEGS_ACC_REUSE=0: a fresh accumulation scratch per backward, filled with zeros by the call (the behaviour before round 6); default:
# one scratch per (device, stream), handed back zeroed by the backward itself (RadegsBwdArgs.acc_reuse)
ACC_REUSE = os.environ.get("RADEGS_ACC_REUSE", "1") != "0"
ACC_REUSE_MAX_BYTES = 256 << 20


class _Resizable:
    """uint8 device tensor grown on request from the native side (the resize lambda of
    DGR/rasterize_points.cu:27-33).  image=True: the tensor is an image-state buffer (see _forget_image)."""

    def __init__(self, device, image=False):
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.error = None
        holder = [0]
        if image:
            weakref.finalize(self.tensor, _forget_image, holder)

        def _cb(_user, nbytes):
            # every request stands for itself: the native side may retry with a smaller size after a failed one (alloc_image's
            # fallback to the tile-wide formulation, rg_launch.inc), and a stale error must not outlive the retry that succeeded
            self.error = None
            try:
                self.tensor.resize_(int(nbytes))
                if _POISON:
                    self.tensor.fill_(0xFF)
                if image and self.tensor.data_ptr() != holder[0]:
                    _forget_image(holder)
                    holder[0] = self.tensor.data_ptr()
                return self.tensor.data_ptr()
            except Exception as ex:  # surfaces as RADEGS_ERR_ALLOC
                self.error = ex
                return 0

        self.cb = _ALLOC_FN(_cb)

    def release(self):
        """Drop the ctypes callback once the native call has returned: it closes over `self`, and the cycle would keep the
        tensor (hundreds of MB of device memory at 1M+ Gaussians) alive until Python's cyclic GC happens to run."""
        self.cb = None


class _Fixed:
    """allocator callback over a tensor that already exists (the cached accumulation scratch)"""

    def __init__(self, tensor):
        self.tensor = tensor
        self.error = None

        def _cb(_user, nbytes):
            if int(nbytes) > self.tensor.numel():
                self.error = RuntimeError(f"accumulation scratch of {self.tensor.numel()} B asked for {int(nbytes)} B")
                return 0
            return self.tensor.data_ptr()

        self.cb = _ALLOC_FN(_cb)

    def release(self):
        self.cb = None


_ACC_SCRATCH = {}   # (device index, stream handle) -> all-zero uint8 tensor, kept zero by the backward itself (RadegsBwdArgs.acc_reuse)
# held for the whole of a backward that uses a cached scratch: ctypes drops the GIL during the native call, and two host threads queueing
# backwards on ONE stream would otherwise interleave their kernels over the same buffer
_ACC_LOCK = threading.Lock()


def _acc_scratch(key, nbytes, device):
    """The zeroed accumulation scratch of this (device, stream), grown (and zeroed again) when a call needs more; at most 8 are kept."""
    t = _ACC_SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        if len(_ACC_SCRATCH) >= 8:
            _ACC_SCRATCH.clear()
        t = torch.zeros(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _ACC_SCRATCH[key] = t
    return t


def _zero_maps(channels, H, W, device):
    """All-zero maps for the outputs a call does not produce (the reference returns torch.full(0) maps whatever the flags,
    rasterize_points.cu:71-77): FRESH memory every call, like the reference -- ONE zero-filled allocation (one ~6 us fill at 1080p
    instead of five) handed out as disjoint tensors that merely share its storage.  They are built with `set_`, not by slicing:
    slices would be autograd VIEWS of one base, and an autograd Function that returns several views of one tensor makes every
    later in-place edit of them raise ("Output N of ... is a view and is being modified inplace"); the reference's separate
    torch.full tensors allow `depth[mask] = 0` / `normal *= x`, so these must too (tests/test_cabi.py)."""
    total = sum(channels)
    if total == 0:
        return []
    flat = torch.zeros(total * H * W, dtype=torch.float32, device=device)
    storage = flat.untyped_storage()
    out, off = [], 0
    for c in channels:
        t = torch.empty(0, dtype=torch.float32, device=device)
        t.set_(storage, off * H * W, (c, H, W))     # own tensor, own version counter, no `_base`: not a view for autograd
        out.append(t)
        off += c
    return out


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, kernel_size, image_height, image_width, sh, degree, campos, prefiltered,
                        require_coord, require_depth, debug):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:60-62
    _require_gpu(means3D, "means3D")
    L = library()
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    require_coord, require_depth = bool(require_coord), bool(require_depth)
    fo = dict(dtype=torch.float32, device=dev)
    live = P != 0

    geo = require_coord or require_depth
    # (channels, produced by this call) in the order color, depth, mdepth, coord, mcoord, alpha, normal
    spec = [(3, True), (1, require_depth), (1, require_depth), (3, require_coord), (3, require_coord), (1, True), (3, geo)]
    # A map the flags do not produce is all-zero in the reference (torch::full(0), rasterize_points.cu:71-77) and FRESH memory on every call.
    # The library zero-fills whatever unproduced map it is handed (include/radegs.h: inside the blend kernel, no separate fill), so a live call
    # allocates all seven uninitialised; only the call that launches nothing (P == 0) fills them here.
    if live:
        maps = [torch.empty((c, H, W), **fo) for c, _ in spec]
    else:
        maps = _zero_maps([c for c, _ in spec], H, W, dev)
    out_color, out_depth, out_mdepth, out_coord, out_mcoord, out_alpha, out_normal = maps
    radii = torch.empty(P, dtype=torch.int32, device=dev) if live else torch.zeros(P, dtype=torch.int32, device=dev)
    if _POISON and live:
        for m in maps:   # the unproduced ones too: the library must overwrite them with zeros
            m.fill_(float("nan"))
        radii.fill_(-0x01010102)
    geom, binning, img = _Resizable(dev), _Resizable(dev), _Resizable(dev, image=True)
    rendered = 0
    if live:
        bg, m3 = _f32(background, "bg"), _f32(means3D, "means3D")
        col, op = _f32(colors, "colors_precomp"), _f32(opacity, "opacities")
        sc, rot, cov = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
        vm, pm, cp, shs = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos"), _f32(sh, "shs")
        M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
        a = RadegsFwdArgs(P, int(degree), M, W, H, _ptr(bg), _ptr(m3), _ptr(shs), _ptr(col), _ptr(op), _ptr(sc), _ptr(rot), _ptr(cov),
                          _ptr(vm), _ptr(pm), _ptr(cp), float(scale_modifier), float(tan_fovx), float(tan_fovy), float(kernel_size),
                          int(bool(prefiltered)), int(require_coord), int(require_depth), int(bool(debug)),
                          _ptr(out_color), _ptr(out_coord), _ptr(out_mcoord), _ptr(out_depth), _ptr(out_mdepth), _ptr(out_alpha),
                          _ptr(out_normal), _ptr(radii))
        with torch.cuda.device(dev):
            rc = L.radegs_forward(ctypes.byref(a), geom.cb, None, binning.cb, None, img.cb, None, _stream(dev))
        for r in (geom, binning, img):
            r.release()
            if r.error is not None:
                raise r.error
        rendered = _check(rc, "radegs_forward")
    return (rendered, out_color, out_coord, out_mcoord, out_alpha, out_normal, out_depth, out_mdepth, radii, geom.tensor,
            binning.tensor, img.tensor)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, kernel_size, dL_dout_color, dL_dout_coord, dL_dout_mcoord,
                                 dL_dout_depth, dL_dout_mdepth, dL_dout_alpha, dL_dout_normal, normalmap, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, alphas, require_coord, require_depth, debug):
    _require_gpu(means3D, "means3D")
    L = library()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    fo = dict(dtype=torch.float32, device=dev)

    grad_alloc = _grad_allocator_for(dev)

    def mk(name, shape):
        if P != 0 and grad_alloc is not None:
            t = grad_alloc(name, shape, torch.float32, dev)
            if t is not None:
                assert t.shape == torch.Size(shape) and t.is_contiguous() and t.dtype == torch.float32 and t.device == dev
                return t
        if P != 0 and _POISON:
            return torch.full(shape, float("nan"), **fo)
        return torch.empty(shape, **fo) if P != 0 else torch.zeros(shape, **fo)

    dL_dmeans3D, dL_dmeans2D, dL_dcolors = mk("dL_dmeans3D", (P, 3)), mk("dL_dmeans2D", (P, 3)), mk("dL_dcolors", (P, 3))
    dL_dopacity, dL_dcov3D = mk("dL_dopacity", (P, 1)), mk("dL_dcov3D", (P, 6))
    drgb = None
    skip_dsh = False
    if P != 0 and M != 0 and grad_alloc is not None:
        drgb = grad_alloc("dL_drgb_clamped", (P, 3), torch.float32, dev)
        skip_dsh = drgb is not None and grad_alloc("dL_dsh", (P, M, 3), torch.float32, dev) is SKIP_GRAD
        if drgb is not None:
            assert drgb.shape == torch.Size((P, 3)) and drgb.is_contiguous() and drgb.dtype == torch.float32 and drgb.device == dev
    dL_dsh = None if skip_dsh else mk("dL_dsh", (P, M, 3))
    dL_dscales, dL_drotations = mk("dL_dscales", (P, 3)), mk("dL_drotations", (P, 4))
    if P != 0:
        bg, m3, col = _f32(background, "bg"), _f32(means3D, "means3D"), _f32(colors, "colors_precomp")
        sc, rot, cov = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
        vm, pm, cp, shs = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos"), _f32(sh, "shs")
        g = [_f32(t, n) for t, n in ((dL_dout_color, "dL_dcolor"), (dL_dout_coord, "dL_dcoord"), (dL_dout_mcoord, "dL_dmcoord"),
                                     (dL_dout_depth, "dL_ddepth"), (dL_dout_mdepth, "dL_dmdepth"), (dL_dout_alpha, "dL_dalpha"),
                                     (dL_dout_normal, "dL_dnormal"))]
        al, nm = _f32(alphas, "alphas"), _f32(normalmap, "normalmap")
        rad = radii.contiguous()
        gb, bb, ib = geomBuffer.contiguous(), binningBuffer.contiguous(), imageBuffer.contiguous()
        if not sc is None and rot is None:
            raise RuntimeError("scales given without rotations")
        # the accumulation scratch: ONE buffer per (device, stream), zeroed once and handed back zeroed by every successful call
        # (RadegsBwdArgs.acc_reuse) -- unless its contents are wanted afterwards (KEEP_ACC), which takes a scratch of its own
        abytes = P * (128 if require_coord else 64)
        akey = (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))
        # (above ACC_REUSE_MAX_BYTES the clearing stores cost the per-Gaussian kernel more than the fill they replace: C4, 5 M Gaussians with
        # the coord map = 640 MB, same-box A/B: preprocess_bwd +0.13 ms against a 0.08-ms fill)
        acc_cached = None if (KEEP_ACC or _POISON or not ACC_REUSE or abytes > ACC_REUSE_MAX_BYTES) else _acc_scratch(akey, abytes, dev)
        acc = _Resizable(dev) if acc_cached is None else _Fixed(acc_cached)
        ready_cb, ready_err = None, []
        owner = getattr(grad_alloc, "__self__", None)
        if drgb is not None and owner is not None and callable(getattr(owner, "drgb_ready", None)) and getattr(owner, "early_drgb", False):
            def _ready(_user, _owner=owner):
                try:
                    _owner.drgb_ready()
                except Exception as ex:  # noqa: BLE001 -- must not unwind through the C frame
                    ready_err.append(ex)
            ready_cb = _READY_FN(_ready)
        # ... and may take the per-Gaussian backward in several launches, told after each one which rows are final (include/radegs.h:
        # grad_chunks / grads_ready): its all-reduce of those rows then runs under the launches that follow
        chunks_cb, nchunks = None, 0
        if owner is not None and callable(getattr(owner, "grads_ready", None)) and int(getattr(owner, "grad_chunks", 0) or 0) > 1 \
                and getattr(owner, "early_grads", False):
            nchunks = int(owner.grad_chunks)

            def _chunk(_user, first, count, _owner=owner):
                try:
                    _owner.grads_ready(int(first), int(count))
                except Exception as ex:  # noqa: BLE001 -- must not unwind through the C frame
                    ready_err.append(ex)
            chunks_cb = _GRADS_READY_FN(_chunk)
        a = RadegsBwdArgs(ctypes.sizeof(RadegsBwdArgs), P, int(degree), M, int(R), W, H, _ptr(bg), _ptr(m3), _ptr(shs), _ptr(col), _ptr(al), _ptr(sc), _ptr(rot),
                          _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), float(scale_modifier), float(tan_fovx), float(tan_fovy),
                          float(kernel_size), _ptr(rad), _ptr(nm), _ptr(gb) if gb.numel() else None, _ptr(bb) if bb.numel() else None,
                          _ptr(ib) if ib.numel() else None, _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), _ptr(g[4]), _ptr(g[5]),
                          _ptr(g[6]), _ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                          _ptr(dL_dsh) if (M and dL_dsh is not None) else None, _ptr(dL_dscales), _ptr(dL_drotations),
                          int(bool(require_coord)), int(bool(require_depth)), int(bool(debug)), _ptr(drgb), int(bool(OPACITY_GRAD_INTENDED)),
                          ctypes.cast(ready_cb, ctypes.c_void_p) if ready_cb is not None else None, None,
                          nchunks, ctypes.cast(chunks_cb, ctypes.c_void_p) if chunks_cb is not None else None, None,
                          int(bool(KEEP_ACC)), int(acc_cached is not None))
        if acc_cached is not None:
            _ACC_LOCK.acquire()
        try:
            with torch.cuda.device(dev):
                rc = L.radegs_backward(ctypes.byref(a), acc.cb, None, _stream(dev))
            acc.release()
            if acc_cached is not None and (rc != 0 or acc.error is not None or ready_err):
                _ACC_SCRATCH.pop(akey, None)   # the scratch is in an unknown state: the next call starts from a fresh one
        finally:
            if acc_cached is not None:
                _ACC_LOCK.release()
        if acc.error is not None:
            raise acc.error
        if ready_err:
            raise ready_err[0]
        _check(rc, "radegs_backward")
        if KEEP_ACC:
            global LAST_ACC
            LAST_ACC = acc.tensor.view(torch.float32)
        if sc is None:  # precomputed covariance: scale/rotation grads are identically zero
            dL_dscales.zero_()
            dL_drotations.zero_()
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def backward_from_sums(sums, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                       tan_fovy, kernel_size, image_height, image_width, sh, degree, campos, geomBuffer, require_coord):
    """Test hook (radegs_backward_from_sums, include/radegs.h): the per-Gaussian half of the backward over caller-supplied per-Gaussian
    sums [P, 16 | 32] (the reference's render-kernel sums, its constant factors included).  Returns the 8-tuple of
    rasterize_gaussians_backward."""
    _require_gpu(means3D, "means3D")
    L = library()
    dev = means3D.device
    P = int(means3D.size(0))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    fo = dict(dtype=torch.float32, device=dev)
    rec = 32 if require_coord else 16
    sm = _f32(sums, "sums")
    if tuple(sm.shape) != (P, rec):
        raise RuntimeError(f"sums must be ({P}, {rec})")
    out = [torch.full(s, float("nan"), **fo) for s in ((P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, max(M, 1), 3), (P, 3), (P, 4))]
    dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = out
    m3, col = _f32(means3D, "means3D"), _f32(colors, "colors_precomp")
    sc, rot, cov = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
    vm, pm, cp, shs = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos"), _f32(sh, "shs")
    rad, gb = radii.contiguous(), geomBuffer.contiguous()
    a = RadegsBwdArgs(ctypes.sizeof(RadegsBwdArgs), P, int(degree), M, 0, int(image_width), int(image_height), None, _ptr(m3), _ptr(shs), _ptr(col), None, _ptr(sc), _ptr(rot),
                      _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), float(scale_modifier), float(tan_fovx), float(tan_fovy), float(kernel_size),
                      _ptr(rad), None, _ptr(gb), None, None, None, None, None, None, None, None, None,
                      _ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                      _ptr(dL_dsh) if M else None, _ptr(dL_dscales), _ptr(dL_drotations), int(bool(require_coord)), 0, 0, None,
                      int(bool(OPACITY_GRAD_INTENDED)), None, None, 0, None, None, 0, 0)
    with torch.cuda.device(dev):
        rc = L.radegs_backward_from_sums(ctypes.byref(a), _ptr(sm), _stream(dev))
    _check(rc, "radegs_backward_from_sums")
    if sc is None:
        dL_dscales.zero_()
        dL_drotations.zero_()
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, (dL_dsh if M else None), dL_dscales, dL_drotations


def sh_grad_from_views(means3D, campos_all, drgb_all, degree, M, scale=1.0, out=None):
    """dL_dsh[P,M,3] = scale * sum_v basis(dir_v) (x) drgb_all[v]  (radegs_sh_grad_from_views; campos_all [V,3], drgb_all [V,P,3])."""
    _require_gpu(means3D, "means3D")
    L = library()
    dev = means3D.device
    P, V = int(means3D.size(0)), int(drgb_all.size(0))
    m3, cp, dr = _f32(means3D, "means3D"), _f32(campos_all, "campos_all"), _f32(drgb_all, "drgb_all")
    if V and (tuple(drgb_all.shape) != (V, P, 3) or tuple(campos_all.shape) != (V, 3)):
        raise RuntimeError("campos_all must be (V,3) and drgb_all (V,P,3)")
    if out is None:
        out = torch.empty((P, int(M), 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.radegs_sh_grad_from_views(P, int(degree), int(M), V, _ptr(m3), _ptr(cp), _ptr(dr), float(scale), _ptr(out), _stream(dev))
    _check(rc, "radegs_sh_grad_from_views")
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    _require_gpu(means3D, "means3D")
    L = library()
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P != 0:
        m3, vm, pm = _f32(means3D, "means3D"), _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix")
        with torch.cuda.device(dev):
            rc = L.radegs_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), ctypes.c_void_p(present.data_ptr()), _stream(dev))
        _check(rc, "radegs_mark_visible")
    return present


def integrate_gaussians_to_points(background, points3D, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                  view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset,
                                  image_height, image_width, sh, degree, campos, prefiltered, debug):
    """IntegrateGaussiansToPointsCUDA (DGR/rasterize_points.cu:269-388).  `view2gaussian_precomp`, `subpixel_offset` and
    `prefiltered` are accepted and -- exactly as upstream's kernels do -- never read.  Returns the upstream 10-tuple
    (num_rendered, out_color[9,H,W], out_alpha_integrated[PN], out_color_integrated[PN,3], out_coordinate2d[PN,2],
    out_sdf[PN], radii[P], geomBuffer, binningBuffer, imgBuffer); the point-state buffer is dropped like upstream's."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if points3D.dim() != 2 or points3D.size(1) != 3:
        raise RuntimeError("points3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    _require_gpu(points3D, "points3D")
    L = library()
    dev = means3D.device
    P, PN, H, W = int(means3D.size(0)), int(points3D.size(0)), int(image_height), int(image_width)
    fo = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty((9, H, W), **fo)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    out_alpha_integrated = torch.empty((PN,), **fo)
    out_color_integrated = torch.empty((PN, 3), **fo)
    out_coordinate2d = torch.empty((PN, 2), **fo)
    out_sdf = torch.empty((PN,), **fo)
    geom, binning, img, pts = _Resizable(dev), _Resizable(dev), _Resizable(dev), _Resizable(dev)
    bg, m3, p3 = _f32(background, "bg"), _f32(means3D, "means3D"), _f32(points3D, "points3D")
    col, op = _f32(colors, "colors_precomp"), _f32(opacity, "opacities")
    sc, rot, cov = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
    vm, pm, cp, shs = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos"), _f32(sh, "shs")
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    a = RadegsIntegrateArgs(P, int(degree), M, PN, W, H, _ptr(bg), _ptr(m3), _ptr(shs), _ptr(col), _ptr(op), _ptr(sc), _ptr(rot),
                            _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), _ptr(p3), float(scale_modifier), float(tan_fovx), float(tan_fovy),
                            float(kernel_size), int(bool(debug)), _ptr(out_color), _ptr(out_alpha_integrated),
                            _ptr(out_color_integrated), _ptr(out_coordinate2d), _ptr(out_sdf), _ptr(radii))
    with torch.cuda.device(dev):
        rc = L.radegs_integrate(ctypes.byref(a), geom.cb, None, binning.cb, None, img.cb, None, pts.cb, None, _stream(dev))
    for r in (geom, binning, img, pts):
        r.release()
        if r.error is not None:
            raise r.error
    rendered = _check(rc, "radegs_integrate")
    global LAST_POINT_STATE
    LAST_POINT_STATE = pts.tensor if KEEP_ACC else None
    return (rendered, out_color, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf, radii, geom.tensor,
            binning.tensor, img.tensor)


def debug_export(name, dtype, numel, P, R, W, H, require_coord, geomBuffer, binningBuffer, imageBuffer):
    """Test hook: copy a private state array (see radegs_debug_export in include/radegs.h)."""
    L = library()
    dev = geomBuffer.device
    dst = torch.empty(numel, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        n = L.radegs_debug_export(name.encode(), int(P), int(R), int(W), int(H), int(bool(require_coord)),
                                  ctypes.c_void_p(geomBuffer.data_ptr()) if geomBuffer.numel() else None,
                                  ctypes.c_void_p(binningBuffer.data_ptr()) if binningBuffer.numel() else None,
                                  ctypes.c_void_p(imageBuffer.data_ptr()) if imageBuffer.numel() else None,
                                  ctypes.c_void_p(dst.data_ptr()), dst.numel() * dst.element_size(), _stream(dev))
    if n < 0:
        raise RuntimeError(L.radegs_last_error().decode())
    return dst


def binning_stats(reset=False):
    """(speculative forwards, forwards redone because the predicted capacity was too small) since the last reset."""
    calls, misses = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    library().radegs_binning_stats(ctypes.byref(calls), ctypes.byref(misses), int(bool(reset)))
    return int(calls.value), int(misses.value)


def reload_env():
    """Have the library read its RADEGS_* environment switches again (it reads them once, at first use)."""
    library().radegs_reload_env()


def last_forward_used_streams():
    """The blend formulation of this thread's last forward: True = sub-tile entry streams, False = tile-wide kernels, None = none yet."""
    v = library().radegs_last_forward_used_streams()
    return None if v < 0 else bool(v)


def profile_enable(on=True, only=None, every=1):
    """Per-stage HIP-event timing on the launch stream.  `only` = a stage name: record that stage alone (every recorded stage
    boundary costs ~10 us of stream bubble, so a timed run selects just the kernel it reports), `every` = n: every n-th launch."""
    L = library()
    sel = -1
    if only is not None:
        names = [L.radegs_profile_stage_name(i).decode() for i in range(L.radegs_profile_num_stages())]
        sel = names.index(only)
    L.radegs_profile_select(sel)
    L.radegs_profile_stride(int(every) if only is not None else 1)
    L.radegs_profile_enable(int(bool(on)))


def profile_collect():
    """{stage name: (total ms, launches)} since the last collect (HIP events on the launch stream)."""
    L = library()
    n = L.radegs_profile_num_stages()
    ms = (ctypes.c_float * n)()
    cnt = (ctypes.c_int * n)()
    L.radegs_profile_collect(ms, cnt, n)
    return {L.radegs_profile_stage_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}
