"""diff_gaussian_rasterization -- MI355X (gfx950) build of RaDe-GS's rasterization operator.

Drop-in for the package the reference imports at gaussian_renderer/__init__.py:14:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Same public names, argument meaning, output tuple and error messages as
DGR/diff_gaussian_rasterization/__init__.py (settings tuple :171-186, module :188-237, autograd
function :44-169); the native side is the hand-written HIP library behind `_C`
(include/radegs.h).  Put `rade-gs_amd/` on PYTHONPATH (or `pip install -e rade-gs_amd`) and
train.py / render.py run unchanged.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

import os

if os.environ.get("RADEGS_BINDING", "ctypes") == "torch":
    # upstream's COMPILED `_C` module (DGR/ext.cpp:15-19) rebuilt over the C ABI: csrc/torch_binding/radegs_torch_binding.cpp, built by
    # `python rade-gs_amd/build.py --torch-binding`.  Same four entry points; this file uses nothing else of `_C`.
    from . import _C_torch as _C
else:
    from . import _C                 # the default: ctypes over the same C ABI (adds the view-parallel hooks and the test inspection calls)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    """Field order and names as upstream (DGR/diff_gaussian_rasterization/__init__.py:171-186)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    require_depth: bool
    require_coord: bool
    debug: bool


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug, dump_name, direction):
    """Runs a `_C` entry point; with debug=True inputs are snapshotted first and written to
    `dump_name` if the native call raises (the upstream debugging aid, :86-93 and :146-153)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {direction}. Please forward {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        native_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                       rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.image_height, rs.image_width, sh, rs.sh_degree,
                       rs.campos, rs.prefiltered, rs.require_coord, rs.require_depth, rs.debug)
        (num_rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geomBuffer, binningBuffer,
         imgBuffer) = _call_native(_C.rasterize_gaussians, native_args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geomBuffer, binningBuffer,
                              imgBuffer, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, coord, mcoord, depth, mdepth, alpha, normal

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_coord, grad_mcoord, grad_depth, grad_mdepth, grad_alpha, grad_normal):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         alpha) = ctx.saved_tensors
        native_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                       rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, grad_color, grad_coord, grad_mcoord, grad_depth,
                       grad_mdepth, grad_alpha, grad_normal, normal, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered,
                       binningBuffer, imgBuffer, alpha, rs.require_coord, rs.require_depth, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_native(_C.rasterize_gaussians_backward, native_args, rs.debug, "snapshot_bw.dump", "backward")
        # one gradient per forward input (the settings tuple gets None), upstream :157-169
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the Gaussians that pass the camera's near-plane test (upstream :193-202)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        """Returns (color[3,H,W], radii[P], coord[3,H,W], median_coord[3,H,W], depth[1,H,W],
        median_depth[1,H,W], alpha[1,H,W], normal[3,H,W]) -- upstream :101,204-237."""
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        def absent():
            return torch.Tensor([])  # empty tensor stands for "not provided" across the native boundary

        shs = absent() if shs is None else shs
        colors_precomp = absent() if colors_precomp is None else colors_precomp
        scales = absent() if scales is None else scales
        rotations = absent() if rotations is None else rotations
        cov3D_precomp = absent() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)

    def integrate(self, points3D, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None, view2gaussian_precomp=None):
        """GOF point integration used by mesh_extract_tetrahedra.py through gaussian_renderer.integrate (upstream :239-306).
        Returns (color[9,H,W], alpha_integrated[PN], color_integrated[PN,3], point_coordinate[PN,2], point_sdf[PN],
        radii[P]).  Not differentiable (upstream calls the native function outside any autograd.Function).  Like
        upstream, the 2D filter is switched off here (kernel_size 0.0) and `means2D` is not used."""
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        def absent(t):
            return torch.Tensor([]) if t is None else t

        args = (rs.bg, points3D, means3D, absent(colors_precomp), opacities, absent(scales), absent(rotations), rs.scale_modifier,
                absent(cov3D_precomp), absent(view2gaussian_precomp), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                0.0,    # kernel_size: hard-coded upstream (:283)
                None,   # subpixel_offset: upstream allocates an (H,W,2) zero tensor its kernels never read
                rs.image_height, rs.image_width, absent(shs), rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        # (upstream's debug branch unpacks 8 of the 10 returned values and cannot work; both branches unpack all 10 here)
        out = _call_native(_C.integrate_gaussians_to_points, args, rs.debug, "snapshot_fw.dump", "forward")
        _num_rendered, color, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii = out[:7]
        return color, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii
