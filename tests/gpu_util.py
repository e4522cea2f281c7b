"""Runs the HIP operator on a synthetic Scene through the public drop-in API (test infrastructure)."""
import numpy as np
import torch

from synth_scene import Scene, to_device


def settings_for(s: Scene, device, debug=False, scale_modifier=1.0):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, kernel_size=s.kernel_size, bg=s.bg.to(device),
        scale_modifier=scale_modifier, viewmatrix=s.viewmatrix.to(device), projmatrix=s.projmatrix.to(device), sh_degree=s.sh_degree,
        campos=s.campos.to(device), prefiltered=False, require_depth=s.require_depth, require_coord=s.require_coord, debug=debug)


class HipRun:
    """forward (+ optional backward) of one view; keeps the private state for index checks."""

    def __init__(self, s: Scene, device="cuda:0", colors=None, cov3D=None, debug=False, scale_modifier=1.0):
        import diff_gaussian_rasterization._C as C
        self.C = C
        self.s = s
        self.dev = torch.device(device)
        d = to_device(s, self.dev)
        self.P = s.means3D.shape[0]
        self.means3D = d.means3D.clone().requires_grad_(True)
        self.means2D = torch.zeros_like(d.means3D, requires_grad=True)
        self.opacities = d.opacities.clone().requires_grad_(True)
        self.shs = None if colors is not None else d.shs.clone().requires_grad_(True)
        self.colors = None if colors is None else colors.to(self.dev).clone().requires_grad_(True)
        self.scales = None if cov3D is not None else d.scales.clone().requires_grad_(True)
        self.rotations = None if cov3D is not None else d.rotations.clone().requires_grad_(True)
        self.cov3D = None if cov3D is None else cov3D.to(self.dev).clone().requires_grad_(True)
        self.rs = settings_for(s, self.dev, debug, scale_modifier)
        self.state = None
        C.reload_env()   # the library reads its RADEGS_* switches once; tests flip them (monkeypatch.setenv) before building a HipRun

    def forward(self):
        """through the autograd operator; also captures the state buffers via a direct `_C` call"""
        from diff_gaussian_rasterization import GaussianRasterizer
        self.C.reload_env()
        r = GaussianRasterizer(self.rs)
        self.out = r(self.means3D, self.means2D, self.opacities, shs=self.shs, colors_precomp=self.colors, scales=self.scales,
                     rotations=self.rotations, cov3D_precomp=self.cov3D)
        return self.out

    def forward_native(self):
        e = torch.Tensor([])
        rs = self.rs
        self.C.reload_env()
        res = self.C.rasterize_gaussians(rs.bg, self.means3D.detach(), e if self.colors is None else self.colors.detach(),
                                         self.opacities.detach(), e if self.scales is None else self.scales.detach(),
                                         e if self.rotations is None else self.rotations.detach(), rs.scale_modifier,
                                         e if self.cov3D is None else self.cov3D.detach(), rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                         rs.tanfovy, rs.kernel_size, rs.image_height, rs.image_width,
                                         e if self.shs is None else self.shs.detach(), rs.sh_degree, rs.campos, False,
                                         rs.require_coord, rs.require_depth, rs.debug)
        self.state = res
        return res

    def export(self, name, dtype, numel):
        R, geom, binning, img = self.state[0], self.state[9], self.state[10], self.state[11]
        return self.C.debug_export(name, dtype, numel, self.P, R, self.s.W, self.s.H, self.s.require_coord, geom, binning, img).cpu().numpy()

    def backward(self, g):
        color, radii, coord, mcoord, depth, mdepth, alpha, normal = self.out
        dev = self.dev
        self.C.reload_env()
        loss = (color * g["color"].to(dev)).sum() + (alpha * g["alpha"].to(dev)).sum()
        loss = loss + (coord * g["coord"].to(dev)).sum() + (mcoord * g["mcoord"].to(dev)).sum()
        loss = loss + (depth * g["depth"].to(dev)).sum() + (mdepth * g["mdepth"].to(dev)).sum()
        loss = loss + (normal * g["normal"].to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize(dev)

        def n(t):
            return None if t is None or t.grad is None else t.grad.detach().cpu().numpy()

        return dict(dL_dmeans2D=n(self.means2D), dL_dcolors=n(self.colors), dL_dopacity=n(self.opacities), dL_dmeans3D=n(self.means3D),
                    dL_dcov3D=n(self.cov3D), dL_dsh=n(self.shs), dL_dscales=n(self.scales), dL_drotations=n(self.rotations))


def outputs_numpy(out):
    return [o.detach().cpu().numpy() for o in out]


# ---- the per-Gaussian sums between the two halves of the backward -------------------------------------------------
def reference_sums(get, P, coord, raw_opacity="dL_dopacity_raw"):
    """[P, 16 | 32] float32 record of the render kernel's per-Gaussian sums in the order radegs_backward_from_sums takes them
    (include/radegs.h), from a checker's arrays: `get` = Ref.get (compiled reference; raw_opacity 'dL_dopacity_raw') or Oracle.get
    (raw_opacity 'acc_dopacity')."""
    rec = np.zeros((P, 32 if coord else 16), np.float32)
    rec[:, 0:3] = get("dL_dcolors").reshape(P, 3)
    rec[:, 3] = get("dL_dts").reshape(P)
    rec[:, 4:6] = get("dL_dray_planes").reshape(P, 2)
    rec[:, 6:9] = get("dL_dnormals").reshape(P, 3)
    rec[:, 9:12] = get("dL_dmeans2D").reshape(P, 3)
    dc = get("dL_dconic").reshape(P, 4)
    rec[:, 12], rec[:, 13], rec[:, 14] = dc[:, 0], dc[:, 1], dc[:, 3]
    rec[:, 15] = get(raw_opacity).reshape(P)
    if coord:
        rec[:, 16:19] = get("dL_dview_points").reshape(P, 3)
        rec[:, 19:25] = get("dL_dcamera_planes").reshape(P, 6)
    return rec


def hip_sums_as_reference(acc, s):
    """The blend backward's accumulator records (_C.LAST_ACC, [P, 16 | 32]) with the constant factors it leaves to the per-Gaussian
    kernel applied in float64 (1/focal on the plane sums, W/2 and H/2 on mean2D: backward.cu:917-922, 939-940, 1002-1003), i.e. in the
    reference's units, for comparison with reference_sums()."""
    a = acc.detach().cpu().numpy().astype(np.float64).reshape(s.means3D.shape[0], -1).copy()
    fx, fy = s.W / (2.0 * s.tanfovx), s.H / (2.0 * s.tanfovy)
    a[:, 4] /= np.float32(fx); a[:, 5] /= np.float32(fy)
    a[:, 9] *= 0.5 * s.W; a[:, 10] *= 0.5 * s.H
    if a.shape[1] == 32:
        a[:, 19:25:2] /= np.float32(fx); a[:, 20:25:2] /= np.float32(fy)
    return a


def backward_from_sums(h, sums):
    """HipRun `h` after forward_native(): the per-Gaussian half of the backward over `sums` (numpy [P, rec]) -> dict of numpy gradients."""
    import torch
    C, rs = h.C, h.rs
    C.reload_env()
    e = torch.Tensor([])
    st = h.state
    out = C.backward_from_sums(torch.from_numpy(np.ascontiguousarray(sums, dtype=np.float32)).to(h.dev), h.means3D.detach(), st[8],
                               e if h.colors is None else h.colors.detach(), e if h.scales is None else h.scales.detach(),
                               e if h.rotations is None else h.rotations.detach(), rs.scale_modifier,
                               e if h.cov3D is None else h.cov3D.detach(), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                               rs.kernel_size, rs.image_height, rs.image_width, e if h.shs is None else h.shs.detach(), rs.sh_degree,
                               rs.campos, st[9], rs.require_coord)
    torch.cuda.synchronize(h.dev)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    return {k: (None if t is None else t.cpu().numpy()) for k, t in zip(names, out)}
