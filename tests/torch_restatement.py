"""TEST INFRASTRUCTURE: an independent float64 PyTorch restatement of the rasterizer's forward, differentiated by autograd.

Purpose: a second opinion on the oracle (oracle/radegs_oracle.cpp), written from the reference's forward only
-- per-Gaussian preprocess (DGR/cuda_rasterizer/forward.cu:23-74 SH, :77-264 computeCov2D, :270-304 computeCov3D, :307-423
preprocessCUDA, auxiliary.h:57-72,155-180) and a DENSE, un-tiled-in-memory blend (forward.cu:428-693) -- and never from the
reference's or the oracle's hand-derived backward (backward.cu:145-1016): every gradient here comes out of torch.autograd.
Matrices are written in plain column-vector mathematics, not in glm's column-major constructor order, and every map is a
vectorised tensor expression, so the two restatements share formulas but no code structure.

Small problems only (tens of Gaussians, a few tiles): everything is O(P x pixels) dense tensors.
"""
import numpy as np
import torch


def f32(x):
    """A literal the reference writes with an `f` suffix: its value is the float32 rounding, also when the arithmetic around it runs
    in float64 (the oracle's precision=64 mode keeps the same constants)."""
    return float(np.float32(x))


C0 = f32(0.28209479177387814)
C1 = f32(0.4886025119029199)
C2 = tuple(f32(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396))
C3 = tuple(f32(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                            1.445305721320277, -0.5900435899266435))
TILE = 16


def sh_to_rgb(deg, shs, means, campos):
    """computeColorFromSH (forward.cu:23-74): colour = max(SH(dir) + 0.5, 0), dir = normalize(mean - campos)."""
    d = means - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * shs[:, 0]
    if deg > 0:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6] + C2[3] * xz * shs[:, 7]
               + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10] + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
               + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
               + C3[5] * z * (xx - yy) * shs[:, 14] + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def cov3d_matrix(scales, rotations, scale_modifier):
    """computeCov3D (forward.cu:270-304): Sigma = R S^2 R^T with R from the quaternion (r,x,y,z) AS GIVEN (not normalised here)."""
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    S = torch.diag_embed(scale_modifier * scales)
    M = R @ S
    return M @ M.transpose(1, 2)


def sym_from6(c):
    return torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)


def render(means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg, *, shs=None, sh_degree=0, colors=None,
           scales=None, rotations=None, cov3D=None, scale_modifier=1.0, kernel_size=0.0, require_coord=False, require_depth=False,
           ndc_offset=None):
    """Returns (dict of the 7 image outputs + radii, aux).  `ndc_offset` (P,2), normally zeros with requires_grad: added to the
    projected NDC position, its gradient is what the operator returns as dL_dmeans2D[:, :2]."""
    dt = torch.float64
    P = means3D.shape[0]
    VM, PM = viewmatrix.to(dt), projmatrix.to(dt)          # stored transposed: row-vector convention
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    p_view = means3D @ VM[:3, :3] + VM[3, :3]               # transformPoint4x3
    visible = ~(p_view[:, 2] <= f32(0.2))                        # in_frustum, auxiliary.h:155-180
    p_hom = means3D @ PM[:3, :] + PM[3, :]
    p_w = 1.0 / (p_hom[:, 3] + f32(0.0000001))
    ndc = p_hom[:, :2] * p_w[:, None]
    if ndc_offset is not None:
        ndc = ndc + ndc_offset
    pix = ((ndc + 1.0) * torch.tensor([W, H], dtype=dt) - 1.0) * 0.5        # ndc2Pix

    Sigma = sym_from6(cov3D) if cov3D is not None else cov3d_matrix(scales, rotations, scale_modifier)
    Rv = VM[:3, :3].T                                        # world -> view rotation, column-vector convention
    # ---- EWA projection (forward.cu:85-121) ----
    limx, limy = f32(1.3) * tanfovx, f32(1.3) * tanfovy
    tz = p_view[:, 2]
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    u, v = tx / tz, ty / tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(-1, 2, 3)
    A = J @ Rv[None]
    cov = A @ Sigma @ A.transpose(1, 2)                      # 2x2 screen-space covariance before the filter
    c00, c01, c11 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = torch.clamp_min(c00 * c11 - c01 * c01, 1e-6)
    det1 = torch.clamp_min((c00 + kernel_size) * (c11 + kernel_size) - c01 * c01, 1e-6)
    coef = torch.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef = torch.where((det0 <= 1e-6) | (det1 <= 1e-6), torch.zeros_like(coef), coef)
    a, b, c = c00 + kernel_size, c01, c11 + kernel_size
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, f32(0.1)))
    lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, f32(0.1)))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2))).detach()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def tiles(lo, n):
        return torch.clamp(torch.trunc(lo / TILE), 0, n).to(torch.int64)
    pxd, pyd = pix[:, 0].detach(), pix[:, 1].detach()
    rx0, ry0 = tiles(pxd - radius, gx), tiles(pyd - radius, gy)
    rx1, ry1 = tiles(pxd + radius + TILE - 1, gx), tiles(pyd + radius + TILE - 1, gy)
    live = visible & (det != 0) & ((rx1 - rx0) * (ry1 - ry0) > 0)
    radii = torch.where(live, radius, torch.zeros_like(radius)).to(torch.int32)

    # ---- ray-space geometry (forward.cu:137-262) ----
    lam3, vec3 = torch.linalg.eigh(Sigma)
    well = lam3[:, 0] > 0.00000001
    Sig_inv = torch.linalg.inv(torch.where(well[:, None, None], Sigma, torch.eye(3, dtype=dt)[None].expand(P, 3, 3)))
    emin = vec3[:, :, 0]
    Vinv = torch.where(well[:, None, None], Sig_inv, emin[:, :, None] * emin[:, None, :])
    cam_inv = Rv[None] @ Vinv @ Rv.T[None]
    uvh = torch.stack([u, v, torch.ones_like(u)], 1)
    uvh_m = (cam_inv @ uvh[:, :, None])[:, :, 0]
    uvh_mn = uvh_m / uvh_m.norm(dim=1, keepdim=True)
    u2, v2, uv = u * u, v * v, u * v
    l = torch.sqrt(tx * tx + ty * ty + tz * tz)
    vbn = (uvh_mn * uvh).sum(1)
    w = uvh_mn / torch.clamp_min(vbn, f32(0.0000001))[:, None]
    plane0 = w[:, 0] * (v2 + 1) - w[:, 1] * uv - w[:, 2] * u
    plane1 = -w[:, 0] * uv + w[:, 1] * (u2 + 1) - w[:, 2] * v
    nl = u2 + v2 + 1
    cam_plane = torch.stack([(-(v2 + 1) * tz + plane0 * tx) / nl / fx, (uv * tz + plane1 * tx) / nl / fy,
                             (uv * tz + plane0 * ty) / nl / fx, (-(u2 + 1) * tz + plane1 * ty) / nl / fy,
                             (tx + plane0 * tz) / nl / fx, (ty + plane1 * tz) / nl / fy], 1)
    ray_plane = torch.stack([plane0 * l / nl / fx, plane1 * l / nl / fy], 1)
    fn = l / nl
    rn = torch.stack([-plane0 * fn, -plane1 * fn, -torch.ones_like(fn)], 1)
    cam_n = torch.stack([rn[:, 0] / tz + rn[:, 2] * tx / l, rn[:, 1] / tz + rn[:, 2] * ty / l,
                         -rn[:, 0] * tx / (tz * tz) - rn[:, 1] * ty / (tz * tz) + rn[:, 2] * tz / l], 1)
    normal = cam_n / cam_n.norm(dim=1, keepdim=True)
    bad = torch.isnan(uvh_mn[:, 0])
    cam_plane = torch.where(bad[:, None], torch.zeros_like(cam_plane), cam_plane)
    ray_plane = torch.where(bad[:, None], torch.zeros_like(ray_plane), ray_plane)
    normal = torch.where(bad[:, None], torch.zeros_like(normal), normal)

    ts = p_view.norm(dim=1)
    op = opacities.reshape(-1) * coef
    rgb = colors if colors is not None else sh_to_rgb(sh_degree, shs, means3D, campos.to(dt))

    # ---- dense blend (forward.cu:428-693): every pixel walks the Gaussians of ITS TILE in (depth, index) order ----
    order = sorted([i for i in range(P) if bool(live[i])], key=lambda i: (float(p_view[i, 2]), i))
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxf, pyf = xs.to(dt), ys.to(dt)
    tile_x, tile_y = xs // TILE, ys // TILE
    T = torch.ones((H, W), dtype=dt)
    done = torch.zeros((H, W), dtype=torch.bool)
    Cc = [torch.zeros((H, W), dtype=dt) for _ in range(3)]
    weight = torch.zeros((H, W), dtype=dt)
    Dep, mDep = torch.zeros((H, W), dtype=dt), torch.zeros((H, W), dtype=dt)
    Nn = [torch.zeros((H, W), dtype=dt) for _ in range(3)]
    Co = [torch.zeros((H, W), dtype=dt) for _ in range(3)]
    mCo = [torch.zeros((H, W), dtype=dt) for _ in range(3)]
    last = torch.zeros((H, W), dtype=torch.bool)
    G_list = []
    geo = require_coord or require_depth
    for i in order:
        in_rect = (tile_x >= rx0[i]) & (tile_x < rx1[i]) & (tile_y >= ry0[i]) & (tile_y < ry1[i])
        dx, dy = pix[i, 0] - pxf, pix[i, 1] - pyf
        power = -0.5 * (conic[i, 0] * dx * dx + conic[i, 2] * dy * dy) - conic[i, 1] * dx * dy
        G = torch.exp(power)
        G.retain_grad()
        G_list.append((i, G, dx, dy))
        alpha = torch.clamp_max(op[i] * G, f32(0.99))
        ok = in_rect & ~done & ~(power > 0) & ~(alpha < float(np.float32(1.0) / np.float32(255.0)))
        test_T = T * (1 - alpha)
        kill = ok & (test_T < f32(0.0001))
        act = ok & ~kill
        done = done | kill
        aT = torch.where(act, alpha * T, torch.zeros_like(T))
        for ch in range(3):
            Cc[ch] = Cc[ch] + rgb[i, ch] * aT
        med = act & (T > 0.5)
        if require_coord:
            for ch in range(3):
                cval = p_view[i, ch] + cam_plane[i, 2 * ch] * dx + cam_plane[i, 2 * ch + 1] * dy
                Co[ch] = Co[ch] + cval * aT
                mCo[ch] = torch.where(med, cval, mCo[ch])
        if require_depth:
            t = ts[i] + (ray_plane[i, 0] * dx + ray_plane[i, 1] * dy)
            Dep = Dep + t * aT
            mDep = torch.where(med, t, mDep)
        if geo:
            for ch in range(3):
                Nn[ch] = Nn[ch] + normal[i, ch] * aT
        weight = weight + aT
        T = torch.where(act, test_T, T)
        last = last | act
    pnx, pny = (pxf - W / 2.0) / fx, (pyf - H / 2.0) / fy
    ln = torch.sqrt(pnx * pnx + pny * pny + 1)
    bgd = bg.to(dt)
    out = {"color": torch.stack([Cc[ch] + T * bgd[ch] for ch in range(3)]), "alpha": weight[None], "radii": radii}
    zero_img = torch.zeros((H, W), dtype=dt)
    safe_w = torch.where(last, weight, torch.ones_like(weight))
    if require_coord:
        out["coord"] = torch.stack([torch.where(last, Co[ch] / safe_w, zero_img) for ch in range(3)])
        out["mcoord"] = torch.stack(mCo)
    else:
        out["coord"] = torch.zeros((3, H, W), dtype=dt)
        out["mcoord"] = torch.zeros((3, H, W), dtype=dt)
    if require_depth:
        out["depth"] = torch.where(last, Dep / ln / safe_w, zero_img)[None]
        out["mdepth"] = (mDep / ln)[None]
    else:
        out["depth"] = torch.zeros((1, H, W), dtype=dt)
        out["mdepth"] = torch.zeros((1, H, W), dtype=dt)
    if geo:
        nlen = torch.sqrt(Nn[0] * Nn[0] + Nn[1] * Nn[1] + Nn[2] * Nn[2])
        nlen = torch.where(last, torch.clamp_min(nlen, 1.0e-12), torch.ones_like(nlen))
        out["normal"] = torch.stack([torch.where(last, Nn[ch] / nlen, zero_img) for ch in range(3)])
    else:
        out["normal"] = torch.zeros((3, H, W), dtype=dt)
    aux = dict(G_list=G_list, conic=conic, op=op, live=live, well=well)
    return out, aux


def abs_grad_sum(aux, W, H, P):
    """dL_dmean2D[:, 2] (backward.cu:1005): sum over pixels of |dL/dG * dG/ddelx * W/2| + |dL/dG * dG/ddely * H/2|, from the
    per-pair dL/dG autograd left on the retained exp(power) tensors."""
    out = torch.zeros(P, dtype=torch.float64)
    conic = aux["conic"].detach()
    for i, G, dx, dy in aux["G_list"]:
        if G.grad is None:
            continue
        dL_dG = G.grad
        Gd = G.detach()
        gdx, gdy = Gd * dx.detach(), Gd * dy.detach()
        dG_ddelx = -gdx * conic[i, 0] - gdy * conic[i, 1]
        dG_ddely = -gdy * conic[i, 2] - gdx * conic[i, 1]
        out[i] = ((dL_dG * dG_ddelx * 0.5 * W).abs() + (dL_dG * dG_ddely * 0.5 * H).abs()).sum()
    return out
