"""Times one full regularised training iteration at C2 scale with every step on this repository's HIP ops (GPU box only):
3D-filter activations -> rasterizer fwd -> L1/SSIM + normal-consistency losses -> backward -> Adam."""
import math, os, sys, time
from collections import namedtuple
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "rade-gs_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import fused_adam, gaussian_model_ops as gmo, graphics_utils as gu, loss_utils as lu
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from synth_scene import make_config, to_device

dev = torch.device("cuda:0")
s = to_device(make_config("C2", filter3d=False), dev)
View = namedtuple("View", "image_width image_height FoVx FoVy")
view = View(s.W, s.H, 2 * math.atan(s.tanfovx), 2 * math.atan(s.tanfovy))
rs = GaussianRasterizationSettings(image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, kernel_size=s.kernel_size, bg=s.bg,
                                   scale_modifier=1.0, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, sh_degree=s.sh_degree, campos=s.campos,
                                   prefiltered=False, require_depth=True, require_coord=False, debug=False)
rast = GaussianRasterizer(rs)
P = s.means3D.shape[0]
filter_3D = (s.means3D.norm(dim=1, keepdim=True) / (s.W / (2 * s.tanfovx)) * math.sqrt(0.2)).contiguous()
params = dict(xyz=s.means3D.clone(), f_dc=s.shs[:, :1].clone(), f_rest=s.shs[:, 1:].clone(), op=torch.logit(s.opacities.clamp(1e-4, 1 - 1e-4)),
              sc=torch.log(s.scales), rot=s.rotations.clone())
params = {k: torch.nn.Parameter(v.contiguous()) for k, v in params.items()}
opt = fused_adam.Adam([{"params": [p], "lr": 1e-4, "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
target = torch.rand(3, s.H, s.W, device=dev)


def iteration():
    scales, opacity = gmo.scaling_n_opacity_with_3D_filter(params["sc"], params["op"], filter_3D)
    shs = torch.cat((params["f_dc"], params["f_rest"]), dim=1)
    out = rast(means3D=params["xyz"], means2D=torch.zeros_like(params["xyz"], requires_grad=True), shs=shs, colors_precomp=None,
               opacities=opacity, scales=scales, rotations=torch.nn.functional.normalize(params["rot"]), cov3D_precomp=None)
    loss = lu.photometric_loss(out[0], target, 0.2) + 0.05 * gu.normal_consistency_loss(view, out[7], out[4], out[5], 0.6)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    iteration()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    iteration()
torch.cuda.synchronize()
print(f"full training iteration at C2 (1M Gaussians, 1080p, SH3, depth+normal regulariser): {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
