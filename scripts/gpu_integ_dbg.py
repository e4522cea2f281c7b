"""Debug aid: locate integrate() mismatches between the HIP path and the oracle (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "rade-gs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from synth_scene import make_scene, to_device
from util import oracle_for
from test_hostcheck import _flat_scene
from test_gpu_integrate import _points
import diff_gaussian_rasterization._C as C
from gpu_util import settings_for

s = _flat_scene(make_scene(3000, 160, 120, sh_degree=1, mu_px=3.0, seed=33, kernel_size=0.0, pose="random", require_coord=False,
                           require_depth=True), frac=0.4, seed=4)
pts = _points(s, 15000, 3)
o = oracle_for(s)
ref = o.integrate(pts)
P = 3000
dev = torch.device("cuda:0")
d = to_device(s, dev)
rs = settings_for(s, dev)
e = torch.Tensor([])
C.KEEP_ACC = True
st = C.integrate_gaussians_to_points(rs.bg, torch.from_numpy(pts).to(dev), d.means3D, e, d.opacities, d.scales, d.rotations, 1.0, e, e,
                                     rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, 0.0, None, s.H, s.W, d.shs, s.sh_degree,
                                     rs.campos, False, False)
torch.cuda.synchronize()
ps = C.LAST_POINT_STATE
rec = ps[: P * 32].view(torch.float32).cpu().numpy().reshape(P, 8)
icr, cond, radii = o.get("invraycov", (P, 6)), o.get("condition"), o.get("radii")
vis = radii > 0
print("icr bit-equal on visible rows:", np.array_equal(rec[vis, :6].view(np.uint32), icr[vis].view(np.uint32)),
      " cond equal:", np.array_equal(rec[vis, 6].astype(np.uint8), cond[vis]))
bad = np.nonzero(np.any(rec[vis, :6].view(np.uint32) != icr[vis].view(np.uint32), axis=1))[0]
print("rows differing:", len(bad), "cond of those:", cond[vis][bad][:20])
for r in bad[:5]:
    print(rec[vis][r], icr[vis][r])
a, b = st[2].cpu().numpy(), ref[1]
mis = np.nonzero(np.abs(a - b) > 1e-4)[0]
print("alpha mismatches:", len(mis), "of", len(a))
coord = ref[3]
for q in mis[:8]:
    print("point", q, "got", a[q], "ref", b[q], "pix", coord[q], "sdf got/ref", st[5][q].item(), ref[4][q])
