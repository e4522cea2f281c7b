// TEST INFRASTRUCTURE (oracle/_ref): C entry points around the reference's own CudaRasterizer::Rasterizer, compiled for the host.
//
// This file plays the role of DGR/rasterize_points.cu (the torch binding, which cannot be built without torch's C++ headers
// and CUDA): it allocates what that file allocates -- outputs and gradients zero-filled (rasterize_points.cu:70-77,185-198),
// `out_alpha_integrated` = 1 and `out_sdf` = -1000 for integrate (:315-318), three (five) growable scratch buffers handed over as
// std::function<char*(size_t)> -- and calls Rasterizer::forward / backward / integrate / markVisible with the arguments in the
// order that file passes them (:98-130, :202-246, :347-388).  Absent tensors are null pointers, as an empty tensor's data_ptr()
// is.  Everything numerical is the reference's code (forward.cu, backward.cu, rasterizer_impl.cu, auxiliary.h), compiled from
// /root/reference by oracle/build_ref.py; see oracle/ref_shim/cuda_on_host.h for what stands in for CUDA.
//
// One deliberate deviation: rasterize_points.cu:320 sizes `condition` by the number of query POINTS while the kernel indexes it by
// GAUSSIAN (forward.cu:369); here it holds max(P, PN) entries so that the host run cannot write out of bounds.
#include <cuda_on_host.h>
#include <map>
#include <string>

#include "config.h"
#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "auxiliary.h"

uint32_t getHigherMsb(uint32_t n);   // rasterizer_impl.cu:35-50 (external linkage there)

namespace {
using CudaRasterizer::BinningState;
using CudaRasterizer::GeometryState;
using CudaRasterizer::ImageState;
using CudaRasterizer::PointState;

struct Ctx {
  int P, D, M, W, H;
  std::vector<float> bg, means3D, shs, colors, opac, scales, rots, cov3Dp, view, proj, campos;
  float scale_modifier, tanfovx, tanfovy, kernel_size;
  bool req_coord, req_depth, prefiltered;
  std::vector<char> geom, binning, img, point, point_binning;
  int R = 0, PN = 0, NI = 0;
  std::vector<float> out_color, out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal;
  std::vector<int> radii;
  std::vector<float> dL_dmeans3D, dL_dview_points, dL_dmeans2D, dL_dcolors, dL_dts, dL_dcamera_planes, dL_dray_planes, dL_dnormals, dL_dconic,
      dL_dopacity, dL_dopacity_raw, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations;
  std::vector<float> out9, accum_alpha, invraycov, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf;
  std::vector<unsigned char> condition;
};

const float* p(const std::vector<float>& v) { return v.empty() ? nullptr : v.data(); }
void take(std::vector<float>& dst, const float* src, size_t n) { if (src && n) dst.assign(src, src + n); }
std::function<char*(size_t)> resizer(std::vector<char>& v) {
  return [&v](size_t n) { v.assign(n, 0); return v.data(); };
}
}  // namespace

extern "C" {

void ref_set_exp_fn(void* fn) { cuda_on_host::exp_fn() = (cuda_on_host::exp_fn_t)fn; }
void ref_set_num_threads(int n) { cuda_on_host::num_threads() = n; }
unsigned ref_higher_msb(unsigned n) { return getHigherMsb(n); }

// forward.cu:126-133: the reference's own known answer for the matrix convention, mat3(1..9) * (1,1,1) = (12, 15, 18)
void ref_kat_mat3(float* out3) {
  glm::mat3 m = glm::mat3(1, 2, 3, 4, 5, 6, 7, 8, 9);
  glm::vec3 v = {1, 1, 1};
  glm::vec3 r = m * v;
  out3[0] = r[0]; out3[1] = r[1]; out3[2] = r[2];
}

// auxiliary.h:217-401 on a symmetric matrix given as (xx, xy, xz, yy, yz, zz); V9 receives the eigenvectors, column i at V9[3i..3i+2]
int ref_sym_eigen3(const float* s, float* ev3, float* V9) {
  glm::mat3 A = glm::mat3(s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]);
  glm::vec3 ev;
  glm::mat3 V;
  int D = glm_modification::findEigenvaluesSymReal(A, ev, V);
  for (int i = 0; i < 3; i++) { ev3[i] = ev[i]; for (int j = 0; j < 3; j++) V9[3 * i + j] = V[i][j]; }
  return D;
}

void* ref_create(int P, int D, int M, int W, int H, const float* bg, const float* means3D, const float* shs, const float* colors,
                 const float* opac, const float* scales, const float* rots, const float* cov3Dp, const float* view, const float* proj,
                 const float* campos, float scale_modifier, float tanfovx, float tanfovy, float kernel_size, int req_coord, int req_depth,
                 int prefiltered) {
  Ctx* c = new Ctx();
  c->P = P; c->D = D; c->M = M; c->W = W; c->H = H;
  take(c->bg, bg, 3); take(c->means3D, means3D, size_t(P) * 3); take(c->shs, shs, size_t(P) * M * 3); take(c->colors, colors, size_t(P) * 3);
  take(c->opac, opac, P); take(c->scales, scales, size_t(P) * 3); take(c->rots, rots, size_t(P) * 4); take(c->cov3Dp, cov3Dp, size_t(P) * 6);
  take(c->view, view, 16); take(c->proj, proj, 16); take(c->campos, campos, 3);
  c->scale_modifier = scale_modifier; c->tanfovx = tanfovx; c->tanfovy = tanfovy; c->kernel_size = kernel_size;
  c->req_coord = req_coord; c->req_depth = req_depth; c->prefiltered = prefiltered;
  return c;
}
void ref_destroy(void* h) { delete (Ctx*)h; }

int ref_forward(void* h) {
  Ctx* c = (Ctx*)h;
  const size_t N = size_t(c->W) * c->H;
  c->out_color.assign(3 * N, 0.f); c->out_depth.assign(N, 0.f); c->out_mdepth.assign(N, 0.f); c->out_coord.assign(3 * N, 0.f);
  c->out_mcoord.assign(3 * N, 0.f); c->out_alpha.assign(N, 0.f); c->out_normal.assign(3 * N, 0.f); c->radii.assign(c->P, 0);
  c->geom.clear(); c->binning.clear(); c->img.clear();
  c->R = 0;
  if (c->P != 0)
    c->R = CudaRasterizer::Rasterizer::forward(resizer(c->geom), resizer(c->binning), resizer(c->img), c->P, c->D, c->M, p(c->bg), c->W, c->H,
                                               p(c->means3D), p(c->shs), p(c->colors), p(c->opac), p(c->scales), c->scale_modifier, p(c->rots),
                                               p(c->cov3Dp), p(c->view), p(c->proj), p(c->campos), c->tanfovx, c->tanfovy, c->kernel_size,
                                               c->prefiltered, c->out_color.data(), c->out_coord.data(), c->out_mcoord.data(), c->out_depth.data(),
                                               c->out_mdepth.data(), c->out_alpha.data(), c->out_normal.data(), c->radii.data(), c->req_coord,
                                               c->req_depth, false);
  return c->R;
}

void ref_backward(void* h, const float* dL_dcolor, const float* dL_dcoord, const float* dL_dmcoord, const float* dL_ddepth, const float* dL_dmdepth,
                  const float* dL_dalpha, const float* dL_dnormal) {
  Ctx* c = (Ctx*)h;
  const size_t P = c->P;
  c->dL_dmeans3D.assign(P * 3, 0.f); c->dL_dview_points.assign(P * 3, 0.f); c->dL_dmeans2D.assign(P * 3, 0.f); c->dL_dcolors.assign(P * 3, 0.f);
  c->dL_dts.assign(P, 0.f); c->dL_dcamera_planes.assign(P * 6, 0.f); c->dL_dray_planes.assign(P * 2, 0.f); c->dL_dnormals.assign(P * 3, 0.f);
  c->dL_dconic.assign(P * 4, 0.f); c->dL_dopacity.assign(P, 0.f); c->dL_dcov3D.assign(P * 6, 0.f); c->dL_dsh.assign(P * c->M * 3, 0.f);
  c->dL_dscales.assign(P * 3, 0.f); c->dL_drotations.assign(P * 4, 0.f);
  c->dL_dopacity_raw.assign(P, 0.f);
  if (P == 0) return;
  // the first kernel of Rasterizer::backward is the render backward (rasterizer_impl.cu:500-541): what it leaves in dL_dopacity is
  // the raw per-Gaussian sum; computeCov2DCUDA, two launches later, multiplies it by the opacity-compensation factor in place
  int launches = 0;
  cuda_on_host::post_launch_hook() = [&]() { if (launches++ == 0) c->dL_dopacity_raw = c->dL_dopacity; };
  struct Unhook { ~Unhook() { cuda_on_host::post_launch_hook() = nullptr; } } unhook;
  CudaRasterizer::Rasterizer::backward(c->P, c->D, c->M, c->R, p(c->bg), c->W, c->H, p(c->means3D), p(c->shs), p(c->colors), c->out_alpha.data(),
                                       p(c->scales), c->scale_modifier, p(c->rots), p(c->cov3Dp), p(c->view), p(c->proj), p(c->campos), c->tanfovx,
                                       c->tanfovy, c->kernel_size, c->radii.data(), c->out_normal.data(), c->geom.data(), c->binning.data(),
                                       c->img.data(), dL_dcolor, dL_dcoord, dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha, dL_dnormal,
                                       c->dL_dmeans2D.data(), c->dL_dview_points.data(), c->dL_dconic.data(), c->dL_dopacity.data(),
                                       c->dL_dcolors.data(), c->dL_dts.data(), c->dL_dcamera_planes.data(), c->dL_dray_planes.data(),
                                       c->dL_dnormals.data(), c->dL_dmeans3D.data(), c->dL_dcov3D.data(), p(c->dL_dsh) ? c->dL_dsh.data() : nullptr,
                                       c->dL_dscales.data(), c->dL_drotations.data(), c->req_coord, c->req_depth, false);
}

int ref_integrate(void* h, int PN, const float* points3D, const float* subpixel_offset) {
  Ctx* c = (Ctx*)h;
  const size_t N = size_t(c->W) * c->H;
  c->PN = PN;
  c->out9.assign(9 * N, 0.f); c->accum_alpha.assign(N, 0.f); c->radii.assign(c->P, 0);
  c->out_alpha_integrated.assign(PN, 1.f); c->out_color_integrated.assign(size_t(PN) * 3, 0.f); c->out_coordinate2d.assign(size_t(PN) * 2, 0.f);
  c->out_sdf.assign(PN, -1000.f); c->invraycov.assign(size_t(c->P) * 6, 0.f);
  c->condition.assign(std::max(c->P, PN), 0);
  c->geom.clear(); c->binning.clear(); c->img.clear(); c->point.clear(); c->point_binning.clear();
  c->R = 0;
  static_assert(sizeof(bool) == 1, "bool");
  if (c->P != 0 && PN != 0)
    c->R = CudaRasterizer::Rasterizer::integrate(resizer(c->geom), resizer(c->binning), resizer(c->img), resizer(c->point), resizer(c->point_binning),
                                                 PN, c->P, c->D, c->M, p(c->bg), c->W, c->H, points3D, p(c->means3D), p(c->shs), p(c->colors),
                                                 p(c->opac), p(c->scales), c->scale_modifier, p(c->rots), p(c->cov3Dp), nullptr, p(c->view), p(c->proj),
                                                 p(c->campos), c->tanfovx, c->tanfovy, c->kernel_size, subpixel_offset, c->prefiltered,
                                                 c->out9.data(), c->accum_alpha.data(), c->invraycov.data(), c->radii.data(),
                                                 c->out_alpha_integrated.data(), c->out_color_integrated.data(), c->out_coordinate2d.data(),
                                                 c->out_sdf.data(), (bool*)c->condition.data(), false);
  return c->R;
}

void ref_mark_visible(int P, const float* means3D, const float* view, const float* proj, unsigned char* present) {
  if (P == 0) return;
  CudaRasterizer::Rasterizer::markVisible(P, (float*)means3D, (float*)view, (float*)proj, (bool*)present);
}

// Copies the named array into dst (if dst != null and nbytes suffices); returns its size in bytes, -1 for an unknown name.
long long ref_get(void* h, const char* name_c, void* dst, long long nbytes) {
  Ctx* c = (Ctx*)h;
  const std::string name(name_c);
  const size_t P = c->P, N = size_t(c->W) * c->H, R = c->R;
  const size_t tiles = size_t((c->W + BLOCK_X - 1) / BLOCK_X) * ((c->H + BLOCK_Y - 1) / BLOCK_Y);
  const void* src = nullptr;
  size_t n = 0;
  auto vec = [&](const auto& v) { src = v.data(); n = v.size() * sizeof(v[0]); };
  if (!c->geom.empty() && P) {
    char* g = c->geom.data();
    GeometryState gs = GeometryState::fromChunk(g, P);
    if (name == "depths") { src = gs.depths; n = P * 4; }
    else if (name == "camera_planes") { src = gs.camera_planes; n = P * 24; }
    else if (name == "ray_planes") { src = gs.ray_planes; n = P * 8; }
    else if (name == "ts") { src = gs.ts; n = P * 4; }
    else if (name == "normals") { src = gs.normals; n = P * 12; }
    else if (name == "clamped") { src = gs.clamped; n = P * 3; }
    else if (name == "means2D") { src = gs.means2D; n = P * 8; }
    else if (name == "view_points") { src = gs.view_points; n = P * 12; }
    else if (name == "cov3D") { src = gs.cov3D; n = P * 24; }
    else if (name == "conic_opacity") { src = gs.conic_opacity; n = P * 16; }
    else if (name == "rgb") { src = gs.rgb; n = P * 12; }
    else if (name == "tiles_touched") { src = gs.tiles_touched; n = P * 4; }
    else if (name == "point_offsets") { src = gs.point_offsets; n = P * 4; }
  }
  if (!src && !c->binning.empty()) {
    char* b = c->binning.data();
    BinningState bs = BinningState::fromChunk(b, R);
    if (name == "keys_sorted") { src = bs.point_list_keys; n = R * 8; }
    else if (name == "point_list") { src = bs.point_list; n = R * 4; }
  }
  if (!src && !c->img.empty()) {
    char* i = c->img.data();
    ImageState is = ImageState::fromChunk(i, N);
    if (name == "ranges") { src = is.ranges; n = tiles * 8; }
    else if (name == "point_ranges") { src = is.point_ranges; n = tiles * 8; }
    else if (name == "n_contrib") { src = is.n_contrib; n = N * 8; }
    else if (name == "accum_coord") { src = is.accum_coord; n = N * 12; }
    else if (name == "accum_depth") { src = is.accum_depth; n = N * 4; }
    else if (name == "normal_length") { src = is.normal_length; n = N * 4; }
  }
  if (!src && !c->point.empty() && c->PN) {
    char* q = c->point.data();
    PointState ps = PointState::fromChunk(q, c->PN);
    if (name == "points2D") { src = ps.points2D; n = size_t(c->PN) * 8; }
    else if (name == "point_depths") { src = ps.depths; n = size_t(c->PN) * 4; }
    else if (name == "point_tiles_touched") { src = ps.tiles_touched; n = size_t(c->PN) * 4; }
  }
  if (!src) {
#define V(x) else if (name == #x) vec(c->x)
    if (false) {}
    V(out_color); V(out_coord); V(out_mcoord); V(out_depth); V(out_mdepth); V(out_alpha); V(out_normal); V(radii);
    V(dL_dmeans3D); V(dL_dview_points); V(dL_dmeans2D); V(dL_dcolors); V(dL_dts); V(dL_dcamera_planes); V(dL_dray_planes); V(dL_dnormals);
    V(dL_dconic); V(dL_dopacity); V(dL_dopacity_raw); V(dL_dcov3D); V(dL_dsh); V(dL_dscales); V(dL_drotations);
    V(out9); V(accum_alpha); V(invraycov); V(out_alpha_integrated); V(out_color_integrated); V(out_coordinate2d); V(out_sdf); V(condition);
#undef V
    else return -1;
  }
  if (dst && nbytes >= (long long)n && n) memcpy(dst, src, n);
  return (long long)n;
}

}  // extern "C"
