// radegs_torch_binding.cpp -- the reference's pybind `_C` module, re-built over the C ABI of include/radegs.h.
//
// Upstream's operator package binds four functions (DGR/ext.cpp:15-19):
//     rasterize_gaussians            DGR/rasterize_points.cu:35-131    -> radegs_forward
//     rasterize_gaussians_backward   DGR/rasterize_points.cu:134-246   -> radegs_backward
//     mark_visible                   DGR/rasterize_points.cu:248-267   -> radegs_mark_visible
//     integrate_gaussians_to_points  DGR/rasterize_points.cu:269-388   -> radegs_integrate
// with torch::Tensor arguments in a fixed positional order.  This file is the one a maintainer who wants to KEEP that compiled
// module (instead of the ctypes binding this repository ships as `_C.py`) would put in the place of rasterize_points.cu: same
// function names, argument order, return tuples, tensor shapes / dtypes and error text, so DGR/diff_gaussian_rasterization/
// __init__.py imports it unchanged.  It is plain host C++ (no device code): torch for tensors, the current stream and the caching
// allocator; everything else goes through libradegs_hip.so.  Built in-tree by rade-gs_amd/build.py as
// diff_gaussian_rasterization/_C_torch*.so; tests/test_torch_binding.py runs it against the ctypes binding bit for bit.
//
// Deliberate differences from upstream's shim (all inside its contract):
//   * outputs are torch::empty where the native side writes every element (the reference zero-fills 14 gradient tensors and 7 maps);
//     maps the flags do not produce come back all-zero because the forward zero-fills them (include/radegs.h);
//   * the image-state buffer is handed out with a deleter that tells the library when it dies (radegs_forget_image);
//   * a call that launches nothing (P == 0) returns zeros, like upstream.
#include <torch/extension.h>

// PyTorch-ROCm presents its devices as DeviceType::CUDA; the plain c10::hip guards / streams / allocator refuse that type, the
// "MasqueradingAsCUDA" variants are the ones a ROCm build of an extension uses (they are what torch's hipify maps c10::cuda::* to)
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "radegs.h"

namespace {

using torch::Tensor;

void require_gpu_f32(const Tensor& t, const char* name) {
  if (t.numel() == 0) return;   // an empty tensor is "not provided" (the reference's convention)
  TORCH_CHECK(t.is_cuda(), "diff_gaussian_rasterization (MI355X build): `", name, "` must be a GPU tensor -- this operator has no CPU implementation");
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, "`", name, "` must be float32");
}

// contiguous float view of an input (kept alive by the holder for the duration of the call); empty -> NULL
struct In {
  Tensor keep;
  const float* p = nullptr;
  In(const Tensor& t, const char* name) {
    require_gpu_f32(t, name);
    if (t.numel() != 0) { keep = t.contiguous(); p = keep.data_ptr<float>(); }
  }
};

// A state buffer the native side sizes through a callback (the resize lambdas of rasterize_points.cu:27-33).  Memory comes
// from torch's caching allocator; an image-state buffer tells the library when it is released.
struct StateBuffer {
  c10::Device device;
  bool image;
  void* ptr = nullptr;
  size_t bytes = 0;
  StateBuffer(c10::Device d, bool is_image) : device(d), image(is_image) {}
  StateBuffer(const StateBuffer&) = delete;
  StateBuffer& operator=(const StateBuffer&) = delete;
  ~StateBuffer() { drop(); }
  void drop() {
    if (!ptr) return;
    if (image) radegs_forget_image(ptr);
    c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::raw_delete(ptr);
    ptr = nullptr; bytes = 0;
  }
  static void* grow(void* user, size_t nbytes) {
    auto* self = static_cast<StateBuffer*>(user);
    try {
      if (nbytes > self->bytes || !self->ptr) {
        self->drop();
        self->ptr = c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::raw_alloc(nbytes ? nbytes : 1);
        self->bytes = nbytes;
      }
      return self->ptr;
    } catch (...) {   // must not unwind through the C frame: surfaces as RADEGS_ERR_ALLOC
      self->ptr = nullptr; self->bytes = 0;
      return static_cast<void*>(nullptr);
    }
  }
  // hands the memory over to a uint8 tensor (empty tensor when nothing was requested)
  Tensor release() {
    auto opts = torch::TensorOptions().dtype(torch::kUInt8).device(device);
    if (!ptr) return torch::empty({0}, opts);
    void* p = ptr;
    const bool img = image;
    const auto n = static_cast<int64_t>(bytes);
    ptr = nullptr; bytes = 0;
    return torch::from_blob(p, {n}, [img](void* q) {
      if (img) radegs_forget_image(q);
      c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::raw_delete(q);
    }, opts);
  }
};

void* current_stream(const c10::Device& d) { return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(d.index()).stream()); }

int checked(int rc, const char* what) {
  TORCH_CHECK(rc >= 0, what, " failed (", rc, "): ", radegs_last_error());
  return rc;
}

// The accumulation scratch of the backward: one all-zero buffer per (device, stream), handed back all-zero by every successful
// call (RadegsBwdArgs.acc_reuse) -- up to 256 MB; larger ones are allocated per call and filled by the library.
constexpr size_t kAccReuseMaxBytes = size_t(256) << 20;
// The mutex is held for the whole of a backward that uses a cached scratch: two host threads queueing backwards on ONE stream would
// otherwise interleave their kernels over the same buffer (each call's kernels must run back to back: the second one clears what the first
// accumulated).  The map is leaked on purpose: tensors destroyed during static destruction would outlive the HIP context.
std::mutex g_acc_mutex;
std::map<std::pair<int, void*>, Tensor>& acc_scratch_map() {
  static auto* m = new std::map<std::pair<int, void*>, Tensor>();
  return *m;
}

struct AccRequest { Tensor t; bool failed = false; };
void* acc_fixed(void* user, size_t nbytes) {
  auto* r = static_cast<AccRequest*>(user);
  if (static_cast<size_t>(r->t.numel()) < nbytes) { r->failed = true; return nullptr; }
  return r->t.data_ptr();
}

}  // namespace

// DGR/rasterize_points.cu:35-131.  Returns (num_rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geomBuffer,
// binningBuffer, imgBuffer).
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_gaussians(const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
                    const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
                    const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const float kernel_size, const int image_height,
                    const int image_width, const Tensor& sh, const int degree, const Tensor& campos, const bool prefiltered,
                    const bool require_coord, const bool require_depth, const bool debug) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build): `means3D` must be a GPU tensor -- this operator has no CPU implementation");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  const int P = static_cast<int>(means3D.size(0)), H = image_height, W = image_width;
  const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
  const bool live = P != 0;
  auto map = [&](int c) { return live ? torch::empty({c, H, W}, f32) : torch::zeros({c, H, W}, f32); };
  Tensor out_color = map(3), out_depth = map(1), out_mdepth = map(1), out_coord = map(3), out_mcoord = map(3), out_alpha = map(1),
         out_normal = map(3);
  Tensor radii = live ? torch::empty({P}, i32) : torch::zeros({P}, i32);
  StateBuffer geom(dev, false), binning(dev, false), img(dev, true);
  int rendered = 0;
  if (live) {
    In bg(background, "bg"), m3(means3D, "means3D"), col(colors, "colors_precomp"), op(opacity, "opacities"), sc(scales, "scales"),
        rot(rotations, "rotations"), cov(cov3D_precomp, "cov3D_precomp"), vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix"),
        cp(campos, "campos"), shs(sh, "shs");
    RadegsFwdArgs a{};
    a.P = P; a.D = degree; a.M = sh.numel() != 0 ? static_cast<int>(sh.size(1)) : 0; a.width = W; a.height = H;
    a.background = bg.p; a.means3D = m3.p; a.shs = shs.p; a.colors_precomp = col.p; a.opacities = op.p; a.scales = sc.p;
    a.rotations = rot.p; a.cov3D_precomp = cov.p; a.viewmatrix = vm.p; a.projmatrix = pm.p; a.cam_pos = cp.p;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
    a.prefiltered = prefiltered ? 1 : 0; a.require_coord = require_coord ? 1 : 0; a.require_depth = require_depth ? 1 : 0; a.debug = debug ? 1 : 0;
    a.out_color = out_color.data_ptr<float>(); a.out_coord = out_coord.data_ptr<float>(); a.out_mcoord = out_mcoord.data_ptr<float>();
    a.out_depth = out_depth.data_ptr<float>(); a.out_mdepth = out_mdepth.data_ptr<float>(); a.out_alpha = out_alpha.data_ptr<float>();
    a.out_normal = out_normal.data_ptr<float>(); a.radii = radii.data_ptr<int>();
    rendered = checked(radegs_forward(&a, StateBuffer::grow, &geom, StateBuffer::grow, &binning, StateBuffer::grow, &img, current_stream(dev)),
                       "radegs_forward");
  }
  return std::make_tuple(rendered, out_color, out_coord, out_mcoord, out_alpha, out_normal, out_depth, out_mdepth, radii, geom.release(),
                         binning.release(), img.release());
}

// DGR/rasterize_points.cu:134-246.  Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
// dL_drotations).
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_gaussians_backward(const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors, const Tensor& scales,
                             const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
                             const Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const float kernel_size,
                             const Tensor& dL_dout_color, const Tensor& dL_dout_coord, const Tensor& dL_dout_mcoord,
                             const Tensor& dL_dout_depth, const Tensor& dL_dout_mdepth, const Tensor& dL_dout_alpha,
                             const Tensor& dL_dout_normal, const Tensor& normalmap, const Tensor& sh, const int degree, const Tensor& campos,
                             const Tensor& geomBuffer, const int R, const Tensor& binningBuffer, const Tensor& imageBuffer,
                             const Tensor& alphas, const bool require_coord, const bool require_depth, const bool debug) {
  TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build): `means3D` must be a GPU tensor -- this operator has no CPU implementation");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  const int P = static_cast<int>(means3D.size(0));
  const int H = static_cast<int>(dL_dout_color.size(1)), W = static_cast<int>(dL_dout_color.size(2));
  const int M = sh.numel() != 0 ? static_cast<int>(sh.size(1)) : 0;
  const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  const bool live = P != 0;
  auto grad = [&](std::initializer_list<int64_t> shape) { return live ? torch::empty(shape, f32) : torch::zeros(shape, f32); };
  Tensor dL_dmeans3D = grad({P, 3}), dL_dmeans2D = grad({P, 3}), dL_dcolors = grad({P, 3}), dL_dopacity = grad({P, 1}), dL_dcov3D = grad({P, 6}),
         dL_dsh = grad({P, M, 3}), dL_dscales = grad({P, 3}), dL_drotations = grad({P, 4});
  if (live) {
    In bg(background, "bg"), m3(means3D, "means3D"), col(colors, "colors_precomp"), sc(scales, "scales"), rot(rotations, "rotations"),
        cov(cov3D_precomp, "cov3D_precomp"), vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix"), cp(campos, "campos"), shs(sh, "shs"),
        g_color(dL_dout_color, "dL_dcolor"), g_coord(dL_dout_coord, "dL_dcoord"), g_mcoord(dL_dout_mcoord, "dL_dmcoord"),
        g_depth(dL_dout_depth, "dL_ddepth"), g_mdepth(dL_dout_mdepth, "dL_dmdepth"), g_alpha(dL_dout_alpha, "dL_dalpha"),
        g_normal(dL_dout_normal, "dL_dnormal"), al(alphas, "alphas"), nm(normalmap, "normalmap");
    TORCH_CHECK(!(sc.p && !rot.p), "scales given without rotations");
    const Tensor rad = radii.contiguous(), gb = geomBuffer.contiguous(), bb = binningBuffer.contiguous(), ib = imageBuffer.contiguous();
    void* stream = current_stream(dev);
    const size_t abytes = static_cast<size_t>(P) * (require_coord ? 128 : 64);
    const std::pair<int, void*> key(dev.index(), stream);
    const bool reuse = abytes <= kAccReuseMaxBytes;
    AccRequest acc;
    StateBuffer acc_fresh(dev, false);
    std::unique_lock<std::mutex> scratch_lock(g_acc_mutex, std::defer_lock);
    if (reuse) {
      scratch_lock.lock();
      auto& cache = acc_scratch_map();
      auto it = cache.find(key);
      if (it == cache.end() || static_cast<size_t>(it->second.numel()) < abytes) {
        if (cache.size() >= 8) cache.clear();
        cache[key] = torch::zeros({static_cast<int64_t>(abytes ? abytes : 1)}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
        it = cache.find(key);
      }
      acc.t = it->second;
    }
    RadegsBwdArgs a{};
    a.struct_size = sizeof(RadegsBwdArgs);
    a.P = P; a.D = degree; a.M = M; a.R = R; a.width = W; a.height = H;
    a.background = bg.p; a.means3D = m3.p; a.shs = shs.p; a.colors_precomp = col.p; a.alphas = al.p; a.scales = sc.p; a.rotations = rot.p;
    a.cov3D_precomp = cov.p; a.viewmatrix = vm.p; a.projmatrix = pm.p; a.cam_pos = cp.p;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
    a.radii = rad.data_ptr<int>(); a.normalmap = nm.p;
    a.geom_buffer = gb.numel() ? gb.data_ptr() : nullptr; a.binning_buffer = bb.numel() ? bb.data_ptr() : nullptr;
    a.image_buffer = ib.numel() ? ib.data_ptr() : nullptr;
    a.dL_dpix = g_color.p; a.dL_dpix_coord = g_coord.p; a.dL_dpix_mcoord = g_mcoord.p; a.dL_dpix_depth = g_depth.p;
    a.dL_dpix_mdepth = g_mdepth.p; a.dL_dalphas = g_alpha.p; a.dL_dpix_normal = g_normal.p;
    a.dL_dmean2D = dL_dmeans2D.data_ptr<float>(); a.dL_dcolor = dL_dcolors.data_ptr<float>(); a.dL_dopacity = dL_dopacity.data_ptr<float>();
    a.dL_dmean3D = dL_dmeans3D.data_ptr<float>(); a.dL_dcov3D = dL_dcov3D.data_ptr<float>(); a.dL_dsh = M ? dL_dsh.data_ptr<float>() : nullptr;
    a.dL_dscale = dL_dscales.data_ptr<float>(); a.dL_drot = dL_drotations.data_ptr<float>();
    a.require_coord = require_coord ? 1 : 0; a.require_depth = require_depth ? 1 : 0; a.debug = debug ? 1 : 0;
    a.acc_reuse = reuse ? 1 : 0;
    const int rc = reuse ? radegs_backward(&a, acc_fixed, &acc, stream) : radegs_backward(&a, StateBuffer::grow, &acc_fresh, stream);
    if (reuse && (rc != 0 || acc.failed)) acc_scratch_map().erase(key);   // unknown state: the next call starts from a fresh one
    if (reuse) scratch_lock.unlock();
    checked(rc, "radegs_backward");
    if (!sc.p) {   // precomputed covariance: scale / rotation gradients are identically zero
      dL_dscales.zero_();
      dL_drotations.zero_();
    }
  }
  return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

// DGR/rasterize_points.cu:248-267
Tensor mark_visible(Tensor& means3D, Tensor& viewmatrix, Tensor& projmatrix) {
  TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build): `means3D` must be a GPU tensor -- this operator has no CPU implementation");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  const int P = static_cast<int>(means3D.size(0));
  Tensor present = torch::zeros({P}, torch::TensorOptions().dtype(torch::kBool).device(dev));
  if (P != 0) {
    In m3(means3D, "means3D"), vm(viewmatrix, "viewmatrix"), pm(projmatrix, "projmatrix");
    checked(radegs_mark_visible(P, m3.p, vm.p, pm.p, reinterpret_cast<unsigned char*>(present.data_ptr<bool>()), current_stream(dev)),
            "radegs_mark_visible");
  }
  return present;
}

// DGR/rasterize_points.cu:269-388.  view2gaussian_precomp, subpixel_offset and prefiltered are accepted and -- exactly as upstream's
// kernels do -- never read.  Returns (num_rendered, out_color[9,H,W], out_alpha_integrated, out_color_integrated, out_coordinate2d,
// out_sdf, radii, geomBuffer, binningBuffer, imgBuffer).
std::tuple<int, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
integrate_gaussians_to_points(const Tensor& background, const Tensor& points3D, const Tensor& means3D, const Tensor& colors, const Tensor& opacity,
                              const Tensor& scales, const Tensor& rotations, const float scale_modifier, const Tensor& cov3D_precomp,
                              const Tensor& view2gaussian_precomp, const Tensor& viewmatrix, const Tensor& projmatrix, const float tan_fovx,
                              const float tan_fovy, const float kernel_size, const Tensor& subpixel_offset, const int image_height,
                              const int image_width, const Tensor& sh, const int degree, const Tensor& campos, const bool prefiltered,
                              const bool debug) {
  (void)view2gaussian_precomp; (void)subpixel_offset; (void)prefiltered;
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
  if (points3D.ndimension() != 2 || points3D.size(1) != 3) AT_ERROR("points3D must have dimensions (num_points, 3)");
  TORCH_CHECK(means3D.is_cuda() && points3D.is_cuda(), "diff_gaussian_rasterization (MI355X build): inputs must be GPU tensors -- this operator has no CPU implementation");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
  const int P = static_cast<int>(means3D.size(0)), PN = static_cast<int>(points3D.size(0)), H = image_height, W = image_width;
  const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  Tensor out_color = torch::empty({9, H, W}, f32), out_alpha_integrated = torch::empty({PN}, f32), out_color_integrated = torch::empty({PN, 3}, f32),
         out_coordinate2d = torch::empty({PN, 2}, f32), out_sdf = torch::empty({PN}, f32);
  Tensor radii = torch::empty({P}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
  StateBuffer geom(dev, false), binning(dev, false), img(dev, false), pts(dev, false);
  In bg(background, "bg"), m3(means3D, "means3D"), p3(points3D, "points3D"), col(colors, "colors_precomp"), op(opacity, "opacities"),
      sc(scales, "scales"), rot(rotations, "rotations"), cov(cov3D_precomp, "cov3D_precomp"), vm(viewmatrix, "viewmatrix"),
      pm(projmatrix, "projmatrix"), cp(campos, "campos"), shs(sh, "shs");
  RadegsIntegrateArgs a{};
  a.P = P; a.D = degree; a.M = sh.numel() != 0 ? static_cast<int>(sh.size(1)) : 0; a.PN = PN; a.width = W; a.height = H;
  a.background = bg.p; a.means3D = m3.p; a.shs = shs.p; a.colors_precomp = col.p; a.opacities = op.p; a.scales = sc.p; a.rotations = rot.p;
  a.cov3D_precomp = cov.p; a.viewmatrix = vm.p; a.projmatrix = pm.p; a.cam_pos = cp.p; a.points3D = p3.p;
  a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size; a.debug = debug ? 1 : 0;
  a.out_color = out_color.data_ptr<float>(); a.out_alpha_integrated = out_alpha_integrated.data_ptr<float>();
  a.out_color_integrated = out_color_integrated.data_ptr<float>(); a.out_coordinate2d = out_coordinate2d.data_ptr<float>();
  a.out_sdf = out_sdf.data_ptr<float>(); a.radii = radii.data_ptr<int>();
  const int rendered = checked(radegs_integrate(&a, StateBuffer::grow, &geom, StateBuffer::grow, &binning, StateBuffer::grow, &img, StateBuffer::grow,
                                                &pts, current_stream(dev)),
                               "radegs_integrate");
  return std::make_tuple(rendered, out_color, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf, radii, geom.release(),
                         binning.release(), img.release());
}

// the module of DGR/ext.cpp:15-19, same four names
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rasterize_gaussians", &rasterize_gaussians);
  m.def("integrate_gaussians_to_points", &integrate_gaussians_to_points);
  m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
  m.def("mark_visible", &mark_visible);
  m.def("radegs_version", []() { return std::string(radegs_version()); });
}
