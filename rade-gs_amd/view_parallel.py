"""View-level data parallelism for the rasterizer: one process per GPU, replicated Gaussians, each rank
renders its own view; the only exchange step is the sum of the parameter gradients.

The reference has no distributed code at all (single process, `cuda:0`); this is the extension
BASELINE.json's north_star asks for.  The six gradient tensors (xyz, SH, opacity, scaling, rotation
= 59 floats = 236 B per Gaussian) are packed into ONE flat bucket so a single RCCL all-reduce moves
them over xGMI; with `backend="gloo"` the same code runs on CPU tensors (tests/test_dist_gloo.py).
"""
from typing import Dict, Iterable, List

import torch
import torch.distributed as dist

GRAD_KEYS = ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations")


def pack(tensors: Iterable[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])


def unpack(flat: torch.Tensor, like: List[torch.Tensor]) -> List[torch.Tensor]:
    out, off = [], 0
    for t in like:
        n = t.numel()
        out.append(flat[off:off + n].view_as(t))
        off += n
    return out


def allreduce_gradients(grads: Dict[str, torch.Tensor], average: bool = True, group=None, force: bool = False) -> Dict[str, torch.Tensor]:
    """Sum (or mean) of the per-view gradients over all ranks, one collective for the whole bucket.
    Mean keeps the single-view learning-rate scale of the reference's batch-1 training loop.
    `force` runs the collective even in a 1-rank group (path check on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return grads
    keys = [k for k in GRAD_KEYS if grads.get(k) is not None]
    flat = pack(grads[k] for k in keys)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    out = dict(grads)
    for k, t in zip(keys, unpack(flat, [grads[k] for k in keys])):
        out[k] = t
    return out


def allreduce_densification_stats(grad_norm_xy: torch.Tensor, grad_norm_abs: torch.Tensor, visible: torch.Tensor,
                                  radii: torch.Tensor, group=None):
    """Keeps densification consistent across ranks (scene/gaussian_model.py:743-747, train.py:187):
    gradient-norm accumulators and visibility counts add up, radii take the max."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grad_norm_xy, grad_norm_abs, visible, radii
    flat = pack([grad_norm_xy.float(), grad_norm_abs.float(), visible.float()])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    a, b, c = unpack(flat, [grad_norm_xy, grad_norm_abs, visible])
    r = radii.clone()
    dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
    return a.to(grad_norm_xy.dtype), b.to(grad_norm_abs.dtype), c.to(visible.dtype), r


def assert_same_on_all_ranks(what: str, values, group=None):
    """Fails LOUDLY on every rank when the ranks disagree on the sizes the collectives below are built from -- a mismatch would
    otherwise show up as a hang inside RCCL (all_gather_into_tensor / all_reduce block forever on unequal counts)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    mine = [int(v) for v in values]
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    if any(v != mine for v in everyone):
        raise RuntimeError(f"view-parallel exchange: ranks disagree on {what}: {everyone} (this rank: {mine}); "
                           "every rank must hold the same replicated Gaussians")


class GradBucket:
    """One flat fp32 buffer holding the parameter gradients of a view back to back (59 floats = 236 B per
    Gaussian at SH degree 3).  Install `bucket.allocator` with `_C.set_grad_allocator(device, ...)` and the backward kernels write
    straight into it; `allreduce()` is then a single RCCL call on the bucket, no packing copy."""

    LAYOUT = ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations")

    def __init__(self, P: int, M: int, device, group=None, timing: bool = False):
        assert_same_on_all_ranks("(P, M)", (P, M), group)
        self.timing = bool(timing) and torch.device(device).type == "cuda"
        self._timings = []
        shapes = {"dL_dmeans3D": (P, 3), "dL_dsh": (P, M, 3), "dL_dopacity": (P, 1), "dL_dscales": (P, 3), "dL_drotations": (P, 4)}
        sizes = [int(torch.Size(shapes[k]).numel()) for k in self.LAYOUT]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
        self.views, off = {}, 0
        for k, n in zip(self.LAYOUT, sizes):
            self.views[k] = self.flat[off:off + n].view(shapes[k])
            off += n

    def allocator(self, name, shape, dtype, device):
        v = self.views.get(name)
        return v if (v is not None and v.shape == torch.Size(shape)) else None

    def allreduce(self, average: bool = True, group=None):
        if dist.is_available() and dist.is_initialized():
            t0 = None
            if self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average and dist.get_world_size(group) > 1:
                self.flat /= dist.get_world_size(group)
            if t0 is not None:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record()
                self._timings.append((t0, t1))
        return self.views

    def collect_timing(self):
        """[(exchange_ms, exposed_ms)] since the last collect (after a device synchronisation): the one all-reduce is issued after the
        backward, so nothing of it is hidden -- both numbers are its duration on the launch stream."""
        out = [(a.elapsed_time(b),) * 2 for a, b in self._timings]
        self._timings = []
        return out


class FactoredGradExchange:
    """The gradient exchange of a view-parallel step with the SH part FACTORED.

    The SH-coefficient gradient is 48 of the 59 floats per Gaussian, but per view it is an outer product
    dL/dsh = basis(normalize(mean - campos)) (x) dL/dRGB (3 floats + the camera position).  Instead of all-reducing
    192 B per Gaussian, every rank all-gathers its 12-B dL/dRGB row (and 3 floats of camera position) and rebuilds the
    batch sum locally with one kernel (`_C.sh_grad_from_views`).  Per-GPU traffic over xGMI at N ranks:
        plain   all-reduce 236 B            -> 2 (N-1)/N * 236 B  = 413 B per Gaussian at N = 8
        here    all-reduce  44 B + all-gather (N-1) * 12 B        = 161 B per Gaussian at N = 8
    The result equals the plain all-reduce up to fp32 summation order.  GPU only (the rebuild is a HIP kernel).

    Overlap, in the order the backward produces its results:
      1. the dL/dRGB rows are final one kernel before the rest of the backward (`drgb_ready`, called by the backward on the host between
         the two kernels): their all-gather -- the larger share of the bytes at N = 8 -- starts there on a side stream and runs under
         the per-Gaussian backward;
      2. the per-Gaussian backward is queued in `grad_chunks` launches over consecutive ranges of Gaussians (`grads_ready(first, count)`,
         called after each launch is queued): the 44-B rows of a finished range are all-reduced under the launches that follow -- with
         two chunks half of the all-reduce is hidden, only the last chunk's part (and the densification statistics, which exist only
         after the backward) waits for the last kernel.
    `exchange()` issues what is still missing and waits.  Every rank issues the same collectives in the same order.

    Usage: `_C.set_grad_allocator(device, ex.allocator)` before the backward; `ex.set_view(campos)` if the collectives should start
    early; `ex.exchange(means3D, campos)` after the backward."""

    SMALL = ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")

    def __init__(self, P: int, M: int, degree: int, device, group=None, grad_chunks: int = 2, timing: bool = False, early: bool = True):
        from diff_gaussian_rasterization import _C
        self._C, self.P, self.M, self.D, self.group = _C, P, M, degree, group
        assert_same_on_all_ranks("(P, M, sh_degree, grad_chunks)", (P, M, degree, grad_chunks), group)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        shapes = {"dL_dmeans3D": (P, 3), "dL_dopacity": (P, 1), "dL_dscales": (P, 3), "dL_drotations": (P, 4)}
        sizes = [int(torch.Size(shapes[k]).numel()) for k in self.SMALL]
        # one flat buffer: the four small gradient tensors, then the densification statistics of the view ([P, 3]: |grad xy|, |grad abs|,
        # visible) -- summed over the ranks by the same collective type on the same stream, no second bucket
        self.flat = torch.zeros(sum(sizes) + 3 * P, dtype=torch.float32, device=device)
        self.small = self.flat[:sum(sizes)]
        self.stats = self.flat[sum(sizes):].view(P, 3)
        self.views, off = {}, 0
        for k, n in zip(self.SMALL, sizes):
            self.views[k] = self.small[off:off + n].view(shapes[k])
            off += n
        self.drgb = torch.empty((P, 3), dtype=torch.float32, device=device)
        self.gathered = torch.empty((self.world, P, 3), dtype=torch.float32, device=device)
        self.campos_all = torch.empty((self.world, 3), dtype=torch.float32, device=device)
        self.dL_dsh = torch.empty((P, M, 3), dtype=torch.float32, device=device)
        self.device = torch.device(device)
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._early = None        # handles of the all-gathers started from drgb_ready()
        self._chunk_handles = []  # handles of the all-reduces started from grads_ready()
        self._rows_done = 0       # rows of the small tensors whose all-reduce has been issued
        self._campos = None
        self.grad_chunks = int(grad_chunks)
        self.early = bool(early)     # False: issue every collective from exchange() (no overlap with the backward)
        self.timing = bool(timing) and self.device.type == "cuda"
        self._t_first = None      # event on the side stream before the step's first collective (timing)
        self._timings = []        # (first collective issued, exchange() entered, exchange() done) event triples, one per step

    def _abandon_step(self):
        """Collectives of a step whose exchange() never ran (its backward raised after the hooks had fired, or the caller skipped the
        exchange): every rank that got as far issued them, so they are waited for -- the side stream and the process group stay in
        order -- and the step's state is dropped.  The rows they reduced in place belong to a gradient nobody will read."""
        pending = list(self._chunk_handles) + (list(self._early) if self._early is not None else [])
        for h in pending:
            h.wait()
        self._early, self._chunk_handles, self._rows_done, self._t_first = None, [], 0, None

    def set_view(self, campos):
        """The camera position of the view about to be differentiated: lets drgb_ready() / grads_ready() start the collectives early.
        Starts a STEP: whatever an unfinished earlier step left behind is waited for and discarded first (rows all-reduced
        by a step without exchange() would otherwise be taken for this step's, and the ranks would diverge silently)."""
        if self._early is not None or self._chunk_handles or self._rows_done:
            self._abandon_step()
        self._campos = campos.reshape(1, 3).to(torch.float32).contiguous()

    @property
    def _early_ok(self):
        if not self.early:
            return False
        return self._campos is not None and dist.is_available() and dist.is_initialized()

    @property
    def early_drgb(self):
        """True when the backward should write the dL/dRGB rows with their own early kernel and call drgb_ready()."""
        return self._early_ok

    @property
    def early_grads(self):
        """True when the backward should queue its per-Gaussian kernel in grad_chunks launches and call grads_ready() after each."""
        return self._early_ok and self.grad_chunks > 1

    def _on_side_stream(self, issue):
        """Run `issue()` (which starts asynchronous collectives) on the side stream, behind everything queued on the launch stream so far."""
        if self._side is None:      # CPU tensors (gloo): asynchronous collectives need no stream
            return issue()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            if self.timing and self._t_first is None:
                self._t_first = torch.cuda.Event(enable_timing=True)
                self._t_first.record(self._side)
            return issue()

    def drgb_ready(self):
        """Called by `_C.rasterize_gaussians_backward` on the host once the kernel that writes the dL/dRGB rows is queued (and the
        per-Gaussian backward is not yet): start their all-gather on the side stream."""
        if not self.early_drgb:
            return
        if self._early is not None:
            raise RuntimeError("FactoredGradExchange: a second backward before exchange() (or set_view()) finished the first one's step")
        self._early = self._on_side_stream(lambda: (
            dist.all_gather_into_tensor(self.gathered.view(-1, 3), self.drgb, group=self.group, async_op=True),
            dist.all_gather_into_tensor(self.campos_all, self._campos, group=self.group, async_op=True)))

    def _allreduce_rows(self, first, count):
        """asynchronous all-reduces of rows [first, first + count) of the four small tensors (contiguous inside each tensor)"""
        return [dist.all_reduce(self.views[k][first:first + count], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for k in self.SMALL]

    def grads_ready(self, first, count):
        """Called by the backward on the host after each launch of its per-Gaussian kernel (include/radegs.h: grads_ready): rows
        [first, first + count) of the small gradient tensors are final once the launch stream reaches this point.  All but the last
        range are all-reduced from here, under the launches that follow; the last one leaves with exchange()."""
        if not self.early_grads:
            return
        if first == 0 and self._rows_done != 0:
            raise RuntimeError("FactoredGradExchange: a second backward before exchange() (or set_view()) finished the first one's step: "
                               f"rows [0, {self._rows_done}) are already on their way")
        if first != self._rows_done or first + count >= self.P:
            return
        self._chunk_handles += self._on_side_stream(lambda: self._allreduce_rows(first, count))
        self._rows_done = first + count

    def allocator(self, name, shape, dtype, device):
        if name == "dL_drgb_clamped":
            return self.drgb
        if name == "dL_dsh":
            return self._C.SKIP_GRAD
        v = self.views.get(name)
        return v if (v is not None and v.shape == torch.Size(shape)) else None

    def exchange(self, means3D, campos, average: bool = True, stats=None, radii=None):
        """Finish the step's exchange.  stats: optional (|grad xy| [P], |grad abs| [P], visible [P]) of this rank's view -- summed over the
        ranks with the last part of the small bucket (scene/gaussian_model.py:743-747 keeps densification consistent this way);
        radii: optional int32 [P], maximum over the ranks.  Returns the gradient dict (+ 'densify_stats' [P, 3] / 'radii_max')."""
        scale = 1.0
        t0 = None
        if self.timing:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(torch.cuda.current_stream(self.device))
        on = dist.is_available() and dist.is_initialized()
        rmax = None
        if on:
            if stats is not None:
                self.stats[:, 0].copy_(stats[0].reshape(-1)); self.stats[:, 1].copy_(stats[1].reshape(-1)); self.stats[:, 2].copy_(stats[2].reshape(-1))
            # output in the concatenated form (world*P, 3): the layout every backend's all_gather_into_tensor accepts
            if self._early is not None:      # started under the per-Gaussian backward (drgb_ready)
                h1, h2 = self._early
            else:
                h1, h2 = self._on_side_stream(lambda: (
                    dist.all_gather_into_tensor(self.gathered.view(-1, 3), self.drgb, group=self.group, async_op=True),
                    dist.all_gather_into_tensor(self.campos_all, campos.reshape(1, 3).to(torch.float32).contiguous(), group=self.group, async_op=True)))
            first = self._rows_done          # rows not yet on their way (everything, without early chunks)

            def rest():
                hs = self._allreduce_rows(first, self.P - first) if first else [dist.all_reduce(self.small, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
                if stats is not None:
                    hs.append(dist.all_reduce(self.stats, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                return hs
            handles = self._chunk_handles + self._on_side_stream(rest)
            if radii is not None:
                rmax = radii.clone()
                handles.append(dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            self._early, self._campos, self._chunk_handles, self._rows_done = None, None, [], 0
            for h in (h1, h2, *handles):
                h.wait()
            if average and self.world > 1:
                self.small /= self.world
                scale = 1.0 / self.world
        else:
            self.gathered[0].copy_(self.drgb)
            self.campos_all[0].copy_(campos.reshape(3))
            rmax = radii
            if stats is not None:
                self.stats[:, 0].copy_(stats[0].reshape(-1)); self.stats[:, 1].copy_(stats[1].reshape(-1)); self.stats[:, 2].copy_(stats[2].reshape(-1))
        self._C.sh_grad_from_views(means3D, self.campos_all, self.gathered, self.D, self.M, scale, out=self.dL_dsh)
        out = dict(self.views)
        out["dL_dsh"] = self.dL_dsh
        if stats is not None:
            out["densify_stats"] = self.stats
        if radii is not None:
            out["radii_max"] = rmax
        if self.timing:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream(self.device))
            self._timings.append((self._t_first if self._t_first is not None else t0, t0, t1))
            self._t_first = None
        return out

    def collect_timing(self):
        """[(exchange_ms, exposed_ms)] of the exchange() calls since the last collect -- call after a device synchronisation (timing=True).
        exchange: first collective of the step issued -> exchange() done, on the GPU's clock; exposed: exchange() entered -> done on the
        launch stream, i.e. what the step waits for on top of the backward."""
        out = [(first.elapsed_time(t1), t0.elapsed_time(t1)) for first, t0, t1 in self._timings]
        self._timings = []
        return out
