"""Device-resident pair pipeline: HBM buffers and streams come from PyTorch (plumbing only), all
compute goes through the smx_dev_* C-ABI on torch's current HIP stream.

One PairPipeline per (shape, slice range) per GPU; nothing is allocated after construction, so a
step is a fixed sequence of kernel launches on one stream.
"""
import ctypes as C

import torch

from . import _lib


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class PairPipeline:
    """main.cu:65-155 for one stereo pair, optionally restricted to slices [s_begin, s_end) of
    both volumes (the D-shard of this rank).  Layout in HBM: gray images u8 [h][w]; per view packed
    WTA keys i64 [h][w]; best/dmap/occlusion/filled f32 [h][w]; workspace = image / guidance planes +
    per slice in flight one aggregated plane and the strip hand-off records (smx_agg_workspace_bytes).
    Every launch goes to the pipeline's own device (torch.cuda.device(self.device)) on that device's
    current stream, whatever device is current in the calling thread."""

    def __init__(self, w, h, size_d, dminl=None, dminr=0, s_begin=0, s_end=None, device="cuda:0",
                 slices_in_flight=None, want_agg=False, params=None, max_ws_bytes=64 << 30, multi_kernel=False):
        self.lib = _lib.lib()
        self.w, self.h, self.size_d = int(w), int(h), int(size_d)
        self.n = self.w * self.h
        self.dminl = -(size_d - 1) if dminl is None else int(dminl)
        self.dminr = int(dminr)
        self.s_begin = int(s_begin)
        self.s_end = self.size_d if s_end is None else int(s_end)
        self.device = torch.device(device)
        self.params = params if params is not None else _lib.default_params()
        local = max(1, self.s_end - self.s_begin)
        sif = local if slices_in_flight is None else max(1, min(local, int(slices_in_flight)))
        # workspace of the path these parameters run (the fused walker: one plane per slice in flight); a forced
        # multi-kernel path (smx_set_agg_path(1)) needs the radius-agnostic bound
        multi = multi_kernel or self.params.radius > 9
        need = (lambda n: self.lib.smx_agg_workspace_bytes(self.w, self.h, n)) if multi else \
               (lambda n: self.lib.smx_agg_workspace_bytes_for(C.byref(self.params), self.w, self.h, n))
        while sif > 1 and 2 * need(sif) > max_ws_bytes:
            sif = (sif + 1) // 2
        self.slices_in_flight = sif
        # pair calls (both views per launch) need twice the single-view workspace
        self.ws_bytes = 2 * int(need(sif))
        dev = self.device
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        # keys[0] = left view, keys[1] = right view: one buffer so the shard merge is ONE all-reduce
        self.keys = torch.empty((2, self.h, self.w), dtype=torch.int64, device=dev)
        f = dict(dtype=torch.float32, device=dev)
        self.best = torch.empty((2, self.h, self.w), **f)
        self.dmap = torch.empty((2, self.h, self.w), **f)
        self.mean = torch.empty((2, self.h, self.w), dtype=torch.uint8, device=dev)
        self.occlusion = torch.empty((self.h, self.w), **f)
        self.filled = torch.empty((self.h, self.w), **f)
        self.agg = (torch.empty((2, local, self.h, self.w), **f) if want_agg else None)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        return torch.cuda.device(self.device)

    # -- stages ------------------------------------------------------------------------------
    def aggregate(self, gray_l, gray_r, cost_l=None, cost_r=None):
        """Cost build (fused unless cost_* given) + guided-filter aggregation + running WTA of this
        rank's slices, both views.  Leaves packed keys in self.keys."""
        # (no smx_dev_init_keys launch: the aggregation presets the keys itself, smx_set_keys_fresh)
        self.lib.smx_set_keys_fresh(1)
        try:
            if cost_l is None and cost_r is None:
                self.aggregate_pair(gray_l, gray_r)
            elif cost_l is not None and cost_r is not None:
                self.aggregate_pair_cost(gray_l, gray_r, cost_l, cost_r)
            else:
                self.aggregate_view(0, gray_l, gray_r, cost_l)
                self.aggregate_view(1, gray_r, gray_l, cost_r)
        finally:
            self.lib.smx_set_keys_fresh(0)

    def aggregate_pair_cost(self, gray_l, gray_r, cost_l, cost_r):
        """Both views per launch from materialised cost volumes of this rank's slices (smx_dev_aggregate_wta_pair_cost):
        the reference's data flow, read p + write q."""
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_set_max_slices_per_launch(self.slices_in_flight))
            try:
                _lib.check(L.smx_dev_aggregate_wta_pair_cost(
                    P, _dp(gray_l), _dp(gray_r), _dp(cost_l), _dp(cost_r), self.w, self.h, self.dminl, self.dminr,
                    self.s_begin, self.s_end, _dp(self.keys), _dp(self.mean), _dp(self.agg), _dp(self.ws), self.ws_bytes, st))
            finally:
                L.smx_set_max_slices_per_launch(0)

    def cost_volumes(self, gray_l, gray_r):
        """The two raw cost volumes of this rank's slices, resident in HBM (smx_dev_cost_volume; main.cu:80-82)."""
        n = self.s_end - self.s_begin
        cl = torch.empty((n, self.h, self.w), dtype=torch.float32, device=self.device)
        cr = torch.empty_like(cl)
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_dev_cost_volume(P, _dp(gray_l), _dp(gray_r), _dp(cl), self.w, self.w, self.h, self.dminl,
                                             self.s_begin, self.s_end, st))
            _lib.check(L.smx_dev_cost_volume(P, _dp(gray_r), _dp(gray_l), _dp(cr), self.w, self.w, self.h, self.dminr,
                                             self.s_begin, self.s_end, st))
        return cl, cr

    def aggregate_pair(self, gray_l, gray_r):
        """Both views per kernel launch (smx_dev_aggregate_wta_pair)."""
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_set_max_slices_per_launch(self.slices_in_flight))
            try:
                _lib.check(L.smx_dev_aggregate_wta_pair(
                    P, _dp(gray_l), _dp(gray_r), self.w, self.h, self.dminl, self.dminr, self.s_begin,
                    self.s_end, _dp(self.keys), _dp(self.mean), _dp(self.agg), _dp(self.ws), self.ws_bytes, st))
            finally:
                L.smx_set_max_slices_per_launch(0)

    def init_keys(self):
        with self._on_device():
            _lib.check(self.lib.smx_dev_init_keys(_dp(self.keys), 2 * self.n, self._stream()))

    def aggregate_view(self, view, guide, other, cost=None):
        dmin = self.dminl if view == 0 else self.dminr
        agg = self.agg[view] if self.agg is not None else None
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_set_max_slices_per_launch(self.slices_in_flight))
            try:
                _lib.check(L.smx_dev_aggregate_wta(
                    P, _dp(guide), _dp(other), _dp(cost), self.w, self.h, dmin, self.s_begin, self.s_end,
                    _dp(self.keys[view]), _dp(self.mean[view]), _dp(agg), _dp(self.ws), self.ws_bytes, st))
            finally:
                L.smx_set_max_slices_per_launch(0)

    def last_chunk(self):
        """(slices per walker launch, walker launches) of this thread's last fused aggregation: what
        `slices_in_flight` came to (guidedFilter.cu:171-238 is a loop over single slices)."""
        c, n = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.smx_last_agg_chunk(C.byref(c), C.byref(n)))
        return c.value, n.value

    def finish(self):
        """Keys -> best/dmap (reference presets, dispSelect rule), LR check, filling: main.cu:112-155 in one
        call (smx_dev_finish_pair: one launch, a row per workgroup)."""
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_dev_finish_pair(P, _dp(self.keys), self.w, self.h, self.dminl, self.dminr,
                                             self.dminl - 100, float(self.dminl), _dp(self.best), _dp(self.dmap),
                                             _dp(self.occlusion), _dp(self.filled), st))

    def finish_per_call(self):
        """The same through the per-stage entry points (the reference's call sequence, seven launches)."""
        with self._on_device():
            L, P, st = self.lib, C.byref(self.params), self._stream()
            _lib.check(L.smx_dev_init_wta(_dp(self.best), _dp(self.dmap), 2 * self.n, st))
            _lib.check(L.smx_dev_apply_keys(_dp(self.keys[0]), self.n, self.dminl, _dp(self.best[0]),
                                            _dp(self.dmap[0]), st))
            _lib.check(L.smx_dev_apply_keys(_dp(self.keys[1]), self.n, self.dminr, _dp(self.best[1]),
                                            _dp(self.dmap[1]), st))
            self.occlusion.copy_(self.dmap[0])                                   # main.cu:141
            _lib.check(L.smx_dev_detect_occlusion(P, _dp(self.occlusion), _dp(self.dmap[1]),
                                                  self.dminl - 100, self.w, self.h, st))  # main.cu:149
            self.filled.copy_(self.occlusion)                                    # main.cu:153
            _lib.check(L.smx_dev_fill_occlusion(_dp(self.filled), self.w, self.h, float(self.dminl), st))

    def run(self, gray_l, gray_r):
        self.aggregate(gray_l, gray_r)
        self.finish()

    def check_status(self):
        """Raise if a workgroup of the fused aggregation gave up waiting for a neighbour."""
        torch.cuda.synchronize(self.device)
        with self._on_device():
            _lib.check(self.lib.smx_dev_agg_status(_dp(self.ws)))

    def results(self):
        """Host copies (numpy) named like the oracle's dict."""
        self.check_status()
        c = lambda t: t.detach().cpu().numpy()
        r = {"bestl": c(self.best[0]), "bestr": c(self.best[1]), "dmapl": c(self.dmap[0]),
             "dmapr": c(self.dmap[1]), "meanl": c(self.mean[0]), "meanr": c(self.mean[1]),
             "occlusion": c(self.occlusion), "filled": c(self.filled)}
        if self.agg is not None:
            r["aggl"], r["aggr"] = c(self.agg[0]), c(self.agg[1])
        return r
