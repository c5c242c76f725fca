"""A hipGraph around a whole train step (torch.cuda.CUDAGraph = hipGraph on ROCm).

    step = GraphedStep(lambda: train_step(), warmup=3)     # runs `warmup` eager steps on a side stream, then captures ONE step
    step()                                                  # replays it: one graph launch instead of ~550 kernel launches

What makes the library's step capturable (each of these is host-side state that an eager step renews and a replay cannot):
  * the zeroed arenas the conv weight gradients and the BN f64 statistics are cut from are re-created INSIDE the capture
    (ops.graph_capture_begin), so their memsets are graph nodes and every replay starts from zeros;
  * the grouped weight-gradient launch stages its problem table in pinned host memory that a captured copy node reads at every replay:
    slots used inside a capture are never recycled (csrc/wgrad_x3.hip);
  * the side stream of the chunked weight gradients joins the capture through its fork event and re-joins at the next flush
    (ops._wgrad_flush; DASS_WGRAD_SIDE_CAPTURE=0: one stream inside the graph), nothing synchronises or allocates through the
    driver, and the launch profile (csrc/prof.hip) must be closed.
The captured step reads its inputs from the tensors the callable closed over: copy new batches INTO them (static inputs), as with any
CUDA graph.  Reference loop: active_train.py:103-107.

What a replay does NOT freeze: the optimizer's lr / momentum / weight_decay (device-resident, refreshed from `param_groups` before
every replay -- the reference's per-iteration poly schedule, active_train.py:101, keeps working).  What it DOES freeze: every other host
scalar and every Python branch taken while capturing (batch shapes, which parameters fired, train/eval mode)."""
import torch

from . import ops
from ._lib import lib

_capture_hooks = None   # the list replay hooks register into while a GraphedStep captures
_TRACE = __import__("os").environ.get("DASS_GRAPH_TRACE") == "1"


def register_replay_hook(fn):
    """code that is being captured calls this for host-side state it needs refreshed before EVERY replay (dass_hip.optim.SGD: the
    device copy of lr / momentum / weight_decay).  Outside a GraphedStep capture it is a no-op."""
    if _capture_hooks is not None and all(h != fn for h in _capture_hooks):
        _capture_hooks.append(fn)


class GraphedStep(object):
    """GraphedStep(fn): `fn` as one graph.  GraphedStep(fn, reduce=r, finish=f): `fn` (zero_grad + forward + loss + backward) as graph
    A, then `r()` EAGERLY on the same stream after every replay of A -- the gradient all-reduce of a multi-process step, which gloo cannot
    and RCCL need not be captured for -- then `f()` (the optimizer step) as graph B out of the same memory pool.  Per step the host
    issues two graph launches and the collectives instead of ~550 kernel launches (active_train.py:82-85 wraps the same loop,
    :103-107, in nn.DataParallel).  `before` runs eagerly ahead of graph A (e.g. utils.loss static_global().exchange(target))."""

    # warmup: eager steps run before the capture.  At least TWO must have run in the process (here or by the caller) before the first
    # capture of a model: step one registers every conv weight's split operands as it meets them, step two refreshes them all through
    # one batched table -- the form a steady-state step, and therefore the capture, uses (its host->device copy cannot be captured).
    STAGING_SLOTS = 48   # pinned problem tables reserved per capture (a R101 step uses ~14: chunks x tile classes)

    def __init__(self, fn, warmup=3, reduce=None, finish=None, before=None):
        global _capture_hooks
        self.fn, self.reduce, self.finish, self.before = fn, reduce, finish, before
        self.hooks, self.token, self.graph, self.graph_b = [], 0, None, None

        def whole():
            if before is not None:
                before()
            out = fn()
            if reduce is not None:
                reduce()
            if finish is not None:
                finish()
            return out

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(0, warmup)):   # (0: the caller has run its own eager steps -- a capture needs the step's allocations,
                whole()                       #  momentum buffers and staging tables to exist)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.token = int(lib.dass_graph_capture_open(self.STAGING_SLOTS))
        if self.token <= 0:
            raise RuntimeError("dass_graph_capture_open failed (pinned staging tables)")
        _capture_hooks = self.hooks
        ok = False
        # with a process group alive, its watchdog thread polls events while this thread captures: "thread_local" keeps another thread's
        # harmless runtime calls from invalidating the capture (the launches of autograd's backward thread are captured in either mode)
        import torch.distributed as dist

        mode = {"capture_error_mode": "thread_local"} if (dist.is_available() and dist.is_initialized()) else {}
        try:
            if before is not None:
                before()
            self.graph = torch.cuda.CUDAGraph()
            ops.graph_capture_begin()
            try:
                with torch.cuda.graph(self.graph, **mode):
                    out = fn()
                # the static result WITHOUT its autograd history: holding the captured loss itself would keep the step's graph nodes -- and
                # the parameters' AccumulateGrad nodes, bound to the capture stream -- alive for as long as this object lives, and a later
                # eager step or capture of the same model would find them on the wrong stream
                self.out = out.detach() if torch.is_tensor(out) else out
                del out
            finally:
                ops.graph_capture_end()
            if finish is not None:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), **mode):
                    finish()
                ops.weights_changed()
            ok = True
        finally:
            _capture_hooks = None
            lib.dass_graph_capture_close()
            if not ok:
                self.release()
        # (the capture did not run anything: the step is recorded, not executed, and `reduce` was not called -- the first replay
        #  recomputes everything from the current weights)

    def release(self):
        """hand the capture's pinned staging tables back (the graph must not be replayed afterwards)"""
        if self.token:
            lib.dass_graph_release(self.token)
            self.token = 0
        self.graph = self.graph_b = None
        self.out = None

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def __call__(self):
        if self.graph is None:
            raise RuntimeError("GraphedStep: released (or its capture failed)")
        if _TRACE:
            return self._traced_call()
        for h in self.hooks:
            h()
        if self.before is not None:
            self.before()
        self.graph.replay()
        if self.reduce is not None:
            self.reduce()
        if self.graph_b is not None:
            self.graph_b.replay()
        ops.weights_changed()   # (the replay stepped the optimizer: cached weight operands of eager code are stale)
        return self.out

    def _traced_call(self):
        """DASS_GRAPH_TRACE=1: the same step with a device synchronisation and a wall-clock stamp behind every phase (diagnostic)"""
        import sys
        import time

        t = [time.perf_counter()]

        def mark():
            torch.cuda.synchronize()
            t.append(time.perf_counter())

        for h in self.hooks:
            h()
        if self.before is not None:
            self.before()
        mark()
        self.graph.replay()
        mark()
        if self.reduce is not None:
            self.reduce()
        mark()
        if self.graph_b is not None:
            self.graph_b.replay()
        mark()
        ops.weights_changed()
        print("GraphedStep: before %.1f ms, graph A %.1f, reduce %.1f, graph B %.1f" % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])), file=sys.stderr, flush=True)
        return self.out
