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
CUDA graph.  Reference loop: active_train.py:103-107."""
import torch

from . import ops


class GraphedStep(object):
    def __init__(self, fn, warmup=3):
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        ops.graph_capture_begin()
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
        finally:
            ops.graph_capture_end()

    def __call__(self):
        self.graph.replay()
        ops.weights_changed()   # (the replay stepped the optimizer: cached weight operands of eager code are stale)
        return self.out
