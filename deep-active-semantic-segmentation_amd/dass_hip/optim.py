"""torch.optim.SGD with the parameter update on the HIP path.

`active_train.py:60-66` builds `torch.optim.SGD(train_params, momentum=..., weight_decay=..., nesterov=...)` over two
learning-rate groups.  This subclass keeps that surface -- constructor, param_groups, `state[p]['momentum_buffer']`,
state_dict()/load_state_dict() round-trip with the stock optimizer -- and replaces the arithmetic of `step()` by
`dass_sgd_step_multi`: every f32 CUDA parameter of the step is updated by a handful of launches (64 tensors per launch
ride in the kernel argument) instead of three foreach passes per group.  Anything it does not cover (nesterov, dampening,
maximize, sparse / non-f32 / non-dense gradients, CPU tensors) goes through the stock implementation.

Inside a hipGraph capture (dass_hip/graph.py) the launch reads {lr, momentum, weight_decay} of its param group from DEVICE memory
(`dass_sgd_step_multi_dev`): the reference calls its poly-LR scheduler before every iteration (active_train.py:101), and the by-value
arguments of a captured launch would freeze the rate of the capture.  The step registers `sync_hyper` with the capture, and
GraphedStep runs it before every replay: `param_groups[i]['lr']` set by any scheduler keeps working, same arithmetic bit for bit."""
import ctypes
import os

import torch

from ._lib import check, lib


def _same_dense_layout(a, b):
    """same element order in memory: equal strides on every dimension that has more than one element (a [K,C,1,1]
    channels_last weight and its gradient may disagree on the meaningless strides of the unit dimensions)"""
    if a.shape != b.shape:
        return False
    return all(n == 1 or sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()))


class SGD(torch.optim.SGD):

    def _hyper_tensor(self, gi, device):
        """device-resident {lr, momentum, weight_decay, 0} of param group gi.  Created by the first EAGER step: memory allocated while a
        stream captures belongs to the graph's pool, and a triple allocated there was overwritten by the replays (measured:
        tools/graph_det_probe.py, lr read back as 1.0 / garbage) -- a capture without a preceding eager step is refused."""
        store = self.__dict__.setdefault("_dass_hyper", {})
        ent = store.get(gi)
        if ent is None or ent["dev"].device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("dass_hip.optim.SGD: run one eager optimizer.step() before capturing it into a hipGraph "
                                   "(GraphedStep(warmup >= 1) does)")
            ent = store[gi] = {"dev": torch.zeros((4,), dtype=torch.float32, device=device), "pushed": None}
        return ent

    def sync_hyper(self):
        """push every captured group's current {lr, momentum, weight_decay} to its device triple (a kernel argument carries the values:
        no staging buffer to race on); skipped when nothing changed since the last push.  Called before each graph replay."""
        for gi, ent in self.__dict__.get("_dass_hyper", {}).items():
            g = self.param_groups[gi]
            vals = (float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]))
            if ent["pushed"] != vals:
                arr = (ctypes.c_float * 3)(*vals)
                stream = ctypes.c_void_p(torch.cuda.current_stream(ent["dev"].device).cuda_stream)
                check(lib.dass_set_floats(ctypes.c_void_p(ent["dev"].data_ptr()), arr, 3, stream), "dass_set_floats")
                ent["pushed"] = vals

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftovers = []
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing() and os.environ.get("DASS_SGD_DEV_HYPER", "1") == "1"
        for gi, group in enumerate(self.param_groups):
            plain = (group["momentum"] != 0 and group["dampening"] == 0 and not group["nesterov"]
                     and not group.get("maximize", False))
            ps, gs, bs, ns = [], [], [], []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                ok = (plain and p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse
                      and _same_dense_layout(p, g)
                      and (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)))
                if not ok:
                    leftovers.append((group, p))
                    continue
                st = self.state[p]
                buf = st.get("momentum_buffer")
                if buf is None or not _same_dense_layout(buf, p):
                    # zeros: momentum * 0 + g == torch's first-step copy of g, bit for bit
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p)
                gs.append(g)
                bs.append(buf)
                ns.append(p.numel())
            if ps:
                n = len(ps)
                vp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ps])
                vg = (ctypes.c_void_p * n)(*[t.data_ptr() for t in gs])
                vb = (ctypes.c_void_p * n)(*[t.data_ptr() for t in bs])
                vn = (ctypes.c_int64 * n)(*ns)
                stream = ctypes.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
                if capturing:
                    # the captured launch reads its hyper-parameters from device memory; the capture's owner refreshes them before
                    # every replay (graph.register_replay_hook -> sync_hyper), so a scheduler's `group['lr'] = ...` is honoured
                    from . import graph

                    ent = self._hyper_tensor(gi, ps[0].device)
                    graph.register_replay_hook(self.sync_hyper)
                    check(lib.dass_sgd_step_multi_dev(vp, vg, vb, vn, n, ctypes.c_void_p(ent["dev"].data_ptr()), stream), "dass_sgd_step_multi_dev")
                else:
                    self._hyper_tensor(gi, ps[0].device)   # (exists before any capture of this step)
                    vl = (ctypes.c_float * n)(*([float(group["lr"])] * n))
                    check(lib.dass_sgd_step_multi(vp, vg, vb, vn, vl, n, float(group["momentum"]), float(group["weight_decay"]), stream),
                          "dass_sgd_step_multi")
                # the kernel writes through raw pointers: tell autograd the tensors changed, or every cache keyed on
                # (data_ptr, _version) -- split / transposed weight operands, eval-BN vectors -- keeps serving step-0 values
                torch.autograd.graph.increment_version(ps)
                torch.autograd.graph.increment_version(bs)
        if leftovers:  # stock arithmetic for what the kernel does not cover
            for group, p in leftovers:
                g = p.grad
                if group["weight_decay"] != 0:
                    g = g.add(p, alpha=group["weight_decay"])
                if group["momentum"] != 0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        buf = st["momentum_buffer"] = torch.clone(g).detach()
                    else:
                        buf.mul_(group["momentum"]).add_(g, alpha=1 - group["dampening"])
                    g = g.add(buf, alpha=group["momentum"]) if group["nesterov"] else buf
                p.add_(g, alpha=(group["lr"] if not group.get("maximize", False) else -group["lr"]) * -1)
        return loss
