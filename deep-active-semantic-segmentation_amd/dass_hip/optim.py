"""torch.optim.SGD with the parameter update on the HIP path.

`active_train.py:60-66` builds `torch.optim.SGD(train_params, momentum=..., weight_decay=..., nesterov=...)` over two
learning-rate groups.  This subclass keeps that surface -- constructor, param_groups, `state[p]['momentum_buffer']`,
state_dict()/load_state_dict() round-trip with the stock optimizer -- and replaces the arithmetic of `step()` by
`dass_sgd_step_multi`: every f32 CUDA parameter of the step is updated by a handful of launches (64 tensors per launch
ride in the kernel argument) instead of three foreach passes per group.  Anything it does not cover (nesterov, dampening,
maximize, sparse / non-f32 / non-dense gradients, CPU tensors) goes through the stock implementation."""
import ctypes

import torch

from ._lib import check, lib


def _same_dense_layout(a, b):
    """same element order in memory: equal strides on every dimension that has more than one element (a [K,C,1,1]
    channels_last weight and its gradient may disagree on the meaningless strides of the unit dimensions)"""
    if a.shape != b.shape:
        return False
    return all(n == 1 or sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()))


class SGD(torch.optim.SGD):

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftovers = []
        for group in self.param_groups:
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
                vl = (ctypes.c_float * n)(*([float(group["lr"])] * n))
                stream = ctypes.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
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
