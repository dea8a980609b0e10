"""Flat-buffer optimiser state for the ARCO step.

Parameters of a module group are re-homed into ONE contiguous fp32 buffer (each
nn.Parameter becomes a view), gradients likewise, so SGD-Nesterov, EMA and the
data-parallel gradient all-reduce are each one kernel / one collective.
"""
import torch

from . import _lib as L
from . import ops


def flatten_params(params):
    """Move `params` into one flat fp32 buffer; returns the buffer.  p.data become views."""
    params = list(params)
    n = sum(p.numel() for p in params)
    flat = torch.empty(n, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        k = p.numel()
        flat[off:off + k].copy_(p.data.reshape(-1))
        p.data = flat[off:off + k].view(p.shape)
        off += k
    return flat


def _weights_changed(self):
    """Owner-side invalidation: with PackPlans attached (`self.plans`) only they are refreshed (one launch each);
    otherwise every cached packed weight is declared stale."""
    plans = getattr(self, "plans", None)
    if plans:
        for pl in plans:
            pl.refresh()
    else:
        ops.bump_weight_epoch()


class EmaPair:
    _weights_changed = _weights_changed
    """teacher <- m*teacher + (1-m)*student over parameters() (model_2D.py:176-182,
    train_arco_2d.py:306-308)."""

    def __init__(self, student, teacher):
        sp = list(student.parameters()) if hasattr(student, "parameters") else list(student)
        tp = list(teacher.parameters()) if hasattr(teacher, "parameters") else list(teacher)
        assert len(sp) == len(tp)
        self.sp, self.tp = sp, tp
        # the teacher may already live in an optimiser's flat buffer (stage 1: the key heads are trained too)
        self.flat_t = self._flat_of(tp)
        if self.flat_t is None:
            self.flat_t = flatten_params(tp)
        self.flat_s = None

    def _student_flat(self):
        return self._flat_of(self.sp)

    @staticmethod
    def _flat_of(params):
        # contiguous views, in order, of ONE storage -> that stretch as a flat tensor; else None
        first = params[0]
        n = sum(p.numel() for p in params)
        base = first.data.untyped_storage()
        contiguous = True
        off = first.data.storage_offset()
        for p in params:
            if p.data.untyped_storage().data_ptr() != base.data_ptr() or p.data.storage_offset() != off \
                    or not p.data.is_contiguous():
                contiguous = False
                break
            off += p.numel()
        if contiguous:
            return first.data.as_strided((n,), (1,), first.data.storage_offset())
        return None

    @torch.no_grad()
    def update(self, m):
        s = self._student_flat()
        if s is None and not any(hasattr(p, "_arco_grad_view") for p in self.sp):
            s = flatten_params(self.sp)     # no optimiser owns them (stage 2: ISD's query heads): re-home once, no cat per step
        if s is None:
            s = torch.cat([p.data.reshape(-1) for p in self.sp])
        t = self._flat_of(self.tp)          # (an optimiser built after this pair may have re-homed the teacher)
        if t is None and any(hasattr(p, "_arco_grad_view") for p in self.tp):
            # an optimiser owns part of the teacher module (stage 1 with a frozen encoder / decoder): re-homing would pull
            # those parameters out of its flat buffer - it would keep stepping storage the module no longer reads.  EMA on
            # a gathered copy and write the rows back instead (one cat + one split per step, no ownership change).
            t = torch.cat([p.data.reshape(-1) for p in self.tp])
            L.call("arco_ema", L.ptr(t), L.ptr(s), t.numel(), float(m))
            off = 0
            for p in self.tp:
                p.data.copy_(t[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._weights_changed()
            return
        if t is None:
            t = flatten_params(self.tp)
        self.flat_t = t
        L.call("arco_ema", L.ptr(self.flat_t), L.ptr(s), self.flat_t.numel(), float(m))
        self._weights_changed()


class SGDNesterov:
    _weights_changed = _weights_changed
    """torch.optim.SGD(params, lr, momentum=0.9, weight_decay=1e-4, nesterov=True)
    (train_arco_2d.py:248) over flat buffers: parameters, gradients (p.grad are views that
    autograd accumulates into) and momentum live in three contiguous fp32 arrays, and a step
    is one kernel per run of consecutive parameters that received a gradient (one launch in
    steady state).  Parameters without a gradient are skipped, like torch.  `param_groups[0]['lr']`
    is honoured like the reference's poly-LR loop (train_arco_2d.py:433-435)."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0001, nesterov=True):
        assert momentum > 0
        self._kernel = "arco_sgd_nesterov" if nesterov else "arco_sgd_momentum"     # nesterov=False: the stage-1 trainers
        self.params = [p for p in params]
        self.flat_p = flatten_params(self.params)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_buf = torch.zeros_like(self.flat_p)
        self.offsets = []
        off = 0
        for i, p in enumerate(self.params):
            k = p.numel()
            self.offsets.append((off, k))
            p.grad = self.flat_g[off:off + k].view(p.shape)
            p.register_post_accumulate_grad_hook(self._mark(i))
            # direct-write path of arco_amd.ops (contiguous torch-layout weights only)
            p._arco_grad_view = p.grad
            p._arco_mark = self._marker(i)
            off += k
        self.param_groups = [dict(lr=lr, momentum=momentum, weight_decay=weight_decay, params=self.params)]
        self._touched = set()
        self._started = [False] * len(self.params)

    def _mark(self, i):
        def hook(_p):
            self._touched.add(i)
        return hook

    def _marker(self, i):
        def mark():
            self._touched.add(i)
        return mark

    def second_grad_views(self):
        """A second flat gradient buffer with the layout of the first: {id(param): view}.  A graph-replayed pass that runs
        CONCURRENTLY with another backward pass over the same parameters (train_arco_2d: the warped student pass on a side
        stream) accumulates its parameter gradients here - two kernels doing `+=` on one buffer from two streams would race -
        and merge_second() adds the buffer into the first once both passes are done.  a + b is commutative in fp32: the merged
        gradient is bit-identical to the sequential accumulation."""
        if getattr(self, "flat_g2", None) is None:
            self.flat_g2 = torch.zeros_like(self.flat_g)
            self._g2_views = {id(p): self.flat_g2[off:off + k].view(p.shape) for (off, k), p in zip(self.offsets, self.params)}
            self._g2_dirty = False
        return self._g2_views

    def merge_second(self, n_elems=None):
        if getattr(self, "flat_g2", None) is not None and self._g2_dirty:
            n = self.flat_g.numel() if n_elems is None else int(n_elems)
            self.flat_g[:n].add_(self.flat_g2[:n])
            self.flat_g2[:n].zero_()
            self._g2_dirty = False

    def touch_from(self, start_elem):
        """Mark every parameter at flat offset >= start_elem as having received a (zero) gradient this step.  The reference's
        degenerate-batch loss is `0.0 * rep.sum()` (loss_helper_3d.py:417-424): attached to the graph, it hands every head
        parameter a ZERO gradient - and torch.optim.SGD then still applies weight decay and momentum to them, whereas a
        parameter without a gradient is skipped.  The trainers' zero path calls this after backward."""
        for i, (off, _k) in enumerate(self.offsets):
            if off >= start_elem:
                self._touched.add(i)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        self._touched.clear()
        for (off, k), p in zip(self.offsets, self.params):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + k].view(p.shape)

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        # the parameters must still be views of flat_p (someone re-homing them - flatten_params on a module this optimiser
        # owns part of - would leave this step updating storage nobody reads): first and last are checked every step
        for i in (0, len(self.params) - 1):
            if self.params[i].data_ptr() != self.flat_p.data_ptr() + 4 * self.offsets[i][0]:
                raise RuntimeError("SGDNesterov: a parameter no longer lives in this optimiser's flat buffer "
                                   "(it was re-homed after the optimiser was built)")
        # fold back gradients autograd may have re-homed
        for i, ((off, k), p) in enumerate(zip(self.offsets, self.params)):
            if p.grad is not None and p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                self.flat_g[off:off + k].copy_(p.grad.reshape(-1))
                self._touched.add(i)
        runs, cur = [], None
        for i in sorted(self._touched):
            first = not self._started[i]
            if cur is not None and cur[1] == i and cur[2] == first:
                cur[1] = i + 1
            else:
                cur = [i, i + 1, first]
                runs.append(cur)
        for a, b, first in runs:
            off = self.offsets[a][0]
            n = self.offsets[b - 1][0] + self.offsets[b - 1][1] - off
            L.call(self._kernel, L.ptr(self.flat_p[off:]), L.ptr(self.flat_g[off:]), L.ptr(self.flat_buf[off:]),
                   n, float(g['lr']), float(g['momentum']), float(g['weight_decay']), 1 if first else 0)
            for i in range(a, b):
                self._started[i] = True
        self._weights_changed()
