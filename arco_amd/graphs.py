"""HIP-graph capture of no-grad forwards (launch-bound chains of ~150 small kernels per U-Net pass).

`GraphedForward(fn)` runs `fn(x)` eagerly for the first calls, then captures it once per input
shape on torch's capture stream (the C-ABI kernels launch on torch's current stream, allocations
come from the graph's private pool) and replays it afterwards: one host launch per forward.
Only for torch.no_grad() forwards whose parameters are updated IN PLACE (flat buffers), which is
how the teacher (EMA) and the BN-statistics-only student pass are used in the step.  Dropout masks
stay fresh on every replay through a device-resident salt per instance (bound to ops.SEED_DEV during capture) bumped before each replay.
"""
import torch

from . import ops

# Stream-capture error mode of every capture below.  "global" (torch's default) forbids capture-unsafe HIP calls from ANY thread
# while a capture is open - and ProcessGroupNCCL's watchdog thread polls its pending collectives with hipEventQuery: with an `nccl`
# process group alive the first capture of a step aborted the process ("operation not permitted when stream is capturing" raised
# in the watchdog; found by the one-rank ARCO_FORCE_DIST run of round 5 - the gloo rehearsals have no such thread).
# "thread_local" restricts the check to the capturing thread, which issues nothing but the captured launches.
CAPTURE_MODE = "thread_local"


class GraphedForward:
    """Every instance owns its dropout salt (ops.SEED_DEV is swapped to it while the instance captures, as GraphedTrain does):
    two instances replayed on two streams in one step (the statistics pass beside the teacher's first pass) must not
    read-modify-write one device word from two queues.  Graphs are keyed by the input AND by the module-level switches the
    captured launches bake in (BatchNorm groups, the deferred running-statistics slot, the boundary-cast mode)."""

    def __init__(self, fn, warmup=2, enabled=True):
        self.fn, self.warmup, self.enabled = fn, warmup, enabled
        self.calls = 0
        self.graphs = {}
        self.salt = None

    @torch.no_grad()
    def __call__(self, x):
        if not self.enabled:
            return self.fn(x)
        key = (tuple(x.shape), x.dtype, ops.BN_GROUPS, ops.BN_DEFER, ops.FM_CAST)
        self.calls += 1
        if key not in self.graphs:
            if self.calls <= self.warmup:
                return self.fn(x)
            if self.salt is None:
                self.salt = torch.zeros(1, dtype=torch.int64, device=x.device)
            static_in = x.clone()
            prev_salt = ops.SEED_DEV
            ops.SEED_DEV = self.salt
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    out = self.fn(static_in)
            finally:
                ops.SEED_DEV = prev_salt
            self.graphs[key] = (g, static_in, out)
        g, static_in, out = self.graphs[key]
        self.salt.add_(1)
        static_in.copy_(x)
        g.replay()
        return out



class _GraphedTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, x, *params):
        gt.static_in.copy_(x)
        gt.salt.add_(1)
        gt.fwd_g.replay()
        ctx.gt = gt
        return tuple(o.detach() for o in gt.flat_outs)

    @staticmethod
    def backward(ctx, *grads):
        gt = ctx.gt
        for i, j in enumerate(gt.diff_idx):
            g, sg = grads[j], gt.static_grads[i]
            if g is None:
                if gt.grad_live[i]:
                    sg.zero_()
                    gt.grad_live[i] = False
            else:
                if g.data_ptr() != sg.data_ptr():
                    sg.copy_(g)
                    gt.grad_live[i] = True
                elif sg._version != gt.sink_version[i]:
                    # same storage (a consumer scattered its gradient straight into the sink, ops.grad_sink, and re-zeroes the
                    # rows it touched through gt.cleanup below) - but autograd ALSO accumulated into the buffer in place (the
                    # map has a second consumer with a dense gradient): the rows the cleanup does not know are dirty, so the
                    # next sparse use pays a dense zero (raw-pointer kernels do not bump the version counter, tensor ops do)
                    gt.grad_live[i] = True
        gt.bwd_g.replay()
        for fn in gt.cleanup:
            fn()
        gt.cleanup.clear()
        for k in range(len(gt.sink_busy)):
            gt.sink_busy[k] = False
        for mark in gt.markers:
            mark()
        return (None, gt.static_dx) + tuple(gt.static_pgrads)


class GraphedTrain:
    """Forward AND backward of `module(x)` as two HIP graphs behind one autograd node (the scheme of
    torch.cuda.make_graphed_callables, specialised to this package's ops): per step the ~100 forward and ~200
    backward launches of a U-Net / V-Net pass cost two host calls.  Requirements met by the trainers:
    fixed input shape per instance, parameters and their gradient buffers at fixed addresses (flat buffers;
    weight / BN gradients are accumulated straight into them by the captured kernels), packed weights served
    by an ops.PackPlan, BN statistics and dropout salts updated on the device.  One instance per call site:
    the activations saved for backward live in the instance's private pool until its backward has replayed.
    The capture is taken w.r.t. leaf aliases of the parameters (see _capture), so live autograd graphs over
    the real parameters (optimizer hooks, the other pass of the same step) do not leak into it."""

    def __init__(self, module, warmup=2, enabled=True, grad_views=None):
        """grad_views: {id(parameter): tensor} - the captured backward accumulates the parameter gradients THERE instead of in
        the parameters' own .grad views (optim.SGDNesterov.second_grad_views: a pass replayed on a side stream beside another
        backward pass over the same parameters)."""
        self.module, self.warmup, self.enabled = module, warmup, enabled
        self.grad_views = grad_views
        self.calls = 0
        self.captured = False

    def will_replay(self, x):
        """Will the next call with input x capture-or-replay the graphs (rather than run the module eagerly)?"""
        return (self.enabled and self.calls >= self.warmup and torch.is_grad_enabled()
                and (not self.captured or tuple(x.shape) == self.shape))

    def _capture(self, x):
        from torch.utils import _pytree as pytree
        dev = x.device
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        self.markers = [p._arco_mark for p in self.params if hasattr(p, "_arco_mark")]
        self.salt = torch.zeros(1, dtype=torch.int64, device=dev)
        self.static_in = x.detach().clone().requires_grad_(x.requires_grad)
        prev_salt = ops.SEED_DEV
        ops.SEED_DEV = self.salt
        try:
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            self.fwd_g = torch.cuda.CUDAGraph()
            with torch.enable_grad(), torch.cuda.graph(self.fwd_g, pool=pool, capture_error_mode=CAPTURE_MODE):
                # The capture differentiates w.r.t. fresh leaf ALIASES of the parameters (same storage, same
                # flat-gradient views / packed-weight plans): their gradient accumulators are born on the
                # capture stream.  The real parameters' accumulators may be alive on the legacy stream (optimizer
                # hooks, another pass of the same step); autograd would wait on that stream inside the capture.
                alias = {}
                for n, p in named:
                    a = p.detach().requires_grad_(True)
                    for k, v in p.__dict__.items():
                        if k.startswith("_arco"):
                            setattr(a, k, v)
                    gv = self.grad_views.get(id(p)) if self.grad_views is not None else None
                    if gv is not None:
                        a._arco_grad_view = gv
                        a.grad = gv
                    elif p.grad is not None:
                        a.grad = p.grad
                    alias[n] = a
                self.alias = alias
                outs = torch.func.functional_call(self.module, alias, (self.static_in,))
            self.flat_outs, self.spec = pytree.tree_flatten(outs)
            self.diff_idx = [i for i, o in enumerate(self.flat_outs) if o.requires_grad]
            self.static_grads = [torch.zeros_like(self.flat_outs[i]) for i in self.diff_idx]
            self.grad_live = [False] * len(self.diff_idx)
            # gradient sinks (ops.grad_sink): a consumer with a sparse gradient scatters straight into static_grads[k]
            self.sink_busy = [False] * len(self.diff_idx)
            self.sink_version = [0] * len(self.diff_idx)       # static_grads[k]._version when the sink was handed out
            self.cleanup = []
            self.sink_uses = 0
            for k, i in enumerate(self.diff_idx):
                ops.GRAD_SINKS[self.flat_outs[i].data_ptr()] = (self, k)
            inputs = ([self.static_in] if self.static_in.requires_grad else []) + [alias[n] for n, _ in named]
            torch.cuda.synchronize()
            self.bwd_g = torch.cuda.CUDAGraph()
            # single-threaded autograd: every captured launch is issued by the capturing thread
            with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(self.bwd_g, pool=pool, capture_error_mode=CAPTURE_MODE):
                grads = torch.autograd.grad([self.flat_outs[i] for i in self.diff_idx], inputs, self.static_grads,
                                            allow_unused=True)
                ops.join_side()          # the weight-gradient branch (ops._wgrad) joins the capture stream: a graph edge
                grads = list(grads)
                self.static_dx = grads.pop(0) if self.static_in.requires_grad else None
                if self.grad_views is not None:
                    # gradients autograd returned as tensors (biases without a BatchNorm behind them): accumulated into the second
                    # buffer inside the captured graph - handing them to autograd would let AccumulateGrad `+=` the parameters'
                    # own .grad on this pass's stream while the other pass does the same on its stream
                    for i, (n, p) in enumerate(named):
                        if grads[i] is not None and self.grad_views.get(id(p)) is not None:
                            self.grad_views[id(p)].add_(grads[i])
                            grads[i] = None
                            if hasattr(p, "_arco_mark") and p._arco_mark not in self.markers:
                                self.markers.append(p._arco_mark)
            self.static_pgrads = grads         # None where the kernels wrote into the flat gradient buffer
        finally:
            ops.SEED_DEV = prev_salt
        self.captured = True
        self.shape = tuple(x.shape)

    def __call__(self, x):
        self.calls += 1
        if not self.enabled or self.calls <= self.warmup or not torch.is_grad_enabled():
            return self.module(x)
        if not self.captured:
            self._capture(x)
        if tuple(x.shape) != self.shape:
            return self.module(x)
        from torch.utils import _pytree as pytree
        outs = _GraphedTrainFn.apply(self, x, *self.params)
        return pytree.tree_unflatten(list(outs), self.spec)


def set_enabled(owner, flag):
    """Enable / disable every GraphedForward / GraphedTrain held by `owner` (a trainer step object); returns the previous
    flags so a caller (bench.py's eager roofline pass: HIP-event timing needs the kernels outside graphs) can restore them."""
    prev = {}
    for name, v in vars(owner).items():
        if isinstance(v, (GraphedForward, GraphedTrain)):
            prev[name] = v.enabled
            v.enabled = flag if isinstance(flag, bool) else bool(flag.get(name, v.enabled))
    return prev
