"""Autograd operators over the C-ABI HIP kernels (channels-last fp32 activations).

Activations are torch tensors with logical shape [N, C, H, W] and channels-last
memory (rows of C floats per pixel); a channel slice of such a tensor is consumed
in place through its row stride.  PyTorch only allocates memory and chains the
autograd graph - every arithmetic op below is a hand-written gfx950 kernel.
"""
import weakref

import torch

from . import _lib as L
from ._contrast import rows_view

_drop_gen = None
SEED_DEV = None     # device uint64 salt for dropout masks of graph-captured forwards (graphs.py)
PROFILE = None       # bench.py sets a dict: kernel instantiation id -> {n, flop, timed: [(ev0, ev1, flop, (taps, M, N, K))]}
_cfg_cache = {}
WORK = None          # bench.py sets {"bytes": 0.0, "flop": 0.0, "launches": 0}: ALGORITHMIC HBM bytes (each operand once) and FLOP of every
                     # launch of the convolution / weight-gradient / BatchNorm / pooling / resize families, summed over a step (step_roofline)


def _work(nbytes, flop=0.0, launches=1):
    if WORK is not None and not torch.cuda.is_current_stream_capturing():
        WORK["bytes"] += float(nbytes); WORK["flop"] += float(flop); WORK["launches"] += launches
BN_GROUPS = 1         # see bn_groups()
POOL_FUSE = int(__import__('os').environ.get('ARCO_POOL_FUSE', '1'))             # A/B switch: 0 = separate max-pool pass in the U-Net encoder
CONV_MMA = int(__import__('os').environ.get('ARCO_CONV_MMA', '3'))          # MFMA mode of the convolutions / GEMMs (forward and data gradient; --conv_mma of the trainers):
                      # 3 (default, "f32x3"): fp32-accurate products on the bf16 matrix cores - every fp32 operand is split
                      #    exactly into three bf16 terms and six v_mfma_f32_16x16x32_bf16 replace eight v_mfma_f32_16x16x4_f32
                      #    (error per product <= 2^-23, the size of one fp32 rounding; csrc/igemm.hip, MMA = 3);
                      # 0 ("f32"): the native fp32 MFMA (bitwise an fma chain);
                      # 1 / 2 ("f16" / "bf16"): the 3x3x3 convolutions round their operands to f16 / bf16 (BASELINE configs[4])
HEAD_MMA = 0         # 1 / 2: every 1x1 / 1x1x1 GEMM on fp32 tensors (FeatureExtractor_3d, q_representation, the row-sparse heads: model_3D.py:37-63,
                     # train_arco_3d.py:206-209) rounds its operands to f16 / bf16 in registers (v_mfma_f32_16x16x16_f16, fp32 accumulate; gradient
                     # operands bf16 for range) - the "contrastive" half of BASELINE configs[4]'s "fp16 MFMA conv + contrastive"; set by
                     # train_arco_3d --act_dtype f16 (--head_mma).  0: the GEMMs follow CONV_MMA
PROFILE_EVERY = 1    # time every n-th conv launch of an instantiation (bench.py: 7, prime vs the per-step launch counts)
ACT_HALF = False     # f16 ACTIVATION STORAGE of the volume path (--act_dtype f16 of train_arco_3d, BASELINE configs[4]): the V-Net's
                     # first layer writes f16 and every operator below follows its input's dtype (csrc/conv_h.hip, the *_h entry points);
                     # weights, BatchNorm statistics, parameter gradients, loss and optimizer stay fp32
LOSS_SCALE = 16384.0  # gradients enter the f16 region multiplied by this (from_half's backward); the trainer divides the
                      # region's parameter gradients by it before the optimiser step


def _is_half(t):
    return t is not None and t.dtype == torch.float16


class bn_groups:
    """`with ops.bn_groups(g):` - inside, a batch is treated by every train-mode BatchNorm as g independent batches
    of equal size (images [i*B/g, (i+1)*B/g) form batch i): per-group statistics, running statistics updated group
    after group.  One launch sequence then does the work of g separate forwards (and backwards) of B/g images -
    the trainers run the labelled and the unlabelled half of a step this way."""

    def __init__(self, g):
        self.g = int(g)

    def __enter__(self):
        global BN_GROUPS
        self.prev, BN_GROUPS = BN_GROUPS, self.g

    def __exit__(self, *exc):
        global BN_GROUPS
        BN_GROUPS = self.prev


BN_DEFER = None       # (from_group, slot) inside `with bn_defer(...)`, else None
_DEFERRED = {}        # slot -> {running_mean.data_ptr() -> dict(rm, rv, buf, C, n, momentum)}, in registration order
_DEFER_TABLE = {}     # slot -> (keys, device descriptor table) of the last apply_deferred_bn(slot)


class bn_defer:
    """`with ops.bn_defer(g0, slot):` - inside, the groups >= g0 of every train-mode BatchNorm forward compute and use their
    batch statistics as usual but POSTPONE their running-statistics momentum update; `ops.apply_deferred_bn(slot)` applies
    the postponed updates later (one launch for all layers).  The momentum updates do not commute: the reference
    runs model(l), model(cj2_l), model(u) (train_arco_2d.py:310-312) while the trainer here runs (l, u) as one grouped
    pass and cj2_l before or after it - with the u group deferred (slot 0) and, when it runs first, the cj2_l pass deferred
    as well (slot 1), the running statistics receive the three updates in the reference's order: l inside the grouped
    pass, then apply_deferred_bn(1), then apply_deferred_bn(0)."""

    def __init__(self, from_group=0, slot=0):
        self.v = (int(from_group), int(slot))

    def __enter__(self):
        global BN_DEFER
        self.prev, BN_DEFER = BN_DEFER, self.v

    def __exit__(self, *exc):
        global BN_DEFER
        BN_DEFER = self.prev


def _defer_args(running_mean, running_var, co, G, momentum):
    """(defer_from, deferred buffer or None) for the arco_bn_finalize call of one BN layer."""
    if BN_DEFER is None or running_mean is None or BN_DEFER[0] >= G:
        return 0, None
    g0, slot = BN_DEFER
    n = G - g0
    key = running_mean.data_ptr()
    tab = _DEFERRED.setdefault(slot, {})
    e = tab.get(key)
    if e is None or e["n"] != n or e["C"] != co:
        e = dict(rm=running_mean, rv=running_var, C=co, n=n, momentum=float(momentum),
                 buf=torch.zeros(n * 2 * co + 1, dtype=torch.float32, device=running_mean.device))
        tab[key] = e
        _DEFER_TABLE.pop(slot, None)
    return g0, e["buf"]


def apply_deferred_bn(slot=0):
    """Apply every postponed running-statistics update of a slot (see bn_defer); layers with nothing pending are skipped on
    the device (a flag per layer), so calling it when nothing was deferred is harmless."""
    tab = _DEFERRED.get(slot)
    if not tab:
        return
    keys = tuple(tab)
    if slot not in _DEFER_TABLE or _DEFER_TABLE[slot][0] != keys:
        import struct
        raw = b"".join(struct.pack("<QQQiifi", e["rm"].data_ptr(), e["rv"].data_ptr(), e["buf"].data_ptr(), e["C"], e["n"],
                                   e["momentum"], 0) for e in tab.values())
        assert len(raw) == len(keys) * L.query("arco_bn_defer_desc_bytes")
        dev = next(iter(tab.values()))["buf"].device
        _DEFER_TABLE[slot] = (keys, torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev))
    L.call("arco_bn_apply_deferred", L.ptr(_DEFER_TABLE[slot][1]), len(keys))


def _next_seed():
    """Dropout seeds come from a private generator so the torch CPU default generator
    (which defines the bit-exact sampler sequence) is never touched."""
    global _drop_gen
    if _drop_gen is None:
        _drop_gen = torch.Generator()
        _drop_gen.manual_seed(torch.initial_seed() ^ 0x5DEECE66D)
    return int(torch.randint(0, 2 ** 62, (1,), generator=_drop_gen))


def reseed_dropout(seed):
    global _drop_gen
    _drop_gen = torch.Generator()
    _drop_gen.manual_seed(seed)


def _ceil16(x):
    return (x + 15) // 16 * 16


def new_act(nb, c, h, w, dev):
    """Fresh channels-last activation, logical [nb, c, h, w]."""
    return torch.empty((nb, h, w, c), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)


def new_act_nd(n, c, spatial, dev, dtype=torch.float32):
    """Fresh channels-last activation, logical [n, c, *spatial] (2-D or 3-D)."""
    t = torch.empty((n, *spatial, c), dtype=dtype, device=dev)
    return t.movedim(-1, 1)


def _geom(x):
    nb, c, h, w = x.shape
    r, ld = rows_view(x)
    return r, ld, int(nb), int(c), int(h), int(w)


def _geom_nd(x):
    """rows, ld, n_volumes, planes-per-volume (1 for 2-D), H, W, C, spatial tuple."""
    r, ld = rows_view(x)
    sp = tuple(int(v) for v in x.shape[2:])
    if len(sp) == 2:
        return r, ld, int(x.shape[0]), 1, sp[0], sp[1], int(x.shape[1]), sp
    return r, ld, int(x.shape[0]), sp[0], sp[1], sp[2], int(x.shape[1]), sp


def _taps(weight):
    t = 1
    for k in weight.shape[2:]:
        t *= int(k)
    return t


# packed weights are reused until the weights change: optimizers / EMA bump WEIGHT_EPOCH.
# The cache lives ON the weight tensor object (no stale hits when memory is recycled).
WEIGHT_EPOCH = 0


_plans = weakref.WeakSet()
_PLAN_BY_PTR = {}       # data_ptr of a plan-owned GEMM-form weight W2 -> {mode: (plan flag, buf, sbuf, hbuf)} (the tensor object that
                        # reaches pack_weight is an autograd output sharing W2's storage: attributes do not travel, addresses do).
                        # The entries hold the plan's validity FLAG, not the plan, and leave with the plan (weakref.finalize below):
                        # a dropped stepper's W2 / pack buffers are freed and its addresses cannot serve a later tensor


class _PlanFlag:
    __slots__ = ("valid",)

    def __init__(self):
        self.valid = False


def _forget_plan_ptrs(pairs):
    for ptr, ent in pairs:
        if _PLAN_BY_PTR.get(ptr) is ent:
            del _PLAN_BY_PTR[ptr]


def bump_weight_epoch():
    """Weights changed in place: cached packs are stale; plans must be refreshed by their owners."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    for p in _plans:
        p.valid = False


class PackPlan:
    """Packed copies of every conv weight of a module, refreshed by ONE kernel launch (`refresh()`), instead of
    one pack launch per layer per forward.  The owner calls refresh() whenever the weights changed (optimiser
    step / EMA update); pack_weight() then serves the plan's buffers."""

    @property
    def valid(self):
        return self._flag.valid

    @valid.setter
    def valid(self, v):
        self._flag.valid = bool(v)

    def __init__(self, modules, with_dgrad, half=None):
        """half: one flag per top-level module - True: the module runs on f16 activations (ops.ACT_HALF: the V-Net body), its
        Conv3d weights get the f16 pack ONLY (the one-channel first layer keeps the fp32 pack its kernel reads); False / None:
        fp32 activations (heads), the fp32 / split-bf16 packs."""
        import struct
        recs, self.entries = [], []
        self._flag = _PlanFlag()
        by_ptr = []
        dev = None
        total = 0
        tops = modules if isinstance(modules, (list, tuple)) else [modules]
        flags = list(half) if half is not None else [False] * len(tops)
        assert len(flags) == len(tops)
        mods = [(m, bool(f)) for top, f in zip(tops, flags) for m in top.modules()]
        self.gemm_entries = []
        for m, m_half in mods:
            w = getattr(m, "weight", None)
            if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)) and w is not None and tuple(w.shape[2:]) == (2, 2, 2) \
                    and tuple(m.stride) == (2, 2, 2):
                # k2 s2 (transposed) convolution = space-to-depth / depth-to-space + a GEMM over W2 [N2][K2] (vnetWithArgs.py
                # DownsamplingConvBlock / UpsamplingDeconvBlock): W2, its packs (and the 8x repeated bias of the transposed
                # form) are persistent buffers of the plan, gathered from the torch layout by the plan's one launch
                up = isinstance(m, torch.nn.ConvTranspose3d)
                dev = w.device
                if up:
                    ci_, co_ = int(w.shape[0]), int(w.shape[1])
                    n2, k2, gm, g = 8 * co_, ci_, 2, co_
                else:
                    co_, ci_ = int(w.shape[0]), int(w.shape[1])
                    n2, k2, gm, g = co_, 8 * ci_, 1, ci_
                if n2 % 16 or k2 % 16:
                    continue                                  # (W2 doubles as the zero-copy [N][K] operand: whole 16-blocks only)
                w2 = torch.empty((n2, k2, 1, 1, 1), dtype=torch.float32, device=dev)
                recs.append((w, w2, n2, k2, g, gm << 3, n2, k2, total)); total += n2 * k2
                ent = {}
                for mode in ((0, 1) if with_dgrad and w.requires_grad else (0,)):
                    n, k = (n2, k2) if mode == 0 else (k2, n2)
                    buf = sbuf = hbuf = None
                    if m_half:
                        kp32 = (k + 31) // 32 * 32
                        hbuf = torch.empty((1, n, kp32), dtype=torch.float16, device=dev)
                        recs.append((w, hbuf, n2, k2, g, (gm << 3) | mode | 4, n, kp32, total)); total += n * kp32
                    else:
                        if mode == 1:
                            buf = torch.empty((1, n, k), dtype=torch.float32, device=dev)
                            recs.append((w, buf, n2, k2, g, (gm << 3) | mode, n, k, total)); total += n * k
                        if CONV_MMA == 3:
                            kp32 = (k + 31) // 32 * 32
                            sbuf = torch.empty((1, n, kp32 * 3 // 2), dtype=torch.float32, device=dev)
                            recs.append((w, sbuf, n2, k2, g, (gm << 3) | mode | 2, n, kp32, total)); total += n * kp32
                    ent[mode] = (self._flag, buf, sbuf, hbuf)
                bias8 = None
                if up and m.bias is not None:
                    bias8 = torch.empty(8 * co_, dtype=torch.float32, device=dev)
                    recs.append((m.bias, bias8, 1, 8 * co_, co_, 4 << 3, 1, 8 * co_, total)); total += 8 * co_
                ent["shape"] = (n2, k2)      # (the lookup checks it: an address match alone must not hand out another weight's packs)
                _PLAN_BY_PTR[w2.data_ptr()] = ent
                by_ptr.append((w2.data_ptr(), ent))
                self.gemm_entries.append((w, w2, bias8))
                w._arco_gemm = (self, w2, bias8)
                continue
            if not isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)) or w is None:
                continue
            dev = w.device
            co, ci = int(w.shape[0]), int(w.shape[1])
            taps = _taps(w)
            if taps not in (1, 9, 27):
                continue
            for mode in ((0, 1) if with_dgrad and w.requires_grad else (0,)):
                n, k = (co, ci) if mode == 0 else (ci, co)
                npad, kpad = _ceil16(n), _ceil16(k)
                buf = sbuf = hbuf = None
                if m_half and isinstance(m, torch.nn.Conv3d) and k % 8 == 0 and taps in (1, 27):   # f16 pack (csrc/conv_h.hip)
                    kp32 = (k + 31) // 32 * 32
                    hbuf = torch.empty((taps, npad, kp32), dtype=torch.float16, device=dev)
                    recs.append((w, hbuf, co, ci, taps, mode | 4, npad, kp32, total))
                    total += taps * npad * kp32
                    self.entries.append((w, mode, None, None, hbuf))
                    continue
                if not (mode == 0 and taps == 1 and co % 16 == 0 and ci % 16 == 0):   # else served zero-copy by pack_weight
                    buf = torch.empty((taps, npad, kpad), dtype=torch.float32, device=dev)
                    recs.append((w, buf, co, ci, taps, mode, npad, kpad, total))
                    total += buf.numel()
                if CONV_MMA == 3 and k % 4 == 0:       # split-bf16 operand format (3 bf16 per element, K padded to 32)
                    kp32 = (k + 31) // 32 * 32
                    sbuf = torch.empty((taps, npad, kp32 * 3 // 2), dtype=torch.float32, device=dev)
                    recs.append((w, sbuf, co, ci, taps, mode | 2, npad, kp32, total))
                    total += taps * npad * kp32
                self.entries.append((w, mode, buf, sbuf, hbuf))
        self.total, self.n = total, len(recs)
        self.valid = False
        _plans.add(self)
        weakref.finalize(self, _forget_plan_ptrs, by_ptr)
        if self.n:
            raw = b"".join(struct.pack("<QQiiiiiiq", w.data_ptr(), b.data_ptr(), co, ci, taps, mode, npad, kpad, first)
                           for (w, b, co, ci, taps, mode, npad, kpad, first) in recs)
            assert len(raw) == self.n * L.query("arco_pack_desc_bytes")
            self.desc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            self.ptrs = [(w.data_ptr(), w) for (w, *_r) in recs]
            for w, mode, buf, sbuf, hbuf in self.entries:
                d = getattr(w, "_arco_plan", None)
                if d is None:
                    d = {}
                    w._arco_plan = d
                d[mode] = (self, buf, sbuf, hbuf)

    def refresh(self):
        """Re-pack every weight of the plan (one launch).  Called by whoever changed the weights."""
        if self.n:
            for ptr, w in self.ptrs:
                if w.data_ptr() != ptr:
                    raise RuntimeError("PackPlan: a parameter moved after the plan was built (build plans after "
                                       "the optimiser has flattened the parameters)")
            L.call("arco_pack_many", L.ptr(self.desc), self.n, self.total)
        self.valid = True


def conv_sp_set(on):
    """Switch the pipelined split-bf16 3x3 kernels on / off (A/B runs, tests) and return the previous setting.  The
    per-shape answers cached from the library (`arco_conv_split_ok`, tile configurations) depend on the switch for the
    few-channel and 16->16 shapes, so the cache is dropped with it - a stale "split ok" would hand split-packed weights
    to a kernel that no longer takes them (ARCO_ERR_UNSUPPORTED instead of the fp32 fallback)."""
    prev = L.query("arco_conv_sp_set", int(on))
    _cfg_cache.clear()
    return prev


def conv3d_fl_set(on):
    """Switch the pipelined flat-tile 3x3x3 kernel (csrc/conv3d_fl.hip) on / off (A/B runs, tests); returns the previous setting."""
    prev = L.query("arco_conv3d_fl_set", int(on))
    _cfg_cache.clear()
    return prev


def _split_ok(taps, nbd, h, w, k, n, ld):
    key = ("split", taps, nbd, h, w, k, n, ld)
    r = _cfg_cache.get(key)
    if r is None:
        r = _cfg_cache[key] = bool(L.query("arco_conv_split_ok", taps, nbd, h, w, k, n, ld))
    return r


def _plan_by_ptr(weight):
    """The plan entry of a plan-owned GEMM-form weight W2 (ops.gemm_weight hands out autograd outputs that share W2's storage)."""
    ent = _PLAN_BY_PTR.get(weight.data_ptr())
    if ent is not None and ent.get("shape") == (int(weight.shape[0]), int(weight.shape[1])) and weight.dim() == 5:
        return ent
    return None


def _pack_now(w, co, ci, taps, mode, split, half=False):
    n, k = (co, ci) if mode == 0 else (ci, co)
    if half:
        wp = torch.empty((taps, _ceil16(n), (k + 31) // 32 * 32), dtype=torch.float16, device=w.device)
        L.call("arco_pack_conv_weight", L.ptr(w), co, ci, taps, mode | 4, L.ptr(wp))
        return wp
    if split:
        wp = torch.empty((taps, _ceil16(n), (k + 31) // 32 * 32 * 3 // 2), dtype=torch.float32, device=w.device)
    else:
        wp = torch.empty((taps, _ceil16(n), _ceil16(k)), dtype=torch.float32, device=w.device)
    L.call("arco_pack_conv_weight", L.ptr(w), co, ci, taps, mode | (2 if split else 0), L.ptr(wp))
    return wp


def _pack_half(weight, taps, mode):
    """The f16 pack [taps][Npad][ceil32(K)] of a conv weight (f16 activation storage): from the PackPlan, else per weight-epoch."""
    plan = getattr(weight, "_arco_plan", None) or _plan_by_ptr(weight)
    if plan is not None and mode in plan and plan[mode][0].valid and plan[mode][3] is not None:
        return plan[mode][3]
    capturing = torch.cuda.is_current_stream_capturing()
    cache = getattr(weight, "_arco_pack_h", None)
    key = (mode, weight._version)
    if cache is not None and not capturing:
        hit = cache.get(key)
        if hit is not None and hit[0] == WEIGHT_EPOCH:
            return hit[1]
    co, ci = int(weight.shape[0]), int(weight.shape[1])
    wp = _pack_now(weight.detach().contiguous(), co, ci, taps, mode, False, half=True)
    if not capturing:
        try:
            if cache is None:
                cache = weight._arco_pack_h = {}
            cache.clear() if len(cache) > 4 else None
            cache[key] = (WEIGHT_EPOCH, wp)
        except AttributeError:
            pass
    return wp


def pack_weight(weight, taps, mode, half=False):
    """torch [Cout, Cin, kh, kw] -> Wp[taps][Npad][Kpad]; mode 0 forward, 1 dgrad (flipped+transposed).
    Served from the module's PackPlan when one is valid, else cached per weight-epoch.  With CONV_MMA == 3 the returned
    tensor carries `_arco_split`: the same weights in the split-bf16 operand format (conv_raw picks the one the kernel takes).
    half=True: the f16 pack of the f16-storage kernels."""
    if half:
        return _pack_half(weight, taps, mode)
    co, ci = int(weight.shape[0]), int(weight.shape[1])
    w = weight.detach()
    want_split = CONV_MMA == 3 and (ci if mode == 0 else co) % 4 == 0
    zero_copy = mode == 0 and taps == 1 and co % 16 == 0 and ci % 16 == 0 and w.is_contiguous()
    plan = getattr(weight, "_arco_plan", None) or _plan_by_ptr(weight)
    if plan is not None and mode in plan and plan[mode][0].valid:
        _, buf, sbuf, _h = plan[mode]
        if (buf is not None or zero_copy) and (sbuf is not None or not want_split):
            wp = buf if buf is not None else w.view(co, ci)
            wp._arco_split = sbuf if want_split else None
            return wp
    capturing = torch.cuda.is_current_stream_capturing()
    cache = getattr(weight, "_arco_pack", None)
    key = (mode, weight._version, want_split)
    if cache is not None and not capturing:
        hit = cache.get(key)
        if hit is not None and hit[0] == WEIGHT_EPOCH:
            return hit[1]
    w = w.contiguous()
    wp = w.view(co, ci) if zero_copy else _pack_now(w, co, ci, taps, mode, False)      # zero copy: already [N][K]
    wp._arco_split = _pack_now(w, co, ci, taps, mode, True) if want_split else None
    if not capturing:
        if cache is None:
            cache = {}
            try:
                weight._arco_pack = cache
            except AttributeError:
                return wp
        cache.clear() if len(cache) > 4 else None
        cache[key] = (WEIGHT_EPOCH, wp)
    return wp


def use_half(x, taps, ci):
    """Does a convolution of x run on the f16-storage kernels?  Yes when x is f16, or when x is the fp32 one-channel volume
    entering the V-Net's first 3x3x3 layer in f16 mode (that layer's output opens the f16 region)."""
    return _is_half(x) or (ACT_HALF and taps == 27 and ci == 1 and x.dim() == 5)


def conv_raw(xr, ld, k, wp, n, nb, h, w, taps, bias=None, residual=None, ld_res=0, stats=False, d3=1, sp=None,
             stat_groups=1, grad=False, half=False, pro=None, out=None, pro_groups=1):
    """out[pix][0..n) = conv(x)(+bias)(+residual); returns (out channels-last, stat slabs or None).
    2-D: nb images of h x w -> out [nb,n,h,w].  3-D (d3 > 1): nb volumes of d3 planes -> out [nb,n,d3,h,w].
    half: f16 activation storage - out (and xr, unless k == 1) are f16, wp is the f16 pack (k == 1: the fp32 pack).
    pro: an _lib.act_pro descriptor - xr is the PRE-activation of the producing layer, activated in the loader (the caller has
    asked pro_ok).  out: write into this channels-last tensor (may be a channel slice of a wider buffer)."""
    odt = torch.float16 if half else torch.float32
    outr = None
    if out is not None:
        outr, ld_out = rows_view(out)
        if outr.data_ptr() != out.data_ptr():
            raise RuntimeError("arco_amd: conv_raw(out=...) needs a channels-last tensor (or channel slice) with 16-byte aligned rows")
    elif sp is not None:
        out = new_act_nd(nb, n, sp, xr.device, odt)    # keeps the caller's rank (a 3-D volume may have depth 1)
    elif d3 > 1:
        out = new_act_nd(nb, n, (d3, h, w), xr.device, odt)
    else:
        out = new_act(nb, n, h, w, xr.device)
    if outr is None:
        outr, ld_out = out, n
    mma = 0
    if half:
        if residual is not None or (d3 <= 1 and sp is None):
            raise RuntimeError("arco_amd: f16 activation storage covers the volume path's convolutions without a residual operand")
        mma = 4
    elif HEAD_MMA and taps == 1 and pro is None:
        mma = 2 if grad else HEAD_MMA                     # (wp is the fp32 pack: the operands are rounded in registers)
    elif CONV_MMA == 3:
        sp_ = getattr(wp, "_arco_split", None)
        if sp_ is not None and _split_ok(taps, nb * d3, h, w, k, n, ld):
            wp, mma = sp_, 3
    elif CONV_MMA and taps in (1, 27) and d3 > 1:
        mma = 2 if grad else CONV_MMA                     # gradient operands: bf16 (range)
    ssum = ssq = None
    nmb = 0
    if stats:
        if pro is not None and taps == 27:       # the 3x3x3 forms with the activation in their loaders tile differently
            nmb = L.query("arco_conv_mblocks_pro", taps, nb * d3, h, w, k, n, ld, stat_groups, mma, pro_groups)
        else:
            nmb = L.query("arco_conv_mblocks_mma", taps, nb * d3, h, w, k, n, ld, stat_groups, mma)
        ssum = torch.empty((n, nmb), dtype=torch.float32, device=xr.device)
        ssq = torch.empty((n, nmb), dtype=torch.float32, device=xr.device)
    if WORK is not None:
        esz = 2 if half else 4
        mpix = nb * d3 * h * w
        _work(mpix * (k + n) * esz + taps * n * k * 4 + (mpix * n * esz if residual is not None else 0), 2.0 * taps * mpix * n * k)
    prof = cfg = None
    if PROFILE is not None and not torch.cuda.is_current_stream_capturing():
        # every launch is counted; every PROFILE_EVERY-th one is bracketed by HIP events on the launch stream
        # (an event pair per launch costs ~1.5 ms/step of stream bubbles at ~350 conv launches per step)
        key = (taps, nb * d3, h, w, k, n, ld, mma)
        cfg = _cfg_cache.get(key)
        if cfg is None:        # kernel instantiation id + 1e8 * matrix-core mode
            cfg = _cfg_cache[key] = L.query("arco_conv_config_mma", *key) + 100000000 * mma
        if pro is not None:
            cfg += 50000000          # the PRO instantiation (consumer-side activation in the loader) is a kernel of its own in the records
        rec = PROFILE.setdefault(cfg, {"n": 0, "flop": 0.0, "timed": []})
        rec["n"] += 1
        rec["flop"] += 2.0 * taps * nb * d3 * h * w * n * k
        if rec["n"] % PROFILE_EVERY == 1 % PROFILE_EVERY:
            prof = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            prof[0].record()
    if pro is not None:
        if mma != 3:
            raise RuntimeError("arco_amd: a consumer-side activation needs the split-bf16 kernels (ops.pro_ok)")
        L.call("arco_conv3d_fwd_pro", L.ptr(xr), ld, k, L.ptr(wp), n, L.ptr(outr), ld_out, L.ptr(bias), L.ptr(residual), ld_res,
               L.ptr(ssum), L.ptr(ssq), taps, nb, d3, h, w, stat_groups if stats else 1, mma, pro)
    else:
        L.call("arco_conv3d_fwd", L.ptr(xr), ld, k, L.ptr(wp), n, L.ptr(outr), ld_out, L.ptr(bias), L.ptr(residual), ld_res,
               L.ptr(ssum), L.ptr(ssq), taps, nb, d3, h, w, stat_groups if stats else 1, mma)
    if prof is not None:
        prof[1].record()
        PROFILE[cfg]["timed"].append((prof[0], prof[1], 2.0 * taps * nb * d3 * h * w * n * k, (taps, nb * d3 * h * w, n, k)))
    return out, (ssum, ssq, nmb)


def conv_wgrad(dzr, ldz, co, xr, ldx, ci, taps, nb, h, w, like, d3=1, pro=None):
    ws = torch.empty(L.query("arco_wgrad_ws_floats", co, ci, taps, nb * d3 * h * w), dtype=torch.float32,
                     device=dzr.device)

    def compute(out, accumulate):
        prof = None
        if WORK is not None:
            mpix = nb * d3 * h * w
            _work(mpix * (co + ci) * (2 if _is_half(dzr) else 4) + taps * co * ci * 4, 2.0 * taps * mpix * co * ci, 2)      # + the slab reduction
        if PROFILE is not None and not torch.cuda.is_current_stream_capturing():
            flop = 2.0 * taps * nb * d3 * h * w * co * ci
            rec = PROFILE.setdefault(("wgrad", taps, co, ci), {"n": 0, "flop": 0.0, "timed": []})
            rec["n"] += 1
            rec["flop"] += flop
            if rec["n"] % PROFILE_EVERY == 1 % PROFILE_EVERY:
                prof = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                prof[0].record()
        if _is_half(dzr):     # f16 activation storage (xr is f16 too, or the first layer's fp32 one-channel volume)
            mma = 4
        else:
            mma = 3 if (CONV_MMA == 3 and taps in (9, 27)) else (2 if (CONV_MMA in (1, 2) and taps == 27) else 0)
        if pro is not None:
            L.call("arco_conv3d_wgrad_pro", L.ptr(dzr), ldz, co, L.ptr(xr), ldx, ci, taps, nb, d3, h, w, L.ptr(ws), L.ptr(out),
                   accumulate, mma, pro)
        else:
            L.call("arco_conv3d_wgrad", L.ptr(dzr), ldz, co, L.ptr(xr), ldx, ci, taps, nb, d3, h, w, L.ptr(ws), L.ptr(out),
                   accumulate, mma)
        if prof is not None:
            prof[1].record()
            rec["timed"].append((prof[0], prof[1], flop, (taps, nb * d3 * h * w, co, ci)))
    return _grad_into(like, compute)


# ---- weight gradients on a second stream (round 4) --------------------------------------------------------------------------
# In a layer's backward the weight gradient depends only on dZ and the saved input; the critical chain is
# BN-backward -> data gradient -> the next layer's BN-backward.  The weight-gradient launches (wgrad + slab reduction: 2.4 ms of
# the 2-D step, 108 launches that each fill the chip for 10-50 us with start-up / tail phases) are queued on a side stream,
# forked behind the kernel that produced dZ and joined once at the end of the backward pass (join_side: GraphedTrain's
# capture, the trainers after loss.backward(), the data-parallel bucket hook) - inside a HIP-graph capture the fork / join
# become graph edges.  Only gradients that go straight into the optimiser's flat buffer take the side stream (nobody reads
# them before the join); accumulation into one parameter from two passes stays ordered (one side stream, launch order).
# Tensors the side stream reads are held until the join, so the caching allocator cannot hand their memory to a kernel on
# the main stream while the side stream still reads it.  A/B: ARCO_WGRAD_SIDE=0..3.
# Measured (round 4, one box, alternating runs; profiles/r04_notes.md): the graph-replayed 2-D step gets SLOWER with it (13.4 ->
# 14.2 ms forked before the data gradient, 13.6-13.9 forked behind it: the persistent one-workgroup-per-CU kernels lose more
# to a co-resident weight-gradient workgroup than the tails give back); the 3-D LA step gains 3 % only with EAGER student passes
# (30.1 -> 29.2 ms: two queues hide host launch time), nothing under the default graph replay (30.1-30.6 both ways; LiTS f16
# 20.5 vs 20.7).  Hence: off by default, kept as a knob (ARCO_WGRAD_SIDE=1..3).
# Round 6: with the pipelined 3x3x3 kernels the LA step gains 0.3-0.5 ms from mode 3 under graph replay (24.9 / 24.8 / 25.7 -> 24.6 / 24.4 / 25.2 ms,
# same box, alternating; LiTS-f16 level, the 2-D step 11.0-11.2 -> 12.2-12.3): the 3-D trainer's constructor sets 3, the 2-D trainer's 0,
# unless ARCO_WGRAD_SIDE is given.
_WGRAD_SIDE_ENV = __import__('os').environ.get('ARCO_WGRAD_SIDE')
WGRAD_SIDE = int(_WGRAD_SIDE_ENV) if _WGRAD_SIDE_ENV is not None else 0
_side = {"stream": None, "keep": [], "dirty": False}


def _wgrad(dzr, ldz, co, xr, ldx, ci, taps, nb, h, w, like, d3=1, keep=(), pro=None):
    view = getattr(like, "_arco_grad_view", None)
    if not (WGRAD_SIDE and view is not None and like.grad is not None and like.grad.data_ptr() == view.data_ptr()):
        return conv_wgrad(dzr, ldz, co, xr, ldx, ci, taps, nb, h, w, like, d3=d3, pro=pro)
    cur = torch.cuda.current_stream()
    if _side["stream"] is None:
        _side["stream"] = torch.cuda.Stream()
    sd = _side["stream"]
    sd.wait_stream(cur)                                  # fork: behind everything queued so far (dZ is complete)
    with torch.cuda.stream(sd):
        out = conv_wgrad(dzr, ldz, co, xr, ldx, ci, taps, nb, h, w, like, d3=d3, pro=pro)
    _side["keep"].append((dzr, xr) + tuple(keep))
    if not _side["dirty"]:
        # joined at the end of THIS backward pass whoever runs it (loss.backward(), autograd.grad inside a graph capture, a
        # test): the engine calls back on the calling thread once every node has run
        torch.autograd.Variable._execution_engine.queue_callback(join_side)
    _side["dirty"] = True
    return out


def join_side():
    """The current stream waits for the side stream's weight gradients (no-op when none were queued)."""
    if _side["dirty"]:
        torch.cuda.current_stream().wait_stream(_side["stream"])
        _side["keep"].clear()
        _side["dirty"] = False


_zero_cache = {}


def _zeros_cached(shape, device):
    """Shared read-only zero tensor (never written): the gradient of a conv bias under train-mode BN."""
    key = (tuple(shape), device)
    z = _zero_cache.get(key)
    if z is None:
        z = torch.zeros(shape, dtype=torch.float32, device=device)
        _zero_cache[key] = z
    return z


def _grad_into(param, compute):
    """Weight gradients go straight into the optimiser's flat gradient view when the parameter has one
    (param._arco_grad_view, installed by optim.SGDNesterov): `compute(out, accumulate)` writes there with
    accumulate=1 and autograd gets None - no per-parameter AccumulateGrad add kernel.  Otherwise the gradient
    is returned normally."""
    view = getattr(param, "_arco_grad_view", None)
    if view is not None and param.grad is not None and param.grad.data_ptr() == view.data_ptr():
        compute(view, 1)
        param._arco_mark()
        return None
    out = torch.empty_like(param, memory_format=torch.contiguous_format)
    compute(out, 0)
    return out


def colsum(xr, ld, m, c):
    ws = torch.empty(1024 * c, dtype=torch.float32, device=xr.device)
    out = torch.empty(c, dtype=torch.float32, device=xr.device)
    L.call("arco_colsum_h" if _is_half(xr) else "arco_colsum", L.ptr(xr), ld, m, c, L.ptr(ws), L.ptr(out), 0)
    return out


# Row-sparse backward of the dense per-pixel layers.  The reference's trainer calls FeatureExtractor and q_representation on whole
# feature maps (model_2D.py:51-53, train_arco_2d.py:231-234, 324-326: [16 x 65 536] rows of 496 channels) and its loss reads `rep` at
# <= num_queries sampled rows per class (loss_helper_3d.py:455-457): d loss / d rep is zero in every other row, and because these
# layers are per-pixel so is every gradient down to the first resampling.  Autograd hands ConvFn.backward the dense tensor; one pass
# over it finds the rows that hold anything (arco_row_nonzero), and when they are few the data gradient and the weight gradient run
# on those rows alone: the same kernels on [n, C] instead of [M, C] (data-gradient rows bit-identical - a row's result does not depend
# on the other rows; weight gradients differ by fp32 summation order only), the rest of dx is a memset.  Needs the count on the host
# (one synchronisation per layer): eager backward only, never inside a graph capture.  ARCO_SPARSE_BWD=0: always dense.
SPARSE_BWD = int(__import__('os').environ.get('ARCO_SPARSE_BWD', '1'))
SPARSE_BWD_MIN_ROWS = 65536
SPARSE_BWD_MIN_CH = 128          # the heads' widths (496 ... 384; 3-D 240 ... 128); the U-Net's own 1x1 convs stay below it
SPARSE_BWD_MAX_FRAC = 0.125
sparse_bwd_stats = {"sparse": 0, "dense": 0}      # how often each route was taken (tests, bench)


# A layer whose incoming gradient keeps turning out DENSE stops paying for the probe (one pass over dy + one host synchronisation per
# backward): after SPARSE_BWD_GIVE_UP dense verdicts in a row the probe is skipped for SPARSE_BWD_RETRY backward calls of that
# weight, then tried again.  The verdict lives on the weight tensor object.
SPARSE_BWD_GIVE_UP, SPARSE_BWD_RETRY = 3, 64


def _sparse_probe_due(weight):
    st = getattr(weight, "_arco_sparse_probe", None)
    if st is None or st[1] <= 0:
        return True
    st[1] -= 1
    return False


def _sparse_probe_result(weight, was_sparse):
    try:
        st = getattr(weight, "_arco_sparse_probe", None)
        if st is None:
            st = weight._arco_sparse_probe = [0, 0]          # [dense verdicts in a row, backward calls to skip]
        st[0] = 0 if was_sparse else st[0] + 1
        if st[0] >= SPARSE_BWD_GIVE_UP:
            st[0], st[1] = 0, SPARSE_BWD_RETRY
    except AttributeError:
        pass


def _conv1x1_backward_on_nonzero_rows(ctx, dy, dyr, ldy, co, xr, ldx, ci, x, weight):
    M = int(dyr.shape[0])
    dev = dyr.device
    flag = torch.empty(M, dtype=torch.uint8, device=dev)
    L.call("arco_row_nonzero", L.ptr(dyr), ldy, co, M, L.ptr(flag))
    idx = torch.nonzero(flag).view(-1)                   # ascending row ids; synchronises (the count sizes the launches below)
    n = int(idx.shape[0])
    if n > SPARSE_BWD_MAX_FRAC * M:
        sparse_bwd_stats["dense"] += 1
        return None
    sparse_bwd_stats["sparse"] += 1
    n_pad = max(16, (n + 15) // 16 * 16)                 # whole 16-row MFMA tiles; the padding rows are zero
    dyc = torch.zeros((n_pad, co), dtype=torch.float32, device=dev)
    dx = dw = db = None
    if n:
        L.call("arco_gather_rows", L.ptr(dyr), ldy, co, None, L.ptr(idx), None, 0, n, L.ptr(dyc), co)
    if ctx.needs_input_grad[1]:
        xc = torch.zeros((n_pad, ci), dtype=torch.float32, device=dev)
        if n:
            L.call("arco_gather_rows", L.ptr(xr), ldx, ci, None, L.ptr(idx), None, 0, n, L.ptr(xc), ci)
        dw = _wgrad(dyc, co, co, xc, ci, ci, 1, 1, 1, n_pad, weight, keep=(dyc, xc))
    if ctx.needs_input_grad[0]:
        wd = pack_weight(weight, 1, 1)
        dxc, _ = conv_raw(dyc, co, co, wd, ci, 1, 1, n_pad, 1, grad=True, residual=dyc if ctx.residual else None,
                          ld_res=co if ctx.residual else 0)
        dxc_r, ldc = rows_view(dxc)
        dx = torch.zeros((int(x.shape[0]), *[int(v) for v in x.shape[2:]], ci), dtype=torch.float32, device=dev).movedim(-1, 1)
        if n:
            dxr, lddx = rows_view(dx)
            L.call("arco_put_rows", L.ptr(dxc_r), ldc, ci, L.ptr(idx), n, L.ptr(dxr), lddx)
    if ctx.has_bias and ctx.needs_input_grad[2]:
        db = _zeros_cached((co,), dev) if ctx.bias_grad_zero else colsum(dyc, co, n_pad, co)
    return dx, dw, db, (dy if ctx.res_tensor and ctx.needs_input_grad[3] else None), None


class ConvFn(torch.autograd.Function):
    """y = conv(x) (+ bias) (+ x when residual) for 1x1 / 3x3 (2-D) and 1x1x1 / 3x3x3 (3-D) kernels.
    Reference: nn.Conv2d in unetWithArgs.py:72,139 / model_2D.py:25-33 / train_arco_2d.py:231-234;
    nn.Conv3d in vnetWithArgs.py:182, model_3D.py:25-35, train_arco_3d.py:206-209."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bias_grad_zero=False):
        L.require_gpu(x, weight)
        ctx.bias_grad_zero = bool(bias_grad_zero)
        taps = _taps(weight)
        xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
        co = int(weight.shape[0])
        half = use_half(x, taps, ci)
        wp = pack_weight(weight, taps, 0, half=half and ci != 1)
        if isinstance(residual, torch.Tensor):              # y = conv(x) + R  (R: any tensor of y's shape)
            rr, ldr = rows_view(residual)
            res_self = False
        else:                                               # True: y = conv(x) + x
            rr, ldr = (xr, ld) if residual else (None, 0)
            res_self = bool(residual)
        y, _ = conv_raw(xr, ld, ci, wp, co, nv, h, w, taps, bias=bias, residual=rr, ld_res=ldr, d3=d3, sp=sp, half=half)
        ctx.save_for_backward(x, weight)
        ctx.residual, ctx.has_bias, ctx.taps = res_self, bias is not None, taps
        ctx.res_tensor = isinstance(residual, torch.Tensor)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        taps = ctx.taps
        dyr, ldy, nv, d3, h, w, co, sp = _geom_nd(dy)
        xr, ldx = rows_view(x)
        ci = int(x.shape[1])
        dx = dw = db = None
        if taps == 1 and SPARSE_BWD and nv * d3 * h * w >= SPARSE_BWD_MIN_ROWS and co >= SPARSE_BWD_MIN_CH and ci >= SPARSE_BWD_MIN_CH and not _is_half(dy) \
                and xr.dtype == torch.float32 and not torch.cuda.is_current_stream_capturing() and _sparse_probe_due(weight):
            r = _conv1x1_backward_on_nonzero_rows(ctx, dy, dyr, ldy, co, xr, ldx, ci, x, weight)
            _sparse_probe_result(weight, r is not None)
            if r is not None:
                return r
        if ctx.needs_input_grad[1]:      # first: it forks to the side stream and runs beside the data gradient
            dw = _wgrad(dyr, ldy, co, xr, ldx, ci, taps, nv, h, w, weight, d3=d3, keep=(dy, x))
        if ctx.needs_input_grad[0]:
            half = _is_half(dy)
            wd = pack_weight(weight, taps, 1, half=half)
            dx, _ = conv_raw(dyr, ldy, co, wd, ci, nv, h, w, taps, grad=True, residual=dyr if ctx.residual else None,
                             ld_res=ldy if ctx.residual else 0, d3=d3, sp=sp, half=half)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # bias_grad_zero: the output feeds a train-mode BatchNorm, whose backward makes the column sums of dy
            # exactly zero in exact arithmetic (see ConvBnActFn.backward) - no pass over dy for rounding noise
            db = _zeros_cached((co,), dy.device) if ctx.bias_grad_zero else colsum(dyr, ldy, nv * d3 * h * w, co)
        return dx, dw, db, (dy if ctx.res_tensor and ctx.needs_input_grad[3] else None), None


def _bn_apply(zr, ldz, m, co, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, out, ld_out=None, groups=1):
    seed_dev = SEED_DEV if (p > 0 and torch.cuda.is_current_stream_capturing()) else None
    _work(2 * m * co * (2 if _is_half(zr) else 4))
    L.call("arco_bn_act_fwd_h" if _is_half(zr) else "arco_bn_act_fwd", L.ptr(zr), ldz, m, co, L.ptr(mean), L.ptr(istd),
           L.ptr(gamma), L.ptr(beta), float(slope), int(drop_mode), float(p), seed, P, L.ptr(out), co if ld_out is None else ld_out,
           L.ptr(seed_dev), groups)
    return seed_dev


def _bn_backward(da, z, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, seed_dev=None, groups=1):
    """Returns (dz, dgamma, dbeta) for a = drop(lrelu(BN(z)))."""
    dar, ldd = rows_view(da)
    zr, ldz = rows_view(z)
    co = int(z.shape[1])
    m = zr.shape[0]
    nblk = L.query("arco_chan_stats_blocks", m // groups)
    ws = torch.empty(groups * (2 * co * nblk + 2 * co), dtype=torch.float32, device=da.device)
    half = _is_half(zr)
    if half != _is_half(dar):          # a gradient that reached an f16 activation in fp32 (or the reverse): follow the activation
        dar = dar.to(zr.dtype)
        ldd = dar.stride(0)
    dz = new_act_nd(int(z.shape[0]), co, tuple(int(v) for v in z.shape[2:]), da.device, zr.dtype)
    dgamma = dbeta = None
    acc = 0
    if gamma is not None:
        gv, bv = getattr(gamma, "_arco_grad_view", None), getattr(beta, "_arco_grad_view", None)
        if (gv is not None and bv is not None and gamma.grad is not None and beta.grad is not None
                and gamma.grad.data_ptr() == gv.data_ptr() and beta.grad.data_ptr() == bv.data_ptr()):
            dg_t, db_t, acc = gv, bv, 1               # straight into the optimiser's flat gradient buffer
            gamma._arco_mark(); beta._arco_mark()
        else:
            dg_t = dgamma = torch.empty_like(gamma)
            db_t = dbeta = torch.empty_like(beta)
    else:
        dg_t = db_t = None
    _work(5 * m * co * (2 if half else 4), 0.0, 3)     # reduce: dA, Z; apply: dA, Z -> dZ (algorithmic: dA, Z -> dZ would be 3)
    L.call("arco_bn_act_bwd_h" if half else "arco_bn_act_bwd", L.ptr(dar), ldd, L.ptr(zr), ldz, m, co, L.ptr(mean), L.ptr(istd),
           L.ptr(gamma), L.ptr(beta), slope, drop_mode, p, seed, P, L.ptr(ws), L.ptr(dg_t), L.ptr(db_t), acc, L.ptr(dz), co,
           L.ptr(seed_dev), groups)
    return dz, dgamma, dbeta


class ConvBnActFn(torch.autograd.Function):
    """a = dropout(lrelu(BN_train(conv(x) + bias))) as conv (+fused BN partial stats in the epilogue) ->
    finalize -> one apply pass.  2-D 3x3 (ConvBlock stages, unetWithArgs.py:36-44), 3-D 3x3x3 and the
    GEMM-form k2s2 down conv (vnetWithArgs.py:16-25,67-91)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, slope, p, drop_mode, momentum, eps,
                nbt=None, cat_room=0, pool=False):
        global _LAST_CAT_BUF
        L.require_gpu(x, weight)
        taps = _taps(weight)
        xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
        co = int(weight.shape[0])
        m = nv * d3 * h * w
        half = use_half(x, taps, ci)
        wp = pack_weight(weight, taps, 0, half=half and ci != 1)
        G = BN_GROUPS
        if G > 1 and nv % G != 0:
            raise RuntimeError(f"arco_amd: bn_groups({G}) needs a batch that is a multiple of {G}, got {nv}")
        z, (ssum, ssq, nmb) = conv_raw(xr, ld, ci, wp, co, nv, h, w, taps, bias=bias, stats=True, d3=d3, sp=sp,
                                       stat_groups=G, half=half)
        mean, istd = _finalize_bn(ssum, ssq, nmb, co, m, eps, momentum, running_mean, running_var, nbt, G, x.device)
        seed = _next_seed() if p > 0 else 0
        if cat_room:        # leave room behind the channels for a later in-place channel concat (ops.upcat)
            buf = new_act_nd(nv, co + int(cat_room), sp, x.device)
            a, ld_a = buf[:, :co], co + int(cat_room)
            _LAST_CAT_BUF = buf
        else:
            a, ld_a = new_act_nd(nv, co, sp, x.device, z.dtype), co
        zr, ldz = rows_view(z)
        ctx.pool = bool(pool)
        ctx.groups = G
        ctx.cfg = (taps, float(slope), float(p), int(drop_mode), seed, bias is not None)
        ctx.bias_param = bias
        if pool:            # the activation AND its 2x2 max-pool in one pass (encoder blocks: next DownBlock + decoder skip)
            if p > 0 or d3 != 1:
                raise RuntimeError("arco_amd: conv_bn_act(pool=True) is the 2-D, dropout-free last stage of a ConvBlock")
            pooled = new_act(nv, co, h // 2, w // 2, x.device)
            _work((2 * m + m // 4) * co * 4)
            L.call("arco_bn_act_pool_fwd", L.ptr(zr), ldz, nv, h, w, co, L.ptr(mean), L.ptr(istd), L.ptr(gamma), L.ptr(beta),
                   float(slope), L.ptr(a), ld_a, L.ptr(pooled), co, G)
            ctx.seed_dev = None
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(x, weight, z, mean, istd, gamma, beta, a)
            return a, pooled
        ctx.seed_dev = _bn_apply(zr, ldz, m, co, mean, istd, gamma, beta, slope, drop_mode, p, seed, d3 * h * w, a,
                                 ld_a, G)
        ctx.save_for_backward(x, weight, z, mean, istd, gamma, beta)
        return a

    @staticmethod
    def backward(ctx, da, dpool=None):
        if ctx.pool:
            x, weight, z, mean, istd, gamma, beta, a = ctx.saved_tensors
        else:
            x, weight, z, mean, istd, gamma, beta = ctx.saved_tensors
        taps, slope, p, drop_mode, seed, has_bias = ctx.cfg
        xr, ldx, nv, d3, h, w, ci, sp = _geom_nd(x)
        co = int(weight.shape[0])
        if ctx.pool and dpool is not None:       # d a = d skip + maxpool2_bwd(d pooled), summed inside the pooling backward
            ar, lda_ = rows_view(a)
            dpr, ldp = rows_view(dpool)
            dsum = new_act(nv, co, h, w, x.device)
            _work((nv * h * w * (3 if da is not None else 2) + nv * h * w // 4) * co * 4)
            if da is None:
                L.call("arco_maxpool2_bwd", L.ptr(ar), lda_, nv, h, w, co, L.ptr(dpr), ldp, L.ptr(dsum), co)
            else:
                sr, lds = rows_view(da)
                L.call("arco_maxpool2_bwd_add", L.ptr(ar), lda_, nv, h, w, co, L.ptr(dpr), ldp, L.ptr(sr), lds, L.ptr(dsum), co)
            da = dsum
        dz, dgamma, dbeta = _bn_backward(da, z, mean, istd, gamma, beta, slope, drop_mode, p, seed, d3 * h * w,
                                         ctx.seed_dev, ctx.groups)
        dzr, ldzz = rows_view(dz)
        dx = dw = db = None
        # WGRAD_SIDE 1: the weight gradient forks first and runs beside this layer's data gradient; 2: it forks behind the data
        # gradient (beside the next layer's BatchNorm backward passes); 3: as 2, and the next data gradient waits for it
        if ctx.needs_input_grad[1] and WGRAD_SIDE < 2:
            dw = _wgrad(dzr, ldzz, co, xr, ldx, ci, taps, nv, h, w, weight, d3=d3, keep=(dz, x))
        if ctx.needs_input_grad[0]:
            half = _is_half(dz)
            wd = pack_weight(weight, taps, 1, half=half)
            if WGRAD_SIDE == 3 and _side["dirty"]:
                torch.cuda.current_stream().wait_stream(_side["stream"])
            dx, _ = conv_raw(dzr, ldzz, co, wd, ci, nv, h, w, taps, d3=d3, sp=sp, grad=True, half=half)
        if ctx.needs_input_grad[1] and WGRAD_SIDE >= 2:
            dw = _wgrad(dzr, ldzz, co, xr, ldx, ci, taps, nv, h, w, weight, d3=d3, keep=(dz, x))
        if has_bias and ctx.needs_input_grad[2]:
            # a conv bias under train-mode BN has an analytically ZERO gradient (BN removes the channel mean):
            # sum(dz) = -gamma*istd*mean(dy*xhat)*sum(xhat) and sum(xhat) == 0.  The reference's autograd
            # produces fp32 rounding noise (~1e-7) here; we return exact zeros instead of a column-sum pass.
            b = ctx.bias_param
            view = getattr(b, "_arco_grad_view", None)
            if view is not None and b.grad is not None and b.grad.data_ptr() == view.data_ptr():
                b._arco_mark()          # += 0 into the flat gradient: nothing to launch, the optimiser still steps it
            else:
                db = _zeros_cached((co,), da.device)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def conv_block3d_nograd(x, stages):
    """A V-Net ConvBlock (vnetWithArgs.py:5-31: n_stages x [Conv3d(3, pad 1) - BatchNorm3d(train) - ReLU]) in a GRADIENT-FREE pass (the
    teacher's forwards, the warped student pass of the equivariance term): only the last stage writes its activation.  Stage i + 1 reads
    stage i's pre-activation z_i through the consumer-side activation of conv3d_fc_kernel's loaders (arco_conv3d_fwd_pro; the arithmetic
    is bn_act_fwd_kernel's: results bit-identical to the staged route, tests/test_block_fuse_gpu.py) - one launch and one write + read of
    the activation less per link.  stages: [(conv, bn), ...]; the caller has checked block3d_fusable."""
    L.require_gpu(x)
    xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
    G = BN_GROUPS
    if G > 1 and nv % G != 0:
        raise RuntimeError(f"arco_amd: bn_groups({G}) needs a batch that is a multiple of {G}, got {nv}")
    m = nv * d3 * h * w
    pro = None
    zr, ldz, k = xr, ld, ci
    for conv, bn in stages:
        co = int(conv.weight.shape[0])
        z, (ssum, ssq, nmb) = conv_raw(zr, ldz, k, pack_weight(conv.weight, 27, 0), co, nv, h, w, 27, bias=conv.bias, stats=True, d3=d3, sp=sp,
                                       stat_groups=G, pro=pro, pro_groups=G)
        mean, istd = _finalize_bn(ssum, ssq, nmb, co, m, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked, G, x.device)
        zr, ldz = rows_view(z)
        k = co
        pro = L.act_pro(mean, istd, bn.weight, bn.bias, 0.0, G, 0, 0.0, 0, None)
        last = (z, mean, istd, bn, co)
    z, mean, istd, bn, co = last
    a = new_act_nd(nv, co, sp, x.device)
    _bn_apply(zr, ldz, m, co, mean, istd, bn.weight, bn.bias, 0.0, 0, 0.0, 0, d3 * h * w, a, co, G)
    block_fuse_stats["fused3d"] = block_fuse_stats.get("fused3d", 0) + len(stages) - 1
    return a


def block3d_fusable(x, stages):
    """Can conv_block3d_nograd run this block?  fp32 activations, the split-bf16 mode, no gradient, at least two stages, and every
    stage -> stage link a shape conv3d_fc_kernel's loaders take (arco_conv_pro_ok)."""
    if not BLOCK_FUSE3D or torch.is_grad_enabled() or len(stages) < 2 or CONV_MMA != 3 or x.dtype != torch.float32 or x.dim() != 5 or not x.is_cuda:
        return False
    xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
    G = BN_GROUPS
    if G > 1 and nv % G != 0:
        return False
    k = int(stages[0][0].weight.shape[0])
    if not _split_ok(27, nv * d3, h, w, ci, k, ld):
        return False
    for conv, _ in stages[1:]:
        co = int(conv.weight.shape[0])
        if not pro_ok(27, nv, d3, h, w, k, co, k, G):
            return False
        k = co
    return True


# A/B switch of conv_block3d_nograd; OFF by default: measured level with the staged route (the loaders' prologue costs what the apply
# pass of these small tensors costs: tools/micro/fl_pro_bench.py, profiles/r06_notes.md section 10; LA step 24.9-25.0 -> 25.1-25.2 ms)
BLOCK_FUSE3D = int(__import__('os').environ.get('ARCO_BLOCK_FUSE3D', '0'))
BLOCK_FUSE = int(__import__('os').environ.get('ARCO_BLOCK_FUSE', '1'))    # A/B switch: 0 = every stage writes its activation (rounds 1-5)
block_fuse_stats = {"fused": 0, "unfused": 0}       # how often conv_block took each route (tests, bench)


def pro_ok(taps, nv, d3, h, w, ci, co, ld, groups):
    """Does a convolution of this shape take its input as a pre-activation + consumer-side activation (arco_conv_pro_ok)?"""
    key = ("pro", taps, nv, d3, h, w, ci, co, ld, groups, CONV_MMA)
    r = _cfg_cache.get(key)
    if r is None:
        r = _cfg_cache[key] = bool(CONV_MMA == 3 and L.query("arco_conv_pro_ok", taps, nv, d3, h, w, ci, co, ld, 3, groups)
                                   and _split_ok(taps, nv * d3, h, w, ci, co, ld))
    return r


def _finalize_bn(ssum, ssq, nmb, co, m, eps, momentum, running_mean, running_var, nbt, G, dev):
    mean = torch.empty(G * co, dtype=torch.float32, device=dev)      # [G][co]
    istd = torch.empty(G * co, dtype=torch.float32, device=dev)
    d0, dbuf = _defer_args(running_mean, running_var, co, G, momentum)
    _work(2 * co * nmb * 4)
    L.call("arco_bn_finalize", L.ptr(ssum), L.ptr(ssq), nmb, co, m, float(eps), float(momentum), L.ptr(mean),
           L.ptr(istd), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), G, d0, L.ptr(dbuf))
    return mean, istd


class ConvBlockFn(torch.autograd.Function):
    """A whole ConvBlock of the U-Net (unetWithArgs.py:31-47: conv3x3 - BN - LeakyReLU - Dropout(p) - conv3x3 - BN - LeakyReLU) as ONE
    autograd node whose first stage never writes its activation: conv1 stores z1 (+ BN partial statistics), the statistics are
    finalised, and conv2 reads z1 through the consumer-side activation of its loader (arco_conv3d_fwd_pro).  Backward: the second
    stage as ConvBnActFn's, with the weight gradient of conv2 reading z1 the same way (arco_conv3d_wgrad_pro); the first stage's
    BatchNorm backward recomputes its activation from z1 as it always did.  Results are bit-identical to two ConvBnActFn stages
    (tests/test_block_fuse_gpu.py); per block and pass one launch and two of the five HBM crossings of z1 / a1 are gone."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, rm1, rv1, nbt1, w2, b2, g2, be2, rm2, rv2, nbt2, slope1, p1, slope2, mom1, eps1, mom2, eps2,
                cat_room=0, pool=False):
        global _LAST_CAT_BUF
        L.require_gpu(x, w1, w2)
        xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
        cm, co = int(w1.shape[0]), int(w2.shape[0])
        m = nv * h * w
        G = BN_GROUPS
        if G > 1 and nv % G != 0:
            raise RuntimeError(f"arco_amd: bn_groups({G}) needs a batch that is a multiple of {G}, got {nv}")
        dev = x.device
        # stage 1: z1 = conv1(x) + b1, statistics
        z1, (s1, q1, nmb1) = conv_raw(xr, ld, ci, pack_weight(w1, 9, 0), cm, nv, h, w, 9, bias=b1, stats=True, stat_groups=G)
        mean1, istd1 = _finalize_bn(s1, q1, nmb1, cm, m, eps1, mom1, rm1, rv1, nbt1, G, dev)
        seed1 = _next_seed() if p1 > 0 else 0
        seed_dev = SEED_DEV if (p1 > 0 and torch.cuda.is_current_stream_capturing()) else None
        z1r, ldz1 = rows_view(z1)
        pro = L.act_pro(mean1, istd1, g1, be1, slope1, G, 1 if p1 > 0 else 0, p1, seed1, seed_dev)
        # stage 2: z2 = conv2(act(z1)) + b2 with the activation applied in the loader
        z2, (s2, q2, nmb2) = conv_raw(z1r, ldz1, cm, pack_weight(w2, 9, 0), co, nv, h, w, 9, bias=b2, stats=True, stat_groups=G, pro=pro)
        mean2, istd2 = _finalize_bn(s2, q2, nmb2, co, m, eps2, mom2, rm2, rv2, nbt2, G, dev)
        if cat_room:
            buf = new_act_nd(nv, co + int(cat_room), sp, dev)
            a, ld_a = buf[:, :co], co + int(cat_room)
            _LAST_CAT_BUF = buf
        else:
            a, ld_a = new_act_nd(nv, co, sp, dev), co
        z2r, ldz2 = rows_view(z2)
        ctx.pool, ctx.groups = bool(pool), G
        ctx.cfg = (float(slope1), float(p1), seed1, float(slope2))
        ctx.seed_dev = seed_dev
        ctx.params = (b1, b2)
        if pool:
            pooled = new_act(nv, co, h // 2, w // 2, dev)
            _work((2 * m + m // 4) * co * 4)
            L.call("arco_bn_act_pool_fwd", L.ptr(z2r), ldz2, nv, h, w, co, L.ptr(mean2), L.ptr(istd2), L.ptr(g2), L.ptr(be2),
                   float(slope2), L.ptr(a), ld_a, L.ptr(pooled), co, G)
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(x, w1, z1, mean1, istd1, g1, be1, w2, z2, mean2, istd2, g2, be2, a)
            return a, pooled
        _bn_apply(z2r, ldz2, m, co, mean2, istd2, g2, be2, slope2, 0, 0.0, 0, h * w, a, ld_a, G)
        ctx.save_for_backward(x, w1, z1, mean1, istd1, g1, be1, w2, z2, mean2, istd2, g2, be2)
        return a

    @staticmethod
    def backward(ctx, da, dpool=None):
        if ctx.pool:
            x, w1, z1, mean1, istd1, g1, be1, w2, z2, mean2, istd2, g2, be2, a = ctx.saved_tensors
        else:
            x, w1, z1, mean1, istd1, g1, be1, w2, z2, mean2, istd2, g2, be2 = ctx.saved_tensors
        slope1, p1, seed1, slope2 = ctx.cfg
        b1, b2 = ctx.params
        G = ctx.groups
        xr, ldx, nv, d3, h, w, ci, sp = _geom_nd(x)
        cm, co = int(w1.shape[0]), int(w2.shape[0])
        dev = x.device
        if ctx.pool and dpool is not None:       # d a = d skip + maxpool2_bwd(d pooled), summed inside the pooling backward
            ar, lda_ = rows_view(a)
            dpr, ldp = rows_view(dpool)
            dsum = new_act(nv, co, h, w, dev)
            _work((nv * h * w * (3 if da is not None else 2) + nv * h * w // 4) * co * 4)
            if da is None:
                L.call("arco_maxpool2_bwd", L.ptr(ar), lda_, nv, h, w, co, L.ptr(dpr), ldp, L.ptr(dsum), co)
            else:
                sr, lds = rows_view(da)
                L.call("arco_maxpool2_bwd_add", L.ptr(ar), lda_, nv, h, w, co, L.ptr(dpr), ldp, L.ptr(sr), lds, L.ptr(dsum), co)
            da = dsum
        # ---- stage 2
        dz2, dg2, dbe2 = _bn_backward(da, z2, mean2, istd2, g2, be2, slope2, 0, 0.0, 0, h * w, None, G)
        dz2r, lddz2 = rows_view(dz2)
        z1r, ldz1 = rows_view(z1)
        dw2 = None
        if ctx.needs_input_grad[8]:
            pro = L.act_pro(mean1, istd1, g1, be1, slope1, G, 1 if p1 > 0 else 0, p1, seed1, ctx.seed_dev)
            dw2 = _wgrad(dz2r, lddz2, co, z1r, ldz1, cm, 9, nv, h, w, w2, keep=(dz2, z1, mean1, istd1), pro=pro)
        da1, _ = conv_raw(dz2r, lddz2, co, pack_weight(w2, 9, 1), cm, nv, h, w, 9, grad=True)
        # ---- stage 1 (its activation is recomputed from z1 inside the BatchNorm backward kernels)
        dz1, dg1, dbe1 = _bn_backward(da1, z1, mean1, istd1, g1, be1, slope1, 1 if p1 > 0 else 0, p1, seed1, h * w, ctx.seed_dev, G)
        dz1r, lddz1 = rows_view(dz1)
        dw1 = dx = None
        if ctx.needs_input_grad[1]:
            dw1 = _wgrad(dz1r, lddz1, cm, xr, ldx, ci, 9, nv, h, w, w1, keep=(dz1, x))
        if ctx.needs_input_grad[0]:
            dx, _ = conv_raw(dz1r, lddz1, cm, pack_weight(w1, 9, 1), ci, nv, h, w, 9, grad=True)
        # conv biases under train-mode BN: analytically zero gradients (see ConvBnActFn.backward)
        dbs = []
        for b, cn, need in ((b1, cm, ctx.needs_input_grad[2]), (b2, co, ctx.needs_input_grad[9])):
            db = None
            if b is not None and need:
                view = getattr(b, "_arco_grad_view", None)
                if view is not None and b.grad is not None and b.grad.data_ptr() == view.data_ptr():
                    b._arco_mark()
                else:
                    db = _zeros_cached((cn,), dev)
            dbs.append(db)
        return (dx, dw1, dbs[0], dg1, dbe1, None, None, None, dw2, dbs[1], dg2, dbe2, None, None, None) + (None,) * 9


def conv_block(x, conv1, bn1, act1, p1, conv2, bn2, act2, cat_room=0, pool=False):
    """The train-mode ConvBlock: ConvBlockFn when both convolutions run on the pipelined kernels (ops.pro_ok), else None (the
    caller runs the two stages separately)."""
    global _LAST_CAT_BUF
    if not BLOCK_FUSE or x.dim() != 4 or x.dtype != torch.float32:
        return None
    xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
    cm, co = int(conv1.weight.shape[0]), int(conv2.weight.shape[0])
    if _taps(conv1.weight) != 9 or _taps(conv2.weight) != 9 or not pro_ok(9, nv, 1, h, w, cm, co, cm, BN_GROUPS):
        block_fuse_stats["unfused"] += 1
        return None
    block_fuse_stats["fused"] += 1
    y = ConvBlockFn.apply(x, conv1.weight, conv1.bias, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                          conv2.weight, conv2.bias, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                          getattr(act1, "negative_slope", 0.0), float(p1), getattr(act2, "negative_slope", 0.0),
                          bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, cat_room, pool)
    if cat_room:
        (y[0] if pool else y)._arco_cat_buf, _LAST_CAT_BUF = _LAST_CAT_BUF, None
    return y


class BnActFn(torch.autograd.Function):
    """a = lrelu(BN_train(z)) for a tensor that did not come out of the conv kernel (the depth-to-space'd
    transposed conv of UpsamplingDeconvBlock, vnetWithArgs.py:94-118).  gamma=None: plain dropout/activation."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, slope, p, drop_mode, momentum, eps, nbt=None, residual=None):
        zr, ldz = rows_view(z)
        co = int(z.shape[1])
        m = zr.shape[0]
        sp = tuple(int(v) for v in z.shape[2:])
        P = 1
        for v in sp:
            P *= v
        mean = istd = None
        G = 1
        if gamma is not None:
            G = BN_GROUPS
            if G > 1 and int(z.shape[0]) % G != 0:
                raise RuntimeError(f"arco_amd: bn_groups({G}) needs a batch that is a multiple of {G}")
            nblk = L.query("arco_chan_stats_blocks", m // G)
            ssum = torch.empty((co, G * nblk), dtype=torch.float32, device=z.device)
            ssq = torch.empty((co, G * nblk), dtype=torch.float32, device=z.device)
            L.call("arco_chan_stats_h" if _is_half(zr) else "arco_chan_stats", L.ptr(zr), ldz, m, co, L.ptr(ssum), L.ptr(ssq), G)
            mean = torch.empty(G * co, dtype=torch.float32, device=z.device)
            istd = torch.empty(G * co, dtype=torch.float32, device=z.device)
            d0, dbuf = _defer_args(running_mean, running_var, co, G, momentum)
            L.call("arco_bn_finalize", L.ptr(ssum), L.ptr(ssq), G * nblk, co, m, float(eps), float(momentum), L.ptr(mean),
                   L.ptr(istd), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), G, d0, L.ptr(dbuf))
        seed = _next_seed() if p > 0 else 0
        a = new_act_nd(int(z.shape[0]), co, sp, z.device, zr.dtype)
        ctx.has_res = residual is not None
        if residual is not None:        # a = lrelu(BN(z)) + residual in the apply pass (the V-Net decoder's skip additions)
            if gamma is None or p > 0 or residual.dtype != z.dtype or tuple(residual.shape) != tuple(z.shape):
                raise RuntimeError("arco_amd: bn_act(residual=...) is the dropout-free BatchNorm apply with a same-shape, same-dtype addend")
            rr, ldr = rows_view(residual)
            L.call("arco_bn_act_add_fwd_h" if _is_half(zr) else "arco_bn_act_add_fwd", L.ptr(zr), ldz, m, co, L.ptr(mean), L.ptr(istd),
                   L.ptr(gamma), L.ptr(beta), float(slope), L.ptr(rr), ldr, L.ptr(a), co, G)
            ctx.seed_dev = None
        else:
            ctx.seed_dev = _bn_apply(zr, ldz, m, co, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, a, None, G)
        ctx.groups = G
        ctx.save_for_backward(z, mean, istd, gamma, beta)
        ctx.cfg = (float(slope), float(p), int(drop_mode), seed, P)
        return a

    @staticmethod
    def backward(ctx, da):
        z, mean, istd, gamma, beta = ctx.saved_tensors
        slope, p, drop_mode, seed, P = ctx.cfg
        dz, dgamma, dbeta = _bn_backward(da, z, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, ctx.seed_dev,
                                         ctx.groups)
        # (the activation is recomputed from z in the backward kernels, so the added residual never enters them; its gradient is da)
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, (da if ctx.has_res else None)


class BnActD2sFn(torch.autograd.Function):
    """a = lrelu(BN_train(depth_to_space(y))) (+ residual) WITHOUT the depth-to-space pass: y = [N, 8C, X, Y, Z] is the GEMM form of
    the k2 s2 transposed conv of UpsamplingDeconvBlock (vnetWithArgs.py:94-118, tap-major channels); read as (voxel, tap) rows of C
    channels it is the pre-activation in another row order.  Statistics over those rows; the apply pass writes each row to its
    voxel of a = [N, C, 2X, 2Y, 2Z]; the backward reads da from there and returns dy in y's layout (csrc/elementwise.hip, D2S)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, slope, momentum, eps, nbt, residual):
        yr, ldy = rows_view(y)
        n, c8 = int(y.shape[0]), int(y.shape[1])
        x2, y2, z2 = (int(v) for v in y.shape[2:])
        c = c8 // 8
        if ldy != c8 or c8 % 8 or c % 4:
            raise RuntimeError("arco_amd: bn_act_d2s needs a dense [N, 8C, X, Y, Z] channels-last tensor with C % 4 == 0")
        half = _is_half(yr)
        m8 = yr.shape[0] * 8
        G = BN_GROUPS
        if G > 1 and n % G != 0:
            raise RuntimeError(f"arco_amd: bn_groups({G}) needs a batch that is a multiple of {G}")
        nblk = L.query("arco_chan_stats_blocks", m8 // G)
        ssum = torch.empty((c, G * nblk), dtype=torch.float32, device=y.device)
        ssq = torch.empty((c, G * nblk), dtype=torch.float32, device=y.device)
        L.call("arco_chan_stats_h" if half else "arco_chan_stats", L.ptr(yr), c, m8, c, L.ptr(ssum), L.ptr(ssq), G)
        mean = torch.empty(G * c, dtype=torch.float32, device=y.device)
        istd = torch.empty(G * c, dtype=torch.float32, device=y.device)
        d0, dbuf = _defer_args(running_mean, running_var, c, G, momentum)
        L.call("arco_bn_finalize", L.ptr(ssum), L.ptr(ssq), G * nblk, c, m8, float(eps), float(momentum), L.ptr(mean),
               L.ptr(istd), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), G, d0, L.ptr(dbuf))
        a = new_act_nd(n, c, (2 * x2, 2 * y2, 2 * z2), y.device, yr.dtype)
        rr, ldr = (None, 0)
        if residual is not None:
            if residual.dtype != y.dtype or tuple(residual.shape) != tuple(a.shape):
                raise RuntimeError("arco_amd: bn_act_d2s(residual=...) needs an addend of the activation's shape and dtype")
            rr, ldr = rows_view(residual)
        L.call("arco_bn_act_d2s_fwd_h" if half else "arco_bn_act_d2s_fwd", L.ptr(yr), m8, c, L.ptr(mean), L.ptr(istd), L.ptr(gamma),
               L.ptr(beta), float(slope), L.ptr(rr), ldr, L.ptr(a), c, x2, y2, z2, G)
        ctx.save_for_backward(y, mean, istd, gamma, beta)
        ctx.cfg = (float(slope), G, nblk, residual is not None)
        return a

    @staticmethod
    def backward(ctx, da):
        y, mean, istd, gamma, beta = ctx.saved_tensors
        slope, G, nblk, has_res = ctx.cfg
        yr, _ = rows_view(y)
        n, c8 = int(y.shape[0]), int(y.shape[1])
        x2, y2, z2 = (int(v) for v in y.shape[2:])
        c = c8 // 8
        half = _is_half(yr)
        dar, ldd = rows_view(da)
        if _is_half(dar) != half:
            dar = dar.to(yr.dtype)
            ldd = dar.stride(0)
        dy = new_act_nd(n, c8, (x2, y2, z2), y.device, yr.dtype)
        ws = torch.empty(G * (2 * c * nblk + 2 * c), dtype=torch.float32, device=y.device)
        dgamma = dbeta = None
        acc = 0
        gv, bv = getattr(gamma, "_arco_grad_view", None), getattr(beta, "_arco_grad_view", None)
        if (gv is not None and bv is not None and gamma.grad is not None and beta.grad is not None
                and gamma.grad.data_ptr() == gv.data_ptr() and beta.grad.data_ptr() == bv.data_ptr()):
            dg_t, db_t, acc = gv, bv, 1               # straight into the optimiser's flat gradient buffer
            gamma._arco_mark(); beta._arco_mark()
        else:
            dg_t = dgamma = torch.empty_like(gamma)
            db_t = dbeta = torch.empty_like(beta)
        L.call("arco_bn_act_d2s_bwd_h" if half else "arco_bn_act_d2s_bwd", L.ptr(dar), ldd, L.ptr(yr), yr.shape[0] * 8, c, L.ptr(mean),
               L.ptr(istd), L.ptr(gamma), L.ptr(beta), slope, L.ptr(ws), L.ptr(dg_t), L.ptr(db_t), acc, L.ptr(dy), x2, y2, z2, G)
        return dy, dgamma, dbeta, None, None, None, None, None, None, (da if has_res else None)


def bn_act_d2s(y, gamma, beta, running_mean, running_var, slope=0.0, momentum=0.1, eps=1e-5, num_batches_tracked=None, residual=None):
    return BnActD2sFn.apply(y, gamma, beta, running_mean, running_var, slope, momentum, eps, num_batches_tracked, residual)


D2S_FUSE = int(__import__('os').environ.get('ARCO_D2S_FUSE', '1'))      # A/B switch: 0 = separate depth-to-space pass in UpsamplingDeconvBlock


class S2D3Fn(torch.autograd.Function):
    """[N,C,2X,2Y,2Z] -> [N,8C,X,Y,Z] (tap-major channels) and back (inverse=True): pure data movement that
    turns the k=2,s=2 (transposed) Conv3d of the V-Net into a 1x1x1 GEMM."""

    @staticmethod
    def forward(ctx, x, inverse):
        ctx.inverse = inverse
        return _s2d3(x, inverse)

    @staticmethod
    def backward(ctx, dy):
        return _s2d3(dy, not ctx.inverse), None


class S2dSkipFn(torch.autograd.Function):
    """(space_to_depth(x), x-as-skip) for an encoder activation that feeds BOTH the next DownsamplingConvBlock and the decoder's skip
    connection (vnetWithArgs.py:186-201,224-236): the backward forms depth_to_space(d xs) + d skip in one pass (arco_d2s3_add) instead
    of the inverse permutation followed by a full-tensor add (the 3-D sibling of MaxPoolSkipFn)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return _s2d3(x, False), x.view_as(x)

    @staticmethod
    def backward(ctx, dxs, dskip):
        if dxs is None:
            return dskip
        if dskip is None or dskip.dtype != dxs.dtype:
            d = _s2d3(dxs, True)
            return d if dskip is None else d + dskip
        n, c = ctx.shape[0], ctx.shape[1]
        x2, y2, z2 = (int(v) for v in dxs.shape[2:])
        pr, ldp = rows_view(dxs)
        ar, lda = rows_view(dskip)
        out = new_act_nd(n, c, ctx.shape[2:], dxs.device, pr.dtype)
        L.call("arco_d2s3_add_h" if _is_half(pr) else "arco_d2s3_add", L.ptr(pr), ldp, n, x2, y2, z2, c, L.ptr(ar), lda, L.ptr(out), c)
        return out


S2D_SKIP = int(__import__('os').environ.get('ARCO_S2D_SKIP', '1'))      # A/B switch: 0 = inverse permutation + tensor-library add


def s2d_skip(x):
    """(space_to_depth3(x), alias of x to use as the skip connection) - see S2dSkipFn."""
    if not S2D_SKIP:
        return S2D3Fn.apply(x, False), x
    return S2dSkipFn.apply(x)


def _s2d3(x, inverse):
    n, c = int(x.shape[0]), int(x.shape[1])
    sp = [int(v) for v in x.shape[2:]]
    xr, ld = rows_view(x)
    e = 2 if _is_half(xr) else 1          # the kernel permutes 4-byte words: an f16 row of C channels is a row of C / 2 words
    if not inverse:
        x2, y2, z2 = sp[0] // 2, sp[1] // 2, sp[2] // 2
        out = new_act_nd(n, 8 * c, (x2, y2, z2), x.device, xr.dtype)
        L.call("arco_s2d3", L.ptr(xr), ld // e, n, x2, y2, z2, c // e, L.ptr(out), 8 * c // e, 0)
    else:
        cv = c // 8
        out = new_act_nd(n, cv, (2 * sp[0], 2 * sp[1], 2 * sp[2]), x.device, xr.dtype)
        L.call("arco_s2d3", L.ptr(out), cv // e, n, sp[0], sp[1], sp[2], cv // e, L.ptr(xr), ld // e, 1)
    return out


class TrilinearFn(torch.autograd.Function):
    """nn.Upsample(size, mode='trilinear', align_corners=True) (model_3D.py:46-58)."""

    @staticmethod
    def forward(ctx, x, do, ho, wo):
        xr, ld = rows_view(x)
        n, c, di, hi, wi = (int(v) for v in x.shape)
        y = new_act_nd(n, c, (do, ho, wo), x.device)
        L.call("arco_trilinear_fwd", L.ptr(xr), ld, n, di, hi, wi, c, do, ho, wo, L.ptr(y), c)
        ctx.dims = (n, c, di, hi, wi, do, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, di, hi, wi, do, ho, wo = ctx.dims
        dyr, ldy = rows_view(dy)
        dx = new_act_nd(n, c, (di, hi, wi), dy.device)
        L.call("arco_trilinear_bwd", L.ptr(dyr), ldy, n, di, hi, wi, c, do, ho, wo, L.ptr(dx), c)
        return dx, None, None, None


class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2) (unetWithArgs.py:55-58)."""

    @staticmethod
    def forward(ctx, x):
        xr, ld, nb, c, h, w = _geom(x)
        y = new_act(nb, c, h // 2, w // 2, x.device)
        L.call("arco_maxpool2_fwd", L.ptr(xr), ld, nb, h, w, c, L.ptr(y), c)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        xr, ld, nb, c, h, w = _geom(x)
        dyr, ldy = rows_view(dy)
        dx = new_act(nb, c, h, w, x.device)
        L.call("arco_maxpool2_bwd", L.ptr(xr), ld, nb, h, w, c, L.ptr(dyr), ldy, L.ptr(dx), c)
        return dx


class MaxPoolSkipFn(torch.autograd.Function):
    """(maxpool2(x), x) for an activation that feeds BOTH the next DownBlock and the decoder's skip concat
    (unetWithArgs.py:109-116,142-158): the backward adds the two incoming gradients inside the max-pool backward kernel
    instead of a separate full-tensor add (4 levels x every student pass)."""

    @staticmethod
    def forward(ctx, x):
        xr, ld, nb, c, h, w = _geom(x)
        y = new_act(nb, c, h // 2, w // 2, x.device)
        L.call("arco_maxpool2_fwd", L.ptr(xr), ld, nb, h, w, c, L.ptr(y), c)
        ctx.save_for_backward(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        (x,) = ctx.saved_tensors
        xr, ld, nb, c, h, w = _geom(x)
        if dy is None:
            return dskip
        dyr, ldy = rows_view(dy)
        dx = new_act(nb, c, h, w, x.device)
        if dskip is None:
            L.call("arco_maxpool2_bwd", L.ptr(xr), ld, nb, h, w, c, L.ptr(dyr), ldy, L.ptr(dx), c)
        else:
            sr, lds = rows_view(dskip)
            L.call("arco_maxpool2_bwd_add", L.ptr(xr), ld, nb, h, w, c, L.ptr(dyr), ldy, L.ptr(sr), lds, L.ptr(dx), c)
        return dx


class BilinearFn(torch.autograd.Function):
    """nn.Upsample(size, mode='bilinear', align_corners=True) (model_2D.py:43-52, unetWithArgs.py:74-75)."""

    @staticmethod
    def forward(ctx, x, ho, wo):
        xr, ld, nb, c, h, w = _geom(x)
        y = new_act(nb, c, ho, wo, x.device)
        _work(nb * c * (h * w + ho * wo) * 4)
        L.call("arco_bilinear_fwd", L.ptr(xr), ld, nb, h, w, c, ho, wo, L.ptr(y), c)
        ctx.dims = (nb, c, h, w, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        nb, c, h, w, ho, wo = ctx.dims
        dyr, ldy = rows_view(dy)
        dx = new_act(nb, c, h, w, dy.device)
        _work(nb * c * (h * w + ho * wo) * 4)
        L.call("arco_bilinear_bwd", L.ptr(dyr), ldy, nb, h, w, c, ho, wo, L.ptr(dx), c, 0)
        return dx, None, None


class _GemmWeightFn(torch.autograd.Function):
    """W2 = the GEMM form of a k2 s2 (transposed) conv weight, served from the PackPlan's persistent buffer (refreshed with the
    plan: no permute / copy / pack launches per forward); the backward un-permutes dW2 into the torch layout (views)."""

    @staticmethod
    def forward(ctx, w, w2buf, up):
        ctx.up, ctx.wshape = bool(up), tuple(w.shape)
        return w2buf.detach()

    @staticmethod
    def backward(ctx, dw2):
        a, b = ctx.wshape[0], ctx.wshape[1]
        if ctx.up:          # W2 [(t, co)][ci] <- W [ci][co][t]
            dw = dw2.reshape(2, 2, 2, b, a).permute(4, 3, 0, 1, 2)
        else:               # W2 [co][(t, ci)] <- W [co][ci][t]
            dw = dw2.reshape(a, 2, 2, 2, b).permute(0, 4, 1, 2, 3)
        return dw, None, None


class _Bias8Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, b8buf):
        return b8buf.detach()

    @staticmethod
    def backward(ctx, db8):
        return db8.view(8, -1).sum(0), None


def gemm_weight(conv):
    """(W2, bias') of a k2 s2 Conv3d / ConvTranspose3d in GEMM form: W2 [co][8 ci] resp. [8 co][ci] (tap-major), bias' = the
    bias (Conv3d) resp. the bias repeated over the 8 taps (ConvTranspose3d).  From the module's PackPlan when it is valid,
    otherwise built here (permute + reshape, as round 3 did at every forward)."""
    w, b = conv.weight, conv.bias
    up = isinstance(conv, torch.nn.ConvTranspose3d)
    ent = getattr(w, "_arco_gemm", None)
    if ent is not None and ent[0].valid:
        _, w2buf, b8buf = ent
        grad = torch.is_grad_enabled()
        w2 = _GemmWeightFn.apply(w, w2buf, up) if (grad and w.requires_grad) else w2buf
        if up and b is not None:
            b = _Bias8Fn.apply(b, b8buf) if (grad and b.requires_grad) else b8buf
        return w2, b
    if up:
        ci, co = w.shape[0], w.shape[1]
        return w.permute(2, 3, 4, 1, 0).reshape(8 * co, ci, 1, 1, 1), (b.repeat(8) if b is not None else None)
    co, ci = w.shape[0], w.shape[1]
    return w.permute(0, 2, 3, 4, 1).reshape(co, 8 * ci, 1, 1, 1), b


class ConvUpResFn(torch.autograd.Function):
    """y = conv1x1x1(x; W) + trilinear_align_corners(lo -> x's size): FeatureExtractor_3d's `fea_i(cat(up(x), f_i)) + cat(up(x), f_i)`
    with the wide block of the weights evaluated below the upsample (model_3D.py:46-58; arco_amd/model_3D.py forward_lowres2).
    Forward: ONE launch - the upsampled low-resolution product is sampled in the GEMM's epilogue (arco_conv1x1_upres_fwd) instead of
    being written by a resize kernel and read back as the residual (2 x 450 MB at the LA size); where the pipelined kernel does
    not take the shape the two-launch route runs - the results are bit-identical.  Backward: the conv's data and weight gradients,
    and the resize adjoint of the SAME incoming gradient for `lo`."""

    @staticmethod
    def forward(ctx, x, weight, lo):
        L.require_gpu(x, weight, lo)
        xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
        co = int(weight.shape[0])
        lor, ldlo = rows_view(lo)
        ud, uh, uw = (int(v) for v in lo.shape[2:])
        od, oh, ow = (int(v) for v in x.shape[2:])
        wp = pack_weight(weight, 1, 0)
        spw = getattr(wp, "_arco_split", None)
        y = None
        if UPRES_FUSE and CONV_MMA == 3 and spw is not None and x.dtype == torch.float32 and lo.dtype == torch.float32 and int(lo.shape[1]) == co:
            y = new_act_nd(nv, co, (od, oh, ow), x.device)
            rc = getattr(L.load(), "arco_conv1x1_upres_fwd")(L.ptr(xr), ld, ci, L.ptr(spw), co, L.ptr(y), co, L.ptr(lor), ldlo, nv, ud, uh, uw,
                                                             od, oh, ow, L.stream())
            if rc == -3:
                y = None                                   # shape not taken by the pipelined kernel
            elif rc != 0:
                raise RuntimeError(f"arco_amd: arco_conv1x1_upres_fwd failed with code {rc}")
        if y is None:
            up = new_act_nd(nv, co, (od, oh, ow), x.device)
            L.call("arco_trilinear_fwd", L.ptr(lor), ldlo, nv, ud, uh, uw, co, od, oh, ow, L.ptr(up), co)
            upr, ldu = rows_view(up)
            y, _ = conv_raw(xr, ld, ci, wp, co, nv, h, w, 1, residual=upr, ld_res=ldu, d3=d3, sp=sp)
        ctx.save_for_backward(x, weight)
        ctx.dims = (nv, co, ud, uh, uw, od, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        nv, co, ud, uh, uw, od, oh, ow = ctx.dims
        dyr, ldy, _, d3, h, w, _, sp = _geom_nd(dy)
        xr, ldx = rows_view(x)
        ci = int(x.shape[1])
        dx = dw = dlo = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dyr, ldy, co, xr, ldx, ci, 1, nv, h, w, weight, d3=d3, keep=(dy, x))
        if ctx.needs_input_grad[0]:
            wd = pack_weight(weight, 1, 1)
            dx, _ = conv_raw(dyr, ldy, co, wd, ci, nv, h, w, 1, grad=True, d3=d3, sp=sp)
        if ctx.needs_input_grad[2]:
            dlo = new_act_nd(nv, co, (ud, uh, uw), dy.device)
            L.call("arco_trilinear_bwd", L.ptr(dyr), ldy, nv, ud, uh, uw, co, od, oh, ow, L.ptr(dlo), co)
        return dx, dw, dlo


UPRES_FUSE = True      # A/B switch of ConvUpResFn's one-launch route


def conv_upres(x, weight, lo):
    """conv1x1x1(x; weight) + trilinear(lo -> x's size)  (see ConvUpResFn)."""
    if tuple(int(v) for v in lo.shape[2:]) == tuple(int(v) for v in x.shape[2:]) or x.dim() != 5 or _is_half(x):
        return conv(x, weight, None, residual=trilinear(lo, x.shape[-3:]))
    return ConvUpResFn.apply(x, weight, lo)


def conv(x, weight, bias=None, residual=False, bias_grad_zero=False):
    """bias_grad_zero=True: the caller normalises the result with a train-mode BatchNorm, so the bias gradient is
    analytically zero and is returned as exact zeros instead of a reduction over the output gradient."""
    return ConvFn.apply(x, weight, bias, residual, bias_grad_zero)


_LAST_CAT_BUF = None


def conv_bn_act(x, weight, bias, gamma, beta, running_mean, running_var, slope=0.01, p=0.0, drop_mode=1,
                momentum=0.1, eps=1e-5, num_batches_tracked=None, cat_room=0, pool=False):
    """`num_batches_tracked` (int64 buffer) is incremented inside the BN finalize kernel.
    cat_room > 0: the result is written as the leading channels of a buffer with `cat_room` more channels
    (`result._arco_cat_buf`), so that `upcat` can append an upsampled tensor behind it without a copy.
    pool=True: returns (result, maxpool2(result)) from one apply pass (2-D, p == 0)."""
    global _LAST_CAT_BUF
    y = ConvBnActFn.apply(x, weight, bias, gamma, beta, running_mean, running_var, slope, p, drop_mode,
                          momentum, eps, num_batches_tracked, cat_room, pool)
    if cat_room:
        (y[0] if pool else y)._arco_cat_buf, _LAST_CAT_BUF = _LAST_CAT_BUF, None
    return y


class UpCatFn(torch.autograd.Function):
    """cat([skip, bilinear_x2(x)], dim=1) (unetWithArgs.py:80-83) without the concat copy: `skip` already is the
    leading channel block of `buf` (conv_bn_act(cat_room=...)); the upsample writes the trailing block in place."""

    @staticmethod
    def forward(ctx, x, skip, buf):
        xr, ld, nb, c, h, w = _geom(x)
        c2, ho, wo = int(skip.shape[1]), int(skip.shape[2]), int(skip.shape[3])
        ctot = int(buf.shape[1])
        _work(nb * c * (h * w + ho * wo) * 4)
        L.call("arco_bilinear_fwd", L.ptr(xr), ld, nb, h, w, c, ho, wo, L.ptr(buf[:, c2:]), ctot)
        ctx.dims = (nb, c, h, w, ho, wo, c2)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        nb, c, h, w, ho, wo, c2 = ctx.dims
        dr, ldd = rows_view(dbuf)
        dx = new_act(nb, c, h, w, dbuf.device)
        _work(nb * c * (h * w + ho * wo) * 4)
        L.call("arco_bilinear_bwd", L.ptr(dr[:, c2:]), ldd, nb, h, w, c, ho, wo, L.ptr(dx), c, 0)
        return dx, dbuf[:, :c2], None


def upcat(x, skip):
    """cat([skip, bilinear(x, skip.shape[-2:])], 1); in place when `skip` came with concat room."""
    buf = getattr(skip, "_arco_cat_buf", None)
    if (buf is None or buf.shape[1] != skip.shape[1] + x.shape[1] or buf.data_ptr() != skip.data_ptr()
            or buf.shape[0] != x.shape[0]):
        return torch.cat([skip, bilinear(x, skip.shape[-2:])], dim=1)
    return UpCatFn.apply(x, skip, buf)


def conv_bn_act_eval(x, weight, bias, gamma, beta, running_mean, running_var, slope=0.01, eps=1e-5):
    """Inference-mode stage (BN uses running statistics); no autograd."""
    with torch.no_grad():
        taps = _taps(weight)
        xr, ld, nv, d3, h, w, ci, sp = _geom_nd(x)
        co = int(weight.shape[0])
        half = use_half(x, taps, ci)
        z, _ = conv_raw(xr, ld, ci, pack_weight(weight, taps, 0, half=half and ci != 1), co, nv, h, w, taps, bias=bias, d3=d3, sp=sp,
                        half=half)
        return bn_act_eval(z, gamma, beta, running_mean, running_var, slope, eps)


def bn_act_eval(z, gamma, beta, running_mean, running_var, slope=0.0, eps=1e-5):
    """lrelu(BN_eval(z)) with the running statistics (inference; no autograd); follows z's storage type."""
    with torch.no_grad():
        zr, ldz = rows_view(z)
        co = int(z.shape[1])
        sp = tuple(int(v) for v in z.shape[2:])
        P = 1
        for v in sp:
            P *= v
        istd = torch.rsqrt(running_var + eps)
        a = new_act_nd(int(z.shape[0]), co, sp, z.device, zr.dtype)
        L.call("arco_bn_act_fwd_h" if _is_half(zr) else "arco_bn_act_fwd", L.ptr(zr), ldz, zr.shape[0], co, L.ptr(running_mean),
               L.ptr(istd), L.ptr(gamma), L.ptr(beta), float(slope), 0, 0.0, 0, P, L.ptr(a), co, None, 1)
        return a


def bn_act(z, gamma, beta, running_mean, running_var, slope=0.0, p=0.0, drop_mode=0, momentum=0.1, eps=1e-5,
           num_batches_tracked=None, residual=None):
    """residual: a tensor of z's shape added AFTER the activation in the same pass (vnetWithArgs.py:224-236, the decoder's skips)."""
    return BnActFn.apply(z, gamma, beta, running_mean, running_var, slope, p, drop_mode, momentum, eps,
                         num_batches_tracked, residual)


def dropout3d(x, p):
    """nn.Dropout3d(p) in training mode: whole (sample, channel) volumes are dropped (vnetWithArgs.py:195,238)."""
    return BnActFn.apply(x, None, None, None, None, 1.0, p, 2, 0.1, 1e-5)


class _FromHalfFn(torch.autograd.Function):
    """The boundary of the f16 region: an f16 activation leaves it as fp32; the fp32 gradient enters it multiplied by
    LOSS_SCALE (gradients of 1e-7 would flush in f16) - every parameter gradient produced inside the region carries the
    factor, and the trainer divides it out of the flat gradient buffer before the optimiser step."""

    @staticmethod
    def forward(ctx, x):
        xr, ld = rows_view(x)
        c = int(x.shape[1])
        assert ld == c, "from_half: dense channels-last activation expected"
        y = new_act_nd(int(x.shape[0]), c, tuple(int(v) for v in x.shape[2:]), x.device)
        L.call("arco_cast_h2f", L.ptr(xr), xr.numel(), L.ptr(y))
        return y

    @staticmethod
    def backward(ctx, dy):
        dyr, ld = rows_view(dy)
        c = int(dy.shape[1])
        if ld != c:
            dyr = dyr.contiguous()
        dx = new_act_nd(int(dy.shape[0]), c, tuple(int(v) for v in dy.shape[2:]), dy.device, torch.float16)
        L.call("arco_cast_f2h", L.ptr(dyr), dyr.numel(), float(LOSS_SCALE), L.ptr(dx))
        return dx


def from_half(x):
    """fp32 view of an activation for the consumers outside the f16 region (identity for fp32 tensors)."""
    return _FromHalfFn.apply(x) if _is_half(x) else x


FM_CAST = True        # True | False (logits_only) | 'lowres' (fm_rows_half)


class logits_only:
    """`with ops.logits_only():` - a V-Net forward inside hands its feature maps out as they are (f16 in f16 mode) instead of
    casting them to fp32: for the passes of the step that only read the logits (the pseudo-label pass, the warped pass)."""

    def __enter__(self):
        global FM_CAST
        self.prev, FM_CAST = FM_CAST, False

    def __exit__(self, *exc):
        global FM_CAST
        FM_CAST = self.prev


class fm_rows_half(logits_only):
    """`with ops.fm_rows_half():` - a V-Net forward inside (f16 mode) casts its three LOW-resolution feature maps to fp32 and hands
    the two full-resolution ones out as stored (f16): the row-sparse heads (arco_amd.head.lazy_head3d, LazyTeacher3D) read rows /
    weighted row sums of those two straight from the f16 maps and return a row-sparse f16 gradient - no dense cast of a
    full-resolution map in either direction (they were 0.64 ms of the LiTS-shaped step)."""

    def __enter__(self):
        global FM_CAST
        self.prev, FM_CAST = FM_CAST, 'lowres'


def space_to_depth3(x):
    return S2D3Fn.apply(x, False)


def depth_to_space3(x):
    return S2D3Fn.apply(x, True)


def trilinear(x, size):
    if tuple(int(v) for v in x.shape[2:]) == tuple(int(v) for v in size):
        return x                                   # align_corners resize to the same size is the identity
    return TrilinearFn.apply(x, int(size[0]), int(size[1]), int(size[2]))


def maxpool2(x):
    return MaxPool2Fn.apply(x)


def maxpool2_skip(x):
    """(maxpool2(x), x-as-skip): use the returned alias as the skip connection so that both gradients of x meet inside one
    kernel (MaxPoolSkipFn).  The concat room of x (conv_bn_act(cat_room=...)) travels with the alias."""
    y, skip = MaxPoolSkipFn.apply(x)
    buf = getattr(x, "_arco_cat_buf", None)
    if buf is not None:
        skip._arco_cat_buf = buf
    return y, skip


def bilinear(x, size):
    return BilinearFn.apply(x, int(size[0]), int(size[1]))


class BoundaryTensor(torch.Tensor):
    """A channels-last activation as the drop-in boundary hands it to REFERENCE code.  The reference flattens such outputs with
    `rep_u.view(rep_u.shape[0], -1)` (train_arco_2d.py:127,129,400; train_arco_3d.py:124,126,363) - legal on the NCHW-contiguous
    result of torch's own convolution, a RuntimeError on channels-last strides.  On this subclass `.view` falls back to `.reshape`
    when the strides do not allow a view: the statement then yields exactly the (c, h, w) flattening it yields in the reference,
    at the price of the copy `.reshape` makes.  Everything else is torch.Tensor's own behaviour (the subclass survives slicing,
    `.detach()`, arithmetic; autograd is unaffected)."""

    def view(self, *shape, **kw):
        try:
            return super().view(*shape, **kw)
        except RuntimeError:
            if kw or (len(shape) == 1 and isinstance(shape[0], torch.dtype)):
                raise
            return self.reshape(*shape)

    # The subclass survives only the operations that hand the SAME channels-last map on (batch slicing / indexing, detach, narrow):
    # everything else returns plain tensors, so the Python-level dispatch below is paid by the first operation on a boundary
    # output and not by the whole graph behind it (a reference-style user's step runs hundreds of torch ops downstream)
    _KEEP = frozenset(("__getitem__", "detach", "narrow", "requires_grad_", "contiguous"))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **(kwargs or {}))
        if getattr(func, "__name__", "") in cls._KEEP and type(out) is torch.Tensor and out.dim() >= 3:
            return out.as_subclass(cls)
        return out


def boundary(t):
    """Mark a channels-last output that leaves the package through a reference-named module (see BoundaryTensor)."""
    return t.as_subclass(BoundaryTensor) if type(t) is torch.Tensor else t


def to_channels_last(x):
    """API-boundary layout conversion (memory plumbing): NCHW-contiguous -> channels-last."""
    if x.dim() == 4:
        return x.contiguous(memory_format=torch.channels_last)
    return x.contiguous(memory_format=torch.channels_last_3d)


class GnActFn(torch.autograd.Function):
    """a = lrelu(GroupNorm(z)) / lrelu(InstanceNorm(z)): statistics per sample over sets of `cpg` consecutive channels
    (nn.GroupNorm(16, C): cpg = C // 16; nn.InstanceNorm3d(C): cpg = 1, no affine) - the `normalization='groupnorm' |
    'instancenorm'` variants of the V-Net blocks (vnetWithArgs.py:19-22,48-51,76-79).  Per-(sample, channel) slab sums with
    the BatchNorm machinery (groups = samples), a set-wise finalize, the BatchNorm apply pass unchanged; backward: the
    BatchNorm reduce + finalize, the set-wise merge of the gamma-weighted sums, the apply pass in its GroupNorm form."""

    @staticmethod
    def forward(ctx, z, gamma, beta, cpg, slope, eps):
        L.require_gpu(z)
        zr, ldz = rows_view(z)
        nb, co = int(z.shape[0]), int(z.shape[1])
        m = zr.shape[0]
        sp = tuple(int(v) for v in z.shape[2:])
        if gamma is None:                               # InstanceNorm3d default: affine=False
            gamma = torch.ones(co, dtype=torch.float32, device=z.device)
            beta = torch.zeros(co, dtype=torch.float32, device=z.device)
            ctx.affine = False
        else:
            ctx.affine = True
        nblk = L.query("arco_chan_stats_blocks", m // nb)
        ssum = torch.empty((co, nb * nblk), dtype=torch.float32, device=z.device)
        ssq = torch.empty((co, nb * nblk), dtype=torch.float32, device=z.device)
        L.call("arco_chan_stats", L.ptr(zr), ldz, m, co, L.ptr(ssum), L.ptr(ssq), nb)
        mean = torch.empty(nb * co, dtype=torch.float32, device=z.device)      # [N][co]
        istd = torch.empty(nb * co, dtype=torch.float32, device=z.device)
        L.call("arco_gn_finalize", L.ptr(ssum), L.ptr(ssq), nb * nblk, co, int(cpg), nb, m // nb, float(eps), L.ptr(mean), L.ptr(istd))
        a = new_act_nd(nb, co, sp, z.device)
        _bn_apply(zr, ldz, m, co, mean, istd, gamma, beta, slope, 0, 0.0, 0, 1, a, None, nb)
        ctx.save_for_backward(z, mean, istd, gamma, beta)
        ctx.cfg = (int(cpg), float(slope))
        return a

    @staticmethod
    def backward(ctx, da):
        z, mean, istd, gamma, beta = ctx.saved_tensors
        cpg, slope = ctx.cfg
        dar, ldd = rows_view(da)
        zr, ldz = rows_view(z)
        nb, co = int(z.shape[0]), int(z.shape[1])
        m = zr.shape[0]
        nblk = L.query("arco_chan_stats_blocks", m // nb)
        ws = torch.empty(nb * (2 * co * nblk + 2 * co), dtype=torch.float32, device=da.device)
        dz = new_act_nd(nb, co, tuple(int(v) for v in z.shape[2:]), da.device)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        L.call("arco_gn_act_bwd", L.ptr(dar), ldd, L.ptr(zr), ldz, m, co, L.ptr(mean), L.ptr(istd), L.ptr(gamma), L.ptr(beta),
               slope, L.ptr(ws), L.ptr(dgamma), L.ptr(dbeta), 0, L.ptr(dz), co, nb, cpg)
        if not ctx.affine:
            dgamma = dbeta = None
        return dz, dgamma, dbeta, None, None, None


def gn_act(z, gamma, beta, num_groups=16, slope=0.0, eps=1e-5):
    """relu(nn.GroupNorm(num_groups, C)(z)) - vnetWithArgs.py:19-20."""
    c = int(z.shape[1])
    assert c % num_groups == 0, (c, num_groups)
    return GnActFn.apply(z, gamma, beta, c // num_groups, slope, eps)


def in_act(z, slope=0.0, eps=1e-5):
    """relu(nn.InstanceNorm3d(C)(z)) (no affine, batch statistics always) - vnetWithArgs.py:21-22."""
    return GnActFn.apply(z, None, None, 1, slope, eps)


def act(z, slope=0.0):
    """relu / leaky relu alone (`normalization='none'`)."""
    return BnActFn.apply(z, None, None, None, None, slope, 0.0, 0, 0.1, 1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# Glue that replaces chains of tensor-library launches of the step (profiles/r02_h: ~0.6 ms of fills / copies / adds)
# ---------------------------------------------------------------------------------------------------------------------
GRAD_SINKS = {}      # data_ptr of a graph-replayed pass's output -> (GraphedTrain, index into its static_grads)


def grad_sink(ptr, shape):
    """The persistent gradient-input buffer of the graph-replayed pass that produced the tensor at `ptr` (graphs.GraphedTrain
    keeps one per differentiable output, zero by invariant between steps), or None.  A consumer whose gradient is SPARSE
    (the row-sparse head) scatters straight into it and returns it - no dense zeros tensor, no copy into the graph's input -
    and registers a cleanup that re-zeroes the rows it touched after the backward graph has replayed."""
    hit = GRAD_SINKS.get(ptr)
    if hit is None:
        return None, None
    gt, k = hit
    buf = gt.static_grads[k]
    if not gt.captured or tuple(buf.shape) != tuple(shape) or gt.sink_busy[k]:
        return None, None
    if gt.grad_live[k]:                  # a dense gradient was copied in by an earlier step: restore the zero invariant
        buf.zero_()
        gt.grad_live[k] = False
    gt.sink_busy[k] = True
    gt.sink_version[k] = buf._version
    gt.sink_uses += 1
    return gt, k


class _SplitBatchFn(torch.autograd.Function):
    """x -> (x[:n], x[n:]) as views; the backward writes both halves' gradients into ONE buffer (the producing graph's
    gradient sink when there is one) instead of two zero-padded full-size tensors and an add."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.ptr, ctx.shape = n, x.data_ptr(), tuple(x.shape)
        ctx.like = x
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        n = ctx.n
        gt, k = grad_sink(ctx.ptr, ctx.shape)
        if gt is not None:
            buf = gt.static_grads[k]        # fully overwritten below; dense from now on (grad_live: re-zeroed before a sparse use)
            gt.grad_live[k] = True
        else:
            buf = torch.empty_like(ctx.like)
        for half, g in ((buf[:n], ga), (buf[n:], gb)):
            if g is None:
                half.zero_()
            else:
                half.copy_(g)
        ctx.like = None
        return buf, None


def split_batch(x, n):
    return _SplitBatchFn.apply(x, int(n))


class _FoldResidualFn(torch.autograd.Function):
    """W [n, n, 1, 1(, 1)] -> (W + I)[:, :c] as [n, c, 1, 1(, 1)], (W + I)[:, c:] as [n, n - c, 1, 1(, 1)] in one launch
    (model_2D.FeatureExtractor.forward_lowres1/2: the residual folded into the 1x1 weights); backward: one launch."""

    @staticmethod
    def forward(ctx, w, c):
        n = int(w.shape[0])
        wc = w.detach().contiguous()
        ones = (1,) * (w.dim() - 2)                      # [n, n, 1, 1] (2-D) or [n, n, 1, 1, 1] (3-D) 1x1 conv weights
        lo = torch.empty((n, c) + ones, dtype=torch.float32, device=w.device)
        hi = torch.empty((n, n - c) + ones, dtype=torch.float32, device=w.device)
        L.call("arco_fold_residual", L.ptr(wc), n, c, L.ptr(lo), L.ptr(hi))
        ctx.geom = (n, c, tuple(w.shape))
        return lo, hi

    @staticmethod
    def backward(ctx, dlo, dhi):
        n, c, shape = ctx.geom
        dlo = dlo.contiguous() if dlo is not None else None
        dhi = dhi.contiguous() if dhi is not None else None
        ref = dlo if dlo is not None else dhi
        dw = torch.empty(shape, dtype=torch.float32, device=ref.device)
        L.call("arco_unfold_residual", L.ptr(dlo), L.ptr(dhi), n, c, L.ptr(dw))
        return dw, None


def fold_residual(weight, c):
    return _FoldResidualFn.apply(weight, int(c))


class _CombineTermsFn(torch.autograd.Function):
    """sum_i w_i * term_i over device scalars in one launch (the step's loss combination); backward one launch."""

    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes
        n = len(terms)
        ts = [t.detach().reshape(()).float() for t in terms]
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        ws = (ctypes.c_float * n)(*[float(w) for w in weights])
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        L.call("arco_combine_terms", ptrs, ws, n, L.ptr(out))
        ctx.ws, ctx.n = ws, n
        return out

    @staticmethod
    def backward(ctx, g):
        grads = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
        L.call("arco_combine_terms_bwd", ctx.ws, ctx.n, L.ptr(g.contiguous().float()), L.ptr(grads))
        return (None,) + tuple(grads[i] for i in range(ctx.n))


def combine_terms(weights, terms):
    """sum_i weights[i] * terms[i] (0-d device tensors, python-float weights), differentiable w.r.t. the terms."""
    assert 1 <= len(terms) <= 8 and len(weights) == len(terms)
    dev = next((t.device for t in terms if torch.is_tensor(t) and t.is_cuda), None)
    if dev is None:
        raise RuntimeError("arco_amd.combine_terms: no term lives on the GPU")
    # a drop-in caller may hand in a python float or a CPU 0-d tensor (a constant-zero fallback): the kernel dereferences
    # device pointers, so every term is brought to the device first (constants carry no gradient; tensors keep theirs)
    terms = [t if (torch.is_tensor(t) and t.device == dev) else torch.as_tensor(t, dtype=torch.float32).to(dev) for t in terms]
    L.require_gpu(*terms)
    return _CombineTermsFn.apply(tuple(weights), *terms)
