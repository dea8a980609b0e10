"""Stage-2 ARCO trainer, 3-D (LA heart V-Net 112x112x80), on MI355X - drop-in for the reference's
code/train_arco_3d.py.  Every flag of train_arco_3d.py:26-87 is accepted with the same name / type /
default (they equal the 2-D trainer's except --patch_size [112,112,80], --func asmc, --k5 0.1).

`ArcoStep3D.step` follows train_arco_3d.py:257-415: V-Net student / teacher forwards, FeatureExtractor_3d, the two
1x1x1 q_representation convs (D=16), the 5-D contrastive loss (arco_amd.loss_helper), CE + Dice, unsupervised CE, the
mixing strategy of --apply_aug, the equivariance block (--eqv_pass), the opt-in revisiting loss (--revisit),
SGD-Nesterov, EMA.  batch_transform is the identity in the reference's 3-D pipeline (augment_3d.py:133-159).
--synthetic 0 trains from an LA dataset directory (build_loaders); --conv_mma selects reduced-precision MFMA operands.
"""
import contextlib
import logging
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn

from . import _contrast as C_
from . import dist as adist
from . import augment, glue, graphs, head, ops, optim
from .tps.rand_tps_3d import RandTPS as RandTPS3D
from .model_3D import ISD_3d, FeatureExtractor_3d
from .train_arco_2d import build_parser as _build_parser_2d

# pass-level concurrency on a second stream (see train_arco_2d.TEACHER_SIDE): 1 = the teacher's grouped pass beside the student
# forward, 2 = + the gradient-free warped student pass beside the heads / InfoNCE / backward.  ARCO_TEACHER_SIDE=0: off.
# 3 (default; tools/step_timeline3d.py) = + the teacher's FeatureExtractor on the side stream behind its pass (beside the masks and the
# student's FeatureExtractor), and the warped pass no longer queued behind the main stream's heads / row lists / loss forwards: it
# starts as soon as the host has drawn the warp - beside that low-occupancy stretch instead of beside the backward pass.
PASS_SIDE = min(3, int(os.environ.get("ARCO_TEACHER_SIDE", "4")))
LISTS_SIDE = int(os.environ.get("ARCO_LISTS_SIDE", "1"))
# lazy levels of the row-sparse heads: 2 = fea3 / fea4 on rows over the dense 56x56x40 map of fea2 (rounds 2-5), 3 = fea2 on rows too - the
# 224-channel map at 56x56x40 (450 MB at the LA size) is never written (head.LazyHead3dL3Fn, round 6)
HEAD_LEVELS = int(os.environ.get("ARCO_HEAD3D_LEVELS", "3"))
# ARCO_U0_SIDE=1: the teacher's first pass beside the grouped student pass (cutout / cutmix; see ArcoStep3D.step).  Opt-in: measured level
# (LA 21.2-21.3 -> 21.3-21.4 ms, LiTS-f16 13.7 -> 13.6-13.7) - the teacher's two passes stay serial on the second stream, which is then the
# longer one; running them beside each other as well needs their BatchNorm running-statistics updates deferred (profiles/r06_notes.md section 20)
U0_SIDE = int(os.environ.get("ARCO_U0_SIDE", "0"))
# T_MERGE (default; ARCO_T_MERGE=0: the serial order of rounds 2-5): the teacher's first pass and its grouped pass as ONE pass over
# cat(u, l, u_aug) with three BatchNorm groups (running statistics updated group after group: u, l, u_aug - the reference's order,
# train_arco_3d.py:260-262, 286-287), on the second stream beside the student's grouped pass.  Possible with cutout / cutmix, whose mixed
# images need the host-drawn boxes only (classmix masks are the pseudo-labels': serial order).  LA 21.1-21.2 -> 20.3 ms, LiTS-f16 13.5-13.6 -> 13.2
# on the same box (profiles/r06_notes.md section 20)
T_MERGE = int(os.environ.get("ARCO_T_MERGE", "1"))
FM_ROWS_HALF = int(os.environ.get("ARCO_FM_ROWS_HALF", "1"))     # --act_dtype f16: heads read the full-resolution maps as f16 (ops.fm_rows_half)
FEA_DIM_3D = [128, 64, 32, 16, 16]
REP_DIM_3D = 16                                  # train_arco_3d.py:148,207


def build_parser():
    p = _build_parser_2d()
    p.set_defaults(patch_size=[112, 112, 80], func='asmc', k5=0.1, exp='LA/example_training', model='vnet', max_iterations=6000,
                   root_path='/home/weicheng/selfLearning/DTC/data/2018LA_Seg_Training Set')       # train_arco_3d.py:27-36
    next(a for a in p._actions if a.dest == 'conv_mma').choices = ['f32x3', 'f32', 'f16', 'bf16']   # the volume kernels' extra modes
    p.add_argument('--act_dtype', type=str, default='f32', choices=['f32', 'f16'],
                   help="f16: the V-Net's activations and activation gradients are stored as f16 (BASELINE configs[4], 'fp16 MFMA "
                        "conv'): f16 matrix cores with fp32 accumulation, fp32 weights / BatchNorm statistics / loss / optimizer; "
                        "the heads keep fp32 tensors, their GEMM operands follow --head_mma")
    p.add_argument('--head_mma', type=str, default='auto', choices=['auto', 'f32x3', 'f16', 'bf16'],
                   help="matrix-core operands of the heads' GEMMs (FeatureExtractor_3d, q_representation, the row-sparse heads).  auto: f16 "
                        "with --act_dtype f16 (BASELINE configs[4] 'fp16 MFMA conv + contrastive': operands rounded to f16 in registers, "
                        "fp32 accumulate, gradient operands bf16; the full-resolution maps are read as stored f16 rows), else --conv_mma's mode")
    p.add_argument('--loss_scale', type=float, default=16384.0,
                   help='--act_dtype f16: gradients enter the f16 region multiplied by this power of two')
    p.add_argument('--eqv_pass', type=int, default=1,
                   help='1: run the equivariance block of train_arco_3d.py:368-388 (warp + one more student forward); '
                        'its loss only enters the objective at iteration 0 there (:390-393), afterwards it is a logged '
                        'value and a BatchNorm running-statistics update')
    return p


class ArcoStep3D:
    """State + one training step of the 3-D hot path (train_arco_3d.py:144-151,195-232,257-415)."""

    def __init__(self, args, device="cuda"):
        self.args = args
        self.dev = torch.device(device)
        C = args.num_classes
        ops.CONV_MMA = {"f32": 0, "f16": 1, "bf16": 2, "f32x3": 3}[getattr(args, "conv_mma", "f32x3")]
        ops.ACT_HALF = getattr(args, "act_dtype", "f32") == "f16"          # before the PackPlans: they carry the f16 packs
        hm = getattr(args, "head_mma", "auto")
        ops.HEAD_MMA = {"auto": 1 if ops.ACT_HALF else 0, "f32x3": 0, "f16": 1, "bf16": 2}[hm]
        ops.LOSS_SCALE = float(getattr(args, "loss_scale", 16384.0))
        self.memobank, self.queue_ptrlis, self.queue_size = [], [], []
        for i in range(C):                                                # :144-151
            self.memobank.append([torch.randn(1, REP_DIM_3D)])
            self.queue_size.append(args.queue_size if args.queue_size > 0 else 30000)
            self.queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
        if args.queue_size <= 0:
            self.queue_size[0] = 50000
        self.random_pool = None
        if getattr(args, "revisit", 0):                                   # :153-156 (drawn right after the banks)
            args.dense_head = 1
            assert args.K % args.batch_size == 0, "--K must be a multiple of --batch_size (train_arco_3d.py:110)"
            self.random_pool = glue.RevisitPool(args.K, REP_DIM_3D, args.patch_size, self.dev)
        else:      # the pool's normals are not needed, its place in the CPU-generator sequence is (weight init, samplers, warps)
            from . import samplers
            samplers.skip_randn(args.K * REP_DIM_3D * int(np.prod(args.patch_size)))
        self.isd = ISD_3d(K=args.K, m=0.99, Ts=0.01, Tt=0.1, num_classes=C,
                          latent_pooling_size=args.latent_pooling_size, latent_feature_size=args.latent_feature_size,
                          output_pooling_size=args.output_pooling_size, train_encoder=True, train_decoder=True).to(self.dev)
        self.model, self.ema_model = self.isd.model, self.isd.ema_model
        self.q_representation = nn.Sequential(nn.Conv3d(REP_DIM_3D, REP_DIM_3D, kernel_size=1, bias=False),
                                              nn.Conv3d(REP_DIM_3D, REP_DIM_3D, kernel_size=1, bias=False)).to(self.dev)
        self.k_feature_extractor = FeatureExtractor_3d(fea_dim=FEA_DIM_3D, output_dim=REP_DIM_3D).to(self.dev)
        self.q_feature_extractor = FeatureExtractor_3d(fea_dim=FEA_DIM_3D, output_dim=REP_DIM_3D).to(self.dev)
        adist.broadcast_module_states([self.isd, self.q_representation, self.q_feature_extractor,
                                       self.k_feature_extractor])
        params = [p for p in self.model.parameters() if p.requires_grad]
        params_rep = [p for p in self.q_representation.parameters() if p.requires_grad]
        params_fea = [p for p in self.q_feature_extractor.parameters() if p.requires_grad]
        self.heads_start = sum(p.numel() for p in params)     # flat_g[heads_start:] = the heads' gradient bucket (dist.mark_heads_done)
        self.optimizer = optim.SGDNesterov(params + params_rep + params_fea, lr=args.base_lr, weight_decay=0.0001,
                                           momentum=0.9, nesterov=True)
        with torch.no_grad():
            for t, s in zip(self.k_feature_extractor.parameters(), self.q_feature_extractor.parameters()):
                t.data.copy_(s.data)
                t.requires_grad = False
        self.k_fe_ema = optim.EmaPair(self.q_feature_extractor, self.k_feature_extractor)
        for m in (self.model, self.ema_model, self.q_representation, self.k_feature_extractor,
                  self.q_feature_extractor):
            m.train()
        # packed conv weights: one launch per weight owner per step (ops.PackPlan), refreshed by the owner
        plan_s = ops.PackPlan([self.model, self.q_representation, self.q_feature_extractor], True, half=[ops.ACT_HALF, False, False])
        self.optimizer.plans = [plan_s]
        pairs = self.isd._ensure_ema_pairs()
        pairs[0].plans = [ops.PackPlan([self.ema_model], False, half=[ops.ACT_HALF])]
        for pr in pairs[1:]:
            pr.plans = [ops.PackPlan([], False)]
        self.k_fe_ema.plans = [ops.PackPlan([self.k_feature_extractor], False)]
        self.plans = [plan_s] + [pl for pr in pairs for pl in pr.plans] + self.k_fe_ema.plans
        self.iter_num = 0
        self._side, self._tps_pending = None, False
        if ops._WGRAD_SIDE_ENV is None:        # weight gradients on the side stream behind their data gradient: -0.3 .. -0.5 ms on the LA step with
            ops.WGRAD_SIDE = 3                 # round 6's 3x3x3 kernels (level on LiTS-f16; the 2-D step loses 1 ms with it: train_arco_2d resets it)
        self._ovf_host, self._ovf_event, self.overflow_steps, self._clean_steps = None, None, 0, 0     # f16 overflow guard
        self.keep_debug = False          # tests: keep the last step's plan and anchor rows (self.debug)
        use_graphs = bool(getattr(args, "graphs", 1))
        g_train = use_graphs and bool(getattr(args, "graph_train", 0))
        self.s_train_u = graphs.GraphedTrain(self.model, enabled=g_train)    # student passes: fwd + bwd graphs
        self.s_train_l = graphs.GraphedTrain(self.model, enabled=g_train)
        self.tps = None
        if getattr(args, "eqv_pass", 1):                                 # :231-237 (the constructor draws one warp)
            self.tps = RandTPS3D(args.patch_size[0], args.patch_size[1], args.patch_size[2], batch_size=2 * args.batch_size,
                                 sigma=args.tps_sigma, border_padding=False, random_mirror=True, random_scale=(0.8, 1.2),
                                 mode='affine', device=device)
        self.batched_passes = bool(getattr(args, "batched_passes", 1))
        self.s_train_lu = graphs.GraphedTrain(self.model, enabled=g_train)
        self.t_fwd_lu = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_u0 = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_ulu = graphs.GraphedForward(self.ema_model, enabled=use_graphs)      # T_MERGE: (u, l, u_aug) as one three-group pass
        self.t_fwd_l = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_u = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        # the warped student pass carries no gradient after iteration 0 (:390-393): replayed as one graph - its ~300 eager
        # launches sat right behind the sampler stage, the stretch of the step where the GPU waits for the host
        self.s_fwd_tps = graphs.GraphedForward(self.model, enabled=use_graphs)

    def _unscale_and_guard(self):
        """--act_dtype f16: divide the loss scale out of the V-Net's stretch of the flat gradient and guard the step against an
        overflow of the f16 backward (an inf / NaN in flat_g would go straight into SGD, the EMA teacher and,
        a step later, the memory banks - silently and for good).  All on the device, no host synchronisation in the step:
        `ok` = every V-Net gradient finite (on every rank); non-finite values are replaced by zeros and the V-Net's gradient is multiplied
        by ok, so an overflowed step leaves the V-Net with weight decay and momentum only instead of poisoning the run (the heads' fp32
        gradients do not pass through the f16 backward; they are guarded behind their all-reduce: _guard_heads_and_publish).
        The flag is copied to pinned memory and read at the START of the next step (long complete by then): an overflow halves
        the loss scale (floor 1), 500 clean steps double it again up to --loss_scale (dynamic loss scaling)."""
        gv = self.optimizer.flat_g[:self.heads_start]
        gv.mul_(1.0 / ops.LOSS_SCALE)
        ok = torch.isfinite(gv).all()
        torch.nan_to_num_(gv, nan=0.0, posinf=0.0, neginf=0.0)
        okf = ok.to(torch.float32).view(1)
        if adist.is_dist():
            # every rank must take the same decision (a rank that zeroed alone would leave the replicas' loss scales, graphs and -
            # through momentum - weights apart): MIN over ranks, issued by every rank every f16 step, in front of the V-Net bucket
            adist.allreduce_min(okf)
        # Only the V-Net's stretch is written: the heads' gradients are fp32, produced UPSTREAM of the f16 backward (always
        # finite), and under data parallelism their bucket flat_g[heads_start:] is already inside an asynchronous all-reduce
        # started by dist.mark_heads_done's backward hook - nothing may write it before allreduce_grads has waited.
        gv.mul_(okf)
        return okf > 0

    def _guard_heads_and_publish(self, ok):
        """Second half of the guard, AFTER allreduce_grads has waited for the heads' bucket: an f16 forward overflow (an inf
        in a feature-map row or in the loss) makes the heads' fp32 gradients NaN although they never pass through the f16 backward.
        The reduced bucket is identical on every rank, so every rank takes the same decision without another collective: non-finite
        values -> zeros, the bucket multiplied by its own finite flag, and the flag folded into the overflow flag that drives the
        dynamic loss scale (copied to pinned memory, read at the start of the next step)."""
        gh = self.optimizer.flat_g[self.heads_start:]
        okh = torch.isfinite(gh).all()
        torch.nan_to_num_(gh, nan=0.0, posinf=0.0, neginf=0.0)
        gh.mul_(okh.to(torch.float32))
        ok = ok.view(1) & okh.view(1)
        if self._ovf_host is None:
            self._ovf_host = torch.ones(1, dtype=torch.bool).pin_memory()
        self._ovf_host.copy_(ok, non_blocking=True)
        self._ovf_event = torch.cuda.Event()
        self._ovf_event.record()

    def _loss_scale_update(self):
        """Host side of the guard: consume last step's flag (see _unscale_and_guard)."""
        ev = self._ovf_event
        if ev is None:
            return
        ev.synchronize()              # recorded a whole step ago: returns at once
        self._ovf_event = None
        if not bool(self._ovf_host[0]):
            self.overflow_steps += 1
            self._clean_steps = 0
            ops.LOSS_SCALE = max(1.0, ops.LOSS_SCALE / 2.0)
            self._recapture_train_graphs()
            logging.warning("f16 backward overflowed at iteration %d: step reduced to a zero-gradient step, loss scale -> %g",
                            self.iter_num - 1, ops.LOSS_SCALE)
        else:
            self._clean_steps += 1
            if self._clean_steps >= 500 and ops.LOSS_SCALE < float(getattr(self.args, "loss_scale", 16384.0)):
                ops.LOSS_SCALE *= 2.0
                self._clean_steps = 0
                self._recapture_train_graphs()

    def _recapture_train_graphs(self):
        """The loss scale is a kernel argument of the boundary cast inside the captured backward graphs: a new scale needs a
        new capture (the next call of each GraphedTrain re-captures; rare)."""
        for v in vars(self).values():
            if isinstance(v, graphs.GraphedTrain):
                v.captured = False

    @staticmethod
    def _lazy_teacher(kfe, fm_t):
        if HEAD_LEVELS == 3:
            return head.LazyTeacher3DL3(*kfe.forward_lowres1(fm_t), kfe.fea2.weight, kfe.fea3.weight, kfe.fea4.weight)
        return head.LazyTeacher3D(*kfe.forward_lowres2(fm_t), kfe.fea3.weight, kfe.fea4.weight)

    def q_rep(self, x):
        x = ops.conv(x, self.q_representation[0].weight)
        return ops.conv(x, self.q_representation[1].weight)

    def step(self, l_data, l_label, u_data, epoch_num=0, max_epoch=1):
        a = self.args
        C = a.num_classes
        if ops.ACT_HALF:
            self._loss_scale_update()
        for pl in self.plans:                                            # stale only if someone else touched weights
            if not pl.valid:
                pl.refresh()
        # f16 activation storage with the row-sparse heads: the two full-resolution feature maps are consumed as stored (f16 rows,
        # f16 row-sparse gradients) - no dense cast of a full-resolution map in either direction (ops.fm_rows_half)
        rows_half = ops.ACT_HALF and FM_ROWS_HALF and not getattr(a, "dense_head", 0) and self.random_pool is None
        fm_ctx = ops.fm_rows_half if rows_half else contextlib.nullcontext
        # (opt-in, U0_SIDE) The teacher's first pass (pseudo-labels, :260-262) runs ALONE at the head of the step (2.2 of 20.8 ms at the LA size).
        # With cutout / cutmix the mixed IMAGES need only the boxes - host draws -, not the pseudo-labels: the boxes are drawn at the
        # reference's point of the generator order, the images are mixed at once, the grouped student pass starts on this stream while
        # the teacher's first pass and, behind it on the same (second) stream, its grouped pass run beside it; labels and logits are
        # mixed with the same boxes once the teacher is done.  classmix (its masks are the pseudo-labels') keeps the serial order.
        u0_side = ((U0_SIDE or T_MERGE) and PASS_SIDE >= 1 and a.apply_aug in ("cutout", "cutmix") and self.batched_passes
                   and l_data.shape == u_data.shape)
        t_merge = bool(u0_side and T_MERGE)
        if t_merge:
            if self._side is None:
                self._side = torch.cuda.Stream()
            mix_desc = augment.draw_boxes(int(u_data.shape[0]), tuple(int(v) for v in u_data.shape[2:]))
            u_aug = augment.mix_images(u_data, a.apply_aug, mix_desc)
        elif u0_side:
            if self._side is None:
                self._side = torch.cuda.Stream()
            mix_desc = augment.draw_boxes(int(u_data.shape[0]), tuple(int(v) for v in u_data.shape[2:]))
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side), torch.no_grad(), ops.logits_only():      # :260-262
                pred_u0, _, _ = self.t_fwd_u0(u_data)
                pseudo_logits, pseudo_labels = glue.softmax_max(pred_u0)
            u_aug = augment.mix_images(u_data, a.apply_aug, mix_desc)
        else:
            with torch.no_grad(), ops.logits_only():                         # :260-262
                pred_u0, _, _ = self.t_fwd_u0(u_data)
                pseudo_logits, pseudo_labels = glue.softmax_max(pred_u0)
            if self.keep_debug:      # tests: the teacher's decisions before the mixing (cutout writes -1 into the labels in place)
                dbg_pseudo = (pseudo_labels.clone(), pseudo_logits.clone())
            # :268-278: the mixing strategy of --apply_aug on the GPU (train_arco_3d.py:270-271); the PIL transforms are identity
            u_aug, u_aug_label, u_aug_logits = augment.generate_unsup_data_3d(u_data, pseudo_labels, pseudo_logits, mode=a.apply_aug)
        self.k_fe_ema.update(0.99)                                      # :279-281
        batched = self.batched_passes and l_data.shape == u_aug.shape
        lazy_t_side = None
        if batched:     # labelled + unlabelled volumes as one pass with two BatchNorm groups (see train_arco_2d.py)
            lu = torch.cat((l_data, u_aug))
            nb_l = int(l_data.shape[0])
            t_side = None
            if PASS_SIDE >= 1:      # the teacher's grouped pass on a second stream, beside the student forward (see train_arco_2d.TEACHER_SIDE)
                if self._side is None:
                    self._side = torch.cuda.Stream()
                t_side = self._side
                t_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(t_side), torch.no_grad():
                    if t_merge:
                        nb_u = int(u_data.shape[0])
                        with ops.bn_groups(3), fm_ctx():
                            pred_ulu, _, fm_ulu = self.t_fwd_ulu(torch.cat((u_data, lu)))      # :260-262 and :286-287 in one pass
                        pred_u0, pred_t, fm_t = pred_ulu[:nb_u], pred_ulu[nb_u:], [f[nb_u:] for f in fm_ulu]
                        pseudo_logits, pseudo_labels = glue.softmax_max(pred_u0)
                    else:
                        with ops.bn_groups(2), fm_ctx():
                            pred_t, _, fm_t = self.t_fwd_lu(lu)              # :286-287
                    t_done = t_side.record_event()
                    if PASS_SIDE >= 3 and not getattr(a, "dense_head", 0):   # :292-293 (the teacher's heads: joined before the row lists)
                        kfe = self.k_feature_extractor
                        lazy_t_side = self._lazy_teacher(kfe, fm_t)
            with ops.bn_groups(2), fm_ctx():
                pred_all, _, fm_s = self.s_train_lu(lu)                  # :283-284
            if t_side is not None:
                torch.cuda.current_stream().wait_event(t_done)
            if u0_side:          # (t_done lies behind the teacher's first pass on the same stream)
                if self.keep_debug:
                    dbg_pseudo = (pseudo_labels.clone(), pseudo_logits.clone())
                _, u_aug_label, u_aug_logits = augment.generate_unsup_data_3d(u_data, pseudo_labels, pseudo_logits, mode=a.apply_aug, desc=mix_desc)
            pred_l, pred_u = ops.split_batch(pred_all, nb_l)
            if PASS_SIDE >= 3:     # the warped pass's inputs (volumes, mixed labels, the grouped pass's logits) exist from here on
                self._fwd_ready = torch.cuda.current_stream().record_event()
        else:
            with ops.bn_defer(0), fm_ctx():                              # running statistics: l first (:283), then u
                pred_u, _, u_fm = self.s_train_u(u_aug)                  # :284
        with torch.no_grad():
            if batched:
                if PASS_SIDE < 1:
                    with ops.bn_groups(2), fm_ctx():
                        pred_t, _, fm_t = self.t_fwd_lu(lu)              # :286-287
                pred_l_t, pred_u_t = pred_t[:nb_l], pred_t[nb_l:]
            else:
                with fm_ctx():
                    pred_l_t, _, l_fm_t = self.t_fwd_l(l_data)           # :286
                    pred_u_t, _, u_fm_t = self.t_fwd_u(u_aug)            # :287
            alpha_t = 20 * (1 - epoch_num / max_epoch)
            label_l = glue.label_onehot(l_label, C)
            label_u = glue.label_onehot(u_aug_label, C)
            prob_l_t = glue.softmax(pred_l_t)
            prob_u_t = glue.softmax(pred_u_t)
            low_mask_all, high_mask_all = glue.entropy_masks(pred_u, l_label, u_aug_label, alpha_t)
        plan = C_.contrast_masks(label_l, label_u, prob_l_t, prob_u_t, low_mask_all, high_mask_all,
                                 delta_n=a.strong_threshold_u2pl)
        lists_done = None
        if lazy_t_side is not None and LISTS_SIDE:
            # row lists and prototypes need the masks and the TEACHER's heads only: on the side stream (behind those heads), beside
            # the student's FeatureExtractor on this one, instead of in line behind it
            self._side.wait_event(torch.cuda.current_stream().record_event())
            with torch.cuda.stream(self._side):
                C_.contrast_lists_protos(plan, None, lazy_t_side)
                lists_done = self._side.record_event()
        if self.keep_debug:      # tests: the step's gradient-free decision inputs (tests/test_step3d_parity_gpu.py)
            self.decisions = dict(pseudo_labels=dbg_pseudo[0], pseudo_logits=dbg_pseudo[1], low=low_mask_all, high=high_mask_all,
                                  prob_l_t=prob_l_t, prob_u_t=prob_u_t)
        dense = getattr(a, "dense_head", 0)
        if not batched:
            with fm_ctx():
                pred_l, _, l_fm = self.s_train_l(l_data)                 # :283
            ops.apply_deferred_bn()
            fm_t = [torch.cat((x, y)) for x, y in zip(l_fm_t, u_fm_t)]
            fm_s = [torch.cat((x, y)) for x, y in zip(l_fm, u_fm)]
        kfe, qfe = self.k_feature_extractor, self.q_feature_extractor
        with torch.no_grad():                                            # :292-293
            if dense:
                rep_all_teacher, lazy_t = kfe(fm_t), None
            elif lazy_t_side is not None:
                rep_all_teacher, lazy_t = None, lazy_t_side
            else:       # teacher rows are only needed as class means (prototypes) and <= queue_size keys per class
                rep_all_teacher, lazy_t = None, self._lazy_teacher(kfe, fm_t)
        fm_s = adist.mark_heads_done(fm_s, self.optimizer, self.heads_start)     # data parallel: heads' gradient bucket reduced early
        if dense:
            rep_all = self.q_rep(qfe(fm_s))                              # :289-296,301
        else:
            s_maps = qfe.forward_lowres1(fm_s) if HEAD_LEVELS == 3 else qfe.forward_lowres2(fm_s)
        if lists_done is not None:
            torch.cuda.current_stream().wait_event(lists_done)
        else:
            if lazy_t_side is not None:
                torch.cuda.current_stream().wait_stream(self._side)      # the teacher's FeatureExtractor (side stream)
            C_.contrast_lists_protos(plan, rep_all_teacher, lazy_t)     # row lists, prototypes: device-side inputs only
        # the loss forwards need neither counters nor samples: queued before the host blocks (see train_arco_2d.py)
        loss_ce, loss_dice = glue.supervised_loss(pred_l, l_label)       # :306-310
        unsup_loss = glue.compute_unsupervised_loss(pred_u, u_aug_label, u_aug_logits, a.strong_threshold)
        # counters -> [sample-independent GPU work] -> sampler replay on the host -> anchors (see train_arco_2d.py)
        C_.contrast_counts(plan, self.memobank, self.queue_size,
                           adist.anchors_for_rank(a.num_queries, getattr(a, "anchors_per_rank", "split")), a.num_negatives)
        tps_early = (PASS_SIDE >= 3 and batched and getattr(a, "eqv_pass", 1) and self.iter_num > 0 and self.s_fwd_tps.enabled)

        def enqueue():                     # teacher key rows -> banks (no generator draws, no use of the sampled indices)
            C_.contrast_enqueue(plan, rep_all_teacher, self.memobank, self.queue_ptrlis, self.queue_size, lazy_teacher=lazy_t,
                                defer_anchor_pix=True)
        if not tps_early:
            enqueue()
        C_.contrast_draw(plan, a.func, defer=True)     # indices collected by contrast_anchor_pix below
        loss_eqv = None
        if getattr(a, "eqv_pass", 1):
            # :368-388.  The warp is drawn after the samplers (same torch-generator order as the reference).
            nb2 = int(l_data.shape[0]) + int(u_aug.shape[0])
            if self.tps is None or self.tps.batch_size != nb2:
                self.tps = RandTPS3D(a.patch_size[0], a.patch_size[1], a.patch_size[2], batch_size=nb2, sigma=a.tps_sigma,
                                     border_padding=False, random_mirror=True, random_scale=(0.8, 1.2), mode='affine',
                                     device=l_data.device)

            def warp_inputs():
                with torch.no_grad():
                    eq_mask = glue.eqv_mask(torch.cat((l_label, u_aug_label)),
                                            torch.cat((u_aug_logits.new_ones(l_label.shape), u_aug_logits)), a.weak_threshold)
                    self.tps.reset_control_points()                      # :377
                    return (self.tps(torch.cat((l_data, u_aug))), self.tps(eq_mask, padding_mode='zeros'),
                            self.tps(torch.cat((pred_l.detach(), pred_u.detach())), padding_mode='zeros'))
            side_ok = PASS_SIDE >= 2 and self.iter_num > 0 and self.s_fwd_tps.enabled
            if side_ok and self._side is None:
                self._side = torch.cuda.Stream()
            if tps_early:
                # after iteration 0 the warped pass is a logged value and a running-statistics update (:390-393): nothing of this
                # step waits for it.  Warps, pass and loss run on the second stream from the moment the host has drawn the warp -
                # beside the heads' forwards, row lists and loss forwards of the main stream (a stretch of small launches), the
                # InfoNCE and the start of the backward pass - behind the grouped pass (running statistics: l, u, then this pass),
                # and are joined before the optimiser touches the weights.
                self._side.wait_event(self._fwd_ready)
                with torch.cuda.stream(self._side), torch.no_grad(), ops.logits_only():
                    images_tps, mask_tps, pred_tps_org = warp_inputs()
                    pred_tps = self.s_fwd_tps(images_tps)[0]
                    loss_eqv = glue.eqv_loss(pred_tps, pred_tps_org, mask_tps)
                self._tps_pending = True
                enqueue()
            else:
                images_tps, mask_tps, pred_tps_org = warp_inputs()
                if side_ok:    # (PASS_SIDE 2: behind everything queued on the main stream so far, beside InfoNCE and backward)
                    self._side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._side), torch.no_grad(), ops.logits_only():
                        pred_tps = self.s_fwd_tps(images_tps)[0]
                        loss_eqv = glue.eqv_loss(pred_tps, pred_tps_org, mask_tps)
                    self._tps_pending = True
                else:
                    with torch.set_grad_enabled(self.iter_num == 0), ops.logits_only():   # only iteration 0 back-propagates it (:390-393)
                        pred_tps = (self.model if self.iter_num == 0 else self.s_fwd_tps)(images_tps)[0]                 # :380
                        loss_eqv = glue.eqv_loss(pred_tps, pred_tps_org, mask_tps)
        C_.contrast_anchor_pix(plan)
        zero_path = plan.valid_seg <= 1 or not plan.entries
        if zero_path:
            reco_loss = self.q_representation[1].weight.sum() * 0.0
        elif dense:
            A_all = C_.GatherRowsFn.apply(rep_all, plan.anchor_pix)
            reco_loss, _ = C_.contrast_infonce(plan, A_all, self.memobank, temp=0.5)
        else:
            if HEAD_LEVELS == 3:
                A_all = head.lazy_head3d_l3(*s_maps, qfe.fea2.weight, qfe.fea3.weight, qfe.fea4.weight, self.q_representation[0].weight,
                                            self.q_representation[1].weight, plan.anchor_pix)
            else:
                A_all = head.lazy_head3d(*s_maps, qfe.fea3.weight, qfe.fea4.weight, self.q_representation[0].weight,
                                         self.q_representation[1].weight, plan.anchor_pix)
            reco_loss, _ = C_.contrast_infonce(plan, A_all, self.memobank, temp=0.5)
        if self.keep_debug and plan.valid_seg > 1 and plan.entries:
            self.debug = dict(plan=plan, A_all=A_all.detach(), banks=[m[0] for m in self.memobank])
        first = self.iter_num == 0 and loss_eqv is not None
        if first:
            ws, terms = [1.0, 1.0, 1.0, 1.0], [unsup_loss, loss_dice, loss_ce, loss_eqv]      # :393 (iter_num / max_iterations == 0)
        else:                                                                               # :391 (k4*loss_q only with --revisit 1)
            ws = [a.k1 * adist.anchor_weight(a.num_queries, getattr(a, "anchors_per_rank", "split")), a.k3, 1.0, 1.0]
            terms = [reco_loss, unsup_loss, loss_dice, loss_ce]
        loss_q = None
        if self.random_pool is not None:      # :304 (before the pool update) and :365; constant w.r.t. every parameter
            nb_l = int(l_data.shape[0])
            loss_q = glue.get_revisiting_loss(self.random_pool, rep_all[nb_l:], rep_all_teacher[nb_l:], topk=a.topk)
            glue.revisit_enqueue(rep_all_teacher[nb_l:], self.random_pool)
            if not first:
                ws.append(a.k4); terms.append(loss_q)
        loss = ops.combine_terms(ws, terms)           # one launch (and one for its backward) instead of a chain of 0-d ops
        self.optimizer.zero_grad()
        loss.backward()
        ops.join_side()                     # weight gradients queued on the side stream (ops._wgrad)
        if self._tps_pending:               # the warped pass on the second stream reads the weights the optimiser is about to change
            torch.cuda.current_stream().wait_stream(self._side)
            self._tps_pending = False
        if zero_path and not first:      # `0 * rep.sum()` (loss_helper.py:588-595): zero gradients for every head parameter
            self.optimizer.touch_from(self.heads_start)
        if ops.ACT_HALF:       # the V-Net's parameter gradients carry the loss scale of the f16 region
            ok_vnet = self._unscale_and_guard()
        adist.allreduce_grads(self.optimizer)
        if ops.ACT_HALF:
            self._guard_heads_and_publish(ok_vnet)
        self.optimizer.step()
        self.isd._momentum_update_key_encoder()
        lr_ = a.base_lr * (1.0 - self.iter_num / a.max_iterations) ** 0.9
        for g in self.optimizer.param_groups:
            g['lr'] = lr_
        self.iter_num += 1
        self.last_terms = dict(ce=loss_ce.detach(), dice=loss_dice.detach(), unsup=unsup_loss.detach(),
                               reco=reco_loss.detach())
        if loss_eqv is not None:
            self.last_terms["eqv"] = loss_eqv.detach()
        if loss_q is not None:
            self.last_terms["loss_q"] = loss_q
        return loss.detach(), reco_loss.detach()


def synthetic_volume_batch(b, patch, n_cls, seed, device):
    """LA-shaped synthetic batch: volumes U[0,1), ellipsoid labels."""
    rs = np.random.RandomState(seed)
    img = torch.from_numpy(rs.uniform(size=(b, 1, *patch)).astype(np.float32))
    lab = np.zeros((b, *patch), dtype=np.int64)
    g = np.mgrid[0:patch[0], 0:patch[1], 0:patch[2]]
    for i in range(b):
        for c in range(1, n_cls):
            ctr = [rs.randint(p // 4, 3 * p // 4) for p in patch]
            r = [max(2, rs.randint(p // 8, p // 3)) for p in patch]
            lab[i][sum(((g[d] - ctr[d]) / r[d]) ** 2 for d in range(3)) < 1.0] = c
    return img.to(device), torch.from_numpy(lab).to(device)


def build_loaders(args, generator=None):
    """The two training loaders of train_arco_3d.py:158-190: LAHeartWithIndex (first --labeled_num cases labeled, the
    rest unlabeled) with RandomRotFlip -> RandomCrop(patch) -> ToTensor, drawn with replacement, last batch dropped."""
    from torch.utils.data import ConcatDataset, DataLoader
    from torch.utils.data.sampler import RandomSampler
    from .dataloaders import Compose
    from .dataloaders.la_heart import LAHeartWithIndex, RandomCrop, RandomRotFlip, ToTensor
    tf = lambda: Compose([RandomRotFlip(), RandomCrop(args.patch_size), ToTensor()])
    db_l = LAHeartWithIndex(base_dir=args.root_path, split="train", num=None, transform=tf(), index=args.labeled_num, label_type=1)
    db_u = LAHeartWithIndex(base_dir=args.root_path, split="train", num=None, transform=tf(), index=args.labeled_num, label_type=0)
    while len(db_l) < len(db_u):                                           # :171-172
        db_l = ConcatDataset([db_l, db_l])
    mk = lambda ds: DataLoader(ds, batch_size=args.batch_size, sampler=RandomSampler(data_source=ds, replacement=True, generator=generator),
                               drop_last=True, pin_memory=True)
    return mk(db_l), mk(db_u)


def train(args, snapshot_path):
    rank, world = adist.init()
    if getattr(args, "dp_local_thresholds", 0):
        glue.state_reduce_hook = None
    dev = torch.device("cuda", adist.local_rank())
    torch.cuda.set_device(dev)
    stepper = ArcoStep3D(args, dev)
    b = args.batch_size
    loaders = None
    if args.synthetic:
        iters_per_epoch = 100
        if world > 1:         # every rank draws its own cutmix boxes / sampler indices / warps (seed + rank), after the broadcast
            adist.seed_data_pipeline(args.seed)
    else:
        # data parallel: every rank draws its own samples / augmentations (seed + rank), after the weight broadcast above
        loaders = build_loaders(args, generator=adist.seed_data_pipeline(args.seed) if world > 1 else None)
        iters_per_epoch = len(loaders[1])
        logging.info("{} iterations per epoch".format(iters_per_epoch))
        resume = "../model/{}_{}_labeledfinal/{}/iter_30000.pth".format(args.resume, args.labeled_num, args.model)
        if os.path.exists(resume):                                      # stage-1 weights (:198-201), when present
            sd = torch.load(resume, map_location="cpu")
            stepper.isd.model.load_state_dict(sd); stepper.isd.ema_model.load_state_dict(sd)
            for pl in stepper.plans:
                pl.valid = False
        else:
            logging.info("no stage-1 checkpoint at {}: training from the random initialisation".format(resume))
    max_epoch = args.max_iterations // iters_per_epoch + 1
    l_iter = u_iter = None
    while stepper.iter_num < args.max_iterations:
        it = stepper.iter_num
        if args.synthetic:
            l_img, l_lab = synthetic_volume_batch(b, args.patch_size, args.num_classes, 2 * it * world + rank, dev)
            u_img, _ = synthetic_volume_batch(b, args.patch_size, args.num_classes, (2 * it + 1) * world + rank, dev)
        else:
            if it % iters_per_epoch == 0:
                l_iter, u_iter = iter(loaders[0]), iter(loaders[1])
            l_next, u_next = next(l_iter), next(u_iter)
            l_img, l_lab = l_next['image'].to(dev, non_blocking=True), l_next['label'].to(dev, non_blocking=True).long()
            u_img = u_next['image'].to(dev, non_blocking=True)
        loss, reco = stepper.step(l_img, l_lab, u_img, it // iters_per_epoch, max_epoch)
        if rank == 0:
            if getattr(args, "revisit", 0):
                logging.info('iteration %d : loss : %f, reco_loss: %f' % (stepper.iter_num, loss.item(), reco.item()))
            else:      # (see train_arco_2d.train: the gradient-free revisiting term of the reference's logged total is opt-in)
                logging.info('iteration %d : loss : %f (without the gradient-free revisiting term k4*loss_q, k4 = %g: --revisit 1 adds it), '
                             'reco_loss: %f' % (stepper.iter_num, loss.item(), args.k4, reco.item()))
            if stepper.iter_num % 1000 == 0:                           # :441-449
                path = os.path.join(snapshot_path, 'iter_' + str(stepper.iter_num) + '.pth')
                # parameters are views into the optimiser's flat buffer: save private copies, not the shared storage
                torch.save({k: v.detach().clone() for k, v in stepper.isd.model.state_dict().items()}, path)
    return "Training Finished!"


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.set_num_threads(min(4, torch.get_num_threads()))
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    snapshot_path = "../model/{}_{}_labeled{}/{}".format(args.exp, args.labeled_num, 'final', args.model)
    try:                                                             # the reference writes next to the repo (../model)
        os.makedirs(snapshot_path, exist_ok=True)
    except OSError:                                                  # read-only parent: keep the run inside the cwd
        snapshot_path = snapshot_path[1:]
        os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(filename=snapshot_path + "/log.txt", level=logging.INFO,
                        format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S')
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    logging.info(str(args))
    return train(args, snapshot_path)


if __name__ == "__main__":
    main()
