"""Stage-2 ARCO trainer, 2-D, on MI355X - drop-in for the reference's code/train_arco_2d.py.

Plugin surface kept identical: every command-line flag of train_arco_2d.py:26-87 (names,
types, defaults; the dead ones are accepted and ignored exactly like the reference),
`--exp` substring dataset dispatch (:91-106), snapshot directory
../model/{exp}_{labeled_num}_labeledfinal/{model} and `iter_N.pth` = state_dict of the
student U-Net with the reference's key names (:462-470, :509-511).

The step itself (train_arco_2d.py:284-435) is `ArcoStep2D.step`: hot path (SURVEY §8a rows
N1-N5, T1, L1-L6, O1) on hand-written HIP kernels.  One process per GPU; with WORLD_SIZE>1
gradients are all-reduced and new negative keys all-gathered over RCCL (arco_amd/dist.py).
Every loss term of the reference step is here: contrastive (k1), unsupervised CE (k3), CE + Dice, the TPS
equivariance term (k2) and - opt-in, --revisit 1, it has no gradient path - the revisiting loss (k4); so is the mixing
strategy of --apply_aug (arco_amd/augment.py) and, with --synthetic 0, the ACDC / MM slice loaders (build_loaders).
Also built: batch_transform (PIL round trip, ColorJitter, GaussianBlur, AdvMorph; --batch_transform).
"""
import argparse
import logging
import contextlib
import os
import random
import sys

import numpy as np
import torch
import torch.nn as nn

from . import dist as adist
from . import _contrast as C_
from . import augment, glue, graphs, head, ops, optim
from .tps import RandTPS
from .model_2D import ISD, FeatureExtractor

FEA_DIM = [256, 128, 64, 32, 16]
REP_DIM = sum(FEA_DIM)          # 496 (train_arco_2d.py:151,232)


def build_parser():
    """Every add_argument of train_arco_2d.py:26-87, same names/types/defaults."""
    p = argparse.ArgumentParser()
    p.add_argument('--root_path', type=str, default='../data/ACDC', help='Name of Experiment')
    p.add_argument('--exp', type=str, default='ACDC/example_training', help='experiment_name')
    p.add_argument('--model', type=str, default='unet', help='model_name')
    p.add_argument('--max_iterations', type=int, default=30000, help='maximum epoch number to train')
    p.add_argument('--batch_size', type=int, default=4, help='batch_size per gpu')
    p.add_argument('--deterministic', type=int, default=1, help='whether use deterministic training')
    p.add_argument('--base_lr', type=float, default=0.01, help='segmentation network learning rate')
    p.add_argument('--patch_size', type=list, default=[256, 256], help='patch size of network input')
    p.add_argument('--seed', type=int, default=1337, help='random seed')
    p.add_argument('--num_classes', type=int, default=4, help='output channel of network')
    p.add_argument('--consistency', type=float, default=0.1, help='consistency')
    p.add_argument('--consistency_rampup', type=float, default=200.0, help='consistency_rampup')
    p.add_argument('--labeled_bs', type=int, default=2, help='labeled_batch_size per gpu')
    p.add_argument('--labeled_num', type=int, default=7, help='labeled data')
    p.add_argument('--strong_threshold', default=0.97, type=float)
    p.add_argument('--strong_threshold_u2pl', default=0.97, type=float)
    p.add_argument('--weak_threshold', default=0.7, type=float)
    p.add_argument('--temp', default=0.5, type=float)
    p.add_argument('--num_negatives', default=512, type=int, help='number of negative keys')
    p.add_argument('--num_queries', default=256, type=int, help='number of queries per segment per image')
    p.add_argument('--mona_feature_size', type=int, default=512, help='the feature size of latent vectors')
    p.add_argument('--apply_aug', default='cutmix', type=str, help='apply semi-supervised method: cutout cutmix classmix')
    p.add_argument('--resume', type=str, default='ACDC/training_pool_latentF512_K36', help='if we should resume from checkpoint')
    p.add_argument('--K', type=int, default=36, help='the size of cache')
    p.add_argument('--k1', type=float, default=0.01, help='the weights for contrastive loss')
    p.add_argument('--k2', type=float, default=1.0, help='the weights for eqv loss')
    p.add_argument('--k3', type=float, default=1.0, help='the weights for unsup loss')
    p.add_argument('--k4', type=float, default=1.0, help='the weights for nn loss')
    p.add_argument('--k5', type=float, default=1.0, help='the weights for nn loss')
    p.add_argument('--topk', type=int, default=5, help='the size of cache')
    p.add_argument('--latent_pooling_size', type=int, default=1, help='the pooling size of latent vector')
    p.add_argument('--latent_feature_size', type=int, default=512, help='the feature size of latent vectors')
    p.add_argument('--output_pooling_size', type=int, default=8, help='the pooling size of output head')
    p.add_argument('--combinations', type=int, default=0, help='0: all, 1: no reco, 2: no unsup')
    p.add_argument('--mask', type=int, default=0, help='0: no mask, 1: use pre_u')
    p.add_argument('--func', type=str, default='smc', help='asmc or smc')
    p.add_argument('--ref_net', type=str, default='vgg19', help='ref_net')
    p.add_argument('--ref_norm', type=bool, default=False, help='ref_norm')
    p.add_argument('--ref_layer1', type=str, default='relu3_2', help='ref_layer1')
    p.add_argument('--ref_layer2', type=str, default='relu5_4', help='ref_layer2')
    p.add_argument('--ref_weight1', type=float, default=0.33, help='ref_weight1')
    p.add_argument('--ref_weight2', type=float, default=1.0, help='ref_weight2')
    p.add_argument('--temperature', type=float, default=.19, help='temperature')
    p.add_argument('--layer_len', type=int, default=-1, help='layer_len')
    p.add_argument('--tps_sigma', type=float, default=0.01, help='tps_sigma')
    # additions of this implementation (not in the reference)
    p.add_argument('--queue_size', type=int, default=0, help='override per-class bank size (0 = reference 50000/30000)')
    p.add_argument('--synthetic', type=int, default=0, help='1: train on synthetic ACDC-shaped tensors')
    p.add_argument('--in_chns', type=int, default=1, help='input channels (reference: 1)')
    p.add_argument('--graphs', type=int, default=1, help='1: replay the no-grad U-Net forwards as HIP graphs')
    p.add_argument('--batched_passes', type=int, default=1,
                   help='1: labelled + unlabelled student (and teacher) forwards as one pass with two BatchNorm groups')
    p.add_argument('--graph_train', type=int, default=1,
                   help='1 (with --graphs 1): the student forward+backward passes are HIP graphs too (a third of the '
                        'host launch work per step, results identical); 0: eager launches')
    p.add_argument('--dense_teacher', type=int, default=0, help='1: materialise the dense teacher representation')
    p.add_argument('--teacher_levels', type=int, default=2,
                   help='lazy teacher depth: 2 = dense fea2 output, prototypes pushed through two bilinear adjoints (measured as '
                        'fast as 3); 3 = no dense fea2 output either')
    p.add_argument('--head_levels', type=int, default=3, help='row-sparse head depth: 1 = from the 128x128 level, 2 = from 64x64, 3 = from 32x32')
    p.add_argument('--dense_head', type=int, default=0, help='1: materialise the dense 496-ch student rep (reference dataflow)')
    p.add_argument('--list_dir', type=str, default='', help='list directory of the npz experiments (default: the reference\'s hard-wired paths)')
    p.add_argument('--dp_local_thresholds', type=int, default=0,
                   help='data parallel only. 0: entropy percentiles of the global batch (5 small all-reduces per step); '
                        '1: every rank thresholds its own batch')
    p.add_argument('--batch_transform', type=int, default=1,
                   help='1 (reference behaviour): batch_transform of train_arco_2d.py:287-304 - 8-bit PIL round trip of the '
                        'images / confidences, ColorJitter + GaussianBlur + AdvMorph on the unlabeled stream - on the GPU; '
                        '0: the identity (no generator draws)')
    p.add_argument('--conv_mma', type=str, default='f32x3', choices=['f32x3', 'f32'],
                   help='matrix-core mode of the convolutions / GEMMs (forward and data gradient).  f32x3 (default): fp32-accurate '
                        'products on the bf16 matrix cores - each fp32 operand is split exactly into three bf16 terms, six bf16 '
                        'MFMAs per 32 k instead of eight fp32 MFMAs per 16 k (2.7x fewer matrix-core cycles, error per product '
                        '<= 2^-23); f32: the native fp32 MFMA (bitwise an fma chain).  The 3-D trainer '
                        'adds f16 / bf16: operands of the 3x3x3 convolutions rounded to f16 / bf16, fp32 accumulate (BASELINE '
                        'configs[4], tolerance 1e-2) - the 2-D kernels have no such mode and this parser rejects it')
    p.add_argument('--anchors_per_rank', type=str, default='split', choices=['split', 'full'],
                   help='data parallel only (SURVEY 8e). split: every rank samples num_queries/world anchors per class, so the '
                        'world draws the same total number of queries as the single-process reference; full: num_queries per rank')
    p.add_argument('--revisit', type=int, default=0,
                   help='1: also compute k4*loss_q, the revisiting loss (train_arco_2d.py:126-136,334,398-400). It has no '
                        'gradient path (it only changes the logged loss) and needs the dense student and teacher '
                        'representations, so it switches --dense_head and --dense_teacher on; --K must be a multiple of '
                        '--batch_size, as in the reference')
    return p


def patients_to_slices(dataset, patiens_num):
    """train_arco_2d.py:91-106."""
    ref_dict = None
    if "ACDC" in dataset:
        ref_dict = {"1": 23, "3": 68, "7": 136, "14": 256, "21": 396, "28": 512, "35": 664, "140": 1312}
    elif "MM" in dataset:
        ref_dict = {"1": 38, "2": 76, "5": 191, "10": 382}
    elif "Syn" in dataset or "syn" in dataset:
        ref_dict = {"1": 44, "3": 66, "5": 111, "10": 221}
    elif "Lits" in dataset or "LiTS" in dataset:
        ref_dict = {"1": 167, "5": 835, "10": 1668, "20": 3336, "50": 8340}
    elif "jhu" in dataset or "JHU" in dataset:
        ref_dict = {"1": 57, "5": 275, "10": 568, "100": 5675}
    else:
        print("Error")
    return ref_dict[str(patiens_num)]


# Pass-level concurrency on a second stream (round 4; profiles/r04_notes.md section 7).  The step is a dependent chain of ~870 kernels,
# most of them 5-40 us: a kernel rarely fills the chip for its whole duration, and one launch floor (~5 us) per link adds up to a
# third of the step.  Running two INDEPENDENT passes side by side fills those gaps - unlike kernel-level forks inside one pass
# (ops._wgrad), which made the persistent kernels of the same pass fight for CUs.
#   1: the teacher's grouped pass beside the student forward                                  13.30 -> 13.16 ms
#   2: + the statistics-only student pass (cj2_l) beside the masks / heads work               -> 12.87 ms
#   3: + the warped student pass (equivariance term) on the side stream: its forward beside the heads / InfoNCE, its BACKWARD
#      beside the main pass's backward, parameter gradients into a second flat buffer          -> 11.37 ms
#   4: + the step's critical path shortened (tools/step_timeline.py): the statistics pass moves to the step's start, beside the
#      teacher's first pass (its running-statistics updates postponed into slot 1 of ops.bn_defer, so they still land between l
#      and u); warps + warped pass start on the side stream as soon as the host has drawn the warp, the bank appends are
#      queued after them                                                  same process, alternating: 11.64 -> 11.27 ms  (default)
# Results are unchanged: the passes were independent already, only their order in time is free; the two gradient buffers are
# summed once (a + b, bit-identical to accumulating in sequence).  ARCO_TEACHER_SIDE=0 restores the single-stream step.
TEACHER_SIDE = int(os.environ.get("ARCO_TEACHER_SIDE", "4"))
# SIDE_SYNC (round 4's host-side wait in front of loss.backward()) is OFF since round 5: the non-reproducible gradient it hid was a
# gfx950 erratum in ONE compiler-generated instruction of arco_lerp4_cat_rows_bwd, fixed in the kernel (csrc/elementwise.hip,
# tests/test_isa_lint.py, profiles/r05_notes.md section 1).  ARCO_SIDE_SYNC=1 restores the wait (A/B only).
SIDE_SYNC = int(os.environ.get("ARCO_SIDE_SYNC", "0"))
IMG_EARLY = int(os.environ.get("ARCO_IMG_EARLY", "1"))     # see ArcoStep2D.step (with TEACHER_SIDE >= 4)


class ArcoStep2D:
    """State + one training step of the 2-D hot path (train_arco_2d.py:147-154,220-253,284-435)."""

    def __init__(self, args, device="cuda"):
        self.args = args
        self.dev = torch.device(device)
        C = args.num_classes
        mma = getattr(args, "conv_mma", "f32x3")
        if mma not in ("f32", "f32x3"):      # no silent fp32 run under a reduced-precision label (the modes exist in 3-D only)
            raise ValueError(f"--conv_mma {mma}: the 2-D step computes in f32x3 or f32; f16 / bf16 operands are a 3-D trainer mode")
        ops.CONV_MMA = {"f32": 0, "f32x3": 3}[mma]
        ops.HEAD_MMA = 0                 # (a 3-D trainer of this process may have set the heads' reduced-precision mode)
        if ops._WGRAD_SIDE_ENV is None:
            ops.WGRAD_SIDE = 0           # (... and the side-stream weight gradients: 11.0-11.2 -> 12.2-12.3 ms on this step)
        self.random_pool = None
        if getattr(args, "revisit", 0):
            # random_pool (:156-159) is drawn before the models are created, like the reference (same CPU-generator order)
            args.dense_head = args.dense_teacher = 1
            assert args.K % args.batch_size == 0, "--K must be a multiple of --batch_size (train_arco_2d.py:113)"
            self.random_pool = glue.RevisitPool(args.K, 256 + 128 + 64 + 32 + 16, args.patch_size, self.dev)
        else:
            # --revisit 0 (default): the pool's 4.7 GB of normals are not needed, but their place in the CPU-generator
            # sequence is - the weight initialisation below, the samplers and the warps then draw what the reference draws
            # for the same seed (samplers.skip_randn: 0.4 s of state regeneration for K = 36 at 256 x 256)
            from . import samplers
            samplers.skip_randn(args.K * REP_DIM * int(args.patch_size[0]) * int(args.patch_size[1]))
        # memory banks (train_arco_2d.py:147-154); device resident from the first enqueue on
        self.memobank, self.queue_ptrlis, self.queue_size = [], [], []
        for i in range(C):
            self.memobank.append([torch.zeros(1, REP_DIM)])
            self.queue_size.append(args.queue_size if args.queue_size > 0 else 30000)
            self.queue_ptrlis.append(torch.zeros(1, dtype=torch.long))
        if args.queue_size <= 0:
            self.queue_size[0] = 50000
        self.isd = ISD(K=args.K, m=0.99, Ts=0.01, Tt=0.1, num_classes=C,
                       latent_pooling_size=args.latent_pooling_size, latent_feature_size=args.latent_feature_size,
                       output_pooling_size=args.output_pooling_size, train_encoder=True, train_decoder=True,
                       in_chns=args.in_chns)
        self.isd = self.isd.to(self.dev)
        self.model, self.ema_model = self.isd.model, self.isd.ema_model
        self.q_representation = nn.Sequential(nn.Conv2d(REP_DIM, REP_DIM, kernel_size=1, bias=False),
                                              nn.Conv2d(REP_DIM, REP_DIM, kernel_size=1, bias=False)).to(self.dev)
        self.k_feature_extractor = FeatureExtractor(fea_dim=FEA_DIM, output_dim=REP_DIM).to(self.dev)
        self.q_feature_extractor = FeatureExtractor(fea_dim=FEA_DIM, output_dim=REP_DIM).to(self.dev)
        adist.broadcast_module_states([self.isd, self.q_representation, self.q_feature_extractor,
                                       self.k_feature_extractor])
        params = [p for p in self.model.parameters() if p.requires_grad]
        params_rep = [p for p in self.q_representation.parameters() if p.requires_grad]
        params_fea = [p for p in self.q_feature_extractor.parameters() if p.requires_grad]
        self.heads_start = sum(p.numel() for p in params)     # flat_g[heads_start:] = the heads' gradient bucket (dist.mark_heads_done)
        self.optimizer = optim.SGDNesterov(params + params_rep + params_fea, lr=args.base_lr, weight_decay=0.0001,
                                           momentum=0.9, nesterov=True)
        with torch.no_grad():                                            # :250-253
            for t, s in zip(self.k_feature_extractor.parameters(), self.q_feature_extractor.parameters()):
                t.data.copy_(s.data)
                t.requires_grad = False
        self.k_fe_ema = optim.EmaPair(self.q_feature_extractor, self.k_feature_extractor)
        for m in (self.model, self.ema_model, self.q_representation, self.k_feature_extractor,
                  self.q_feature_extractor):
            m.train()                                                   # :263-267
        # packed conv weights: one launch per weight owner per step (ops.PackPlan), refreshed by the owner
        plan_s = ops.PackPlan([self.model, self.q_representation, self.q_feature_extractor], True)
        self.optimizer.plans = [plan_s]
        pairs = self.isd._ensure_ema_pairs()
        pairs[0].plans = [ops.PackPlan([self.ema_model], False)]
        for pr in pairs[1:]:
            pr.plans = [ops.PackPlan([], False)]
        self.k_fe_ema.plans = [ops.PackPlan([self.k_feature_extractor], False)]
        self.plans = [plan_s] + [pl for pr in pairs for pl in pr.plans] + self.k_fe_ema.plans
        self.iter_num = 0
        # HIP-event timing of the three contrastive-loss segments: only when a profiler asks for it (bench.py sets
        # profile_loss); a training run records no events (no stream bubbles, nothing accumulates)
        self.profile_loss = False
        self._t_stream = None
        self._stats_on_side = False
        self.keep_debug = False          # tests: keep the last step's plan and anchor rows (self.debug)
        self.loss_events = []
        # no-grad forwards replayed as HIP graphs (one graph per call site: outputs are static buffers)
        use_graphs = bool(getattr(args, "graphs", 1))
        g_train = use_graphs and bool(getattr(args, "graph_train", 0))
        self.s_train_u = graphs.GraphedTrain(self.model, enabled=g_train)    # student passes: fwd + bwd graphs
        self.s_train_l = graphs.GraphedTrain(self.model, enabled=g_train)
        # RandTPS of the equivariance term: built here like the reference (:255-261; the constructor draws one warp
        # from the generators), rebuilt in step() only if the batch size differs
        self.tps = None
        if getattr(args, "k2", 0) != 0:
            self.tps = RandTPS(args.patch_size[0], args.patch_size[1], batch_size=2 * args.batch_size,
                               sigma=args.tps_sigma, border_padding=False, random_mirror=True, random_scale=(0.8, 1.2),
                               mode='affine', device=device)
        self.batched_passes = bool(getattr(args, "batched_passes", 1))
        self.s_train_lu = graphs.GraphedTrain(self.model, enabled=g_train)
        # the equivariance term's student pass (:415).  ARCO_TEACHER_SIDE >= 3: replayed on the side stream - its forward beside the
        # heads / InfoNCE, its backward beside the main pass's backward - with a gradient buffer of its own (optim.second_grad_views)
        self._tps_side = TEACHER_SIDE >= 3 and g_train
        self.s_train_tps = graphs.GraphedTrain(self.model, enabled=g_train,
                                               grad_views=self.optimizer.second_grad_views() if self._tps_side else None)
        self.t_fwd_lu = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_u0 = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_l = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.t_fwd_u = graphs.GraphedForward(self.ema_model, enabled=use_graphs)
        self.s_fwd_stats = graphs.GraphedForward(self.model, enabled=use_graphs)

    def q_rep(self, x):
        x = ops.conv(x, self.q_representation[0].weight)
        return ops.conv(x, self.q_representation[1].weight)

    def _teacher_heads(self, fm_t, dense):
        """(dense teacher representation or None, lazy teacher or None): the teacher's FeatureExtractor (:321-322) in the form the
        loss front end consumes - rows are only needed as class means (prototypes) and <= queue_size keys per class."""
        a, kfe = self.args, self.k_feature_extractor
        if getattr(a, "dense_teacher", 0) or dense:
            return kfe(fm_t), None
        if getattr(a, "head_levels", 3) == 1:
            x3p_t, f4_t = kfe.forward_lowres(fm_t)
            return None, head.LazyTeacher2D(x3p_t, f4_t, kfe.fea4.weight)
        if getattr(a, "teacher_levels", 2) == 3:
            return None, head.LazyTeacher2DL3(*kfe.forward_lowres1(fm_t), kfe.fea2.weight, kfe.fea3.weight, kfe.fea4.weight)
        return None, head.LazyTeacher2DL2(*kfe.forward_lowres2(fm_t), kfe.fea3.weight, kfe.fea4.weight)

    def step(self, l_data, l_label, u_data, epoch_num=0, max_epoch=1):
        """One iteration.  The mixing strategy of --apply_aug (augment.generate_unsup_data) and batch_transform (8-bit PIL
        round trip, ColorJitter, GaussianBlur, AdvMorph: augment.batch_transform) run on the GPU with the reference's host draws.

        Same operations and results as train_arco_2d.py:284-435 restricted to the hot-path loss term;
        the ORDER is arranged for the GPU: everything the host sampler needs (3*C counters) is
        produced first and copied asynchronously, the large teacher/student GEMMs are queued behind
        it, and the torch-CPU-generator index replay runs on the host while they execute."""
        a = self.args
        C = a.num_classes
        for pl in self.plans:                                            # stale only if someone else touched weights
            if not pl.valid:
                pl.refresh()
        dense = getattr(a, "dense_head", 0)
        # randomGeneratorWithLogits (:292-293) is a same-size zoom(order=0) = the identity; then the mixing strategy
        # (:296-297) on the GPU with the reference's host draws; other --apply_aug values leave the batch unchanged
        bt = bool(getattr(a, "batch_transform", 1))
        cj2_l = l_data
        if bt:      # :287-290: two calls without augmentation - images_cj1_logits_l (the constant 255 -> 1.0) and images_cj2_l
            # (queued in front of the teacher's pass below, which consumes nothing of the torch CPU generator: the host draws keep
            # the reference's order)
            augment.draw_batch_transform_params(int(l_data.shape[0]), False)         # (only its generator draws matter)
            cj2_l, _, _ = augment.batch_transform(l_data, l_label, torch.ones_like(l_label, dtype=torch.float32), a.patch_size,
                                                  (1.0, 1.0), False)
        stats_early = TEACHER_SIDE >= 4 and self.batched_passes and l_data.shape == u_data.shape
        if stats_early:
            # Mode 4: the statistics-only student pass on images_cj2_l (:311) depends on nothing but the weights - it runs on the
            # side stream beside the teacher's first pass (two 8-image passes, neither of which fills the chip alone).  Its
            # running-statistics updates are postponed (slot 1) and land between those of the l and the u half of the grouped
            # student pass below: the reference's order l, cj2_l, u.
            if self._t_stream is None:
                self._t_stream = torch.cuda.Stream()
            self._t_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._t_stream), torch.no_grad(), ops.bn_defer(0, 1):
                self.s_fwd_stats(cj2_l)
        # IMG_EARLY (cutout / cutmix): the IMAGE side of the mixing strategy and of the two batch_transform calls (:296-304) reads no
        # pseudo-label - the boxes, ColorJitter / blur parameters and AdvMorph fields are host / generator draws, labels pass through
        # batch_transform unchanged and the confidences are only quantised (augment.batch_transform) - so it is queued on a third stream
        # NOW, beside the teacher's first pass and the statistics pass, instead of alone between them and the grouped passes (0.4 ms of
        # small launches).  Same host draws in the same order (the teacher's pass draws nothing); labels and confidences are mixed with
        # the same boxes once the pseudo-labels exist.
        img_early = bool(IMG_EARLY and stats_early and a.apply_aug in ("cutout", "cutmix"))
        if img_early:
            if getattr(self, "_img_stream", None) is None:
                self._img_stream = torch.cuda.Stream()
            mix_desc = augment.draw_boxes(int(u_data.shape[0]), tuple(int(v) for v in u_data.shape[2:]))
            self._img_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._img_stream):
                u_mix = augment.mix_images(u_data, a.apply_aug, mix_desc)
                cj2_u = u_aug = u_mix
                if bt:
                    zl = torch.zeros(u_mix.shape[:1] + u_mix.shape[2:], dtype=torch.int64, device=u_mix.device)
                    zg = torch.zeros(u_mix.shape[:1] + u_mix.shape[2:], dtype=torch.float32, device=u_mix.device)
                    cj2_u = augment.batch_transform(u_mix, zl, zg, a.patch_size, (1.0, 1.0), True)[0]
                    u_aug = augment.batch_transform(u_mix, zl, zg, a.patch_size, (1.0, 1.0), True)[0]
        with torch.no_grad():                                            # :284-286
            pred_u0, _, _ = self.t_fwd_u0(u_data)
            pseudo_logits, pseudo_labels = glue.softmax_max(pred_u0)
            if self.keep_debug:      # tests: the teacher's decisions before the mixing (cutout writes -1 into the labels in place)
                dbg_pseudo = (pseudo_labels.clone(), pseudo_logits.clone())
        if img_early:
            torch.cuda.current_stream().wait_stream(self._img_stream)
            _, u_aug_label, u_aug_logits = augment.generate_unsup_data(u_data, pseudo_labels, pseudo_logits, mode=a.apply_aug, desc=mix_desc)
            if bt:      # batch_transform's 8-bit round trip of the confidences
                lg = u_aug_logits.to(torch.float32).contiguous()
                u_aug_logits = torch.empty_like(lg)
                ops.L.call("arco_quantize8", ops.L.ptr(lg), lg.numel(), ops.L.ptr(u_aug_logits))
        else:
            u_aug, u_aug_label, u_aug_logits = augment.generate_unsup_data(u_data, pseudo_labels, pseudo_logits, mode=a.apply_aug)
            cj2_u = u_aug
            if bt:      # :299-304: two independent strong augmentations of the mixed unlabeled batch
                cj2_u, _, _ = augment.batch_transform(u_aug, u_aug_label, u_aug_logits, a.patch_size, (1.0, 1.0), True)
                u_aug, u_aug_label, u_aug_logits = augment.batch_transform(u_aug, u_aug_label, u_aug_logits, a.patch_size, (1.0, 1.0), True)
        self.k_fe_ema.update(0.99)                                      # :306-308
        batched = self.batched_passes and l_data.shape == u_aug.shape
        if stats_early:           # (the student's BatchNorm buffers - num_batches_tracked - belong to the passes below from here on)
            torch.cuda.current_stream().wait_stream(self._t_stream)
        if batched:
            # the labelled and the unlabelled student forward (:310,312) as ONE batch-2b pass with two BatchNorm
            # groups (ops.bn_groups: per-half batch statistics, running statistics updated half after half) - same
            # results as two passes, every conv / BN / pooling launch works on twice the pixels (the mid and deep
            # U-Net levels are too small to fill the GPU at b images: 20-35 % less kernel time), one backward.
            lu = torch.cat((l_data, u_aug))
            # The u half postpones its running-statistics update (ops.bn_defer) until the images_cj2_l pass below has
            # made its own: the momentum updates then land in the reference's order l, cj2_l, u (:310-312).
            t_side = None
            if TEACHER_SIDE:      # the teacher's grouped pass (independent of the student's) on a second stream, beside the student forward
                if self._t_stream is None:
                    self._t_stream = torch.cuda.Stream()
                t_side = self._t_stream
                t_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(t_side), torch.no_grad(), ops.bn_groups(2):
                    pred_t, _, fm_t = self.t_fwd_lu(lu)
            with ops.bn_groups(2), ops.bn_defer(1):
                pred_all, _, fm_all = self.s_train_lu(lu)
            if t_side is not None:
                torch.cuda.current_stream().wait_stream(t_side)
            nb_l = int(l_data.shape[0])
            pred_l, pred_u = ops.split_batch(pred_all, nb_l)    # (views; one gradient buffer for both halves in the backward)
        else:
            with ops.bn_defer(0):                                        # running statistics: applied after l and cj2_l
                pred_u, _, u_fm = self.s_train_u(u_aug)                  # :312 (needed first: entropy masks)
        with torch.no_grad():                                            # teacher params carry no grad (:158-160)
            if batched:
                if not TEACHER_SIDE:
                    with ops.bn_groups(2):
                        pred_t, _, fm_t = self.t_fwd_lu(lu)              # :314-315 as one grouped pass
                pred_l_t, pred_u_t = pred_t[:nb_l], pred_t[nb_l:]
            else:
                pred_l_t, _, l_fm_t = self.t_fwd_l(l_data)               # :314
                pred_u_t, _, u_fm_t = self.t_fwd_u(u_aug)                # :315
            alpha_t = 20 * (1 - epoch_num / max_epoch)                   # :342-393
            label_l = glue.label_onehot(l_label, C)
            label_u = glue.label_onehot(u_aug_label, C)
            prob_l_t = glue.softmax(pred_l_t)
            prob_u_t = glue.softmax(pred_u_t)
            low_mask_all, high_mask_all = glue.entropy_masks(pred_u, l_label, u_aug_label, alpha_t)
            if self.keep_debug:      # the step's gradient-free decision inputs (oracle/cpu_step.py `force=`)
                self.decisions = dict(pseudo_labels=dbg_pseudo[0], pseudo_logits=dbg_pseudo[1], low=low_mask_all, high=high_mask_all,
                                      prob_l_t=prob_l_t, prob_u_t=prob_u_t)
        prof = self.profile_loss
        ev = ev2 = ev3 = None
        if prof:
            ev, ev2, ev3 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
            ev[0].record()
        plan = C_.contrast_masks(label_l, label_u, prob_l_t, prob_u_t, low_mask_all, high_mask_all,
                                 delta_n=a.strong_threshold_u2pl)       # :341-401 (counts -> async D2H)
        if prof:
            ev[1].record()
        # ---- large GPU work queued while the host waits for the counters and samples
        if not batched:
            pred_l, _, l_fm = self.s_train_l(l_data)                     # :310
        with torch.no_grad():
            # images_cj2_l forward (:311): BN running statistics only - its FE/q_rep outputs (l_feature_map_2,
            # :319,326) are never read.  One graph launch (~1 ms of GPU work) queued BEFORE the host sync: work for
            # the GPU while the host replays the samplers (14.8 vs 15.1 ms/step when queued after the sync).
            if stats_early:
                ops.apply_deferred_bn(1)                                 # cj2_l (ran beside the teacher's first pass), then
                ops.apply_deferred_bn()                                  # the u half / pass
            elif TEACHER_SIDE >= 2 and batched:    # (mode 2: this statistics-only pass too runs beside the main stream's work)
                if self._t_stream is None:
                    self._t_stream = torch.cuda.Stream()
                self._t_stream.wait_stream(torch.cuda.current_stream())     # behind the student pass: BN buffers in the reference's order
                with torch.cuda.stream(self._t_stream):
                    self.s_fwd_stats(cj2_l)
                    ops.apply_deferred_bn()
                self._stats_on_side = True
            else:
                self.s_fwd_stats(cj2_l)
                ops.apply_deferred_bn()                                  # the u pass's running-statistics update (:312)
            # FeatureExtractor is per-image -> run it once on the batch-concatenated maps (:321-322)
            if not batched:
                fm_t = [torch.cat((x, y)) for x, y in zip(l_fm_t, u_fm_t)]
            rep_all_teacher, lazy_t = self._teacher_heads(fm_t, dense)
        if not batched:
            fm_all = [torch.cat((x, y)) for x, y in zip(l_fm, u_fm)]     # :317-318
        # data parallel: the heads' gradient bucket is all-reduced as soon as the heads' backward is done, under the U-Net's
        fm_all = adist.mark_heads_done(fm_all, self.optimizer, self.heads_start)
        if dense:
            rep_all = self.q_rep(self.q_feature_extractor(fm_all))       # :324-325,330
        elif getattr(a, "head_levels", 3) == 1:
            x3p, f4 = self.q_feature_extractor.forward_lowres(fm_all)
        elif getattr(a, "head_levels", 3) == 3:
            x1p, f2, f3, f4 = self.q_feature_extractor.forward_lowres1(fm_all)
        else:
            x2p, f3, f4 = self.q_feature_extractor.forward_lowres2(fm_all)
        # supervised CE + Dice and confidence-weighted unsupervised CE (:336-340): they need neither the
        # counters nor the samples - queued before the host blocks.  The equivariance term follows the sampler draw below.
        # k4*loss_q (revisiting loss; no gradient path to any parameter) only with --revisit 1.
        loss_ce, loss_dice = glue.supervised_loss(pred_l, l_label)
        unsup_loss = glue.compute_unsupervised_loss(pred_u, u_aug_label, u_aug_logits, a.strong_threshold)
        eqv_in = None
        if a.k2 != 0:
            # inputs of the equivariance term that depend neither on the counters nor on the warp: queued BEFORE the host
            # blocks on the counters (the stretch between the counters' arrival and the warped student pass is the part
            # of the step where the GPU can run dry: tools/draw_critical.py)
            with torch.no_grad():
                labels_all = torch.cat((l_label, u_aug_label))
                # images_cj1_logits_l (:287-288) is the constant 255 pushed through ToTensor = 1.0 everywhere
                logits_all = torch.cat((u_aug_logits.new_ones(l_label.shape), u_aug_logits))
                eqv_in = (glue.eqv_mask(labels_all, logits_all, a.weak_threshold), torch.cat((cj2_l, cj2_u)),
                          pred_all.detach() if batched else torch.cat((pred_l.detach(), pred_u.detach())))
            if TEACHER_SIDE >= 4:          # mode 4: the warps and the warped pass start from here on the side stream
                self._eqv_in_ready = torch.cuda.Event()
                self._eqv_in_ready.record()
        # per-class row lists and prototypes need the class codes / totals on the DEVICE only: queued before the host blocks
        evp = None
        if prof:
            evp = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            evp[0].record()
        C_.contrast_lists_protos(plan, rep_all_teacher, lazy_t)
        if prof:
            evp[1].record()
        # ---- host: wait for the counters; everything that needs the COUNTS but not the sampled INDICES is queued
        #      first (row lists, prototypes, key rows, bank append, the supervised / unsupervised loss forwards), so the
        #      GPU has work while the host replays the samplers (bit-exact torch-CPU-generator sequence, ~2 ms)
        C_.contrast_counts(plan, self.memobank, self.queue_size,
                           adist.anchors_for_rank(a.num_queries, getattr(a, "anchors_per_rank", "split")), a.num_negatives)
        def enqueue():                     # key rows of the teacher -> banks (no generator draws, no use of the sampled indices)
            if prof:
                ev2[0].record()
            C_.contrast_enqueue(plan, rep_all_teacher, self.memobank, self.queue_ptrlis, self.queue_size,
                                lazy_teacher=lazy_t, defer_anchor_pix=True)
            if prof:
                ev2[1].record()
        enqueue_late = TEACHER_SIDE >= 4 and a.k2 != 0    # mode 4: the warped pass is on the step's critical path - queue it first
        if not enqueue_late:
            enqueue()
        C_.contrast_draw(plan, a.func, defer=True)     # indices collected by contrast_anchor_pix below
        tps_on_side = tps_early = False
        loss_eqv = None
        if a.k2 != 0:
            # equivariance term (:404-423).  The warp is drawn AFTER the samplers, as in the reference: both consume
            # the torch CPU generator, and the sampled indices must not depend on whether this term is on.
            nb2 = int(l_data.shape[0]) + int(u_aug.shape[0])
            if self.tps is None or self.tps.batch_size != nb2:
                self.tps = RandTPS(a.patch_size[0], a.patch_size[1], batch_size=nb2, sigma=a.tps_sigma,
                                   border_padding=False, random_mirror=True, random_scale=(0.8, 1.2), mode='affine',
                                   device=l_data.device)                 # :255-261 (draws one warp, like the reference)
            eq_mask, images_cj2, pred_all_d = eqv_in
            tps_on_side = self._tps_side and self.s_train_tps.will_replay(images_cj2)
            tps_early = tps_on_side and TEACHER_SIDE >= 4
            if tps_on_side and self._t_stream is None:
                self._t_stream = torch.cuda.Stream()
            if tps_early:
                # Mode 4: the warps and the warped pass do not queue behind the main stream's row lists / bank appends / index
                # uploads: they run on the side stream from the moment the host has drawn the warp, behind `eqv_in` on the main
                # stream (which is behind the grouped pass and the running-statistics updates above).
                self._t_stream.wait_event(self._eqv_in_ready)
            with torch.cuda.stream(self._t_stream) if tps_early else contextlib.nullcontext():
                with torch.no_grad():
                    self.tps.reset_control_points()                      # :412
                    images_tps = self.tps(images_cj2)                    # :411-413 images_cj2
                    mask_tps = self.tps(eq_mask, padding_mode='zeros')
                    pred_tps_org = self.tps(pred_all_d, padding_mode='zeros')
                if tps_early:
                    pred_tps = self.s_train_tps(images_tps)[0]           # :415
            if tps_early:
                self.optimizer._g2_dirty = True
            elif tps_on_side:      # behind the statistics-only pass on that stream (running statistics: cj2_l, u, then this pass)
                self._t_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._t_stream):
                    pred_tps = self.s_train_tps(images_tps)[0]
                self._stats_on_side = False
                self.optimizer._g2_dirty = True
            else:
                if self._stats_on_side:    # the warped pass updates the same running statistics next
                    torch.cuda.current_stream().wait_stream(self._t_stream)
                    self._stats_on_side = False
                pred_tps = self.s_train_tps(images_tps)[0]               # :415 one more student pass (one BN batch)
                loss_eqv = glue.eqv_loss(pred_tps, pred_tps_org, mask_tps)   # :419-423
        if enqueue_late:
            enqueue()
        if self._stats_on_side:            # (no equivariance pass: the optimiser must not change the weights under the statistics pass)
            torch.cuda.current_stream().wait_stream(self._t_stream)
            self._stats_on_side = False
        if prof:
            ev3[0].record()
        C_.contrast_anchor_pix(plan)
        zero_path = plan.valid_seg <= 1 or not plan.entries
        if zero_path:
            reco_loss = self.q_representation[1].weight.sum() * 0.0      # :417-424 zero attached to the graph
        else:
            if dense:
                A_all = C_.GatherRowsFn.apply(rep_all, plan.anchor_pix)
            elif getattr(a, "head_levels", 3) == 1:
                A_all = head.lazy_head(x3p, f4, self.q_feature_extractor.fea4.weight,
                                       self.q_representation[0].weight, self.q_representation[1].weight,
                                       plan.anchor_pix)
            elif getattr(a, "head_levels", 3) == 3:
                qfe = self.q_feature_extractor
                A_all = head.lazy_head3(x1p, f2, f3, f4, qfe.fea2.weight, qfe.fea3.weight, qfe.fea4.weight,
                                        self.q_representation[0].weight, self.q_representation[1].weight, plan.anchor_pix)
            else:
                A_all = head.lazy_head2(x2p, f3, f4, self.q_feature_extractor.fea3.weight,
                                        self.q_feature_extractor.fea4.weight, self.q_representation[0].weight,
                                        self.q_representation[1].weight, plan.anchor_pix)
            reco_loss, _ = C_.contrast_infonce(plan, A_all, self.memobank, temp=0.5)   # :394-398 (temp default)
            if self.keep_debug:
                self.debug = dict(plan=plan, A_all=A_all.detach(), banks=[m[0] for m in self.memobank])
        if prof:
            ev3[1].record()
            self.loss_events.append((ev, ev2, ev3, evp))  # masks | keys, banks | anchors, head, InfoNCE | lists, prototypes
        if tps_on_side:                    # the heads and the InfoNCE above ran beside the warped pass's forward
            torch.cuda.current_stream().wait_stream(self._t_stream)
            loss_eqv = glue.eqv_loss(pred_tps, pred_tps_org, mask_tps)   # :419-423
        # :426 - one launch for the weighted sum (and one for its backward) instead of a chain of 0-d multiplies and adds
        ws = [a.k1 * adist.anchor_weight(a.num_queries, getattr(a, "anchors_per_rank", "split")), a.k3, 1.0, 1.0]
        terms = [reco_loss, unsup_loss, loss_dice, loss_ce]
        if loss_eqv is not None:
            ws.append(a.k2); terms.append(loss_eqv)
        loss_q = None
        if self.random_pool is not None:
            # :334 (computed before the pool is updated) and :398-400; a constant w.r.t. every parameter
            nb_l = int(l_data.shape[0])
            loss_q = glue.get_revisiting_loss(self.random_pool, rep_all[nb_l:], rep_all_teacher[nb_l:], topk=a.topk)
            glue.revisit_enqueue(rep_all_teacher[nb_l:], self.random_pool)
            ws.append(a.k4); terms.append(loss_q)
        loss = ops.combine_terms(ws, terms)
        self.optimizer.zero_grad()                                       # :429-431
        if tps_on_side and SIDE_SYNC:
            # Round 4 shipped this host-side wait as a workaround: without it 1-3 % of steps had a gradient off by 1e-3..1e-2.  Round 5
            # found the cause - with the host ahead of the GPU the row-sparse head's backward runs BESIDE the warped pass's backward
            # graph (the overlap this schedule wants), and one packed-fp32 instruction the compiler had put into
            # lerp4_cat_rows_bwd_kernel (`v_pk_mul_f32 ... op_sel:[0,1]`, src0 != src1) returns a wrong low half in lanes 48-63 while
            # another wave of the SIMD executes a K-doubled 16x16 MFMA (gfx950 erratum, torch-free reproducer tools/debug/pkmul_repro.hip).
            # The wait only removed the overlap at that point.  The kernel no longer contains the form; 0 of 1200 amplified trials
            # without the wait against 43 of 400 with the old code object on the same box (profiles/r05_notes.md section 1).
            self._t_stream.synchronize()
        loss.backward()
        ops.join_side()                     # weight gradients queued on the side stream (ops._wgrad)
        if tps_on_side:
            # the warped pass's backward graph was replayed on the side stream and hands nothing back to autograd (its parameter
            # gradients go straight into the second buffer), so the engine has no leaf stream to synchronise with at the end of
            # backward(): wait for it here.  (Found as a 1-in-3 flake of test_cfg2_graph_replay_equals_eager_at_full_size: the LAST
            # gradient of that backward, the first layer's weight, was merged before it was complete.)
            torch.cuda.current_stream().wait_stream(self._t_stream)
        self.optimizer.merge_second(self.heads_start)      # the warped pass's parameter gradients (replayed on the side stream)
        if zero_path:     # `0 * rep.sum()` gives EVERY head parameter a zero gradient: SGD still decays / applies momentum to them
            self.optimizer.touch_from(self.heads_start)
        adist.allreduce_grads(self.optimizer)
        self.optimizer.step()
        self.isd._momentum_update_key_encoder()                          # :432
        lr_ = a.base_lr * (1.0 - self.iter_num / a.max_iterations) ** 0.9   # :433-435
        for g in self.optimizer.param_groups:
            g['lr'] = lr_
        self.iter_num += 1
        # values only: nothing returned or kept may hold this step's autograd graph alive into the next step
        # (graphs.GraphedTrain needs the parameters' gradient accumulators recreated on its capture stream)
        self.last_terms = dict(ce=loss_ce.detach(), dice=loss_dice.detach(), unsup=unsup_loss.detach(),
                               reco=reco_loss.detach())
        if loss_eqv is not None:
            self.last_terms["eqv"] = loss_eqv.detach()
        if loss_q is not None:
            self.last_terms["loss_q"] = loss_q
        return loss.detach(), reco_loss.detach()


def synthetic_batch(b, patch, n_cls, seed, device, in_chns=1):
    """ACDC-shaped synthetic batch: images U[0,1), blob labels with ACDC-like imbalance."""
    rs = np.random.RandomState(seed)
    img = torch.from_numpy(rs.uniform(size=(b, in_chns, *patch)).astype(np.float32))
    lab = np.zeros((b, *patch), dtype=np.int64)
    yy, xx = np.mgrid[0:patch[0], 0:patch[1]]
    for i in range(b):
        for c in range(1, n_cls):
            cy, cx = rs.randint(patch[0] // 4, 3 * patch[0] // 4), rs.randint(patch[1] // 4, 3 * patch[1] // 4)
            r = rs.randint(patch[0] // 16, patch[0] // 6)
            lab[i][(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = c
    return img.to(device), torch.from_numpy(lab).to(device)


def build_loaders(args, generator=None):
    """The two training loaders of train_arco_2d.py:161-215 (ACDC / MM slice datasets): the first
    patients_to_slices(exp, labeled_num) slices are the labeled stream, the rest the unlabeled one; RandomGenerator per
    sample; each loader draws with replacement and drops the last incomplete batch.  The Synapse / LiTS / JHU branches
    use the npz slice dataset with the reference's hard-wired list directories (--list_dir overrides them)."""
    from torch.utils.data import ConcatDataset, DataLoader
    from torch.utils.data.sampler import RandomSampler
    from .build_dataset import BaseDataSetsWithIndex, Synapse_datasetWithIndex
    from .dataloaders import Compose
    from .dataloaders.dataset import RandomGenerator
    n_lab = patients_to_slices(args.exp, args.labeled_num)
    tf = lambda: Compose([RandomGenerator(args.patch_size)])
    # npz experiments (:162-187): (sub-directory of --root_path, the reference's hard-wired list directory)
    npz = None
    if "Syn" in args.exp or "syn" in args.exp:
        npz = ('/data/Synapse/train_npz', '/data/data/Synapse/data/lists_Synapse')
    elif "Lits" in args.exp or "LiTS" in args.exp:
        npz = ('/train_npz_40', '/data/data/Lits')
    elif "jhu" in args.exp or "JHU" in args.exp:
        npz = ('', '/data/data/JHUData')
    if npz is not None:
        list_dir = getattr(args, "list_dir", "") or npz[1]
        sets = [Synapse_datasetWithIndex(base_dir=args.root_path + npz[0], split="train", transform=tf(), index=n_lab, label_type=t,
                                         list_dir=list_dir) for t in (1, 0)]
    else:
        sets = [BaseDataSetsWithIndex(base_dir=args.root_path, split="train", num=None, transform=tf(), index=n_lab, label_type=t)
                for t in (1, 0)]
    db_l, db_u = sets
    while len(db_l) < len(db_u):                                           # :196-197
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
    stepper = ArcoStep2D(args, dev)
    b = args.batch_size
    loaders = None
    if args.synthetic:
        iters_per_epoch = 100
        if world > 1:         # every rank draws its own cutmix boxes / sampler indices / warps (seed + rank), after the broadcast
            adist.seed_data_pipeline(args.seed)
    else:
        # data parallel: every rank draws its own samples / augmentations (seed + rank), after the weight broadcast above
        loaders = build_loaders(args, generator=adist.seed_data_pipeline(args.seed) if world > 1 else None)
        iters_per_epoch = len(loaders[1])                              # :217 iterations per epoch = unlabeled batches
        logging.info("{} iterations per epoch".format(iters_per_epoch))
        resume = "../model/{}_{}_labeledfinal/{}/iter_30000.pth".format(args.resume, args.labeled_num, args.model)
        if os.path.exists(resume):                                      # stage-1 weights (:222-225), when present
            sd = torch.load(resume, map_location="cpu")
            stepper.isd.model.load_state_dict(sd); stepper.isd.ema_model.load_state_dict(sd)
            for pl in stepper.plans:                                    # packed weights are stale now
                pl.valid = False
        else:
            logging.info("no stage-1 checkpoint at {}: training from the random initialisation".format(resume))
    max_epoch = args.max_iterations // iters_per_epoch + 1
    l_iter = u_iter = None
    while stepper.iter_num < args.max_iterations:
        it = stepper.iter_num
        if args.synthetic:
            l_img, l_lab = synthetic_batch(b, args.patch_size, args.num_classes, 2 * it * world + rank, dev)
            u_img, _ = synthetic_batch(b, args.patch_size, args.num_classes, (2 * it + 1) * world + rank, dev)
        else:
            if it % iters_per_epoch == 0:                               # :268-270 fresh iterators every epoch
                l_iter, u_iter = iter(loaders[0]), iter(loaders[1])
            l_next, u_next = next(l_iter), next(u_iter)
            l_img, l_lab = l_next['image'].to(dev, non_blocking=True), l_next['label'].to(dev, non_blocking=True).long()
            u_img = u_next['image'].to(dev, non_blocking=True)
        loss, reco = stepper.step(l_img, l_lab, u_img, it // iters_per_epoch, max_epoch)
        if rank == 0:
            if "loss_q" in stepper.last_terms:                          # --revisit 1: the reference's logged total (:426,457)
                logging.info('iteration %d : loss : %f, reco_loss: %f' % (stepper.iter_num, loss.item(), reco.item()))
            else:
                # the reference's logged `loss` also carries k4*loss_q, the revisiting term (:126-136,334,425) - a constant w.r.t. every
                # parameter (no gradient path: weights, banks and every other logged value are unaffected) that needs the dense
                # 496-channel representations of both nets; by default it is NOT computed, and the log line says so instead of
                # printing a total that silently differs from the reference's (--revisit 1 computes and adds it)
                logging.info('iteration %d : loss : %f (without the gradient-free revisiting term k4*loss_q, k4 = %g: --revisit 1 adds it), '
                             'reco_loss: %f' % (stepper.iter_num, loss.item(), args.k4, reco.item()))
            if stepper.iter_num % 1000 == 0:                           # :462-470
                path = os.path.join(snapshot_path, 'iter_' + str(stepper.iter_num) + '.pth')
                # parameters are views into the optimiser's flat buffer: save private copies, not the shared storage
                torch.save({k: v.detach().clone() for k, v in stepper.isd.model.state_dict().items()}, path)
                logging.info("save model to {}".format(path))
    return "Training Finished!"


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.set_num_threads(min(4, torch.get_num_threads()))   # host logic only; avoids OpenMP oversubscription stalls
    random.seed(args.seed)                                               # :505-508
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
