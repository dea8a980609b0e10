"""Stage-1 pre-training, drop-in for the reference's code/pretrain_2D.py on MI355X: the ISD
student / momentum-teacher pair trained with supervised CE + Dice on the labeled part of the batch and two
KL-divergence-to-queue terms (latent vector, patch-wise output embeddings; pretrain_2D.py:235-252).  Its product is the
`iter_<n>.pth` / `iter_<n>_ema.pth` pair that train_arco_2d.py loads (`--resume`, train_arco_2d.py:223-226).

Same flag table (names, defaults), same loss arithmetic, same generator consumption as the reference; both U-Nets run on
the HIP kernels (arco_amd.ops), the optimizer is torch.optim.SGD(momentum=0.9, weight_decay=1e-4) on flat buffers
(arco_sgd_momentum).  `--synthetic 1` replaces the slice loader with seeded ACDC-shaped tensors (no dataset ships with the
reference)."""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

from . import glue, optim
from .model_2D import ISD


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--root_path', type=str, default='/data/data/Synapse', help='Name of Experiment')
    p.add_argument('--exp', type=str, default='Synapse/example_training', help='experiment_name')
    p.add_argument('--model', type=str, default='unet', help='model_name')
    p.add_argument('--max_iterations', type=int, default=6000, help='maximum epoch number to train')
    p.add_argument('--batch_size', type=int, default=6, help='batch_size per gpu')
    p.add_argument('--deterministic', type=int, default=1, help='whether use deterministic training')
    p.add_argument('--base_lr', type=float, default=0.01, help='segmentation network learning rate')
    p.add_argument('--patch_size', type=list, default=[256, 256], help='patch size of network input')
    p.add_argument('--seed', type=int, default=1337, help='random seed')
    p.add_argument('--num_classes', type=int, default=4, help='output channel of network')
    p.add_argument('--labeled_bs', type=int, default=3, help='labeled_batch_size per gpu')
    p.add_argument('--labeled_num', type=int, default=7, help='labeled data')
    p.add_argument('--ema_decay', type=float, default=0.99, help='ema_decay')
    p.add_argument('--consistency_type', type=str, default="mse", help='consistency_type')
    p.add_argument('--consistency', type=float, default=0.1, help='consistency')
    p.add_argument('--consistency_rampup', type=float, default=200.0, help='consistency_rampup')
    p.add_argument('--resume', type=str, default='ACDC/training_pool_latentF512_K36', help='if we should resume from checkpoint')
    p.add_argument('--K', type=int, default=36, help='the size of cache')
    p.add_argument('--train_encoder', type=int, default=1, help='is training encoder?')
    p.add_argument('--train_decoder', type=int, default=1, help='is training decoder?')
    p.add_argument('--k1', type=float, default=1.0, help='the weights for latent contrastive loss')
    p.add_argument('--k2', type=float, default=1.0, help='the weights for output contrastive loss')
    p.add_argument('--latent_pooling_size', type=int, default=1, help='the pooling size of latent vector')
    p.add_argument('--latent_feature_size', type=int, default=512, help='the feature size of latent vectors')
    p.add_argument('--output_pooling_size', type=int, default=8, help='the pooling size of output head')
    p.add_argument('--T_s', type=float, default=0.1, help='temperature for student')
    p.add_argument('--T_t', type=float, default=0.1, help='temperature for teacher')
    p.add_argument('--combinations', type=int, default=2, help='the combination of transformation')
    p.add_argument('--cut_size', type=int, default=64, help='the combination of transformation')
    p.add_argument('--temp_high', default=1.0, type=float)
    # not in the reference
    p.add_argument('--synthetic', type=int, default=0, help='1: seeded ACDC-shaped tensors instead of the slice loader')
    p.add_argument('--save_every', type=int, default=1000, help='checkpoint interval (the reference hard-codes 1000)')
    p.add_argument('--snapshot_path', type=str, default='', help='overrides ../model/<exp>_<labeled_num>_labeled<suffix>/<model>')
    return p


class KLD(torch.nn.Module):
    """pretrain_2D.py:99-103: batch-mean KL(softmax(targets) || softmax(inputs)) over dim 1."""

    def forward(self, inputs, targets):
        return F.kl_div(F.log_softmax(inputs, dim=1), F.softmax(targets, dim=1), reduction='batchmean')


def patients_to_slices(dataset, patiens_num):
    """pretrain_2D.py:105-121 (keyed on --root_path here, on --exp in train_arco_2d.py)."""
    if "ACDC" in dataset:
        ref = {"1": 23, "3": 68, "7": 136, "14": 256, "21": 396, "28": 512, "35": 664, "140": 1312}
    elif "MM" in dataset:
        ref = {"1": 38, "2": 76, "5": 191, "10": 382, "100": 3823}
    elif "Syn" in dataset or "syn" in dataset:
        ref = {"1": 44, "3": 66, "5": 111, "10": 221, "100": 2211}
    elif "Lits" in dataset or "LiTS" in dataset:
        ref = {"1": 167, "5": 835, "10": 1668, "20": 3336, "50": 8340, "100": 16684}
    elif "jhu" in dataset or "JHU" in dataset:
        ref = {"1": 57, "5": 275, "10": 568, "100": 5675}
    else:
        raise KeyError("patients_to_slices: unknown dataset " + dataset)
    return ref[str(patiens_num)]


def get_current_t(epoch_num, max_epoch, T_low=0.1, T_high=1.0):
    return (T_high - T_low) * (1 + np.cos(2 * np.pi * epoch_num / max_epoch)) / 2 + T_low


def make_transform_student():
    """pretrain_2D.py:134-137"""
    from .dataloaders import Compose
    from .dataloaders.dataset_withAug import RandomColorJitter, RandomNoise
    return Compose([RandomColorJitter(p=0.5, color=(0.2, 0.2, 0.2, 0.1)), RandomNoise(p=0.5)])


def student_teacher_batches(sampled_batch, combinations, transform_student):
    """pretrain_2D.py:212-229.  Both names start as the SAME dict: RandomColorJitter edits the image tensor in place
    (both see it), RandomNoise returns a new dict (only the transformed side sees the blur)."""
    teacher_batch = student_batch = sampled_batch
    if combinations == 1:
        student_batch = transform_student(student_batch)
    elif combinations == 2:
        teacher_batch = transform_student(teacher_batch)
    elif combinations != 0:
        teacher_batch = transform_student(teacher_batch)
        student_batch = transform_student(student_batch)
    return student_batch, teacher_batch


class PretrainStep2D:
    """Model, optimizer and one training iteration (pretrain_2D.py:189-252)."""
    sup_scale = 1.0          # pretrain_3D.py:218 halves the supervised term

    def build_model(self, args):
        return ISD(K=args.K, m=0.99, Ts=args.T_s, Tt=args.T_t, num_classes=args.num_classes,
                   latent_pooling_size=args.latent_pooling_size, latent_feature_size=args.latent_feature_size,
                   output_pooling_size=args.output_pooling_size, train_encoder=args.train_encoder,
                   train_decoder=args.train_decoder, patch_size=args.cut_size)

    def __init__(self, args, device):
        self.args, self.device = args, device
        self.model = self.build_model(args).to(device)
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = optim.SGDNesterov(params, lr=args.base_lr, momentum=0.9, weight_decay=0.0001, nesterov=False)
        self.model.train()
        self.kld = KLD()
        self.iter_num = 0
        self.last_terms = {}

    def step(self, student_batch, student_label, teacher_batch):
        a = self.args
        if student_batch.dim() == 3:                       # (:231-234, kept as written)
            student_batch = student_batch.unsqueeze(1)
        if teacher_batch.dim() == 3:
            teacher_batch = teacher_batch.unsqueeze(1)
        outputs, ema_output, ema_latent_logits, latent_logits, ema_output_logits, output_logits = \
            self.model(student_batch.float(), teacher_batch.float())
        loss_ce, loss_dice = glue.supervised_loss(outputs[:a.labeled_bs], student_label[:a.labeled_bs].long())
        supervised_loss = self.sup_scale * (loss_dice + loss_ce)
        loss_latent = self.kld(inputs=latent_logits, targets=ema_latent_logits)
        loss_output = self.kld(inputs=output_logits, targets=ema_output_logits)
        if a.train_encoder == 1 and not a.train_decoder == 1:
            loss = a.k1 * loss_latent
        else:
            loss = supervised_loss + a.k1 * loss_latent + a.k2 * loss_output
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        lr_ = a.base_lr * (1.0 - self.iter_num / a.max_iterations) ** 0.9
        for g in self.optimizer.param_groups:
            g['lr'] = lr_
        self.iter_num += 1
        self.last_terms = dict(loss=loss.detach(), ce=loss_ce.detach(), dice=loss_dice.detach(),
                               latent=loss_latent.detach(), output=loss_output.detach())
        return loss.detach()

    def save(self, snapshot_path):
        """pretrain_2D.py:281-291: student and teacher U-Net state_dicts (cloned: the parameters are views of the
        optimizer's flat buffer)."""
        for sub, tag in ((self.model.model, ''), (self.model.ema_model, '_ema')):
            path = os.path.join(snapshot_path, 'iter_' + str(self.iter_num) + tag + '.pth')
            torch.save({k: v.detach().clone() for k, v in sub.state_dict().items()}, path)
        logging.info("save model to {}".format(path))


def synthetic_batch(b, patch, n_cls, seed):
    from .train_arco_2d import synthetic_batch as sb
    img, lab = sb(b, patch, n_cls, seed, "cpu")
    return {'image': img, 'label': lab}


def build_loader(args):
    """pretrain_2D.py:154-188: one loader over the whole slice list, every batch = labeled_bs labeled + the rest
    unlabeled indices (TwoStreamBatchSampler)."""
    from torch.utils.data import DataLoader
    from .build_dataset import BaseDataSetsWithIndex, Synapse_datasetWithIndex
    from .dataloaders import Compose
    from .dataloaders.dataset import RandomGenerator, TwoStreamBatchSampler
    tf = Compose([RandomGenerator(args.patch_size)])
    if "Syn" in args.exp or "syn" in args.exp:
        db = Synapse_datasetWithIndex(base_dir=args.root_path + '/data/Synapse/train_npz', split="train", transform=tf,
                                      list_dir=args.root_path + '/data/lists_Synapse', index=0, label_type=0)
    elif "Lits" in args.exp or "LITS" in args.exp:
        db = Synapse_datasetWithIndex(base_dir=args.root_path + '/train_npz_40', split="train", transform=tf,
                                      list_dir='/data/data/Lits', index=0, label_type=0)
    elif 'jhu' in args.exp or "JHU" in args.exp:
        db = Synapse_datasetWithIndex(base_dir=args.root_path, split="train", transform=tf, list_dir='/data/data/JHUData',
                                      index=0, label_type=0)
    else:
        db = BaseDataSetsWithIndex(base_dir=args.root_path, split="train", num=None, transform=tf, index=0, label_type=0)
    total, labeled = len(db), patients_to_slices(args.root_path, args.labeled_num)
    print("Total silices is: {}, labeled slices is: {}".format(total, labeled))
    sampler = TwoStreamBatchSampler(list(range(0, labeled)), list(range(labeled, total)), args.batch_size,
                                    args.batch_size - args.labeled_bs)
    return DataLoader(db, batch_sampler=sampler, num_workers=0, pin_memory=True)


def train(args, snapshot_path):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stepper = PretrainStep2D(args, dev)
    transform_student = make_transform_student()
    loader = None if args.synthetic else build_loader(args)
    per_epoch = 50 if args.synthetic else len(loader)
    logging.info("{} iterations per epoch".format(per_epoch))
    max_epoch = args.max_iterations // per_epoch + 1
    for epoch_num in range(max_epoch):
        batches = (synthetic_batch(args.batch_size, args.patch_size, args.num_classes, args.seed + 1000 * epoch_num + i)
                   for i in range(per_epoch)) if args.synthetic else loader
        for sampled_batch in batches:
            sampled_batch = {'image': sampled_batch['image'].to(dev, non_blocking=True),
                             'label': sampled_batch['label'].to(dev, non_blocking=True)}
            student, teacher = student_teacher_batches(sampled_batch, args.combinations, transform_student)
            loss = stepper.step(student['image'], student['label'], teacher['image'])
            t = stepper.last_terms
            logging.info('iteration %d : loss : %f, loss_ce: %f, loss_dice: %f, loss_latent: %f, loss_output: %f' %
                         (stepper.iter_num, loss.item(), t['ce'].item(), t['dice'].item(), t['latent'].item(), t['output'].item()))
            if stepper.iter_num % args.save_every == 0:
                stepper.save(snapshot_path)
            if stepper.iter_num >= args.max_iterations:
                break
        if stepper.iter_num >= args.max_iterations:
            break
    return "Training Finished!"


def main(argv=None):
    args = build_parser().parse_args(argv)
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed); torch.cuda.manual_seed(args.seed)
    if args.train_encoder == 1 and args.train_decoder == 1:
        suffix = 'final'
    elif args.train_encoder == 1:
        suffix = '_train_encoder'
    else:
        suffix = '_train_decoder'
    snapshot_path = args.snapshot_path or "../model/{}_{}_labeled{}/{}".format(args.exp, args.labeled_num, suffix, args.model)
    os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(filename=snapshot_path + "/log.txt", level=logging.INFO,
                        format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S')
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    logging.info(str(args))
    return train(args, snapshot_path)


if __name__ == "__main__":
    main()
