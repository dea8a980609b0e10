"""Stage-1 pre-training of the 3-D model, drop-in for the reference's code/pretrain_3D.py on MI355X:
ISD_3d (V-Net student / momentum teacher, model_3D.py:219-403) trained with 0.5 * (CE + Dice) on the labeled volumes and
the two KL-divergence-to-queue terms; checkpoints iter_<n>.pth / iter_<n>_ema.pth for train_arco_3d.py.  Same flag table
as the reference, both V-Nets on the HIP kernels; see arco_amd/pretrain_2D.py for the shared pieces."""
import logging
import os
import random
import sys

import numpy as np
import torch

from . import pretrain_2D as P2
from .model_3D import ISD_3d

KLD = P2.KLD
get_current_t = P2.get_current_t
student_teacher_batches = P2.student_teacher_batches


def build_parser():
    p = P2.build_parser()
    p.set_defaults(root_path='/home/weicheng/selfLearning/DTC/data/2018LA_Seg_Training Set', exp='LA/stage_2_temp', model='vnet',
                   batch_size=2, patch_size=[112, 112, 80], num_classes=2, labeled_bs=1, labeled_num=4, combinations=1)
    return p


def make_transform_student():
    """pretrain_3D.py:126-129"""
    from .dataloaders import Compose
    from .dataloaders.la_heart import RandomColorJitter, RandomNoise
    return Compose([RandomColorJitter(p=0.5, color=(0.02, 0.02, 0.02, 0.01)), RandomNoise(p=0.5)])


class PretrainStep3D(P2.PretrainStep2D):
    """pretrain_3D.py:163-232: ISD_3d with the hard-wired 20-voxel patch of the output heads; supervised term halved."""
    sup_scale = 0.5

    def build_model(self, args):
        return ISD_3d(K=args.K, m=0.99, Ts=args.T_s, Tt=args.T_t, num_classes=args.num_classes,
                      latent_pooling_size=args.latent_pooling_size, latent_feature_size=args.latent_feature_size,
                      output_pooling_size=args.output_pooling_size, train_encoder=args.train_encoder,
                      train_decoder=args.train_decoder, patch_size=getattr(args, "head_patch", 20))


def synthetic_batch(b, patch, n_cls, seed):
    from .train_arco_3d import synthetic_volume_batch
    img, lab = synthetic_volume_batch(b, patch, n_cls, seed, "cpu")
    return {'image': img, 'label': lab}


def build_loader(args):
    """pretrain_3D.py:140-161: LA volumes, RandomRotFlip + RandomCrop + ToTensor, the first --labeled_num volumes labeled."""
    from torch.utils.data import DataLoader
    from .dataloaders import Compose
    from .dataloaders.dataset import TwoStreamBatchSampler
    from .dataloaders.la_heart import LAHeartWithIndex, RandomCrop, RandomRotFlip, ToTensor
    db = LAHeartWithIndex(base_dir=args.root_path, split='train', index=0, label_type=0,
                          transform=Compose([RandomRotFlip(), RandomCrop(args.patch_size), ToTensor()]))
    total, labeled = len(db), args.labeled_num
    print("Total silices is: {}, labeled slices is: {}".format(total, labeled))
    sampler = TwoStreamBatchSampler(list(range(0, labeled)), list(range(labeled, total)), args.batch_size,
                                    args.batch_size - args.labeled_bs)
    return DataLoader(db, batch_sampler=sampler, num_workers=0, pin_memory=True)


def train(args, snapshot_path):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stepper = PretrainStep3D(args, dev)
    transform_student = make_transform_student()
    loader = None if args.synthetic else build_loader(args)
    per_epoch = 20 if args.synthetic else len(loader)
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
