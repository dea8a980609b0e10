"""Drop-in for code/networks/net_factory_args.py:14-17: only the backbone the ARCO
trainers instantiate ('unet', model_2D.py:59) is provided."""
import torch

from .unetWithArgs import UNet


def net_factory(net_type="unet", in_chns=1, class_num=3, train_encoder=True, train_decoder=True, unfreeze_seg=True):
    if net_type == "unet":
        net = UNet(in_chns=in_chns, class_num=class_num, train_encoder=train_encoder,
                   train_decoder=train_decoder, unfreeze_seg=unfreeze_seg)
        # the reference moves the net to the GPU here (:17); on a host without one (building checkpoints, inspecting
        # state_dicts) the module stays on the CPU - its forward still refuses CPU tensors
        return net.cuda() if torch.cuda.is_available() else net
    raise NotImplementedError(f"net_type={net_type!r}: only 'unet' is on the ARCO hot path")
