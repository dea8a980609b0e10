"""Drop-in for code/networks/net_factory_3dArgs.py:16-18: only the backbone the ARCO 3-D trainer
instantiates ('vnet', batch-norm, dropout; model_3D.py:115) is provided."""
from .vnetWithArgs import VNet


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2):
    if net_type == "vnet":
        return VNet(n_channels=in_chns, n_classes=class_num, normalization='batchnorm', has_dropout=True)
    raise NotImplementedError(f"net_type={net_type!r}: only 'vnet' is on the ARCO hot path")
