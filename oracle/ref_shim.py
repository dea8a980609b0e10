"""Import the ARCO reference (/root/reference/code) on CPU, in THIS container only.

TEST INFRASTRUCTURE — never imported by the product (arco_amd/).  Used by
oracle/gen_golden.py to produce the golden vectors under tests/golden/.
/root/reference does not exist on the GPU box; nothing at test/bench time
may call this module.

Shims: no reference file is modified;
 (1) stub modules for optional backbones that model_2D/model_3D import but the
     hot path never instantiates;
 (2) Tensor.cuda / Module.cuda become identity so the hard-coded .cuda() calls
     (loss_helper_3d.py:427,433,456,466,485,508; net_factory_args.py:17) run on CPU.
"""
import sys
import types
import importlib

REF = "/root/reference/code"


def load():
    import torch
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, attrs in {
        "networks.efficientunet": ["Effi_UNet"],
        "networks.config": ["get_config"],
        "networks.nnunet": ["initialize_network"],
        "networks.vit_seg_modeling": ["VisionTransformer", "CONFIGS"],
        "networks.enet": ["ENet"],
        "networks.pnet": ["PNet2D"],
        "networks.unet_3D": ["unet_3D"],
        "networks.VoxResNet": ["VoxResNet"],
        "networks.attention_unet": ["Attention_UNet"],
    }.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, {} if a == "CONFIGS" else None)
            sys.modules[name] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    mods = {}
    for n in ["loss_helper_3d", "loss_helper", "networks.unetWithArgs",
              "networks.vnetWithArgs", "model_2D", "model_3D"]:
        mods[n] = importlib.import_module(n)
    return mods
