"""transforms.Compose without torchvision (absent in this image): the trainers use arco_amd.dataloaders.Compose."""
from arco_amd.dataloaders import Compose  # noqa: F401
