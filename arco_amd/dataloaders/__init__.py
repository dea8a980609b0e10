class Compose(object):
    """torchvision.transforms.Compose for dict samples (torchvision is not a dependency of this package)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample
