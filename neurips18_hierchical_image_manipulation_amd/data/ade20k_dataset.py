"""reference data/ade20k_dataset.py"""
from .segmentation_dataset import SegmentationDataset

_SKIPPED = (3, 6, 27, 34)     # the classes 2..38 the reference list leaves out (ade20k_dataset.py:8-10)


class ADE20KDataset(SegmentationDataset):
    def initialize(self, opt):
        super(ADE20KDataset, self).initialize(opt)
        self.class_of_interest = [c for c in range(2, 39) if c not in _SKIPPED]

    def name(self):
        return 'ADE20KDataset'
