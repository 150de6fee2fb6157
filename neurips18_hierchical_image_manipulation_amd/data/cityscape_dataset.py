"""reference data/cityscape_dataset.py"""
from .segmentation_dataset import SegmentationDataset


class CityscapeDataset(SegmentationDataset):
    def initialize(self, opt):
        super(CityscapeDataset, self).initialize(opt)
        self.class_of_interest = list(range(24, 34))      # person ... bicycle (cityscape_dataset.py:8)

    def name(self):
        return 'CitiscapeDataset'
