"""Import location of the reference's ``CityscapeDataset`` (data/cityscape_dataset.py); defined next to its base class."""
from .segmentation_dataset import CityscapeDataset  # noqa: F401
