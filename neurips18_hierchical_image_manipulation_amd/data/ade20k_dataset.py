"""Import location of the reference's ``ADE20KDataset`` (data/ade20k_dataset.py); defined next to its base class."""
from .segmentation_dataset import ADE20KDataset  # noqa: F401
