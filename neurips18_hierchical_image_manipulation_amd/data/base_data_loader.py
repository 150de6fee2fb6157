"""Import location of the reference's ``BaseDataLoader`` (data/base_data_loader.py)."""
from .custom_dataset_data_loader import BaseDataLoader  # noqa: F401
