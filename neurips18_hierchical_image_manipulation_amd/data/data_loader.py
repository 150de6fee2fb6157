"""``from data.data_loader import CreateDataLoader`` of the training scripts (reference data/data_loader.py)."""
from .custom_dataset_data_loader import CreateDataLoader, CustomDatasetDataLoader  # noqa: F401
