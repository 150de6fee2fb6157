"""Host-side mirror of the reference's ``models`` package for the mask2image path (same file and class
names, same state_dict keys, same call signatures); all arithmetic goes to libhim_hip.so."""
from .models import create_model  # noqa: F401
