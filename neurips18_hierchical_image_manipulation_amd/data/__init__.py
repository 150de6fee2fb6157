"""Loader side of the hot path: the reference's ``data/`` package re-built so that a batch is born on
the device.  Host: file listing, PNG/JPEG decode (Pillow), the crop-window sampler; device: every pixel operation
(csrc/him_data.hip).  Same module / class / function names as the reference's ``data`` package."""
