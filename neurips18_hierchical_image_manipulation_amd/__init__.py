"""MI355X-native mask2image (layout-to-image GAN) training path of
xcyan/neurips18_hierchical_image_manipulation: hand-written gfx950 kernels (csrc/, C ABI in include/him.h)
behind the reference's own ``models.create_model`` / ``BaseModel`` Python surface."""
__version__ = '0.1.0'
