"""``Pix2PixHDModel_condImgColor``: the colour-conditioned variant
(reference ``models/pix2pixHD_condImgColor_model.py:147-228``): per-image masked mean colour of the object,
times U(0.97,1.03) noise, clamped, tiled over ``mask_in`` -> the conditioning image becomes 6 channels and the
two-stream generator is built with ``extra_embed=True``.  ``forward`` takes ``obj_mask`` (:252)."""
import torch

from .. import ops
from .pix2pixHD_condImg_model import Pix2PixHDModel_condImg


class Pix2PixHDModel_condImgColor(Pix2PixHDModel_condImg):
    color = True

    def __init__(self, opt):
        if opt.netG != 'global_twostream':
            raise NameError('global generator name is not defined properly: %s' % opt.netG)
        super().__init__(opt)

    def name(self):
        return 'Pix2PixHDModel_condImg'     # sic: the reference returns the parent's name (:141-142)

    def get_color_embedding(self, inst_map, image, noise=None):
        """(B,3) masked mean colour; ``noise`` (B,3) multiplies it (None = exactly 1, the parity mode)."""
        return ops.masked_mean_color(image, inst_map, noise)

    def _color_embedding(self, obj_mask, real_image, color_embed, infer):
        if infer and color_embed is not None:
            return self._dev(color_embed)
        assert obj_mask is not None, 'the colour model needs obj_mask'
        noise = None
        if self.opt.color_noise:
            noise = torch.rand(real_image.size(0), 3, device=real_image.device) * 0.06 + 0.97
        return self.get_color_embedding(obj_mask, real_image, noise)
