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

    # The reference's colour model puts obj_mask / color_embed IN FRONT of infer (positional callers such as its
    # vis_mask2image_color-style scripts depend on the order); the parent's methods take them as trailing keywords.
    def encode_input(self, label_map, inst_map=None, real_image=None, feat_map=None, mask_in=None, obj_mask=None,
                     color_embed=None, infer=False, lazy=False):
        """Reference :188-228 (argument order of :188); ``lazy``: see the parent (the trainer's own calls)."""
        return super().encode_input(label_map, inst_map, real_image, feat_map, mask_in=mask_in, infer=infer,
                                    obj_mask=obj_mask, color_embed=color_embed, lazy=lazy)

    def forward(self, label, inst, image, feat, mask_in, mask_out, obj_mask, infer=False):
        """Reference :252 -- ``obj_mask`` is the seventh positional argument, ``infer`` the eighth."""
        return super().forward(label, inst, image, feat, mask_in, mask_out, infer=infer, obj_mask=obj_mask)

    def forward_wrapper(self, data, infer=False):
        """Reference :242-250 forwards ``infer`` in obj_mask's position (dead code upstream: it cannot run); here the
        batch's ``obj_mask`` goes where :252 expects it."""
        return self.forward(data['label'], data['inst'], data['image'], None, data['mask_in'], data['mask_out'],
                            data['obj_mask'], infer)

    def inference(self, label, inst, image, mask_in, color_embed=None, obj_mask=None):
        """Reference :315 -- no ``mask_out``; a given ``color_embed`` (B,3) replaces the object's mean colour."""
        return super().inference(label, inst, image, mask_in, None, obj_mask=obj_mask, color_embed=color_embed)

    def encode_instwise_embedding(self, inst_map, embedding):
        pass        # reference :144-145: a stub there as well

    def encode_global_embedding(self, mask_in, embedding):
        """(B,K) embedding tiled over the box: (B,K,H,W) = embedding[b,k] * mask_in[b,0] (reference :147-160); the
        training path writes the same values straight into the conditioning buffer (ops.encode_channels)."""
        mask_in, embedding = self._dev(mask_in), self._dev(embedding)
        return ops.tile_embedding(embedding, mask_in)

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
