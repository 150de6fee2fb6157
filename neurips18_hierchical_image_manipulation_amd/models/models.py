"""``create_model(opt)`` -- the reference factory (``models/models.py:6-24``) for the mask2image path."""


def create_model(opt, data_size=None):
    from ..options import complete
    if (opt.get('model') if isinstance(opt, dict) else getattr(opt, 'model', None)) == 'AE_maskgen_twostream':
        from .TwoStreamAE_mask import TwoStreamAE_mask          # box2mask trainer (train_box2mask.py)
        return TwoStreamAE_mask(opt)
    opt = complete(opt)
    if opt.model == 'pix2pixHD_condImg':
        from .pix2pixHD_condImg_model import Pix2PixHDModel_condImg
        model = Pix2PixHDModel_condImg(opt)
    elif opt.model == 'pix2pixHD_condImgColor':
        from .pix2pixHD_condImgColor_model import Pix2PixHDModel_condImgColor
        model = Pix2PixHDModel_condImgColor(opt)
    else:
        raise NotImplementedError('the model is not implemented')
    if getattr(opt, 'verbose', False):
        print('model [%s] was created' % model.name())
    # DataParallel wrapping of the reference is replaced by one process per GPU: ``model.module is model``.
    return model
