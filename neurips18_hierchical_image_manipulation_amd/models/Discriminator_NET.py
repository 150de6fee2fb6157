"""Multi-scale PatchGAN on the HIP layer executor (reference ``models/Discriminator_NET.py:11-118``,
getIntermFeat=True; keys ``scale<i>_layer<j>.0.{weight,bias}``)."""
import torch.nn as nn

from ..nn import Conv2d, InstanceNorm2d, BatchNorm2d, LeakyReLU, FusedSequential, AvgPool3s2


class MultiscaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='instance', use_sigmoid=False, num_D=3,
                 getIntermFeat=True):
        super().__init__()
        if norm_layer not in ('instance', 'batch'):
            raise NotImplementedError('normalization layer [%s] is not found' % norm_layer)
        norm = InstanceNorm2d if norm_layer == 'instance' else BatchNorm2d   # 'batch': the box2mask discriminator
        if use_sigmoid:
            raise NotImplementedError('--no_lsgan (sigmoid + BCE) is not on the HIP path; LSGAN only')
        if not getIntermFeat:
            raise NotImplementedError('the mask2image model always asks for intermediate features')
        self.num_D, self.n_layers = num_D, n_layers
        for i in range(num_D):
            blocks = [[Conv2d(input_nc, ndf, 4, 2, 2), LeakyReLU(0.2)]]
            nf = ndf
            for _ in range(1, n_layers):
                nf_prev, nf = nf, min(nf * 2, 512)
                blocks.append([Conv2d(nf_prev, nf, 4, 2, 2), norm(nf), LeakyReLU(0.2)])
            nf_prev, nf = nf, min(nf * 2, 512)
            blocks.append([Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), LeakyReLU(0.2)])
            blocks.append([Conv2d(nf, 1, 4, 1, 2)])
            for j, b in enumerate(blocks):
                setattr(self, 'scale%d_layer%d' % (i, j), FusedSequential(*b))
        self.downsample = AvgPool3s2()

    def forward(self, input):
        result, x = [], input
        for i in range(self.num_D):
            feats, h = [], x
            for j in range(self.n_layers + 2):
                h = getattr(self, 'scale%d_layer%d' % (self.num_D - 1 - i, j))(h)   # :51 index reversal
                feats.append(h)
            result.append(feats)
            if i != self.num_D - 1:
                x = self.downsample(x)
        return result
