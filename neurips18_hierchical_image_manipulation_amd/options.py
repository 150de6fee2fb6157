"""Flag system of the mask2image path: same flag names and defaults as the reference's
``options/mask2image_base_options.py`` / ``mask2image_train_options.py`` (they are the de-facto config
contract), table-driven.  ``complete(opt)`` accepts a dict / Namespace with any subset of them."""
import argparse
import os

# name -> (type|'flag', default)
BASE_FLAGS = [
    ('name', str, 'label2city'), ('gpu_ids', str, '0'), ('checkpoints_dir', str, './checkpoints'),
    ('model', str, 'CVAE_imggen'), ('norm', str, 'instance'), ('use_dropout', 'flag', False),
    ('input_layout', 'flag', False), ('load_image', 'flag', False), ('load_instmap', 'flag', False),
    ('use_bbox', int, 0), ('batchSize', int, 1), ('loadSize', int, 1024), ('fineSize', int, 512),
    ('label_nc', int, 35), ('output_nc', int, 3), ('contextMargin', float, 3.0), ('prob_bg', float, 0.3),
    ('min_box_size', int, 32), ('max_box_size', int, 256), ('random_crop', int, 1),
    ('dataroot', str, './datasets/cityscape/'), ('dataloader', str, 'segmentation_dataset'),
    ('resize_or_crop', str, 'scale_width'), ('serial_batches', 'flag', False), ('no_flip', 'flag', False),
    ('nThreads', int, 2), ('max_dataset_size', int, float('inf')), ('display_winsize', int, 512),
    ('tf_log', 'flag', False), ('netG', str, 'global'), ('ngf', int, 64), ('n_downsample_global', int, 4),
    ('n_blocks_global', int, 9), ('n_blocks_local', int, 3), ('n_local_enhancers', int, 1),
    ('niter_fix_global', int, 0), ('which_encoder', str, 'ctx'), ('use_output_gate', 'flag', False),
    ('use_skip', 'flag', False), ('feat_fusion', str, 'early_add'), ('no_instance', 'flag', False),
    ('instance_feat', 'flag', False), ('label_feat', 'flag', False), ('feat_num', int, 3),
    ('load_features', 'flag', False), ('n_downsample_E', int, 3), ('nef', int, 16), ('n_clusters', int, 10),
    ('z_dim', int, 32), ('z_embed_dim', int, 64),
]
TRAIN_FLAGS = [
    ('display_freq', int, 100), ('print_freq', int, 100), ('save_latest_freq', int, 1000),
    ('save_epoch_freq', int, 10), ('no_html', 'flag', False), ('debug', 'flag', False),
    ('continue_train', 'flag', False), ('load_pretrain', str, ''), ('which_epoch', str, 'latest'),
    ('phase', str, 'train'), ('niter', int, 100), ('niter_decay', int, 100), ('beta1', float, 0.5),
    ('lr', float, 0.0002), ('kl_weight', float, 0.00001), ('kl_decay_rate', float, 0.99),
    ('kl_threshold', float, 0.001), ('num_checkpoint', int, 2), ('no_gan', 'flag', False), ('num_D', int, 2),
    ('n_layers_D', int, 3), ('ndf', int, 64), ('lambda_feat', float, 10.0), ('lambda_rec', float, 0.0),
    ('no_ganFeat_loss', 'flag', False), ('no_vgg_loss', 'flag', False), ('no_lsgan', 'flag', False),
    ('pool_size', int, 0), ('no_imgCond', 'flag', False), ('mask_gan_input', 'flag', False),
    ('use_soft_mask', 'flag', False),
]
# options/mask2image_test_options.py:8-14 (inference / visualisation runs; --phase and --which_epoch re-declared there)
TEST_FLAGS = [
    ('ntest', int, float('inf')), ('results_dir', str, './checkpoints/'), ('aspect_ratio', float, 1.0),
    ('phase', str, 'test'), ('which_epoch', str, 'latest'), ('how_many', int, 50),
    ('cluster_path', str, 'features_clustered_010.npy'),
]
# box2mask (second hot path): options/box2mask_base_options.py:12-83, box2mask_train_options.py:8-36, box2mask_test_options.py:8-16
BOX2MASK_BASE_FLAGS = [
    ('add_dilated_layers', 'flag', False), ('batchSize', int, 64), ('checkpoints_dir', str, './checkpoints'),
    ('cond_in', str, 'ctx'), ('contextMargin', float, 2.0), ('conv_dim', int, 64), ('conv_size', int, 4),
    ('dataloader', str, 'cityscape'), ('dataroot', str, './datasets/cityscape/'), ('display_winsize', int, 512),
    ('embed_dim', int, 1024), ('fineSize', int, 128), ('first_conv_size', int, 5), ('first_conv_stride', int, 1),
    ('fusion_type', str, 'add'), ('gan_weight', float, 1.0), ('gpu_ids', str, '0'), ('label_nc', int, 36),
    ('lambda_feat', float, 1.0), ('loadSize', int, None), ('load_image', int, 0), ('max_box_size', int, 64),
    ('max_dataset_size', int, float('inf')), ('min_box_size', int, 32), ('model', str, 'AE_maskgen'),
    ('nThreads', int, 2), ('n_blocks', int, 4), ('n_blocks_decode', int, 4), ('n_blocks_gt', int, 4),
    ('n_blocks_masked', int, 4), ('name', str, 'box2mask'), ('ndf', int, 64), ('ngf', int, 64),
    ('no_comb', 'flag', False), ('no_flip', 'flag', False), ('no_instance', 'flag', False),
    ('norm_layer', str, 'batch'), ('num_layers', int, 6), ('num_layers_D', int, 4), ('num_resnetblocks', int, 1),
    ('objReconLoss', str, 'bce'), ('output_nc', int, 36), ('prob_bg', float, 0.3), ('random_crop', int, 1),
    ('rec_weight', float, 1.0), ('resize_or_crop', str, 'select_region'), ('serial_batches', 'flag', False),
    ('skip_end', int, 3), ('skip_start', int, 1), ('tf_log', 'flag', False), ('use_bbox', int, 1),
    ('use_dropout', 'flag', False), ('use_gan', 'flag', False), ('use_output_gate', 'flag', False),
    ('use_resnetblock', int, 1), ('use_simpleRes', 'flag', False), ('which_epoch', str, 'latest'),
    ('which_gan', str, 'patch'), ('which_stream', str, 'obj_context'), ('z_dim', int, 512),
]
BOX2MASK_TRAIN_FLAGS = [
    ('beta1', float, 0.9), ('beta2', float, 0.999), ('continue_train', 'flag', False), ('debug', 'flag', False),
    ('display_freq', int, 40), ('enc_lr', float, 1.0), ('load_pretrain', str, ''), ('lr', float, 0.0002),
    ('lr_control', 'flag', False), ('mask_gan_input', 'flag', False), ('niter', int, 200), ('niter_decay', int, 0),
    ('no_html', 'flag', False), ('num_checkpoint', int, 2), ('phase', str, 'train'), ('print_freq', int, 40),
    ('save_epoch_freq', int, 10), ('save_latest_freq', int, 200), ('use_ganFeat_loss', 'flag', False),
]
BOX2MASK_TEST_FLAGS = [
    ('aspect_ratio', float, 1.0), ('gendata_dir', str, 'gen_ae_512p'), ('gtdata_dir', str, 'gt_512p'),
    ('how_many', int, 50), ('ntest', int, float('inf')), ('num_samples', int, 1), ('phase', str, 'test'),
    ('results_dir', str, 'results/'),
]
# additions of this build (absent in the reference)
BUILD_FLAGS = [('vgg_weights', str, ''), ('verbose', 'flag', False), ('color_noise', 'flag', False),
               ('compact_labels', 'flag', False),
               ('sn_D', 'flag', False)]             # spectral-norm convolutions in the multi-scale PatchGAN (sn_utils.py)   # loader: label ids as uint8 (the trainers widen them on the device)


class MaskToImageOptions(object):
    isTrain = False
    tables = [BASE_FLAGS, BUILD_FLAGS]

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        for table in self.tables:
            for name, typ, default in table:
                if typ == 'flag':
                    self.parser.add_argument('--' + name, action='store_true')
                else:
                    self.parser.add_argument('--' + name, type=typ, default=default)
        self.initialized = True

    def parse(self, save=True, default_args=()):
        if not self.initialized:
            self.initialize()
        self.opt = self.parser.parse_args(list(default_args))
        self.opt.isTrain = self.isTrain
        self.opt.gpu_ids = [int(s) for s in self.opt.gpu_ids.split(',') if int(s) >= 0]
        if save and not getattr(self.opt, 'continue_train', False):
            d = os.path.join(self.opt.checkpoints_dir, self.opt.name)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'opt.txt'), 'wt') as f:
                f.write('------------ Options -------------\n')
                for k, v in sorted(vars(self.opt).items()):
                    f.write('%s: %s\n' % (str(k), str(v)))
                f.write('-------------- End ----------------\n')
        return self.opt


class MaskToImageTrainOptions(MaskToImageOptions):
    isTrain = True
    tables = [BASE_FLAGS, TRAIN_FLAGS, BUILD_FLAGS]


class MaskToImageTestOptions(MaskToImageOptions):
    """options/mask2image_test_options.py: the base flags + the test flags, ``isTrain = False``."""
    isTrain = False
    tables = [BASE_FLAGS, TEST_FLAGS, BUILD_FLAGS]


class BoxToMaskOptions(MaskToImageOptions):
    """options/box2mask_base_options.py: the parser of train_box2mask.py / test_box2mask.py (same parse / opt.txt
    behaviour as the mask2image parser)."""
    isTrain = False
    tables = [BOX2MASK_BASE_FLAGS]


class BoxToMaskTrainOptions(BoxToMaskOptions):
    isTrain = True
    tables = [BOX2MASK_BASE_FLAGS, BOX2MASK_TRAIN_FLAGS]


class BoxToMaskTestOptions(BoxToMaskOptions):
    isTrain = False
    tables = [BOX2MASK_BASE_FLAGS, BOX2MASK_TEST_FLAGS]


def complete(opt):
    """dict / Namespace -> Namespace with every hot-path flag present (reference defaults)."""
    if isinstance(opt, dict):
        opt = argparse.Namespace(**opt)
    if not hasattr(opt, 'model'):
        # the reference PARSER's default is 'CVAE_imggen' (options/mask2image_base_options.py:18), a name its own
        # create_model rejects (models/models.py:16-17): every shipped script passes --model.  A partial dict / Namespace
        # handed to this function means the hot-path trainer.
        opt.model = 'pix2pixHD_condImg'
    for table in (BASE_FLAGS, TRAIN_FLAGS, BUILD_FLAGS):
        for name, typ, default in table:
            if not hasattr(opt, name):
                setattr(opt, name, default)
    if not hasattr(opt, 'isTrain'):
        opt.isTrain = True
    if isinstance(opt.gpu_ids, str):
        opt.gpu_ids = [int(s) for s in opt.gpu_ids.split(',') if int(s) >= 0]
    return opt
