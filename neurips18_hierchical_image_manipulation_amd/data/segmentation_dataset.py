"""Segmentation dataset of the two trainers (reference data/segmentation_dataset.py), split into a host stage and a
device stage.

``host_record(index)`` does what needs files and the sampler: open the label / instance / photo files, read the object
annotation, draw the windows (``get_transform_params``, the context ratio) and cut the windows out of the decoded
images as raw bytes.  ``assemble(records)`` turns a list of such records into the batch dictionary of the reference
(same keys, same values) with every tensor on the device: three NEAREST gathers, one BICUBIC pair and one mask kernel
for the whole batch (see :mod:`.device`).  ``__getitem__`` is ``assemble([host_record(i)])`` without the batch axis.
"""
import json
import os.path

import numpy as np
import torch
from PIL import Image

from . import device as dv
from . import resample
from .base_dataset import BaseDataset, get_soft_bbox, get_transform_params
from .image_folder import make_dataset

_TIMES_255 = ('sun_rgbd', 'ade20k')      # loaders whose 8-bit instance maps are scaled back to ids (reference :79-80)


class SegmentationDataset(BaseDataset):
    def initialize(self, opt):
        self.opt = opt
        self.root = opt.dataroot
        self.class_of_interest = []          # set by the child classes
        self.config = {'prob_flip': 0.0 if opt.no_flip else 0.5, 'prob_bg': opt.prob_bg, 'fineSize': opt.fineSize,
                       'preprocess_option': opt.resize_or_crop, 'min_box_size': opt.min_box_size,
                       'max_box_size': opt.max_box_size, 'img_to_obj_ratio': opt.contextMargin,
                       'patch_to_obj_ratio': 1.2, 'min_ctx_ratio': 1.2, 'max_ctx_ratio': 1.5}
        self.check_config(self.config)
        self.use_bbox = bool(getattr(opt, 'use_bbox', False))
        self.load_image = bool(getattr(opt, 'load_image', False))
        self.load_raw = bool(getattr(opt, 'load_raw', False))
        split = lambda suffix: os.path.join(opt.dataroot, opt.phase + suffix)
        self.dir_A = split('_A' if opt.label_nc == 0 else '_label')
        self.A_paths = sorted(make_dataset(self.dir_A))
        if (opt.isTrain and not hasattr(opt, 'use_bbox')) or self.load_image:
            self.dir_B = split('_B' if opt.label_nc == 0 else '_img')
            self.B_paths = sorted(make_dataset(self.dir_B))
        self.dir_inst = split('_inst')
        self.inst_paths = sorted(make_dataset(self.dir_inst))
        self.dir_bbox = split('_bbox')
        self.bbox_paths = sorted(make_dataset(self.dir_bbox))
        self.dataset_size = len(self.A_paths)

    def check_config(self, config):
        assert config['preprocess_option'] in ('scale_width', 'none', 'select_region')
        if self.opt.isTrain:
            assert config['img_to_obj_ratio'] < 5.0

    def name(self):
        return 'SegmentationDataset'

    def __len__(self):
        return len(self.A_paths)

    # -- host stage -------------------------------------------------------------------------------
    def get_raw_inputs(self, index):
        with open(self.bbox_paths[index], 'r') as f:
            inst_info = json.load(f)
        raw = {'label': Image.open(self.A_paths[index]), 'label_path': self.A_paths[index],
               'inst': Image.open(self.inst_paths[index]), 'inst_path': self.inst_paths[index]}
        if self.load_image:
            raw['image'] = Image.open(self.B_paths[index]).convert('RGB')
            raw['image_path'] = self.B_paths[index]
        return raw, inst_info

    def host_record(self, index):
        """Everything about sample ``index`` that is decided or read on the host; windows as raw bytes."""
        raw, inst_info = self.get_raw_inputs(index)
        params = get_transform_params(raw['label'].size, inst_info, self.class_of_interest, self.config,
                                      random_crop=self.opt.random_crop)
        ctx = dv.ImageTransform(self.opt, params, 0, False, True, True)
        rec = {'params': params, 'flip': ctx.flipped(), 'label_path': raw['label_path'],
               'inst_path': raw['inst_path']}
        box, rec['size'] = ctx.window_and_size(raw['label'])
        cut = (lambda im: im.crop(box)) if box is not None else (lambda im: im)
        rec['label'] = dv.map_bytes(cut(raw['label']))
        rec['inst'] = dv.map_bytes(cut(raw['inst']))
        if self.load_image:
            rec['image'] = np.ascontiguousarray(np.asarray(cut(raw['image'])))
            rec['image_path'] = raw['image_path']
        if self.load_raw:
            rec['label_raw'], rec['inst_raw'] = dv.map_bytes(raw['label']), dv.map_bytes(raw['inst'])
            rec['image_raw'] = np.ascontiguousarray(np.asarray(raw['image']))
        if self.config['preprocess_option'] == 'select_region':
            obj_box = resample.pil_crop_box(params['crop_object_pos'])
            rec['label_obj'] = dv.map_bytes(raw['label'].crop(obj_box))
            size = rec['size'][0]                          # the label tensor's .size(1): square windows
            ratio = np.random.uniform(low=self.config['min_ctx_ratio'], high=self.config['max_ctx_ratio'])
            rec['input_bbox'] = np.array(params['bbox_in_context'])
            rec['output_bbox'] = np.array(get_soft_bbox(rec['input_bbox'], size, size, ratio))
            cls = params['bbox_cls']
            rec['cls'] = cls if cls is not None else self.opt.label_nc - 1
        return rec

    # -- device stage -----------------------------------------------------------------------------
    def _ids(self, windows, H, W, flips, times_255):
        """Label-like maps as the reference's tensors: 8-bit maps are ToTensor()/255 scaled back by 255 (exactly the
        ids) or left in [0,1]; integer-mode maps stay integers."""
        if windows[0].dtype == np.uint8:
            return 'float' if times_255 else 'unit'
        return 'int32'

    def assemble(self, records):
        """Batch dictionary of the reference (DataLoader default_collate of ``__getitem__`` results), on the device."""
        dev = dv.device()
        B = len(records)
        W, H = records[0]['size']
        if any(r['size'] != (W, H) for r in records):
            raise ValueError('samples of one batch must share the output size (use batchSize 1 for unscaled images)')
        flips = [r['flip'] for r in records]
        region = 'label_obj' in records[0]
        compact = bool(getattr(self.opt, 'compact_labels', False))

        plans, work = [], []

        def ids(key, times_255, size=(H, W), flip=flips):
            wins = [r[key] for r in records]
            plan = dv.MapPlan(wins, size[0], size[1], flip)
            if compact and key == 'label' and plan.windows[0].dtype != np.uint8:
                raise ValueError('--compact_labels needs 8-bit label files (got %s): the ids would be truncated'
                                 % plan.windows[0].dtype)
            out = 'uint8' if (compact and key == 'label') else self._ids(plan.windows, 0, 0, 0, times_255)
            kind, dtype = dv._MAP_OUT[out]
            dst = torch.empty((B, 1, size[0], size[1]), dtype=dtype, device=dev)
            plans.append(plan)
            work.append(lambda base, stream: plan.run(base, dst, kind, stream))
            return dst

        out = {'label': ids('label', True),
               'inst': ids('inst', self.opt.dataloader in _TIMES_255)}
        if 'image' in records[0]:
            photo = dv.PhotoPlan([r['image'] for r in records], H, W, flips, True)
            out['image'] = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
            plans.append(photo)
            work.append(lambda base, stream: photo.run(base, out['image'], stream))
            out['image_path'] = [r['image_path'] for r in records]
        if region:
            out['label_obj'] = ids('label_obj', True)
        if 'label_raw' in records[0]:
            if B != 1:
                raise ValueError('load_raw returns the unscaled files: batchSize must be 1')
            h, w = records[0]['label_raw'].shape
            out['label_raw'] = ids('label_raw', True, (h, w), [False])
            out['inst_raw'] = ids('inst_raw', False, (h, w), [False])
            rawp = dv.PhotoPlan([records[0]['image_raw']], h, w, [False], True)
            out['image_raw'] = torch.empty((1, 3, h, w), dtype=torch.float32, device=dev)
            plans.append(rawp)
            work.append(lambda base, stream: rawp.run(base, out['image_raw'], stream))
        if region:
            masks = dv.MaskPlan([r['input_bbox'] for r in records], [r['output_bbox'] for r in records],
                                [r['cls'] for r in records], [r['params']['bbox_inst_id'] for r in records])
            plans.append(masks)

        def body(base, stream):
            keep = [fn(base, stream) for fn in work]
            if region:
                # the reference builds the region masks from the tensors it RETURNS: label * 255 (16/32-bit label files:
                # ToTensor() leaves the integers, then * 255.0) and, for ade20k, inst * 255 compared with bbox_inst_id
                label = out['label']
                if label.dtype == torch.int32:
                    label = label.float() * 255.0
                elif label.dtype != torch.float32:
                    label = label.float()
                inst = out['inst']
                if inst.dtype == torch.int32 and self.opt.dataloader in _TIMES_255:
                    inst = inst * 255
                names = ('mask_in', 'mask_object_in', 'mask_context_in', 'mask_out', 'mask_object_out',
                         'mask_object_inst')
                out.update(zip(names, masks.run(base, label, inst, stream)))
            return keep
        dv.run_plans(dev, plans, body)

        if out['label'].dtype == torch.int32:                  # a 16/32-bit label file: ToTensor() * 255.0
            out['label'] = out['label'].float() * 255.0
        if out['inst'].dtype == torch.int32 and self.opt.dataloader in _TIMES_255:
            out['inst'] = out['inst'] * 255
        out['label_path'] = [r['label_path'] for r in records]
        out['inst_path'] = [r['inst_path'] for r in records]
        if region:
            out['input_bbox'] = torch.from_numpy(np.stack([r['input_bbox'] for r in records])).to(dev)
            out['output_bbox'] = torch.from_numpy(np.stack([r['output_bbox'] for r in records])).to(dev)
            out['cls'] = torch.tensor([[r['cls']] for r in records], dtype=torch.int64, device=dev)
        return out

    def __getitem__(self, index):
        batch = self.assemble([self.host_record(index)])
        return {k: v[0] for k, v in batch.items()}


def _dataset_with_classes(cls_name, shown_name, classes, where):
    """The two shipped datasets differ from SegmentationDataset only in the classes an object may be sampled from."""
    def initialize(self, opt):
        SegmentationDataset.initialize(self, opt)
        self.class_of_interest = list(classes)
    return type(cls_name, (SegmentationDataset,), {'initialize': initialize, 'name': lambda self: shown_name,
                                                   '__doc__': 'reference %s' % where})


# Cityscapes: person ... bicycle (trainId-free label ids 24-33, data/cityscape_dataset.py:8)
CityscapeDataset = _dataset_with_classes('CityscapeDataset', 'CitiscapeDataset', range(24, 34),
                                         'data/cityscape_dataset.py')
# ADE20K: ids 2..38 without 3, 6, 27, 34 (data/ade20k_dataset.py:8-10)
ADE20KDataset = _dataset_with_classes('ADE20KDataset', 'ADE20KDataset',
                                      [c for c in range(2, 39) if c not in (3, 6, 27, 34)], 'data/ade20k_dataset.py')
DATASETS = {'cityscape': CityscapeDataset, 'ade20k': ADE20KDataset}
