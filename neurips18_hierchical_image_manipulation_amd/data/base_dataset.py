"""Crop-window sampler and box helpers of the loader (reference data/base_dataset.py).

Host logic only: which object, which windows, which flip.  The draws from ``random`` / ``numpy.random`` happen in the
reference's order and the float expressions keep its association, so a seeded run selects the same windows
(tests/test_data_cpu.py pins this against windows produced by the reference itself).  The pixel work the reference does
next with PIL / torchvision is done on the device by :mod:`.device`.
"""
import random

import numpy as np
import torch
import torch.utils.data as data

from . import device as _device


class BaseDataset(data.Dataset):
    def name(self):
        return 'BaseDataset'

    def initialize(self, opt):
        pass


# ------------------------------------------------------------------------------------------------
# windows (reference :21-71, 73-140)
# ------------------------------------------------------------------------------------------------
def crop_box_with_margin(box, w, h, margin, random_crop=True):
    """Square window of side ``margin`` x the box's longer edge (at most the image's shorter edge) about the box centre,
    jittered by up to a quarter of the slack, pushed back inside the image (reference :117-140)."""
    xmin, ymin, xmax, ymax = box[0], box[1], box[2], box[3]
    longer = max(xmax - xmin, ymax - ymin)
    shorter_image_edge = min(w, h)
    side = min(longer * margin, shorter_image_edge)
    slack = min(longer * (margin - 1.0), shorter_image_edge)
    corner = []
    for lo, hi in ((xmin, xmax), (ymin, ymax)):          # x first, then y: two draws in this order
        start = (hi + lo) * 0.5 - side * 0.5
        if random_crop:
            start = start + (random.random() - 0.5) * slack / 2.0
        corner.append(start)
    x0 = max(min(max(0, corner[0]), w - side - 1), 0)
    y0 = max(min(max(0, corner[1]), h - side - 1), 0)
    return [x0, y0, min(x0 + side, w - 1), min(y0 + side, h - 1)]


def sample_fg_from_full(inst_info, class_of_interest, min_box_size):
    """One annotated object of an interesting class whose longer edge reaches ``min_box_size`` (reference :143-181)."""
    eligible = []
    for key, obj in inst_info.items():
        if obj['cls'] not in class_of_interest:
            continue
        x0, y0, x1, y1 = obj['bbox'][:4]
        if max(x1 - x0, y1 - y0) < min_box_size:
            continue
        eligible.append({'bbox': [x0, y0, x1, y1], 'cls': obj['cls'], 'inst_id': int(key)})
    if not eligible:
        return None
    return eligible[np.random.randint(len(eligible))]


def sample_bg_from_full(min_box_size, max_box_size, w, h):
    """A random box with no class (reference :183-201): four ``np.random.randint`` draws, x0, y0, x1, y1."""
    x0 = np.random.randint(0, w - min_box_size - 1)
    y0 = np.random.randint(0, h - min_box_size - 1)
    x1 = np.random.randint(x0 + min_box_size, min(x0 + max_box_size, w - 1))
    y1 = np.random.randint(y0 + min_box_size, min(y0 + max_box_size, h - 1))
    return {'bbox': [x0, y0, x1, y1], 'cls': None, 'inst_id': None}


def get_bbox_in_context(bbox_selected, crop_pos, target_size):
    """The object box in the pixel grid of the resized image window (reference :204-236)."""
    x0, y0, x1, y1 = bbox_selected['bbox'][:4]
    sx = 1.0 * target_size / (crop_pos[2] - crop_pos[0])
    sy = 1.0 * target_size / (crop_pos[3] - crop_pos[1])
    left = (x0 - crop_pos[0]) * sx
    top = (y0 - crop_pos[1]) * sy
    right = left + (x1 - x0) * sx
    bottom = top + (y1 - y0) * sy
    return [max(int(left), 0), max(int(top), 0), min(int(right), target_size), min(int(bottom), target_size)]


def crop_single_object(inst_info, class_of_interest, w, h, prob_bg, img_to_obj_ratio, patch_to_obj_ratio,
                       min_box_size, max_box_size, target_size, flip, random_crop=True):
    """reference :96-115"""
    chosen = sample_fg_from_full(inst_info['objects'], class_of_interest, min_box_size)
    want_background = random.random() < prob_bg            # drawn whether or not an object was found
    if want_background or chosen is None:
        chosen = sample_bg_from_full(min_box_size, max_box_size, w, h)
    crop_pos = crop_box_with_margin(chosen['bbox'], w, h, img_to_obj_ratio, random_crop)
    crop_object = crop_box_with_margin(chosen['bbox'], w, h, patch_to_obj_ratio, random_crop)
    in_context = get_bbox_in_context(chosen, crop_pos, target_size)
    if flip:
        in_context[0], in_context[2] = target_size - in_context[2], target_size - in_context[0]
    return crop_pos, crop_object, in_context, chosen['cls'], chosen['inst_id']


def crop_single_object_with_bbox(bbox, w, h, img_to_obj_ratio, patch_to_obj_ratio, target_size, random_crop=True):
    """A caller-supplied box (testing / editing), reference :73-94: no flip mirror, instance id 0."""
    crop_pos = crop_box_with_margin(bbox['bbox'], w, h, img_to_obj_ratio, random_crop)
    crop_object = crop_box_with_margin(bbox['bbox'], w, h, patch_to_obj_ratio, random_crop)
    if target_size is not None:
        in_context = get_bbox_in_context(bbox, crop_pos, target_size)
    else:
        b = bbox['bbox']
        left, top = b[0] - crop_pos[0], b[1] - crop_pos[1]
        in_context = [left, top, left + b[2] - b[0], top + b[3] - b[1]]
    return crop_pos, crop_object, in_context, bbox['cls'], 0


def get_transform_params(full_size, inst_info=None, class_of_interest=None, config=None, bbox=None,
                         random_crop=True):
    """The windows of one training sample (reference :21-71): ``crop_pos`` image window, ``crop_object_pos`` tight object
    window (both in full-image pixels), ``bbox_in_context`` object box inside the resized image window, class, instance."""
    flip = random.random() < config['prob_flip']
    full_w, full_h = full_size
    if bbox is None:
        picked = crop_single_object(inst_info, class_of_interest, full_w, full_h, config['prob_bg'],
                                    config['img_to_obj_ratio'], config['patch_to_obj_ratio'],
                                    config['min_box_size'], config['max_box_size'], config['fineSize'], flip,
                                    random_crop)
    else:
        picked = crop_single_object_with_bbox(bbox, full_w, full_h, config['img_to_obj_ratio'],
                                              config['patch_to_obj_ratio'], config['fineSize'], random_crop)
    crop_pos, crop_object, in_context, cls, inst_id = picked
    return {'crop_pos': crop_pos, 'flip': flip, 'crop_object_pos': crop_object, 'bbox_in_context': in_context,
            'bbox_cls': cls, 'bbox_inst_id': inst_id}


def get_soft_bbox(input_tuple, ow, oh, ratio=1.5):
    """The box grown by ``ratio`` about its centre, clipped to (ow, oh) (reference :325-339)."""
    half_w = (input_tuple[2] - input_tuple[0]) * ratio / 2
    half_h = (input_tuple[3] - input_tuple[1]) * ratio / 2
    cx = (input_tuple[0] + input_tuple[2]) / 2
    cy = (input_tuple[1] + input_tuple[3]) / 2
    return [max(int(cx - half_w), 0), max(int(cy - half_h), 0), min(int(cx + half_w), ow), min(int(cy + half_h), oh)]


def transform_box(opt, params, inst_info):
    """Object boxes carried through the scale / crop / flip of the image (reference :270-320)."""
    mode = opt.resize_or_crop
    full_h, full_w = inst_info['imgHeight'], inst_info['imgWidth']
    out = {}
    for key, obj in inst_info['objects'].items():
        box = obj['bbox']
        if 'scale_width' in mode:
            box = [v * (1.0 * opt.loadSize / full_w) for v in box]
            extent = opt.loadSize
        elif 'scale_minaxis' in mode:
            box = [v * (1.0 * opt.loadSize / min(full_w, full_h)) for v in box]
            extent = opt.loadSize
        if 'crop' in mode:
            cx, cy = params['crop_pos']
            extent = opt.fineSize
            if box[2] <= cx or box[3] <= cy or box[0] >= cx + extent or box[1] >= cy + extent:
                continue
            box[0], box[1] = max(box[0] - cx, 0), max(box[1] - cy, 0)
            box[2], box[3] = min(box[2] - cx, extent - 1), min(box[3] - cy, extent - 1)
            if box[2] - box[0] < 1 or box[3] - box[1] < 1:
                continue
        if params['flip']:
            box = [extent - box[2], box[1], extent - box[0], box[3]]
        out[key] = {'bbox': box, 'cls': obj['cls']}
    return out


# ------------------------------------------------------------------------------------------------
# pixel transforms: executed by the device stage
# ------------------------------------------------------------------------------------------------
NEAREST, BICUBIC = 0, 3      # PIL.Image.NEAREST / PIL.Image.BICUBIC


def get_transform_fn(opt, params, method=BICUBIC, normalize=True, is_context=True, resize=True):
    """PIL image -> device tensor, the composition of reference :243-268 (window / scale, flip, ToTensor, Normalize)."""
    return _device.ImageTransform(opt, params, method, normalize, is_context, resize)


def get_raw_transform_fn(normalize=True):
    """ToTensor (+ Normalize) of the untouched image (reference :236-241)."""
    return _device.ImageTransform(None, None, BICUBIC, normalize, True, False)


class _Normalize(object):
    """transforms.Normalize((0.5,)*3, (0.5,)*3) on a (3,H,W) tensor (host or device): (t - 0.5) / 0.5 per channel."""
    mean = std = (0.5, 0.5, 0.5)

    def __call__(self, tensor):
        return (tensor - 0.5) / 0.5


def normalize():
    """Reference :323-324 (unused by its own loaders)."""
    return _Normalize()


def get_masked_image(image_tensor, bbox_tensor, cls2fill=0):
    """(mask, mask*image, (1-mask)*image + mask*cls2fill) for one (C,H,W) map and a (wmin,hmin,wmax,hmax) box
    (reference :342-357), on the device (ops.get_masked_image -> him_masked_image)."""
    from .. import ops
    dev = _device.device()
    img = image_tensor.to(dev, torch.float32)[None]
    box = torch.tensor([[int(v) for v in bbox_tensor[:4]]], dtype=torch.float32, device=dev)
    mask, obj, ctx = ops.get_masked_image(img, box, float(cls2fill))
    return mask[0], obj[0], ctx[0]
