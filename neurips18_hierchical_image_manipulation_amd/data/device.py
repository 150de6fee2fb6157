"""Device stage of the loader: one staging copy + four kernel launches per batch (csrc/him_data.hip).

The host hands over, per sample, the raw bytes of its crop windows (Pillow decode + ``Image.crop`` only) and the
sampler's numbers.  Everything after that in the reference -- ``Image.resize`` NEAREST / BICUBIC, ``FLIP_LEFT_RIGHT``,
``ToTensor``, ``Normalize`` (data/base_dataset.py:243-268), the two ``get_masked_image`` calls and the instance mask
(data/segmentation_dataset.py:86-131) -- runs on the GPU for the whole batch, on its own stream, and the batch is
returned as device tensors in the layout the trainers take.  There is no host implementation of the pixel work.
"""
import numpy as np
import torch

from .._cabi import lib
from . import resample

_STREAMS = {}


def device():
    if not torch.cuda.is_available():
        raise RuntimeError('the loader\'s device stage needs a GPU (no host fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def loader_stream(dev):
    s = _STREAMS.get(dev)
    if s is None:
        s = _STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


_KIND_OF_DTYPE = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.int32): 2}


def map_bytes(img):
    """A label / instance PIL image as a 2-D integer array in one of the three layouts the gather kernel reads."""
    a = np.asarray(img)
    if a.ndim != 2:
        raise ValueError('label / instance maps must be single-channel images, got mode %s' % img.mode)
    if a.dtype == np.uint8 or a.dtype == np.uint16:
        return np.ascontiguousarray(a)
    if a.dtype == np.bool_:
        return np.ascontiguousarray(a.astype(np.uint8) * 255)      # ToTensor of a mode-'1' image
    if a.dtype in (np.int32, np.int16):
        return np.ascontiguousarray(a.astype(np.int32))
    raise ValueError('unsupported map mode %s' % img.mode)


class _Staging(object):
    """Sections laid out with 16-byte alignment; written straight into one pinned buffer and uploaded with one copy."""

    def __init__(self):
        self.parts, self.size = [], 0

    def add(self, array):
        a = np.ascontiguousarray(array)
        at = self.size + ((-self.size) % 16)
        self.parts.append((at, a))
        self.size = at + a.nbytes
        return at

    def upload(self, dev):
        host = torch.empty(max(self.size, 16), dtype=torch.uint8, pin_memory=True)
        view = host.numpy()
        for at, a in self.parts:
            view[at:at + a.nbytes] = a.reshape(-1).view(np.uint8)
        return host.to(dev, non_blocking=True)


class MapPlan(object):
    """NEAREST resize of ``windows[b]`` (2-D integer arrays, possibly of different sizes) to (H, W), flipped per sample."""

    def __init__(self, windows, H, W, flips):
        self.windows, self.H, self.W, self.flips = windows, H, W, flips
        kinds = {_KIND_OF_DTYPE[w.dtype] for w in windows}
        self.sixteen = [w.dtype == np.uint16 for w in windows]   # mode I;16: Pillow's generic-transform path
        if len(kinds) != 1:
            windows = self.windows = [w.astype(np.int32) for w in windows]
            kinds = {2}
        self.kind = kinds.pop()

    def stage(self, st):
        B = len(self.windows)
        xt = np.empty((B, self.W), np.int32)
        yt = np.empty((B, self.H), np.int32)
        offs, pitch = np.empty(B, np.int64), np.empty(B, np.int32)
        for b, w in enumerate(self.windows):
            h_in, w_in = w.shape
            x = resample.nearest_table(w_in, self.W, self.sixteen[b])
            xt[b] = x[::-1] if self.flips[b] else x
            yt[b] = resample.nearest_table(h_in, self.H, self.sixteen[b])
            offs[b], pitch[b] = st.add(w), w_in
        self.at = (st.add(offs), st.add(pitch), st.add(xt), st.add(yt))

    def run(self, base, dst, dst_kind, stream):
        o, p, x, y = (base + a for a in self.at)
        lib.him_data_nearest(base, o, p, x, y, self.kind, dst.data_ptr(), dst_kind, len(self.windows), self.H, self.W,
                             stream)


class PhotoPlan(object):
    """Antialiased BICUBIC resize of RGB byte windows (h, w, 3) to (H, W), flip, /255, optional (t-.5)/.5."""

    def __init__(self, windows, H, W, flips, normalize):
        self.windows, self.H, self.W, self.flips, self.normalize = windows, H, W, flips, normalize

    def stage(self, st):
        B = len(self.windows)
        tabs = [(resample.bicubic_tables(w.shape[1], self.W), resample.bicubic_tables(w.shape[0], self.H))
                for w in self.windows]
        self.ksx = max(t[0][3] for t in tabs)
        self.ksy = max(t[1][3] for t in tabs)
        fx, nx = np.zeros((B, self.W), np.int32), np.zeros((B, self.W), np.int32)
        fy, ny = np.zeros((B, self.H), np.int32), np.zeros((B, self.H), np.int32)
        wx = np.zeros((B, self.W, self.ksx), np.int32)
        wy = np.zeros((B, self.H, self.ksy), np.int32)
        offs, pitch, rows = np.empty(B, np.int64), np.empty(B, np.int32), np.empty(B, np.int32)
        for b, (w, (tx, ty)) in enumerate(zip(self.windows, tabs)):
            fx[b], nx[b], wx[b, :, :tx[3]] = tx[0], tx[1], tx[2]
            fy[b], ny[b], wy[b, :, :ty[3]] = ty[0], ty[1], ty[2]
            offs[b], pitch[b], rows[b] = st.add(w), w.shape[1], w.shape[0]
        self.maxrows = int(rows.max())
        self.at = tuple(st.add(a) for a in (offs, pitch, rows, fx, nx, wx, fy, ny, wy,
                                            np.asarray(self.flips, np.int32)))

    def run(self, base, dst, stream):
        B = len(self.windows)
        o, p, r, fx, nx, wx, fy, ny, wy, fl = (base + a for a in self.at)
        tmp = torch.empty((B, self.maxrows, self.W, 3), dtype=torch.uint8, device=dst.device)
        lib.him_data_bicubic_h(base, o, p, r, fx, nx, wx, self.ksx, tmp.data_ptr(), self.maxrows, B, self.W, stream)
        lib.him_data_bicubic_v(tmp.data_ptr(), self.maxrows, fy, ny, wy, self.ksy, fl, dst.data_ptr(),
                               1 if self.normalize else 0, B, self.H, self.W, stream)
        return tmp


def run_plans(dev, plans, body):
    """Stage every plan into one buffer, upload it with one copy and call ``body(base_ptr, stream_handle)``; everything
    is queued on the caller's current stream (the loader makes that its own stream, see CustomDatasetDataLoader)."""
    st = _Staging()
    for p in plans:
        p.stage(st)
    buf = st.upload(dev)
    return body(buf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)


_MAP_OUT = {'float': (0, torch.float32), 'unit': (1, torch.float32), 'uint8': (2, torch.uint8), 'int32': (3, torch.int32)}


def resize_maps(windows, H, W, flips=None, out='float', dev=None):
    """(B,1,H,W) NEAREST resize of integer maps.  ``out``: 'float' (the ids as fp32 = ToTensor()*255 of an 8-bit map),
    'unit' (ids/255 = ToTensor), 'uint8' (compact ids), 'int32' (ToTensor of an integer-mode map)."""
    dev = dev or device()
    B = len(windows)
    plan = MapPlan(list(windows), H, W, flips or [False] * B)
    kind, dtype = _MAP_OUT[out]
    dst = torch.empty((B, 1, H, W), dtype=dtype, device=dev)
    run_plans(dev, [plan], lambda base, stream: plan.run(base, dst, kind, stream))
    return dst


def resize_photos(windows, H, W, flips=None, normalize=True, dev=None):
    """(B,3,H,W) fp32: Pillow's BICUBIC resize of (h,w,3) byte windows, flip, ToTensor, Normalize(.5,.5)."""
    dev = dev or device()
    B = len(windows)
    plan = PhotoPlan([np.ascontiguousarray(w) for w in windows], H, W, flips or [False] * B, normalize)
    dst = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    run_plans(dev, [plan], lambda base, stream: plan.run(base, dst, stream))
    return dst


class MaskPlan(object):
    """The six mask tensors of ``preprocess_cropping`` (reference data/segmentation_dataset.py:86-131) for a batch:
    boxes as (B,4) integer (wmin,hmin,wmax,hmax), ``fill`` (B) = class written into the input window, ``inst_ids`` =
    the selected instance id or None per sample."""

    def __init__(self, boxes_in, boxes_out, fill, inst_ids):
        B = len(inst_ids)
        self.boxes = np.concatenate([np.asarray(boxes_in, np.int32).reshape(B, 4),
                                     np.asarray(boxes_out, np.int32).reshape(B, 4)], axis=1)
        self.fill = np.asarray(fill, np.float32).reshape(B)
        self.ids = np.array([[0, 0] if i is None else [1, int(i)] for i in inst_ids], np.int32)

    def stage(self, st):
        self.at = (st.add(self.boxes), st.add(self.fill), st.add(self.ids))

    def run(self, base, label, inst, stream):
        B, _, H, W = label.shape
        if inst is not None and inst.dtype not in (torch.float32, torch.int32):
            raise TypeError('instance map must be fp32 or int32 on the device')
        outs = [torch.empty((B, 1, H, W), dtype=torch.float32, device=label.device) for _ in range(6)]
        lib.him_data_region_masks(label.data_ptr(), 0 if inst is None else inst.data_ptr(),
                                  0 if inst is None or inst.dtype == torch.float32 else 1,
                                  base + self.at[0], base + self.at[1], base + self.at[2],
                                  *[t.data_ptr() for t in outs[:5]], 0 if inst is None else outs[5].data_ptr(),
                                  B, H, W, stream)
        if inst is None:
            outs[5].zero_()
        return outs


def region_masks(label, inst, boxes_in, boxes_out, fill, inst_ids):
    """(mask_in, mask_object_in, mask_context_in, mask_out, mask_object_out, mask_object_inst), each (B,1,H,W)."""
    plan = MaskPlan(boxes_in, boxes_out, fill, inst_ids)
    return run_plans(label.device, [plan], lambda base, stream: plan.run(base, label, inst, stream))


class ImageTransform(object):
    """Callable PIL image -> device tensor, the object ``get_transform_fn`` returns (reference
    data/base_dataset.py:243-268): window ('select_region') or width scaling ('scale_width') or power-of-two rounding
    ('none' with a local enhancer), flip, ToTensor, Normalize."""

    def __init__(self, opt, params, method, normalize, is_context, resize):
        self.opt, self.params, self.method, self.normalize = opt, params, method, normalize
        self.is_context, self.resize = is_context, resize

    def window_and_size(self, img):
        """(crop box in image pixels or None, (out_w, out_h)) for this image."""
        opt, (w, h) = self.opt, img.size
        if opt is None or opt.resize_or_crop not in ('select_region', 'none', 'scale_width'):
            if opt is not None:
                raise AssertionError('resize_or_crop must be select_region, none or scale_width')
            return None, (w, h)
        if opt.resize_or_crop == 'scale_width':
            return None, ((w, h) if w == opt.loadSize else (opt.loadSize, int(opt.loadSize * h / w)))
        if opt.resize_or_crop == 'select_region':
            box = resample.pil_crop_box(self.params['crop_pos' if self.is_context else 'crop_object_pos'])
            if self.resize:
                return box, (opt.fineSize, opt.fineSize)
            return box, (box[2] - box[0], box[3] - box[1])
        if opt.netG == 'local':
            base = float(2 ** opt.n_downsample_global) * (2 ** opt.n_local_enhancers)
            return None, (int(round(w / base) * base), int(round(h / base) * base))
        return None, (w, h)

    def flipped(self):
        opt = self.opt
        return bool(opt is not None and opt.isTrain and not opt.no_flip and self.params['flip'])

    def __call__(self, img):
        box, (ow, oh) = self.window_and_size(img)
        if box is not None:
            img = img.crop(box)
        if self.method == 0 or img.mode in ('1', 'P', 'L', 'I', 'I;16'):
            if self.method != 0 and img.mode not in ('1', 'P'):
                raise ValueError('only RGB photographs are resampled with BICUBIC here, got mode %s' % img.mode)
            a = map_bytes(img)
            out = 'int32' if a.dtype == np.int32 or a.dtype == np.uint16 else 'unit'
            t = resize_maps([a], oh, ow, [self.flipped()], out)[0]
            if self.normalize and out == 'unit':
                t = (t - 0.5) / 0.5
            return t
        a = np.asarray(img.convert('RGB'))
        return resize_photos([a], oh, ow, [self.flipped()], self.normalize)[0]
