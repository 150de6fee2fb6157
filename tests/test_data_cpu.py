"""Loader host logic on the CPU: the window sampler against windows drawn by the reference itself, the
resampling tables against Pillow, the file listing.  No GPU, no compute calls into the library."""
import json
import os
import random

import numpy as np
import pytest
from PIL import Image

from neurips18_hierchical_image_manipulation_amd.data import base_dataset as bd
from neurips18_hierchical_image_manipulation_amd.data import resample
from neurips18_hierchical_image_manipulation_amd.data.image_folder import make_dataset

import data_fixture as fx

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_sampler_matches_reference_windows():
    """get_transform_params + the context ratio draw + get_soft_bbox, 240 seeded cases: every number equal (floats
    bit for bit: JSON round-trips doubles exactly) to what the reference's functions returned
    (tests/golden/make_golden_data.py)."""
    with open(os.path.join(GOLD, 'data_sampler.json')) as f:
        answers = json.load(f)
    assert len(answers) == 240
    seen_flip = seen_bg = seen_bbox = 0
    for a in answers:
        c = a['case']
        random.seed(c['seed'])
        np.random.seed(c['seed'])
        p = bd.get_transform_params(tuple(c['full_size']), c['inst_info'], c['class_of_interest'], c['config'],
                                    bbox=c['bbox'], random_crop=c['random_crop'])
        ratio = np.random.uniform(low=1.2, high=1.5)
        soft = bd.get_soft_bbox(np.array(p['bbox_in_context']), c['config']['fineSize'], c['config']['fineSize'], ratio)
        got = json.loads(json.dumps(p, default=float))
        assert got == a['params'], (c['seed'], got, a['params'])
        assert [int(v) for v in soft] == a['soft_bbox']
        seen_flip += bool(p['flip'])
        seen_bg += p['bbox_cls'] is None
        seen_bbox += c['bbox'] is not None
    assert seen_flip > 20 and seen_bg > 20 and seen_bbox > 10      # the cases reach every branch


def _fixed_point_resize(img, out_w, out_h):
    """Pillow's two passes on the tables of resample.py, in numpy integers (what the device kernels do)."""
    h, w, _ = img.shape
    one = 1 << (resample.PRECISION_BITS - 1)
    cur = img.astype(np.int64)
    for axis, (n_in, n_out) in ((1, (w, out_w)), (0, (h, out_h))):
        first, count, weights, _ = resample.bicubic_tables(n_in, n_out)
        cur = np.moveaxis(cur, axis, 0)
        nxt = np.empty((n_out,) + cur.shape[1:], np.int64)
        for i in range(n_out):
            acc = np.full(cur.shape[1:], one, np.int64)
            for k in range(count[i]):
                acc += cur[first[i] + k] * int(weights[i, k])
            nxt[i] = np.clip(acc >> resample.PRECISION_BITS, 0, 255)
        cur = np.moveaxis(nxt, 0, axis)
    return cur.astype(np.uint8)


@pytest.mark.parametrize('seed', range(6))
def test_resample_tables_reproduce_pillow(seed):
    rng = np.random.RandomState(seed)
    for _ in range(5):
        h, w = int(rng.randint(3, 400)), int(rng.randint(3, 400))
        ow, oh = int(rng.choice([17, 64, 96, 128, 300])), int(rng.choice([17, 64, 96, 128, 300]))
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(_fixed_point_resize(img, ow, oh), ref), (h, w, ow, oh)
        lab = rng.randint(0, 256, (h, w)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(lab).resize((ow, oh), Image.NEAREST))
        got = lab[resample.nearest_table(h, oh)][:, resample.nearest_table(w, ow)]
        assert np.array_equal(got, ref), (h, w, ow, oh)
        ids = rng.randint(0, 40000, (h, w)).astype(np.uint16)        # mode I;16 takes another path inside Pillow
        ref = np.asarray(Image.fromarray(ids).resize((ow, oh), Image.NEAREST))
        got = ids[resample.nearest_table(h, oh, True)][:, resample.nearest_table(w, ow, True)]
        assert np.array_equal(got, ref), (h, w, ow, oh)


def test_vectorised_tables_equal_the_scalar_transcription():
    rng = np.random.RandomState(4)
    for _ in range(400):
        a, b = int(rng.randint(1, 2100)), int(rng.randint(1, 600))
        x, y = resample.bicubic_tables(a, b), resample.bicubic_tables_scalar(a, b)
        assert x[3] == y[3] and all(np.array_equal(p, q) for p, q in zip(x[:3], y[:3])), (a, b)


def test_identity_tables():
    first, count, weights, ksize = resample.bicubic_tables(40, 40)
    one = 1 << resample.PRECISION_BITS
    for i in range(40):
        w = dict((first[i] + k, weights[i, k]) for k in range(count[i]) if weights[i, k])
        assert w == {i: one}
    assert list(resample.nearest_table(40, 40)) == list(range(40))
    assert resample.pil_crop_box((0.5, 1.5, 2.5, 3.49)) == (0, 2, 2, 3)     # half to even, like Image.crop


def test_file_listing_and_fixture(tmp_path):
    root = str(tmp_path)
    fx.write_dataset(root, 'city')
    names = sorted(make_dataset(os.path.join(root, 'train_label')))
    assert [os.path.basename(n) for n in names] == ['sample_%02d.png' % i for i in range(4)]
    assert len(make_dataset(os.path.join(root, 'train_bbox'))) == 4       # 'json' is a target extension
    with pytest.raises(AssertionError):
        make_dataset(os.path.join(root, 'missing'))


def test_windows_stay_inside_image():
    random.seed(3)
    for _ in range(200):
        w, h = random.randint(100, 2048), random.randint(100, 1024)
        bw, bh = random.randint(5, w // 2), random.randint(5, h // 2)
        x0, y0 = random.randint(0, w - bw - 1), random.randint(0, h - bh - 1)
        for margin in (1.2, 2.0, 3.0):
            x, y, x1, y1 = bd.crop_box_with_margin([x0, y0, x0 + bw, y0 + bh], w, h, margin, True)
            assert 0 <= x <= x1 <= w - 1 and 0 <= y <= y1 <= h - 1


def test_image_folder_dataset_and_normalize(tmp_path):
    """data/image_folder.py:33-64 (default_loader, ImageFolder) and base_dataset.normalize() (:323-324)."""
    import numpy as np
    import torch
    from neurips18_hierchical_image_manipulation_amd.data.image_folder import ImageFolder, default_loader
    root = str(tmp_path)
    fx.write_dataset(root, 'city')
    folder = os.path.join(root, 'train_img')
    ds = ImageFolder(folder, return_paths=True)
    assert len(ds) == len(make_dataset(folder)) > 0
    img, path = ds[0]
    assert path == ds.imgs[0] and img.mode == 'RGB' and img.size == default_loader(path).size
    to_tensor = lambda im: torch.from_numpy(np.asarray(im).transpose(2, 0, 1).copy()).float() / 255   # noqa: E731
    ds = ImageFolder(folder, transform=lambda im: bd.normalize()(to_tensor(im)))
    t = ds[1]
    assert t.shape[0] == 3 and float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    assert torch.equal(t, (to_tensor(default_loader(ds.imgs[1])) - 0.5) / 0.5)
    os.makedirs(os.path.join(root, 'empty'))
    with pytest.raises(RuntimeError, match='Found 0 images'):
        ImageFolder(os.path.join(root, 'empty'))
