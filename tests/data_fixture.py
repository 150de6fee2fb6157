"""A small synthetic dataset in the reference's directory layout (``<phase>_label / _inst / _img / _bbox``), written
with lossless PNGs from a seeded generator: the golden generator (tests/golden/make_golden_data.py, run where the
reference is present) and the tests (run anywhere) create byte-identical files from it."""
import json
import os

import numpy as np
from PIL import Image

# (name, loader, sizes (w,h), interesting classes used, 16-bit instance ids?)
SETS = {
    'city': dict(dataloader='cityscape', sizes=[(512, 256), (512, 256), (384, 192), (512, 256)],
                 classes=[24, 25, 26, 27, 28, 31], label_nc=35, inst16=True),
    'ade': dict(dataloader='ade20k', sizes=[(300, 225), (256, 341), (320, 240), (200, 267)],
                classes=[2, 4, 5, 7, 8, 12, 20], label_nc=49, inst16=False),
}


def write_dataset(root, name, phase='train', seed=7):
    spec = SETS[name]
    rng = np.random.RandomState(seed)
    for sub in ('_label', '_inst', '_img', '_bbox'):
        os.makedirs(os.path.join(root, phase + sub), exist_ok=True)
    for i, (w, h) in enumerate(spec['sizes']):
        label = np.zeros((h, w), np.uint8)
        # piecewise-constant background
        gh, gw = max(h // 32, 1), max(w // 32, 1)
        coarse = rng.randint(0, 20, (gh, gw)).astype(np.uint8)
        label[:] = np.kron(coarse, np.ones((h // gh + 1, w // gw + 1), np.uint8))[:h, :w]
        inst = label.astype(np.int32) if spec['inst16'] else np.zeros((h, w), np.int32)
        objects = {}
        for k in range(int(rng.randint(3, 7))):
            cls = int(spec['classes'][rng.randint(len(spec['classes']))])
            bw, bh = int(rng.randint(12, w // 3)), int(rng.randint(12, h // 2))
            x0, y0 = int(rng.randint(0, w - bw - 1)), int(rng.randint(0, h - bh - 1))
            iid = cls * 1000 + k if spec['inst16'] else 10 + k
            label[y0:y0 + bh, x0:x0 + bw] = cls
            inst[y0:y0 + bh, x0:x0 + bw] = iid
            objects[str(iid)] = {'bbox': [x0, y0, x0 + bw, y0 + bh], 'cls': cls}
        photo = rng.randint(0, 256, (h // 4 + 1, w // 4 + 1, 3)).astype(np.uint8)
        photo = np.kron(photo, np.ones((4, 4, 1), np.uint8))[:h, :w]
        photo = (photo.astype(np.int32) + rng.randint(-20, 21, (h, w, 3))).clip(0, 255).astype(np.uint8)
        stem = 'sample_%02d' % i
        Image.fromarray(label, 'L').save(os.path.join(root, phase + '_label', stem + '.png'))
        if spec['inst16']:
            Image.fromarray(inst.astype(np.uint16)).save(os.path.join(root, phase + '_inst', stem + '.png'))
        else:
            Image.fromarray(inst.astype(np.uint8), 'L').save(os.path.join(root, phase + '_inst', stem + '.png'))
        Image.fromarray(photo, 'RGB').save(os.path.join(root, phase + '_img', stem + '.png'))
        with open(os.path.join(root, phase + '_bbox', stem + '.json'), 'w') as f:
            json.dump({'imgHeight': h, 'imgWidth': w, 'objects': objects}, f, sort_keys=True)
    return spec


def loader_argv(root, name, fine_size, extra=()):
    spec = SETS[name]
    return ['--dataroot', root, '--dataloader', spec['dataloader'], '--resize_or_crop', 'select_region',
            '--fineSize', str(fine_size), '--label_nc', str(spec['label_nc']), '--load_image', '--no_instance',
            '--batchSize', '2', '--nThreads', '0', '--serial_batches'] + list(extra)
