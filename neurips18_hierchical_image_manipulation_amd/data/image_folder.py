"""File listing of a dataset split (reference data/image_folder.py:11-31) and its generic folder dataset (:33-64)."""
import os

import torch.utils.data as data

TGK_EXTENSIONS = ('.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP', '.tiff', 'json')


def is_target_file(filename):
    return filename.endswith(TGK_EXTENSIONS)


def make_dataset(dir):
    """Every image / json file below ``dir`` in ``sorted(os.walk)`` order (callers sort the result again)."""
    if not os.path.isdir(dir):
        raise AssertionError('%s is not a valid directory' % dir)
    found = []
    for root, _, names in sorted(os.walk(dir)):
        found.extend(os.path.join(root, n) for n in names if is_target_file(n))
    return found


def default_loader(path):
    from PIL import Image
    return Image.open(path).convert('RGB')


class ImageFolder(data.Dataset):
    """Every target file below ``root`` as one sample: ``transform(loader(path))`` [, path] (reference :37-64; none of the
    reference's own loaders use it -- kept for user code that does).  ``transform`` may be the device stage returned by
    ``base_dataset.get_raw_transform_fn``.  Public attributes as upstream: root, imgs, transform, return_paths, loader."""

    def __init__(self, root, transform=None, return_paths=False, loader=default_loader):
        self.root, self.transform, self.return_paths, self.loader = root, transform, return_paths, loader
        self.imgs = make_dataset(root)
        if not self.imgs:
            raise RuntimeError('Found 0 images in: %s\nSupported image extensions are: %s' % (root, ','.join(TGK_EXTENSIONS)))

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        sample = self.loader(self.imgs[index])
        sample = sample if self.transform is None else self.transform(sample)
        return (sample, self.imgs[index]) if self.return_paths else sample
