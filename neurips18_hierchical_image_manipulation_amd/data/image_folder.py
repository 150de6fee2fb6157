"""File listing of a dataset split (reference data/image_folder.py:11-31)."""
import os

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
