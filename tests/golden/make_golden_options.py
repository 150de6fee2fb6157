"""tests/golden/option_defaults.json: the flag contract of the mask2image path -- every option the REAL reference's
parsers define (options/mask2image_base_options.py, mask2image_train_options.py, mask2image_test_options.py), with its
kind (value type or store_true flag) and default, read from the live argparse objects of the imported reference.
Build container only (oracle/ref_shim.py).  Data only: names, type names, default values.

    python tests/golden/make_golden_options.py
"""
import argparse
import io
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from oracle import ref_shim                                              # noqa: E402


def table(cls):
    o = cls()
    stdout, sys.stdout = sys.stdout, io.StringIO()
    try:
        o.initialize()
    finally:
        sys.stdout = stdout
    out = {}
    for a in o.parser._actions:
        if isinstance(a, argparse._HelpAction):
            continue
        if isinstance(a, argparse._StoreTrueAction):
            kind, default = 'flag', False
        else:
            kind = (a.type or str).__name__
            default = a.default
            if isinstance(default, float) and math.isinf(default):
                default = 'inf'
        out[a.dest] = dict(kind=kind, default=default)
    return dict(isTrain=bool(o.isTrain), options=out)


if __name__ == '__main__':
    ref_shim.install()
    from options.mask2image_train_options import MaskToImageTrainOptions
    from options.mask2image_test_options import MaskToImageTestOptions
    from options.box2mask_train_options import BoxToMaskTrainOptions
    from options.box2mask_test_options import BoxToMaskTestOptions
    res = dict(note='flag names, kinds and defaults of the reference parsers (tests/golden/make_golden_options.py)',
               train=table(MaskToImageTrainOptions), test=table(MaskToImageTestOptions),
               box2mask_train=table(BoxToMaskTrainOptions), box2mask_test=table(BoxToMaskTestOptions))
    with open(os.path.join(HERE, 'option_defaults.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print({k: len(v['options']) for k, v in res.items() if k != 'note'})
