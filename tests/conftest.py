import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        # the CPU oracle (torch/oneDNN) is 8x SLOWER on the GPU box's 128 hardware threads than on 16-32 of them
        # (tools/cpu_thread_sweep.py: C1 step 0.8 s at 16-32 threads, 6-8 s at 128)
        torch.set_num_threads(min(32, torch.get_num_threads()))
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
