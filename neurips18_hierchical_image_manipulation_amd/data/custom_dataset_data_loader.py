"""Batch source of the trainers (reference data/custom_dataset_data_loader.py).

``load_data()`` returns an iterable of batch dictionaries like the reference's ``torch.utils.data.DataLoader``, but in
two stages: worker processes (``--nThreads``) run the host stage of the dataset (decode + window sampling, raw bytes
out), and this process turns each list of records into device tensors on the loader's own HIP stream, one batch ahead
of the consumer, so the upload and the pixel kernels of batch i+1 sit under the training step of batch i.
"""
import torch
import torch.utils.data

from . import device as dv


class BaseDataLoader(object):
    """reference data/base_data_loader.py: keeps ``opt``; ``load_data`` is the subclass's."""

    def initialize(self, opt):
        self.opt = opt

    def load_data(self):
        return None


def CreateDataset(opt):
    from .segmentation_dataset import DATASETS
    if opt.dataloader not in DATASETS:
        raise ValueError('unknown --dataloader %r (%s)' % (opt.dataloader, ' | '.join(sorted(DATASETS))))
    dataset = DATASETS[opt.dataloader]()
    print("dataset [%s] was created" % (dataset.name()))
    dataset.initialize(opt)
    return dataset


class _HostStage(torch.utils.data.Dataset):
    """What the worker processes run: ``dataset.host_record`` (no GPU work in workers)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        return self.dataset.host_record(index)


class _DeviceBatches(object):
    """Iterable over device batches with one batch of look-ahead on the loader stream."""

    def __init__(self, dataset, host_loader):
        self.dataset, self.host_loader = dataset, host_loader

    def __len__(self):
        return len(self.host_loader)

    def _assemble(self, records):
        dev = dv.device()
        side = dv.loader_stream(dev)
        with torch.cuda.stream(side):
            batch = self.dataset.assemble(records)
            done = torch.cuda.Event()
            done.record(side)
        return batch, done

    def __iter__(self):
        pending = None
        for records in self.host_loader:
            ahead = self._assemble(records)
            if pending is not None:
                yield self._hand_over(pending)
            pending = ahead
        if pending is not None:
            yield self._hand_over(pending)

    @staticmethod
    def _hand_over(item):
        batch, done = item
        main = torch.cuda.current_stream()
        main.wait_event(done)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(main)       # allocated on the loader stream, consumed on the trainer's
        return batch


class CustomDatasetDataLoader(BaseDataLoader):
    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        BaseDataLoader.initialize(self, opt)
        self.dataset = CreateDataset(opt)
        self.host_loader = torch.utils.data.DataLoader(
            _HostStage(self.dataset), batch_size=opt.batchSize, shuffle=not opt.serial_batches,
            num_workers=int(opt.nThreads), collate_fn=list)
        self.dataloader = _DeviceBatches(self.dataset, self.host_loader)

    def load_data(self):
        return self.dataloader

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)


def CreateDataLoader(opt):
    """reference data/data_loader.py:3-7: the one entry point the training scripts call."""
    loader = CustomDatasetDataLoader()
    print(loader.name())
    loader.initialize(opt)
    return loader
