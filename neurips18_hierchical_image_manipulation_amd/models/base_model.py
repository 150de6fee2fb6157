"""``BaseModel`` with the reference's stub surface and checkpoint helpers
(reference ``models/base_model.py:8-131``): same file naming ``<epoch>_net_<label>.pth`` and the same
strict -> subset -> shape-matched fallback when loading."""
import os

import torch


class BaseModel(torch.nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)

    # the reference wraps the training model in DataParallel and the driver talks to ``model.module``;
    # here every rank owns a whole model, so ``module`` is the model itself.
    @property
    def module(self):
        return self

    def name(self):
        return 'BaseModel'

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def test(self):
        pass

    def get_image_paths(self):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        return self.input

    def get_current_errors(self):
        return {}

    def save(self, label):
        pass

    def _path(self, network_label, epoch_label, save_dir=''):
        return os.path.join(save_dir or self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))

    def save_network(self, network, network_label, epoch_label, gpu_ids=None):
        os.makedirs(self.save_dir, exist_ok=True)
        sd = {k: v.detach().cpu().clone() for k, v in network.state_dict().items()}
        torch.save(sd, self._path(network_label, epoch_label))

    def save_network_dict(self, network_dict, optimizer, network_label, epoch_label, gpu_ids=None):
        os.makedirs(self.save_dir, exist_ok=True)
        out = {'network': {k: {n: t.detach().cpu().clone() for n, t in v.state_dict().items()}
                           for k, v in network_dict.items()},
               'optimizer': optimizer.state_dict()}
        torch.save(out, self._path(network_label, epoch_label))

    def delete_network(self, network_label, epoch_label, gpu_ids=None):
        p = self._path(network_label, epoch_label)
        if os.path.isfile(p):
            os.remove(p)

    def load_network(self, network, network_label, epoch_label, save_dir=''):
        path = self._path(network_label, epoch_label, save_dir)
        if not os.path.isfile(path):
            print('%s not exists yet!' % path)
            if network_label == 'G':
                raise RuntimeError('Generator must exist!')
            return
        self._load_with_fallback(network, torch.load(path, map_location='cpu'), network_label)

    @staticmethod
    def _load_with_fallback(network, loaded, network_label):
        """strict -> the checkpoint's subset of our keys -> shape-matched merge (reference models/base_model.py:76-110)."""
        try:
            network.load_state_dict(loaded)
            return
        except RuntimeError:
            pass
        own = network.state_dict()
        subset = {k: v for k, v in loaded.items() if k in own}
        try:
            network.load_state_dict(subset)
            print('Pretrained network %s has excessive layers; Only loading layers that are used' % network_label)
            return
        except RuntimeError:
            pass
        print('Pretrained network %s has fewer layers; The following are not initialized:' % network_label)
        merged, missing = dict(own), set()
        for k, v in loaded.items():
            if k in own and v.size() == own[k].size():
                merged[k] = v
        for k, v in own.items():
            if k not in loaded or v.size() != loaded[k].size():
                missing.add(k.split('.')[0])
        print(sorted(missing))
        network.load_state_dict(merged)

    def load_network_dict(self, network_dict, optimizer, network_label, epoch_label, save_dir=''):
        path = self._path(network_label, epoch_label, save_dir)
        if not os.path.isfile(path):
            print('%s not exists yet!' % path)
            assert network_label != 'G', 'Generator must exist!'
            return
        ck = torch.load(path, map_location='cpu')
        for k, v in network_dict.items():
            self._load_with_fallback(v, ck['network'][k], '%s.%s' % (network_label, k))
        if optimizer is not None:
            optimizer.load_state_dict(ck['optimizer'])

    def update_learning_rate(self):
        pass
