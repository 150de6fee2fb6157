"""reference data/base_data_loader.py"""


class BaseDataLoader(object):
    def initialize(self, opt):
        self.opt = opt

    def load_data(self):
        return None
