"""``--module=mmgcn`` plugin (reference: track_mm/mmgcn.py:20-53,126-160)."""
from functools import partial

from erc_amd.mmgcn import MMGCNModule, MMGCNTrainer  # noqa: F401
from erc_amd.params import ERCParams, Group
from erc_amd.trainer import run


class MMGCNParams(ERCParams):
    def __init__(self):
        super().__init__()
        self.epoch = 60
        self.train.batch_size = self.test.batch_size = 16                      # mmgcn.py:29-30
        self.dataset = "iemocap-cogmen-6"
        self.optim = Group(name="Adam", lr=0.0003, weight_decay=3e-5)          # mmgcn.py:34
        self.speaker_onehot, self.batch_first = True, False                    # mmgcn.py:39-40

    def iparams(self):
        super().iparams()
        if self.reimplement and "meld" in self.dataset:                         # mmgcn.py:44-50
            self.optim.lr, self.optim.weight_decay = 0.0001, 0
        return self


ParamsType = MMGCNParams
main = partial(run, MMGCNTrainer, ParamsType)
