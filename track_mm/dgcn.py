"""``--module=dgcn`` plugin (reference: track_mm/dgcn.py:25-50,96-137)."""
from functools import partial

from erc_amd.dgcn import DGCNModule, DGCNTrainer  # noqa: F401
from erc_amd.params import ERCParams, Group
from erc_amd.trainer import run


class DGCNParams(ERCParams):
    def __init__(self):
        super().__init__()
        self.train.batch_size = self.val.batch_size = self.test.batch_size = 32   # dgcn.py:33-35
        self.loss_weights = True                                                  # dgcn.py:37
        self.dataset = "iemocap-cogmen-6"
        self.epoch = 55
        self.optim = Group(name="Adam", lr=0.0003, weight_decay=0)                # dgcn.py:41


ParamsType = DGCNParams
main = partial(run, DGCNTrainer, ParamsType)
