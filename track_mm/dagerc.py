"""``--module=dagerc`` plugin (reference: track_mm/dagerc.py:20-70,201-240)."""
from functools import partial

from erc_amd.dagerc import DAGERCModule, DAGERCTrainer  # noqa: F401
from erc_amd.params import ERCParams, Group
from erc_amd.trainer import run


class DAGERCParams(ERCParams):
    def __init__(self):
        super().__init__()
        self.train.batch_size = self.test.batch_size = 8          # dagerc.py:28-29
        self.num_heads, self.gnn_heads, self.gnn_layers, self.dropout = 10, 1, 4, 0
        self.dataset = "iemocap-cogmen-6"
        self.epoch = 30
        self.optim = Group(name="AdamW", lr=1e-3, weight_decay=1e-2)   # torch AdamW default decay, dagerc.py:39
        self.speaker_onehot = True                                 # dagerc.py:41

    def iparams(self):
        super().iparams()
        if self.reimplement:                                       # dagerc.py:45-67
            if "iemocap" in self.dataset:
                self.dropout, self.epoch, self.gnn_layers = 0.2, 55, 4
                self.train.batch_size, self.optim.lr = 16, 0.0005
            elif "meld" in self.dataset:
                self.optim.lr, self.train.batch_size, self.epoch, self.dropout = 0.00001, 64, 70, 0.1
        return self


ParamsType = DAGERCParams
main = partial(run, DAGERCTrainer, ParamsType)
