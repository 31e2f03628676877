"""``--module=cogmen`` plugin (reference: track_mm/cogmen.py:35-58,163-198)."""
from functools import partial

from erc_amd.cogmen import COGMENModule, COGMENTrainer  # noqa: F401
from erc_amd.params import ERCParams, Group
from erc_amd.trainer import run


class COGMENParams(ERCParams):
    def __init__(self):
        super().__init__()
        self.train.batch_size = self.val.batch_size = self.test.batch_size = 32   # cogmen.py:43-45
        self.num_heads = 17
        self.dataset = "iemocap-cogmen-6"
        self.epoch = 55
        self.optim = Group(name="Adam", lr=0.0001, weight_decay=1e-8)             # cogmen.py:50


ParamsType = COGMENParams
main = partial(run, COGMENTrainer, ParamsType)
