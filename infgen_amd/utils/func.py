"""Host-side helpers with the reference's names (infgen/utils/func.py:15,30-62,65-69,80-173,177-196): what the reference's entry
scripts (run.py:12, val.py:8, train.py) and its dataset module import from ``infgen.utils.func``."""
import logging
import math
import os
import time
from typing import Mapping, Optional

import torch
import torch.nn as nn
import yaml

try:                                    # (run.py / scalable_dataset.py print through it; plain text where rich is absent)
    from rich.console import Console
    CONSOLE = Console(width=128)
except ImportError:                     # pragma: no cover
    class _PlainConsole:
        def print(self, *a, **k):
            print(*a)

        log = rule = print
    CONSOLE = _PlainConsole()


class ConfigDict(dict):
    """nested dict with attribute access - what the reference gets from ``easydict.EasyDict`` (``config.Model.decoder...``)"""

    def __init__(self, d: Optional[Mapping] = None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, Mapping) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    __setattr__ = __setitem__

    def __delattr__(self, k):
        del self[k]


def load_config_act(path: str) -> ConfigDict:
    """YAML file -> attribute dict (reference infgen/utils/func.py:65-69; run.py:96, val.py:25)"""
    with open(path, 'r') as f:
        return ConfigDict(yaml.load(f, Loader=yaml.FullLoader))


class Logging:
    """``Logging().log(level='DEBUG')`` -> a logger with a console and a file handler (reference infgen/utils/func.py:80-122;
    val.py:26, train.py).  Log files go to ``$INFGEN_LOG_DIR`` (default ``./logs``), one per process start."""
    FORMAT = '%(asctime)s-%(levelname)s-%(filename)s-Line:%(lineno)d-Message:%(message)s'

    def make_log_dir(self, dirname: str = 'logs') -> str:
        path = os.path.normpath(os.environ.get('INFGEN_LOG_DIR') or os.path.join(os.getcwd(), dirname))
        os.makedirs(path, exist_ok=True)
        return path

    def get_log_filename(self) -> str:
        return os.path.normpath(os.path.join(self.make_log_dir(), time.strftime('%Y-%m-%d-%H%M%S', time.localtime()) + '.log'))

    def add_log(self, logger: logging.Logger, level: str = 'DEBUG') -> logging.Logger:
        logger.setLevel(getattr(logging, level))
        if not logger.handlers:
            fmt = logging.Formatter(self.FORMAT)
            for h in (logging.StreamHandler(), logging.FileHandler(filename=self.get_log_filename(), mode='a', encoding='utf-8')):
                h.setFormatter(fmt)
                logger.addHandler(h)
        return logger

    def log(self, level: str = 'DEBUG', name: str = 'simagent') -> logging.Logger:
        return self.add_log(logging.getLogger(name), level)


def _process_rank() -> int:
    """rank of this process: torch.distributed when initialised, else the launcher's environment (torchrun / Lightning DDP)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    for key in ('RANK', 'LOCAL_RANK', 'SLURM_PROCID'):
        if os.environ.get(key, '').isdigit():
            return int(os.environ[key])
    return 0


class RankedLogger(logging.LoggerAdapter):
    """command-line logger for one process per GPU: messages carry the rank, ``rank_zero_only=True`` keeps rank 0's only,
    ``log(level, msg, rank=r)`` logs on rank r only (reference infgen/utils/func.py:125-173; run.py:14 ``RankedLogger(__name__,
    rank_zero_only=True)``)"""

    def __init__(self, name: str = __name__, rank_zero_only: bool = False, extra: Optional[Mapping[str, object]] = None) -> None:
        super().__init__(logger=logging.getLogger(name), extra=extra)
        self.rank_zero_only = rank_zero_only

    def log(self, level: int, msg: str, rank: Optional[int] = None, *args, **kwargs) -> None:
        if not self.isEnabledFor(level):
            return
        msg, kwargs = self.process(msg, kwargs)
        cur = _process_rank()
        if (self.rank_zero_only and cur != 0) or (rank is not None and cur != rank):
            return
        self.logger.log(level, f'[rank: {cur}] {msg}', *args, **kwargs)


def angle_between_2d_vectors(ctr_vector: torch.Tensor, nbr_vector: torch.Tensor) -> torch.Tensor:
    return torch.atan2(ctr_vector[..., 0] * nbr_vector[..., 1] - ctr_vector[..., 1] * nbr_vector[..., 0],
                       (ctr_vector[..., :2] * nbr_vector[..., :2]).sum(dim=-1))


def wrap_angle(angle: torch.Tensor, min_val: float = -math.pi, max_val: float = math.pi) -> torch.Tensor:
    return min_val + (angle + max_val) % (max_val - min_val)


def weight_init(m: nn.Module) -> None:
    """the distributions the reference initialises with (Linear xavier-uniform / zero bias,
    Embedding N(0, 0.02), LayerNorm 1 / 0)"""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Embedding):
        nn.init.normal_(m.weight, mean=0.0, std=0.02)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
