"""Logging helpers with the reference's API (utils/logger.py:8-25): `get_logger(filename)` and `get_summary_writer(dirname)`."""
import logging
import os

_CONSOLE = dict(format='%(asctime)s - %(levelname)s -   %(message)s', datefmt='%m/%d/%Y %H:%M:%S', level=logging.INFO)
_FILE_FORMAT = '%(asctime)s:%(levelname)s: %(message)s'


def get_logger(filename=None):
    """The named logger 'logger' at DEBUG; the root logger prints INFO to the console and, with `filename`, appends everything
    to that file — upstream's arrangement, log-line formats included."""
    logging.basicConfig(**_CONSOLE)
    log = logging.getLogger('logger')
    log.setLevel(logging.DEBUG)
    if filename is not None:
        to_file = logging.FileHandler(filename)
        to_file.setLevel(logging.DEBUG)
        to_file.setFormatter(logging.Formatter(_FILE_FORMAT))
        logging.getLogger().addHandler(to_file)
    return log


class _NullWriter:
    """The reference creates a TensorBoard SummaryWriter and never writes to it."""

    def __init__(self, log_dir):
        self.log_dir = log_dir

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def get_summary_writer(dirname: str):
    os.makedirs(dirname, exist_ok=True)
    try:
        from torch.utils.tensorboard import SummaryWriter
    except Exception:
        return _NullWriter(dirname)
    try:
        return SummaryWriter(log_dir=dirname)
    except Exception:
        return _NullWriter(dirname)
