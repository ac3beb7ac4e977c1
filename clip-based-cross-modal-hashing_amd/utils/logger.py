"""Logging helpers with the reference's API (utils/logger.py:8-25)."""
import logging
import os


def get_logger(filename=None):
    logger = logging.getLogger('logger')
    logger.setLevel(logging.DEBUG)
    logging.basicConfig(format='%(asctime)s - %(levelname)s -   %(message)s',
                        datefmt='%m/%d/%Y %H:%M:%S', level=logging.INFO)
    if filename is not None:
        handler = logging.FileHandler(filename)
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(logging.Formatter('%(asctime)s:%(levelname)s: %(message)s'))
        logging.getLogger().addHandler(handler)
    return logger


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
        return SummaryWriter(log_dir=dirname)
    except Exception:
        return _NullWriter(dirname)
