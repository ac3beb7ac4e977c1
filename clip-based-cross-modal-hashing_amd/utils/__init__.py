from .utils import *  # noqa: F401,F403
from .logger import get_logger, get_summary_writer  # noqa: F401
