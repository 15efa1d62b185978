import logging
import os

from .misc import check_resume, load_resume_state  # noqa: F401


def get_root_logger(logger_name="basicsr", log_level=logging.INFO, log_file=None):
    return logging.getLogger(logger_name)


def scandir(dir_path, suffix=None, recursive=False, full_path=False):
    for e in sorted(os.listdir(dir_path)):
        if suffix is None or e.endswith(suffix):
            yield os.path.join(dir_path, e) if full_path else e
