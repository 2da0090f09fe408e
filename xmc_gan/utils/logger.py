import logging
import os
import sys


def setup_logger(name, save_dir, distributed_rank, filename='log.txt'):
    """stdout + file logger; ranks > 0 stay silent (same contract as the reference's utils/logger.py:5-23)."""
    log = logging.getLogger(name)
    log.setLevel(logging.DEBUG)
    if distributed_rank > 0 or log.handlers:
        return log
    fmt = logging.Formatter("%(asctime)s %(name)s %(levelname)s: %(message)s")
    handlers = [logging.StreamHandler(stream=sys.stdout)]
    if save_dir:
        handlers.append(logging.FileHandler(os.path.join(save_dir, filename)))
    for h in handlers:
        h.setLevel(logging.DEBUG)
        h.setFormatter(fmt)
        log.addHandler(h)
    return log
