"""The registration drivers log one DEBUG line per EM iteration ("Iteration: i, Criteria: q").

The logger carries the reference package's name so that applications which already configure
``logging.getLogger("probreg")`` keep working after switching packages (reference probreg/log.py:3-6 uses the
same name, INFO level and a stream handler).
"""
import logging


def _make_logger(name="probreg", level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not any(isinstance(h, logging.StreamHandler) for h in logger.handlers):
        logger.addHandler(logging.StreamHandler())
    return logger


log = _make_logger()
