"""Logger of the registration drivers; same name / level / handler as the reference (probreg/log.py:3-6)."""
import logging

log = logging.getLogger("probreg")
log.setLevel(logging.INFO)
if not log.handlers:
    log.addHandler(logging.StreamHandler())
