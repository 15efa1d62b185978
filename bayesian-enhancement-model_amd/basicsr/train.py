"""Resume entry of the training driver (basicsr/train.py:74-94).  The driver loop itself (dataloaders, samplers, validation, logging)
is host-side control plane outside the hot path; what the hot path needs from it -- picking up a run from ``<iter>.state`` -- is here."""
from basicsr.utils.misc import check_resume, load_resume_state  # noqa: F401
