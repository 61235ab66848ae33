"""`models.cavp_model` of the reference (models/cavp_model.py:21-205), served by the MI355X implementation: same class
names, constructor, forward signatures, attribute tree and state_dict keys; all arithmetic in libcavp_hip.so."""
from cavp_amd.cavp_model import CAVP, SoundBank, load_reference_checkpoint  # noqa: F401

__all__ = ["CAVP", "SoundBank", "load_reference_checkpoint"]
