"""Authoring-container-only stub: the reference imports torchvision.models at module import time."""
from . import models  # noqa
