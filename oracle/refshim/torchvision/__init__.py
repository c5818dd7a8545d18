"""Stub: the reference imports torchvision only for save_image/make_grid, unused on the hot path."""
from . import utils  # noqa: F401
