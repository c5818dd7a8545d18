from .twoarmy import Twoarmy_v4, Twoarmy_v6  # noqa: F401
