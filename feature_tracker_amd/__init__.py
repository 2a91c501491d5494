"""MI355X-native visual front-end (pyramid, Scharr gradients, pyramidal LK, grid FAST) behind the
TrackBase/TrackKLT/Grider_FAST call surface of nunuNNN/feature_tracker.

The compute path is the hand-written gfx950 library in ``_lib/`` reached through the C ABI of
``include/ft_device.h``; importing the device modules fails loudly when it has not been built.
"""
from . import device  # noqa: F401

__all__ = ["device"]
