"""dvbs2rx_amd -- host-side mirror of the reference's FEC block surface over libdvbs2_fec_hip.so."""
from . import capi  # noqa: F401
from .blocks import *  # noqa: F401,F403
from . import shard  # noqa: F401,E402
