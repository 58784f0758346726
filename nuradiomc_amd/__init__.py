"""nuradiomc_amd -- MI355X-native hot path for NuRadioMC-style simulations (ray tracing ->
Askaryan emission -> antenna/amplifier response), host side in Python over the libnrhip.so C ABI."""
from .context import Context, ATTENUATION_MODEL_TO_INT  # noqa: F401
from ._lib import NrhipError, LIB_PATH  # noqa: F401
from .station import Station, TabulatedAntenna  # noqa: F401
from . import filters  # noqa: F401
from . import comm, sequencing  # noqa: F401  (comm registers its C-ABI signatures before the library is first loaded)
from .array import StationArray  # noqa: F401

__version__ = "0.1.0"
