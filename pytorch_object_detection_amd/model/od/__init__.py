from . import Fcos, HISFcos, MNFcos  # noqa: F401
from .Fcos import FCOS  # noqa: F401
from .HISFcos import HalfInvertedStageFCOS  # noqa: F401
from .MNFcos import MNFCOS  # noqa: F401

proposed = HISFcos  # the reference's Builder refers to `od.proposed` (bulider.py:23); keep that spelling importable
