from . import utils  # noqa: F401
from . import KSphereDataset, SyntheticImages  # noqa: F401
