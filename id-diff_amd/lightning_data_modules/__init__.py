from . import utils  # noqa: F401
from . import KSphereDataset, SyntheticImages, ImageDatasets, GanDataset  # noqa: F401
