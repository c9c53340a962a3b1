from . import utils  # noqa: F401
from . import BaseSdeGenerativeModel  # noqa: F401
