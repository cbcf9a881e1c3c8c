from .photometric import *  # noqa: F401,F403
from .reconstruction import *  # noqa: F401,F403
from .regression import *  # noqa: F401,F403
