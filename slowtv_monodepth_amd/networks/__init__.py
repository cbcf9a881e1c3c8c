from .encoders import *  # noqa: F401,F403
from .decoders import *  # noqa: F401,F403
from .depth import *  # noqa: F401,F403
from .pose import *  # noqa: F401,F403
