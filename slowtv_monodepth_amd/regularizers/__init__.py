from .smooth import *  # noqa: F401,F403
