from .common import *      # noqa: F401,F403
from .blackbody import *   # noqa: F401,F403
