from .logutils import *       # noqa: F401,F403
from .sourceutils import *    # noqa: F401,F403
