from .types import *  # noqa
from . import lib  # noqa
from .lib import Renderer, HostScene, PtError  # noqa
from . import partition  # noqa
