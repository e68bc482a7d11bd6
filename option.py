"""`from option import args` as in the reference (option.py:360): parsed lazily on first access."""
from r2l_amd.options import parse_args

_args = None


def __getattr__(name):
    global _args
    if name == "args":
        if _args is None:
            _args = parse_args()
        return _args
    raise AttributeError(name)
