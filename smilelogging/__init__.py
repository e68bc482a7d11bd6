"""Import-path shim: reference checkpoints may pickle args.trial as smilelogging.{slutils,utils}.EmptyClass
(/root/reference/smilelogging/slutils.py:172, utils.py:1272); the logger itself is r2l_amd.logger.Logger."""
from r2l_amd.logger import Logger  # noqa: F401
