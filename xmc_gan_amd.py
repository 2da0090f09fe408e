"""Import alias for the engine package, whose directory name ``xmc-gan_amd/`` is not a Python identifier.

``import xmc_gan_amd`` (and ``xmc_gan_amd.<submodule>``) resolves to the files in ``xmc-gan_amd/``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "xmc-gan_amd")]
__version__ = "0.1.0"
