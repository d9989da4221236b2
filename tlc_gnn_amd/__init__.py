"""Import alias for the product package.

The build contract fixes the package directory name as ``tlc-gnn_amd/`` (a hyphen is not
importable), so this alias package re-points its ``__path__`` there: ``import tlc_gnn_amd.sg2dgm``
resolves to ``tlc-gnn_amd/sg2dgm``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tlc-gnn_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError("tlc_gnn_amd: product directory %r is missing" % _real)
__path__ = [_real]

from ._version import __version__  # noqa: E402,F401
