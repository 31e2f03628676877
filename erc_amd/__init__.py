"""Importable alias of the ``emotion-recognition-in-conversation_amd/`` package.

The package directory carries the repository's name (with hyphens, so it is
not a valid Python identifier); this shim makes its modules importable as
``erc_amd.<module>`` by extending ``__path__``.  No code lives here.
"""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "emotion-recognition-in-conversation_amd")
__path__.insert(0, _pkg_dir)
PACKAGE_DIR = _pkg_dir
