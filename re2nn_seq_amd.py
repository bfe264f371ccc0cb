"""Import shim: the package directory is ``re2nn-seq_amd/`` (hyphenated, as the layout
contract names it), which Python cannot import by name.  Importing ``re2nn_seq_amd``
loads this file, which replaces itself in ``sys.modules`` with the real package."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "re2nn-seq_amd")
_spec = _u.spec_from_file_location("re2nn_seq_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["re2nn_seq_amd"] = _mod
_spec.loader.exec_module(_mod)
