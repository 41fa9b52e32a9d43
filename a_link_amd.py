"""Import shim: the package directory is `a-link_amd/` (hyphen, as the build contract names it);
`import a_link_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "a-link_amd")
_spec = importlib.util.spec_from_file_location("a_link_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["a_link_amd"] = _mod
_spec.loader.exec_module(_mod)
