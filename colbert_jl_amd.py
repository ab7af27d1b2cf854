"""Import shim: the package directory is named `colbert.jl_amd/` (after the reference, ColBERT.jl), and
a dot is not legal in a Python module name, so `import colbert_jl_amd` loads that directory as the
package `colbert_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colbert.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "colbert_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["colbert_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
