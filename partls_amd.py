"""Import shim: the product package lives in the directory 'partitionedls.jl_amd' (a dot is not importable by name)."""
import importlib.util
import os
import sys

_NAME = "partitionedls_jl_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "partitionedls.jl_amd")


def package():
    """Return the product package (module name 'partitionedls_jl_amd'), importing it on first use."""
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
