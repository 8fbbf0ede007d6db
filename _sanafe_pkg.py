"""Registers the hyphenated package directory ``sana-fe_amd/`` as module ``sanafe_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "sanafe_amd" in sys.modules:
        return sys.modules["sanafe_amd"]
    pkg_dir = os.path.join(ROOT, "sana-fe_amd")
    spec = importlib.util.spec_from_file_location("sanafe_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sanafe_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
