"""Registers the hyphenated package directory ``sana-fe_amd/`` as module ``sanafe_amd``.

For the tests and bench.py's small configurations the pure-Python twin of the description layer (tests/twin, test
infrastructure, never imported by the product) is attached to the loaded package as ``description`` / ``yaml_io`` /
``to_desc`` when it is present."""
import importlib
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
    tests_dir = os.path.join(ROOT, "tests")
    if os.path.isdir(os.path.join(tests_dir, "twin")):
        if tests_dir not in sys.path:
            sys.path.insert(0, tests_dir)
        mod.description = importlib.import_module("twin.description")
        mod.yaml_io = importlib.import_module("twin.yaml_io")
        mod.to_desc = mod.description.to_desc
    return mod
