"""Loader for the product package, whose directory name (`bevy-signed-distance-mesh-generation_amd`)
is not a valid Python identifier."""
import importlib.util
import os
import sys

_NAME = "bevy_signed_distance_mesh_generation_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bevy-signed-distance-mesh-generation_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
