"""Import helper: the package directory is called ``music-fader-nets_amd`` (not a valid Python identifier),
so it is loaded by path and registered as ``music_fader_nets_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "music-fader-nets_amd")
PKG_NAME = "music_fader_nets_amd"


def load_package():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(PKG_NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
