"""Import shim: the product package lives in the directory ``image-codecs_amd/`` (the name the
project layout prescribes); a hyphen cannot appear in an ``import`` statement, so this module
loads that directory as the package ``image_codecs_amd`` and replaces itself with it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-codecs_amd")
_spec = importlib.util.spec_from_file_location(
    "image_codecs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["image_codecs_amd"] = _mod
_spec.loader.exec_module(_mod)
