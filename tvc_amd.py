"""Importable alias of the product package, whose directory name
``multimodal-detection-consistency_amd`` is not a Python identifier:
``import tvc_amd`` == ``importlib.import_module("multimodal-detection-consistency_amd")``."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("multimodal-detection-consistency_amd")
