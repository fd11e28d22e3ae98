"""Import alias: the package directory is `rotate-yolov3_amd/` (not a valid Python identifier), so
`import rotate_yolov3_amd` loads it from there and installs it under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rotate-yolov3_amd")
_spec = importlib.util.spec_from_file_location(
    "rotate_yolov3_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rotate_yolov3_amd"] = _mod
_spec.loader.exec_module(_mod)
