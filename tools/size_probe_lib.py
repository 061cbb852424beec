"""tools/size_probe.py on another build of the library: SLAM_LIB=path python tools/size_probe_lib.py (development aid)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-experiments_amd"))
from slamhip import _lib
if os.environ.get("SLAM_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SLAM_LIB"])
runpy.run_path(os.path.join(ROOT, "tools", "size_probe.py"), run_name="__main__")
