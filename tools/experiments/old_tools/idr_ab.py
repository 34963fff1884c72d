"""IDR bench against an alternative library build: ISO_DEV_LIB=tools/variants/libiso_X.so python tools/idr_ab.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
if len(sys.argv) < 2:
    sys.argv.append("300000")
exec(open(os.path.join(ROOT, "tools", "idr_bench.py")).read())
