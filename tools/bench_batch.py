"""bench.py with another batch size per GPU (the headline uses 16): tells a host-bound step from a GPU-bound one.
Usage: [OD_LIB=experimental/libod_boxes.so] python tools/bench_batch.py BATCH [bench.py arguments]"""
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _C, synthetic  # noqa: E402

if os.environ.get("OD_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["OD_LIB"])

synthetic.CFG2 = dataclasses.replace(synthetic.CFG2, batch=int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
