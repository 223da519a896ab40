"""Run bench.py's workload functions on a library variant: WB_LIB=<file under tools/probe/variants>
python tools/probe/bench_variant.py conv3x3  (prints the workload's result dict as JSON)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "conv3x3"
if which == "conv3x3":
    r = bench.run_conv(20, 5, 0, 1, False)
    print(os.environ.get("WB_LIB", "libwm.so"), json.dumps({k: r[k] for k in ("fwd_ms", "bwd_wb_ms", "pair_ms")}), "frac", round(r["roofline"]["frac"], 4), "trio", round(r["with_bwd_data"]["trio_ms"], 4), round(r["with_bwd_data"]["roofline"]["frac"], 4))
