"""Which knob moved a configuration's full-depth parity numbers?  Runs tests/test_gpu_configs.py's C4 rows check (8 x 4096, 115 affine layers + extra
context, 512 rows against the fp64 oracle) under a list of `key=value` fc_debug_set settings and prints the table each time.
    python profiles/micro/parity_by_knob.py "" "26=0" "23=0" "8=0"
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_configs as T  # noqa: E402
from flowcompare_amd import engine  # noqa: E402

lib = engine.lib()
for spec in sys.argv[1:] or [""]:
    sets = [tuple(int(x) for x in kv.split("=")) for kv in spec.split(",") if kv]
    for k, v in sets:
        assert lib.fc_debug_set(k, v) == 0, (k, v)
    print(f"==== knobs {spec or '(shipped)'}", flush=True)
    T.test_c4_extra_context_at_8_scenes_of_4096_points()
    defaults = {26: 1, 23: 1, 8: 2, 16: 1, 9: 1, 22: 1}
    for k, _ in sets:
        lib.fc_debug_set(k, defaults[k])
