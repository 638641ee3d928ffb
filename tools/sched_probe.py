"""Developer probe (GPU box): which generator settings give the look-ahead schedule test of tests/test_gpu_world2.py both
kept and failed look-aheads?  python tools/sched_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_world2 as T  # noqa: E402

for data in (((4000, 136, 8, 1, 5, 3, 64, 8), (6e-3, 0.6)), ((4000, 136, 8, 1, 5, 3, 128, 8), (5e-3, 0.5)),
             ((6000, 140, 12, 5, 5, 3, 64, 20), (6e-3, 0.6)), ((3000, 136, 8, 1, 5, 3, 32, 8), (4.5e-3, 0.4))):
    outs = T._run_sched(2, "1000,1000,1000;0,1000,0", {}, data)
    for o in outs:
        print(data, {k: (str(v) if k == "error" else int(v)) for k, v in o.items() if k != "lab"}, flush=True)
