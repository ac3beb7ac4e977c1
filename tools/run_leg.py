"""One per-config bench leg by itself (bench_configs.LEGS), e.g. under rocprofv3:
   rocprofv3 --kernel-trace --stats -d out -- python3 tools/run_leg.py mith_step"""
import json
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
name = sys.argv[1]
sys.argv = ["bench.py"]
import bench            # noqa: E402,F401  (sys.path set-up of the package and the test helpers)
import bench_configs    # noqa: E402

leg = dict(bench_configs.LEGS)[name]
print(json.dumps(leg(torch.device("cuda:0"))))
