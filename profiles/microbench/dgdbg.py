import os, sys
sys.path.insert(0, "/root/repo")
os.environ["AMD_LOG_LEVEL"] = "0"
import numpy as np, torch
from tests.test_depgraph_dev import random_prefix_graph, run_both
rng = np.random.default_rng(1)
leader, number, first, count, deps, own = random_prefix_graph(rng, 3, 300, 0, False)
try:
    print(run_both(3, leader, number, first, count, deps, own)[0][:3])
except Exception as e:
    print("ERR", e)
    try:
        torch.cuda.synchronize()
    except Exception as e2:
        print("sync:", e2)
