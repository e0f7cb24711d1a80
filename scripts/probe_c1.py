"""BASELINE configs[1] alone (CausalBGM binary treatment, N = 1e5, p = 100, z_dims [3,3,6,6], 5000 + 3000 transitions, ITE + intervals):
bench.config_c1_leg on one MI355X, for kernel traces.   python scripts/probe_c1.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
print(json.dumps(bench.config_c1_leg(torch.device("cuda", 0))))
