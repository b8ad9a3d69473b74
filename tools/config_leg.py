"""One config leg of bench.py alone (C2 | C5): python tools/config_leg.py C5"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import scenes  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for name in sys.argv[1:] or ["C5"]:
        print(json.dumps({name: bench.config_leg(name, scenes, dev, steps=100 if name == "C2" else 20, warmup=10 if name == "C2" else 3)}))
