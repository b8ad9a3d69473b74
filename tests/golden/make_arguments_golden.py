"""Generates tests/golden/arguments.json: the default values of the reference's PipelineParams / OptimizationParams / ModelParams
(/root/reference/arguments/__init__.py:47-100), read by instantiating the classes in the build container.  Data only.
    python tests/golden/make_arguments_golden.py"""
import json
import os
import sys
from argparse import ArgumentParser

sys.path.append("/root/reference")
import arguments as A   # noqa: E402

out = {}
for cls in (A.ModelParams, A.PipelineParams, A.OptimizationParams):
    out[cls.__name__] = {k.lstrip("_"): v for k, v in vars(cls(ArgumentParser())).items()}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arguments.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
    f.write("\n")
print(json.dumps(out, sort_keys=True))
