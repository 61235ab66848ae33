"""The reference CAVP state_dict key tree (tests/golden/state_dict_shapes.json, dumped from the reference by
tools/make_golden.py).  "C" stands for num_classes."""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def cavp_state_shapes(num_classes, seg_model="DeepLabV3Plus"):
    name = "state_dict_shapes_pvt.json" if seg_model == "PVT" else "state_dict_shapes.json"
    with open(os.path.join(_HERE, "golden", name)) as f:
        raw = json.load(f)
    return {k: tuple(num_classes if d == "C" else d for d in v) for k, v in raw.items()}
