"""Shared driver pieces of the three entry points (synthetic cohort, K-fold loop, result JSON)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def env_int(name, default):
    return int(os.environ.get(name, default))


def env_float(name, default):
    return float(os.environ.get(name, default))


def setup_device():
    from multimodal_survival_prediction_amd import distributed as D
    world, rank, local = D.init()
    if not torch.cuda.is_available():
        raise SystemExit("these entry points run the HIP hot path and need an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    return world, rank, torch.device("cuda", local)


def save_json(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(obj, f, indent=2)
