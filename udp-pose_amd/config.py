"""Minimal reader of the reference's experiment YAMLs (yacs is not required).

Defaults follow deep_hrnet/lib/config/default.py:17-130 for the keys the hot
path reads; ``load_config(path)`` merges a YAML over them (the reference's
``update_config`` / ``cfg.merge_from_file``, default.py:133-160, minus the
output-directory bookkeeping).
"""
import copy

import yaml


class CfgNode(dict):
    """dict with attribute access (cfg.MODEL.EXTRA and cfg['MODEL']['EXTRA'])."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        return None

    def freeze(self):
        return None

    def merge_from_file(self, path):
        with open(path) as f:
            _merge(self, yaml.safe_load(f) or {})


DEFAULTS = {
    "MODEL": {"NAME": "pose_hrnet", "INIT_WEIGHTS": True, "PRETRAINED": "", "NUM_JOINTS": 17,
              "TAG_PER_JOINT": True, "TARGET_TYPE": "gaussian", "IMAGE_SIZE": [256, 256],
              "HEATMAP_SIZE": [64, 64], "SIGMA": 2, "EXTRA": {}},
    "LOSS": {"USE_OHKM": False, "TOPK": 8, "USE_TARGET_WEIGHT": True, "USE_DIFFERENT_JOINTS_WEIGHT": False,
             "KPD": 4.0},
    "DATASET": {"DATASET": "mpii", "FLIP": True, "SCALE_FACTOR": 0.25, "ROT_FACTOR": 30, "PROB_HALF_BODY": 0.0,
                "NUM_JOINTS_HALF_BODY": 8, "COLOR_RGB": False, "CUTOUT": None, "HIDE_AND_SEEK": None},
    "TEST": {"BATCH_SIZE_PER_GPU": 32, "FLIP_TEST": False, "POST_PROCESS": False, "SHIFT_HEATMAP": False,
             "MODEL_FILE": ""},
    "TRAIN": {"LR": 0.001, "OPTIMIZER": "adam", "BATCH_SIZE_PER_GPU": 32},
}


def _wrap(d):
    return CfgNode({k: _wrap(v) if isinstance(v, dict) else v for k, v in d.items()})


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = _wrap(v) if isinstance(v, dict) else v


def default_config():
    return _wrap(copy.deepcopy(DEFAULTS))


def load_config(path=None, overrides=None):
    cfg = default_config()
    if path:
        cfg.merge_from_file(path)
    if overrides:
        _merge(cfg, overrides)
    return cfg
