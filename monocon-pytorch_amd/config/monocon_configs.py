"""Default configuration tree.  Keys and default values are those of the reference's yacs tree
(config/monocon_configs.py:4-64) so its YAML files merge unchanged; kept here as one nested table."""
from config.cfgnode import CfgNode as CN

_DEFAULTS = {
    "VERSION": "v1.0.3",
    "DESCRIPTION": "MonoCon Default Configuration",
    "OUTPUT_DIR": "",
    "SEED": -1,
    "GPU_ID": 0,
    "USE_BENCHMARK": True,
    "DATA": {
        # 'synthetic' selects the built-in KITTI-shaped synthetic dataset; anything else is a KITTI root directory
        "ROOT": r"/home/user/SSD/KITTI",
        "BATCH_SIZE": 8,
        "NUM_WORKERS": 4,
        "TRAIN_SPLIT": "train",
        "TEST_SPLIT": "val",
        "FILTER": {"MIN_HEIGHT": 25, "MIN_DEPTH": 2, "MAX_DEPTH": 65, "MAX_TRUNCATION": 0.5, "MAX_OCCLUSION": 2},
    },
    "MODEL": {
        "BACKBONE": {"NUM_LAYERS": 34, "IMAGENET_PRETRAINED": True},
        "HEAD": {"NUM_CLASSES": 3, "MAX_OBJS": 30},
    },
    "SOLVER": {
        "OPTIM": {"LR": 2.25e-4, "WEIGHT_DECAY": 1e-5, "NUM_EPOCHS": 200},
        "SCHEDULER": {"ENABLE": True},
        "CLIP_GRAD": {"ENABLE": True, "NORM_TYPE": 2.0, "MAX_NORM": 35},
    },
    "PERIOD": {"EVAL_PERIOD": 10, "LOG_PERIOD": 50},
}


def _tree(table):
    node = CN()
    for key, value in table.items():
        node[key] = _tree(value) if isinstance(value, dict) else value
    return node


_C = _tree(_DEFAULTS)
