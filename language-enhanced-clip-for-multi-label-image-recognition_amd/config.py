"""Minimal dotted-key config node with the reference's key names (yacs is not installed here or on the GPU box).

Key names follow dassl/config/defaults.py and ``extend_cfg`` in the reference's train_caption.py:74-142; merge
order is defaults -> dataset yaml -> trainer yaml -> argparse -> free ``opts`` (train_caption.py:145-166)."""
from __future__ import annotations

import ast
import copy


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError(f"cfg is frozen; cannot set {k}")
        self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        c = CfgNode(copy.deepcopy(dict(self)))
        return c

    def merge_from_dict(self, other: dict):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path: str):
        import yaml
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = val


def get_cfg_default() -> CfgNode:
    """The subset of dassl defaults + project extensions that the hot path's callers read."""
    return CfgNode({
        "VERSION": 1, "OUTPUT_DIR": "./output", "RESUME": "", "SEED": -1, "USE_CUDA": True, "VERBOSE": True,
        "INPUT": {"SIZE": (224, 224), "PIXEL_MEAN": [0.48145466, 0.4578275, 0.40821073],
                  "PIXEL_STD": [0.26862954, 0.26130258, 0.27577711]},
        "DATASET": {"ROOT": "", "NAME": "", "NUM_CLASSES": 80},
        "DATALOADER": {"NUM_WORKERS": 4, "TRAIN_X": {"BATCH_SIZE": 512}, "TEST": {"BATCH_SIZE": 256}},
        "MODEL": {"INIT_WEIGHTS": "", "BACKBONE": {"NAME": "ViT-B/16", "PRETRAINED": True, "PATH": "synthetic:0:cond"}},
        "OPTIM": {"NAME": "sgd", "LR": 0.002, "MAX_EPOCH": 50, "LR_SCHEDULER": "cosine", "WARMUP_EPOCH": 1,
                  "WARMUP_TYPE": "constant", "WARMUP_CONS_LR": 1e-5, "MOMENTUM": 0.9, "WEIGHT_DECAY": 5e-4},
        "TRAIN": {"CHECKPOINT_FREQ": 0, "PRINT_FREQ": 10, "LOSSFUNC": "double_ranking", "MODEL": "CustomCLIP",
                  "MODEL_NAME": "default", "IF_LEARN_SCALE": False, "IF_LEARN_spatial_SCALE": False,
                  "spatial_SCALE_text": 50, "spatial_SCALE_image": 40, "IF_ablation": False, "Caption_num": 0,
                  "ema": False, "momentum": 0.995},
        "TEST": {"EVALUATOR": "MLClassification", "EVALUATOR_ACT": "default", "PER_CLASS_RESULT": False,
                 "COMPUTE_CMAT": False, "NO_TEST": False, "SPLIT": "test", "FINAL_MODEL": "last_step",
                 "SAVE_PREDS": "", "multi_model": False, "multi_scale": False, "save_pth": "", "use_freq": False,
                 "freq_stats": "freq_stats.pkl",
                 # the reference's ./ChatGLM_..._all_caption_text_feats.pkl (Caption_distill_double.py:35-36): a pickled / torch-saved
                 # [N, E] tensor of normalised caption features for DenseCLIP's top-10 mixing at test time; "" = off
                 "caption_text_feats": ""},
        "TRAINER": {"NAME": "Caption_distill_double",
                    "Caption": {"N_CTX": 16, "CSC": False, "CTX_INIT": "", "PREC": "fp16",
                                "CLASS_TOKEN_POSITION": "end", "GL_merge_rate": 0.5, "use_evidence": False}},
    })
