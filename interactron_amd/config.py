"""YAML config objects and factories with the reference's keys (utils/config_utils.py:9-113).

``Config`` applies the reference's coercion rule: every scalar goes through ``float()`` and becomes ``int`` when
integral, otherwise it is left as is -- so ``"1e-3"`` -> 0.001, ``True`` -> 1 and non-numeric strings stay strings.
"""
import argparse
import os

import yaml


class Config:
    def __init__(self, **entries):
        out = {}
        for key, value in entries.items():
            if type(value) is dict:
                out[key] = Config(**value)
                continue
            try:
                value = float(value)
                if value.is_integer():
                    value = int(value)
            except (TypeError, ValueError):
                pass
            out[key] = value
        self.__dict__.update(out)

    def dictionarize(self):
        return {k: (v.dictionarize() if isinstance(v, Config) else v) for k, v in self.__dict__.items()}


def get_config(cfg):
    assert os.path.exists(cfg), "File {} does not exist".format(cfg)
    with open(cfg) as f:
        return Config(**yaml.safe_load(f))


def get_args():
    parser = argparse.ArgumentParser(description="Train Interactron Model")
    parser.add_argument("--config_file", type=str, required=True,
                        help="path to the configuration file for this training run")
    return parser.parse_args()


MODEL_TYPES = ["detr", "detr_multiframe", "interactron_random", "interactron"]


def arg_check(arg, choices, argname):
    assert arg in choices, "{} is not a valid {}. Please select one from {}".format(arg, argname, choices)


def build_model(args):
    arg_check(args.TYPE, MODEL_TYPES, "model")
    from . import episode
    # MODEL.COMPUTE_DTYPE: f32 (default, the parity path) | bf16 (16-bit activations, b16.py) | single_pass / fp16 (fp32 storage,
    # single-pass 16-bit contractions).  It is the MODEL's mode: stored on it and put in force at each of its entry points
    # (hipops.compute_mode) -- building a model neither loads the kernel library nor touches any other model's arithmetic.
    from .hipops import normalize_compute_dtype
    model = getattr(episode, args.TYPE)(args)
    model.compute_dtype = normalize_compute_dtype(getattr(args, "COMPUTE_DTYPE", "f32"))
    return model


def build_trainer(model, args, evaluator=None):
    """reference utils/config_utils.py:79-97 (the ``adaptive*`` types point at modules the reference does not ship)."""
    arg_check(args.TRAINER.TYPE, ["direct_supervision", "interactron_random", "interactron"], "supervisor")
    from . import engine
    cls = {"direct_supervision": engine.DirectSupervisionTrainer, "interactron_random": engine.InteractronRandomTrainer,
           "interactron": engine.InteractronTrainer}[args.TRAINER.TYPE]
    return cls(model, args, evaluator=evaluator)


def build_evaluator(model, args, load_checkpoint=False):
    """reference utils/config_utils.py:100-113."""
    arg_check(args.EVALUATOR.TYPE, ["random_policy_evaluator", "interactive_evaluator"], "evaluator")
    from . import engine
    cls = {"random_policy_evaluator": engine.RandomPolicyEvaluator,
           "interactive_evaluator": engine.InteractiveEvaluator}[args.EVALUATOR.TYPE]
    return cls(model, args, load_checkpoint=load_checkpoint)
