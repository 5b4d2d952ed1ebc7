from .build import LOSS, CrossEntropyAce, build_criterion_from_cfg
