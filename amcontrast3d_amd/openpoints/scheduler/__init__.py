from .cosine_lr import CosineLRScheduler, build_scheduler_from_cfg  # noqa: F401
