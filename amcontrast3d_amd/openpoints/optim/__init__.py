from .optim_factory import add_weight_decay, build_optimizer_from_cfg, get_parameter_groups, optimizer_kwargs  # noqa: F401
