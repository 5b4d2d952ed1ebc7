"""Learning-rate schedule of the trainer: cosine decay with optional linear warm-up
(openpoints/scheduler/cosine_lr.py:18-119, scheduler_factory.py:12-60; the shipped configs: ``sched: cosine``,
``epochs: 100``, ``min_lr: 1e-5``, ``warmup_epochs: 0``, cfgs/s3dis/default.yaml:71-76).

Same driver-facing interface as the reference's scheduler objects -- ``step(epoch)`` once per epoch,
``step_update(num_updates)`` per optimizer step, ``state_dict`` / ``load_state_dict`` -- for the options the AMContrast3D
configs can reach (one cycle or restarts with cycle_mul / cycle_decay, k-decay, warm-up); the lr-noise options are not
built.  Learning rates are written into ``param_group['lr']`` scaled by the group's ``lr_scale`` when present.
"""
import math


class CosineLRScheduler:
    def __init__(self, optimizer, t_initial, lr_min=0., cycle_mul=1., cycle_decay=1., cycle_limit=1, warmup_t=0,
                 warmup_lr_init=0, warmup_prefix=False, t_in_epochs=True, noise_range_t=None, noise_pct=0.67,
                 noise_std=1.0, noise_seed=42, k_decay=1.0, initialize=True):
        if noise_range_t is not None:
            raise NotImplementedError("lr noise is not part of the AMContrast3D configs")
        assert lr_min >= 0
        self.optimizer = optimizer
        for group in optimizer.param_groups:
            if initialize:
                group.setdefault("initial_lr", group["lr"])
            elif "initial_lr" not in group:
                raise KeyError("initial_lr is not specified in param_groups when resuming a scheduler")
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        self.t_initial, self.lr_min = t_initial, lr_min
        self.cycle_mul, self.cycle_decay, self.cycle_limit = cycle_mul, cycle_decay, cycle_limit
        self.warmup_t, self.warmup_lr_init, self.warmup_prefix = warmup_t, warmup_lr_init, warmup_prefix
        self.t_in_epochs, self.k_decay = t_in_epochs, k_decay
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._set([warmup_lr_init] * len(self.base_values))
        else:
            self.warmup_steps = [1 for _ in self.base_values]
            self._set(self.base_values)

    def _set(self, values):
        for group, v in zip(self.optimizer.param_groups, values):
            group["lr"] = v * group.get("lr_scale", 1.0)

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if self.warmup_prefix:
            t = t - self.warmup_t
        if self.cycle_mul != 1:
            i = math.floor(math.log(1 - t / self.t_initial * (1 - self.cycle_mul), self.cycle_mul))
            t_i = self.cycle_mul ** i * self.t_initial
            t_curr = t - (1 - self.cycle_mul ** i) / (1 - self.cycle_mul) * self.t_initial
        else:
            i = t // self.t_initial
            t_i = self.t_initial
            t_curr = t - self.t_initial * i
        if i >= self.cycle_limit:
            return [self.lr_min for _ in self.base_values]
        gamma, k = self.cycle_decay ** i, self.k_decay
        return [self.lr_min + 0.5 * (v * gamma - self.lr_min) * (1 + math.cos(math.pi * t_curr ** k / t_i ** k))
                for v in self.base_values]

    def get_epoch_values(self, epoch):
        return self._get_lr(epoch) if self.t_in_epochs else None

    def get_update_values(self, num_updates):
        return None if self.t_in_epochs else self._get_lr(num_updates)

    def step(self, epoch, metric=None):
        values = self.get_epoch_values(epoch)
        if values is not None:
            self._set(values)

    def step_update(self, num_updates, metric=None):
        values = self.get_update_values(num_updates)
        if values is not None:
            self._set(values)

    def get_cycle_length(self, cycles=0):
        cycles = max(1, cycles or self.cycle_limit)
        if self.cycle_mul == 1.0:
            return self.t_initial * cycles
        return int(math.floor(-self.t_initial * (self.cycle_mul ** cycles - 1) / (1 - self.cycle_mul)))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)


def build_scheduler_from_cfg(args, optimizer, return_epochs=False):
    """scheduler_factory.py:12-60 for ``sched: cosine``"""
    if getattr(args, "sched", "cosine") != "cosine":
        raise NotImplementedError(f"scheduler {args.sched!r}: the AMContrast3D configs use 'cosine'")
    num_epochs = args.epochs
    min_lr = args.min_lr if getattr(args, "min_lr", False) else args.lr / 1000.
    sched = CosineLRScheduler(optimizer, t_initial=getattr(args, "t_max", num_epochs), lr_min=min_lr,
                              warmup_lr_init=getattr(args, "warmup_lr", 1.0e-6), warmup_t=getattr(args, "warmup_epochs", 0),
                              k_decay=getattr(args, "lr_k_decay", 1.0), cycle_mul=getattr(args, "lr_cycle_mul", 1.),
                              cycle_decay=getattr(args, "lr_cycle_decay", 0.1), cycle_limit=getattr(args, "lr_cycle_limit", 1))
    num_epochs = sched.get_cycle_length() + getattr(args, "cooldown_epochs", 0)
    return (sched, num_epochs) if return_epochs else sched
