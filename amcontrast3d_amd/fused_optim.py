"""AdamW + gradient-norm clipping as two launches over all parameter tensors (csrc/optim.hip).

Drop-in for the pair the trainer calls every step (examples/segmentation/main_AA.py:586-592):

    torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_norm_clip, norm_type=2)
    optimizer.step()                                   # torch.optim.AdamW from openpoints/optim/optim_factory.py

`FusedAdamW` is a torch.optim.Optimizer: same constructor arguments and parameter groups as torch.optim.AdamW (per-group lr,
betas, eps, weight_decay), same state_dict layout ('step', 'exp_avg', 'exp_avg_sq' per parameter), so
openpoints.utils.ckpt_util saves / resumes it unchanged.  `step(max_grad_norm=...)` folds the clipping in (the total norm of
the unclipped gradients is returned as a device scalar, like clip_grad_norm_ does); plain `step()` is AdamW alone.

The moments of all parameters live in one flat buffer each (the per-parameter state tensors are views), the tensors are
described to the kernels by a table in device memory that is rebuilt only when a .grad tensor moves, and the learning rates
are part of that table: a scheduler's new lr is copied into it by the next eager `step()`; a step captured in a hipGraph reads
the table at every replay, so `sync_hyperparameters()` between two replays (what pipeline.GraphPipeline does whenever a
group's lr changed) is all a schedule needs.  GPU fp32 parameters only; one value of betas / eps for all groups."""
import ctypes

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if not 0.0 <= lr or not 0.0 < eps or not all(0.0 <= b < 1.0 for b in betas) or not 0.0 <= weight_decay:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            raise ValueError("FusedAdamW: no parameters")
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("FusedAdamW updates contiguous fp32 GPU parameters (there is no CPU path)")
        if len({(tuple(g["betas"]), g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("FusedAdamW: betas and eps must be the same in every parameter group")
        self._dev = ps[0].device
        self._chunk = int(_lib.load().amc3d_adamw_chunk())
        total = sum(p.numel() for p in ps)
        self._m = torch.zeros(total, dtype=torch.float32, device=self._dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=self._dev)
        self._steps = torch.zeros(len(ps), dtype=torch.float32, device=self._dev)  # one count per parameter, as torch keeps
        self._norm = torch.zeros(1, dtype=torch.float32, device=self._dev)
        off = 0
        for i, p in enumerate(ps):
            n = p.numel()
            self.state[p] = {"step": self._steps[i], "exp_avg": self._m[off:off + n].view_as(p),
                             "exp_avg_sq": self._v[off:off + n].view_as(p)}
            off += n
        self._key = None      # what the device table was built from
        self._table = self._map = self._partial = None

    # ---- device table ---------------------------------------------------------------------------
    def _entries(self):
        out = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if not (p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.device == p.device):
                        raise RuntimeError("FusedAdamW: gradients must be contiguous fp32 tensors on the parameter's device")
                    out.append((p, float(g["lr"]), float(g["weight_decay"])))
        return out

    def _build(self, entries):
        import numpy as np
        rec = np.zeros(len(entries), dtype=np.dtype([("param", "<u8"), ("grad", "<u8"), ("m", "<u8"), ("v", "<u8"), ("step", "<u8"),
                                                     ("numel", "<i8"), ("wd", "<f4"), ("lr", "<f4")]))
        blocks = []
        for i, (p, lr, wd) in enumerate(entries):
            st = self.state[p]
            rec[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                      st["step"].data_ptr(), p.numel(), wd, lr)
            blocks += [(i, c) for c in range(-(-p.numel() // self._chunk))]
        assert rec.dtype.itemsize == 56
        self._table = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self._dev)
        self._map = torch.tensor(blocks, dtype=torch.int32).reshape(-1).to(self._dev)
        self._partial = torch.empty(len(blocks), dtype=torch.float64, device=self._dev)
        self._nblocks = len(blocks)

    def sync_hyperparameters(self):
        """Write the groups' current lr / weight_decay into the device table IN PLACE (same tensor, so a hipGraph that
        captured step() reads the new values at its next replay) -- what a scheduler's step needs between two replays of a
        captured update.  A no-op while nothing changed; the tensors themselves must not have moved (else: prepare())."""
        if self._table is None:
            return False
        entries = self._entries()
        key = tuple((p.data_ptr(), p.grad.data_ptr(), lr, wd) for p, lr, wd in entries)
        if key == self._key:
            return False
        if self._key is None or [k[:2] for k in key] != [k[:2] for k in self._key]:
            raise RuntimeError("FusedAdamW.sync_hyperparameters: a parameter or .grad tensor moved; call prepare()")
        table, nmap, partial = self._table, self._map, self._partial
        self._build(entries)  # same sizes: copy the fresh table into the old tensor and keep the old tensors (a graph holds them)
        table.copy_(self._table)
        self._table, self._map, self._partial = table, nmap, partial
        self._key = key
        return True

    def prepare(self):
        """Describe the current parameter / .grad tensors and learning rates to the kernels now (no update is made).  Call it
        before capturing step() in a hipGraph when the gradients have moved since the last eager step -- e.g. right after
        capturing the backward pass, whose .grad tensors live in the graph's memory pool."""
        entries = self._entries()
        key = tuple((p.data_ptr(), p.grad.data_ptr(), lr, wd) for p, lr, wd in entries)
        if entries and key != self._key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdamW: a .grad tensor or a learning rate changed inside a stream capture; call "
                                   "prepare() (or one eager step()) with the final gradient tensors first")
            self._build(entries)
            self._key = key
        return entries

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """AdamW step; with max_grad_norm, preceded by clip_grad_norm_(all parameters with a gradient, max_grad_norm, 2) --
        returns that call's result (the total norm before clipping, a device scalar), else the closure's loss / None"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.prepare():
            return loss
        g0 = self.param_groups[0]
        clip = float(max_grad_norm) if max_grad_norm else 0.0
        with torch.cuda.device(self._dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
            _lib.check(_lib.load().amc3d_adamw_step(ctypes.c_void_p(self._table.data_ptr()), ctypes.c_void_p(self._map.data_ptr()),
                                                    self._nblocks, float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
                                                    clip, ctypes.c_void_p(self._partial.data_ptr()),
                                                    ctypes.c_void_p(self._norm.data_ptr()), stream), "adamw_step")
        return self._norm[0] if max_grad_norm else loss

    def zero_grad(self, set_to_none=False):
        """Zero the gradients IN PLACE by default (torch's default drops the tensors): the device table holds every .grad
        address, and fresh tensors after each backward would rebuild it -- a host stall and two blocking copies per step"""
        return super().zero_grad(set_to_none=set_to_none)

    # ---- checkpoints: torch's layout in, the flat buffers stay the storage ------------------------
    def load_state_dict(self, state_dict):
        flat = {id(p): (st["exp_avg"], st["exp_avg_sq"], st["step"]) for p, st in self.state.items()}
        super().load_state_dict(state_dict)  # replaces the per-parameter state tensors by the loaded ones
        # every parameter of the groups, not only those the checkpoint has state for: torch.optim.AdamW creates state lazily,
        # so a parameter that never received a gradient has no entry in its checkpoints -- its moments restart from zero
        for g in self.param_groups:
            for p in g["params"]:
                m, v, step = flat[id(p)]
                st = self.state.get(p)
                if st:
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    step.copy_(torch.as_tensor(st["step"], dtype=torch.float32))
                else:
                    m.zero_(); v.zero_(); step.zero_()
                self.state[p] = {"step": step, "exp_avg": m, "exp_avg_sq": v}
        self._key = None  # learning rates / decays of the groups may have changed
