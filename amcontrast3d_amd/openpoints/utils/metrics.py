"""Segmentation metrics of the evaluation loops (drop-in for openpoints/utils/metrics.py:30-183).

    AverageMeter      :30-47    running mean of a scalar
    ConfusionMatrix   :50-170   bincount-accumulated (true, pred) matrix and everything derived from it
    get_mious         :173-181  IoU / accuracy from (tp, union, count) vectors (after the cross-rank all-reduce)

Integer work on the GPU (a scatter-add histogram of true * C + pred, no host read-back); the matrix stays an int64 device tensor so that
`dist.all_reduce(cm.tp)` etc. keep working as in examples/segmentation/main_AA.py:460-462.
Deliberate differences: `update` does not overwrite the caller's `pred` / `true` tensors where
true == ignore_index (the reference's flatten() views make its in-place writes visible outside); a label or prediction
outside [0, num_classes) that is not ignore_index makes the reference's bincount / view raise at once -- here such entries are
left out of the histogram (an out-of-range scatter index would fault the GPU), counted on the device, and the error is raised
by the first summary that reads the matrix back (all_acc / all_metrics / check()): update() never waits for the GPU.
"""
import torch


class AverageMeter:
    """last value, running sum, count and mean"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class ConfusionMatrix:
    """value[t, p] = number of points of class t predicted as p.  `ignore_index` (a label < 0 or >= num_classes) is
    mapped to one extra row/column that is dropped again."""

    def __init__(self, num_classes, ignore_index=None):
        self.value = 0
        self.invalid = 0  # entries outside the class range seen so far (device scalar once update() has run)
        self.num_classes = num_classes
        self.virtual_num_classes = num_classes + (1 if ignore_index is not None else 0)
        self.ignore_index = ignore_index

    @torch.no_grad()
    def update(self, pred, true):
        true, pred = true.reshape(-1), pred.reshape(-1)
        v = self.virtual_num_classes
        if self.ignore_index is not None:
            ignored = true == self.ignore_index
            true = torch.where(ignored, v - 1, true)
            pred = torch.where(ignored, v - 1, pred)
        # a histogram by scatter-add rather than torch.bincount: bincount reads its maximum back to the host, which
        # would drain the GPU once per training step (train_one_epoch updates the matrix every iteration)
        valid = (true >= 0) & (true < v) & (pred >= 0) & (pred < v)
        key = torch.where(valid, true * v + pred, 0)
        bins = torch.zeros(v * v, dtype=torch.int64, device=key.device)
        bins.scatter_add_(0, key, valid.to(torch.int64))
        self.value = self.value + bins.view(v, v)[:self.num_classes, :self.num_classes]
        self.invalid = self.invalid + (~valid).sum()

    @torch.no_grad()
    def update_from_logits(self, logits, true):
        """update(logits.argmax(dim=1), true) -- what the trainer does every iteration (main_AA.py:414-415) -- as ONE launch on
        the GPU (ops.confusion_update: arg-max + histogram); anything else goes through update()"""
        v = self.virtual_num_classes
        if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3 and true.dtype == torch.int64
                and logits.shape[1] == self.num_classes and v <= 64 and true.shape == (logits.shape[0], logits.shape[2])):
            return self.update(logits.argmax(dim=1), true)
        from amcontrast3d_amd import ops
        if not torch.is_tensor(self.value):
            self._full = torch.zeros(v, v, dtype=torch.int64, device=logits.device)
            self.invalid = torch.zeros(1, dtype=torch.int64, device=logits.device)
            self.value = self._full[:self.num_classes, :self.num_classes]  # a view: the kernel adds into the full matrix
        elif getattr(self, "_full", None) is None or self.value.data_ptr() != self._full.data_ptr():
            return self.update(logits.argmax(dim=1), true)  # update() replaced the matrix in between: stay on that path
        if not torch.is_tensor(self.invalid) or self.invalid.dim() == 0:
            self.invalid = torch.zeros(1, dtype=torch.int64, device=logits.device) + self.invalid
        ops.confusion_update(self._full, self.invalid, logits, true, self.ignore_index)

    @torch.no_grad()
    def add_counts(self, full, invalid):
        """counts collected elsewhere in update_from_logits' layout -- full (v, v) int64 with the ignore row / column last,
        invalid (1) int64 -- e.g. by the captured training step (amcontrast3d_amd/train.py); the tensors are not kept"""
        self.value = self.value + full[:self.num_classes, :self.num_classes]
        self.invalid = self.invalid + invalid.sum()
        self._full = None

    def check(self):
        """raise if update() met a label / prediction outside the class range (reads one scalar back)"""
        n = int(self.invalid)
        if n:
            raise ValueError(f"ConfusionMatrix: {n} entries with a label or prediction outside [0, {self.num_classes})"
                             + (f" other than ignore_index = {self.ignore_index}" if self.ignore_index is not None else ""))

    def reset(self):
        self.value = 0
        self.invalid = 0
        self._full = None

    # ---- per-class vectors --------------------------------------------------------------------
    @property
    def tp(self):
        return self.value.diag()

    @property
    def actual(self):
        return self.value.sum(dim=1)

    @property
    def predicted(self):
        return self.value.sum(dim=0)

    @property
    def fn(self):
        return self.actual - self.tp

    @property
    def fp(self):
        return self.predicted - self.tp

    @property
    def tn(self):
        return self.actual.sum() + self.tp - (self.actual + self.predicted)

    @property
    def count(self):
        return self.actual

    @property
    def union(self):
        return self.predicted + self.actual - self.tp

    @property
    def frequency(self):
        c = self.actual
        return c / c.sum().clamp(min=1)

    @property
    def total(self):
        return self.value.sum()

    @property
    def overall_accuray(self):  # (sic) the name the reference's loops print
        return self.tp.sum() / self.total

    # ---- summaries ----------------------------------------------------------------------------
    @staticmethod
    def cal_acc(tp, count):
        per_class = tp / count.clamp(min=1) * 100
        overall = tp.sum() / count.sum() * 100
        return torch.mean(per_class).item(), overall.item(), per_class.cpu().numpy()

    def all_acc(self):
        self.check()
        return self.cal_acc(self.tp, self.count)

    def all_metrics(self):
        self.check()
        tp = self.tp
        iou = tp / self.union.clamp(min=1) * 100
        acc = tp / self.count.clamp(min=1) * 100
        overall = tp.sum() / self.total * 100
        return torch.mean(iou).item(), torch.mean(acc).item(), overall.item(), iou.cpu().numpy(), acc.cpu().numpy()


def get_mious(tp, union, count):
    """(mIoU, mAcc, OA, IoU per class, accuracy per class), in per cent; 1e-10 guards empty classes (they count
    as 100 %, as in the reference)"""
    iou = (tp + 1e-10) / (union + 1e-10) * 100
    acc = (tp + 1e-10) / (count + 1e-10) * 100
    overall = tp.sum() / count.sum() * 100
    return torch.mean(iou).item(), torch.mean(acc).item(), overall.item(), iou.cpu().numpy(), acc.cpu().numpy()
