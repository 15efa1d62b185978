"""Learning-rate schedules of the shipped option files (basicsr/models/lr_scheduler.py:186-230): host-side arithmetic only."""
import bisect
import math

from torch.optim.lr_scheduler import _LRScheduler


class CosineAnnealingRestartCyclicLR(_LRScheduler):
    """Cosine annealing with restarts and a minimum learning rate per cycle:
    lr = eta_min_i + w_i/2 (base_lr - eta_min_i) (1 + cos(pi (t - start_i) / period_i)) inside cycle i."""

    def __init__(self, optimizer, periods, restart_weights=(1,), eta_mins=(0,), last_epoch=-1):
        if len(periods) != len(restart_weights):
            raise AssertionError("periods and restart_weights should have the same length.")
        self.periods, self.restart_weights, self.eta_mins = list(periods), list(restart_weights), list(eta_mins)
        self.ends = [sum(self.periods[:i + 1]) for i in range(len(self.periods))]
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        i = bisect.bisect_left(self.ends, self.last_epoch)          # first cycle whose end is >= t (lr_scheduler.py:8-23)
        start = 0 if i == 0 else self.ends[i - 1]
        w, lo, per = self.restart_weights[i], self.eta_mins[i], self.periods[i]
        c = 1 + math.cos(math.pi * (self.last_epoch - start) / per)
        return [lo + w * 0.5 * (b - lo) * c for b in self.base_lrs]
