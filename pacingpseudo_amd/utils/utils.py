"""Scalar schedule helpers with the reference's names and semantics (utils/utils.py:7-84).

These set the learning rate, the loss ramp-up weights and the epoch meters of the training driver; they are plain
host arithmetic (a handful of flops per epoch), so they stay in Python."""
import math

_POLICIES = {
    'linear': lambda s, n, lr, g: (1 - s / n) * lr,
    'cosine': lambda s, n, lr, g: 0.5 * (1 + math.cos(s * math.pi / n)) * lr,
    'poly': lambda s, n, lr, g: lr * (1 - s / n) ** g,
}


def _decay(policy, optimizer, step, num_steps, base_lr, gamma=0.9):
    new_lr = _POLICIES[policy](step, num_steps, base_lr, gamma)
    for group in optimizer.param_groups:
        group['lr'] = new_lr
    return optimizer, new_lr


def linear_lr_decay(optimizer, step, num_steps, base_lr):
    """new_lr = (1 - step/num_steps) * base_lr  (utils/utils.py:7-20)."""
    return _decay('linear', optimizer, step, num_steps, base_lr)


def cosine_lr_decay(optimizer, step, num_steps, base_lr):
    """new_lr = (1 + cos(pi*step/num_steps))/2 * base_lr  (utils/utils.py:22-35)."""
    return _decay('cosine', optimizer, step, num_steps, base_lr)


def poly_lr_decay(optimizer, step, num_steps, base_lr, gamma=0.9):
    """new_lr = base_lr * (1 - step/num_steps)**gamma  (utils/utils.py:37-51)."""
    return _decay('poly', optimizer, step, num_steps, base_lr, gamma)


def gaussian_ramp_up(t, base_value, max_t=80, scale=5.):
    """base_value * exp(-scale * (1 - t/max_t)) while t < max_t, base_value afterwards (utils/utils.py:53-65)."""
    return base_value * math.exp(-scale * (1 - t / max_t)) if t < max_t else base_value


class AvgMeter(object):
    """Running weighted mean (utils/utils.py:67-84)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
