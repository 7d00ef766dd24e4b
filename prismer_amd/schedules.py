"""Learning-rate schedules of the reference training scripts (utils.py:13-31), as pure functions of the step so that the native
Trainer can evaluate them on the host each iteration (the value travels to the fused AdamW through a pinned scalar).

  fine-tuning  (train_caption.py:127, train_vqa.py:119): cosine over ITERATIONS, from init_lr to min_lr
  pre-training (train_pretrain.py:113-122): cosine over EPOCHS, set at every epoch start; during the first `warmup_steps`
               steps the linear warm-up overrides it each step (and its last value stays until the next epoch start)
"""
import math


def cosine_lr(it, total, init_lr, min_lr):
    """utils.py:13-17 (the reference passes iterations for fine-tuning, epochs for pre-training)."""
    return (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * it / total)) + min_lr


def warmup_lr(step, max_step, init_lr, max_lr):
    """utils.py:20-24."""
    return min(max_lr, init_lr + (max_lr - init_lr) * step / max_step)


def step_lr(epoch, init_lr, min_lr, decay_rate):
    """utils.py:27-31."""
    return max(min_lr, init_lr * (decay_rate ** epoch))


def finetune_schedule(total_steps, init_lr, min_lr=0.0):
    return lambda it: cosine_lr(it, total_steps, init_lr, min_lr)


def pretrain_schedule(steps_per_epoch, max_epoch, init_lr, min_lr, warmup_init_lr, warmup_steps):
    """lr(it) of the pre-training loop (train_pretrain.py:112-122), it = global step counted from 0."""
    def lr(it):
        epoch = it // steps_per_epoch
        if it < warmup_steps:
            return warmup_lr(it, warmup_steps, warmup_init_lr, init_lr)
        if epoch * steps_per_epoch < warmup_steps:          # this epoch started inside the warm-up: its cosine value was overridden
            return warmup_lr(warmup_steps - 1, warmup_steps, warmup_init_lr, init_lr)
        return cosine_lr(epoch, max_epoch, init_lr, min_lr)
    return lr
