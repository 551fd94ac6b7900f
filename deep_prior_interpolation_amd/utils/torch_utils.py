"""torch-side helpers (drop-in for reference utils/torch.py): weight init, noise, converters, device
selection on ROCm, seeding, early stopping."""
import os

import numpy as np
import torch

__all__ = ["init_weights", "get_noise", "np_to_torch", "torch_to_np", "batch_channel_add", "batch_channel_del",
           "set_gpu", "get_gpu_name", "set_seed", "EarlyStopping"]


def init_weights(net, init_type="normal", init_gain=0.02, verbose=False):
    """Conv*/Linear weights: normal | xavier | kaiming | orthogonal, bias 0;
    BatchNorm weight ~ N(10, 10*init_gain) (sic: the reference uses mean 10, utils/torch.py:52), bias 0."""

    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, "weight") and ("Conv" in name or "Linear" in name):
            if init_type == "normal":
                torch.nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                torch.nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                torch.nn.init.kaiming_normal_(m.weight.data, a=0.2, mode="fan_in")
            elif init_type == "orthogonal":
                torch.nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if getattr(m, "bias", None) is not None:
                torch.nn.init.constant_(m.bias.data, 0.0)
        elif "BatchNorm" in name:
            torch.nn.init.normal_(m.weight.data, 10.0, init_gain * 10)
            torch.nn.init.constant_(m.bias.data, 0.0)

    if init_type != "default":
        net.apply(init_func)
        if verbose:
            print("parameters initialized with %s" % init_type)


def get_noise(shape, noise_type):
    x = torch.zeros(shape)
    if noise_type == "u":
        x.uniform_()
    elif noise_type == "n":
        x.normal_()
    elif noise_type == "c":
        x.cauchy_()
    else:
        raise ValueError("Noise type has to be one of [u, n, c]")
    return x


def batch_channel_add(t):
    return t.unsqueeze(0).unsqueeze(0)


def batch_channel_del(t):
    return t.squeeze(0).squeeze(0)


def np_to_torch(in_content, bc_add=True):
    t = torch.from_numpy(in_content.copy())
    return batch_channel_add(t) if bc_add else t


def torch_to_np(in_content, bc_del=True):
    a = in_content.detach().cpu().numpy()
    return a.squeeze() if bc_del else a


def set_gpu(id=-1):
    """Select the HIP device (None = refuse: the engine has no CPU path).  -1 picks the device with most free HBM."""
    if id is None:
        raise RuntimeError("--gpu is required: deep_prior_interpolation_amd has no CPU path")
    n = torch.cuda.device_count()
    if n == 0:
        raise RuntimeError("no HIP device visible")
    if id == -1 or id >= n:
        free = []
        for i in range(n):
            f, _ = torch.cuda.mem_get_info(i)
            free.append(f)
        id = int(np.argmax(free))
    torch.cuda.set_device(id)
    print("GPU selected: %d - %s" % (id, torch.cuda.get_device_name(id)))
    return id


def get_gpu_name(id=None):
    if not torch.cuda.is_available():
        return "CPU"
    id = torch.cuda.current_device() if id is None else id
    return "%s (%d)" % (torch.cuda.get_device_name(id), id)


def set_seed(seed=0):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


class EarlyStopping:
    """Stop when the metric has not improved by min_delta (absolute, or percent of the best) for `patience` steps;
    a NaN metric stops immediately (reference utils/torch.py:216-275)."""

    def __init__(self, patience=10, max=False, min_delta=0, percentage=False):
        self.mode = "max" if max else "min"
        self.min_delta, self.patience, self.percentage = min_delta, patience, percentage
        self.best = None
        self.num_bad_epochs = 0
        self.msg = "\nEarly stopping called, terminating..."

    def is_better(self, a, best):
        if self.patience == 0:
            return True
        delta = best * self.min_delta / 100 if self.percentage else self.min_delta
        return a < best - delta if self.mode == "min" else a > best + delta

    def step(self, metrics):
        if self.patience == 0:
            return False
        m = float(metrics)
        if self.best is None:
            self.best = m
            return False
        if m != m:
            print("Metrics is NaN, terminating...")
            return True
        if self.is_better(m, self.best):
            self.num_bad_epochs = 0
            self.best = m
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs >= self.patience:
            print(self.msg)
            return True
        return False
