"""SNR / Pearson correlation and the per-iteration History container (drop-in for reference utils/metrics.py).

Inside the optimisation loop the same quantities come out of the fused loss kernel (dpi_masked_loss);
these host/torch versions serve offline evaluation of reconstructed volumes."""
import numpy as np
import torch

from .generic import ten_digit

__all__ = ["snr", "pcorr", "History", "HistoryReg"]


def _lib(output, target):
    if target.shape != output.shape:
        raise ValueError("There is something wrong with the dimensions!")
    return torch if isinstance(output, torch.Tensor) and isinstance(target, torch.Tensor) else np


def snr(output, target):
    """10 log10( sum t^2 / sum (t-o)^2 ) in dB, on the full (unmasked) target."""
    xp = _lib(output, target)
    return 10 * xp.log10(xp.sum(target ** 2) / xp.sum((target - output) ** 2))


def pcorr(output, target):
    xp = _lib(output, target)
    td, od = target - xp.mean(target), output - xp.mean(output)
    return xp.sum(td * od) / (xp.sqrt(xp.sum(td ** 2)) * xp.sqrt(xp.sum(od ** 2)))


class History:
    """loss / snr / pcorr / lr lists, pickled into <patch>_run.npy (reference main.py:226-235)."""

    def __init__(self, epochs):
        self.loss, self.snr, self.pcorr, self.lr = [], [], [], []
        self.msg = "Iter %s, Loss = %+.2e, SNR = %+2.2f dB, PCORR = %+.2f %%"
        self.zfill = ten_digit(epochs)

    def __getitem__(self, i):
        return self.loss[i], self.snr[i], self.pcorr[i]

    def __setitem__(self, i, values):
        self.loss[i], self.snr[i], self.pcorr[i] = values

    def append(self, values):
        l, s, p = values
        self.loss.append(l)
        self.snr.append(s)
        self.pcorr.append(p)

    def __len__(self):
        assert len(self.loss) == len(self.snr) == len(self.pcorr) == len(self.lr)
        return len(self.loss)

    def log_message(self, idx):
        return self.msg % (str(idx + 1).zfill(self.zfill), self.loss[idx], self.snr[idx], self.pcorr[idx] * 100)

    def __str__(self):
        return "Loss : %s\nSNR  : %s\nPCORR: %s" % (self.loss, self.snr, self.pcorr)

    __repr__ = __str__


class HistoryReg:
    """History with the data-fidelity and regularisation terms split (reference utils/metrics.py:88-137; main_pocs.py:35)."""

    def __init__(self, epochs):
        self.loss, self.snr, self.pcorr, self.lr, self.df, self.reg = [], [], [], [], [], []
        self.msg = "Iter %s, Loss = %+.2e, DF = %.2e, REG = %.2e, SNR = %+.2f dB, PCORR = %+.2f %%"
        self.zfill = ten_digit(epochs)

    def __getitem__(self, i):
        return self.loss[i], self.reg[i], self.snr[i], self.pcorr[i]

    def __setitem__(self, i, values):
        self.loss[i], self.df[i], self.reg[i], self.snr[i], self.pcorr[i] = values

    def append(self, values):
        l, d, r, s, p = values
        self.loss.append(l)
        self.df.append(d)
        self.reg.append(r)
        self.snr.append(s)
        self.pcorr.append(p)

    def __len__(self):
        assert len(self.loss) == len(self.snr) == len(self.pcorr) == len(self.lr) == len(self.df) == len(self.reg)
        return len(self.loss)

    def log_message(self, idx):
        return self.msg % (str(idx + 1).zfill(self.zfill), self.loss[idx], self.df[idx], self.reg[idx], self.snr[idx], self.pcorr[idx] * 100)

    def __str__(self):
        return "Loss : %s\nReg  : %s\nSNR  : %s\nPCORR: %s" % (self.loss, self.reg, self.snr, self.pcorr)

    __repr__ = __str__
