"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's muscle-condition action maps
(/root/reference/myosuite/envs/myo/base_v0.py:83-109 and envs/myo/fatigue.py:8-108), used by tests/ as the checker for the
action-map stage of the HIP step kernel.  Parity unpinned (the reference holds no golden vectors for these)."""
import numpy as np


def sigmoid_map(a):
    return 1.0 / (1.0 + np.exp(-5.0 * (np.asarray(a, np.float64) - 0.5)))       # base_v0.py:87-91


class Fatigue3CCr:
    """CumulativeFatigue (fatigue.py:8-108): compartments MA / MR / MF per muscle, batched over leading axes."""

    def __init__(self, tauact, taudeact, dt, shape):
        self.r, self.F, self.R = 10 * 15, 0.00912, 0.1 * 0.00094                  # fatigue.py:10-18
        self.tauact, self.taudeact, self.dt = np.asarray(tauact, float), np.asarray(taudeact, float), float(dt)
        self.MA, self.MR, self.MF = np.zeros(shape), np.ones(shape), np.zeros(shape)

    def compute_act(self, act):                                                   # fatigue.py:61-108
        TL = np.asarray(act, np.float64)
        MA, MR, MF, dt = self.MA, self.MR, self.MF, self.dt
        LD = 1 / self.tauact * (0.5 + 1.5 * MA)
        LR = (0.5 + 1.5 * MA) / self.taudeact
        C = np.where(MA < TL, np.where(MR > TL - MA, LD * (TL - MA), LD * MR), LR * (TL - MA))
        rR = np.where(MA >= TL, self.r * self.R, self.R)
        C = np.clip(C, np.maximum(-MA / dt + self.F * MA, (MR - 1) / dt + rR * MF), np.minimum((1 - MA) / dt + self.F * MA, MR / dt + rR * MF))
        self.MA = MA + (C - self.F * MA) * dt
        self.MR = MR + (-C + rR * MF) * dt
        self.MF = MF + (self.F * MA - rR * MF) * dt
        return self.MA


def reafferentation_map(ctrl, epl, eip):                                          # base_v0.py:105-109
    c = np.array(ctrl, copy=True)
    c[..., epl] = c[..., eip]
    c[..., eip] = 0
    return c
