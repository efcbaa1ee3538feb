"""The two audio_dspy 0.0.4 helpers lpf.py uses (lpf.py:12,58,61); the package itself is not in
the image.  Own implementations of the documented behaviour: an exponential sine sweep and a
first-order lowpass designed with the bilinear transform."""
import numpy as np


def sweep_log(f0, f1, duration, fs):
    """Logarithmic sine sweep from f0 to f1 Hz over `duration` seconds."""
    n = int(duration * fs)
    t = np.arange(n) / fs
    k = np.log(f1 / f0)
    return np.sin(2.0 * np.pi * f0 * duration / k * (np.exp(t / duration * k) - 1.0))


def design_LPF1(fc, fs):  # noqa: N802
    """(b, a) of a first-order lowpass with cutoff fc (bilinear transform, pre-warped)."""
    c = 1.0 / np.tan(np.pi * fc / fs)
    a0 = c + 1.0
    return np.array([1.0 / a0, 1.0 / a0]), np.array([1.0, (1.0 - c) / a0])
