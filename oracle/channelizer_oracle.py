"""CPU ORACLE for the channeliser (test infrastructure only -- see oracle/lora_oracle.h for the rule: only tests/,
__graft_entry__.smoke() and bench legs may import this; the product path is gr_lora_amd/csrc/lora_channelizer.hip).

Restates gr::lora::channelizer (reference lib/channelizer_impl.cc:46-71):
    d_lpf = firdes::low_pass(1.0, samp_rate, (bandwidth/2)+15000, 10000, WIN_HAMMING, 6.67)          (:46)
    d_freq_offset = channel_list[0] - center_freq                                                     (:47)
    freq_xlating_fir_filter_ccf(decimation, d_lpf, d_freq_offset, samp_rate)                          (:48)
    apply_cfo(cfo): d_cfo += cfo; set_center_freq(d_freq_offset + d_cfo)                              (:68-71)
GNU Radio (3.9, CMakeLists.txt:87 of the reference) is not under /root/reference; its published algorithms are
restated: firdes::low_pass (gr-filter/lib/firdes.cc: ntaps = (int)(53 fs / (22 tw)) made odd, Hamming-windowed
sinc in float, normalised to unit DC gain) and freq_xlating_fir_filter (band-pass taps h[k] e^{+j theta k}, decimating
FIR, output rotator e^{-j theta D m}), which is y[m] = sum_k h[k] x[mD-k] e^{-j theta (mD-k)} with zero initial
history.  PARITY UNPINNED at the sample level: the reference holds no vectors for this block; what pins it is the
end-to-end known answer (README.md:75-85 trace through channeliser + decoder) and the filter's design properties.
Arithmetic here is float64 throughout (the taps are rounded to float like GNU Radio's), so it is the exact value
the float32 device kernel is compared against with a stated tolerance.
"""
from __future__ import annotations

import numpy as np


def firdes_low_pass(gain: float, fs: float, cutoff: float, transition: float) -> np.ndarray:
    ntaps = int(53.0 * fs / (22.0 * transition))
    if ntaps % 2 == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    fw = 2.0 * np.pi * cutoff / fs
    w = (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(ntaps) / (ntaps - 1))).astype(np.float32)   # fft::window::hamming
    n = np.arange(-m, m + 1, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        taps = np.where(n == 0, fw / np.pi * w, np.sin(n * fw) / (n * np.pi) * w).astype(np.float32)
    fmax = float(taps[m]) + 2.0 * float(taps[m + 1:].astype(np.float64).sum())
    return (taps.astype(np.float64) * (gain / fmax)).astype(np.float32)


class Channelizer:
    """Streaming restatement; one instance per channel (the reference translates channel_list[0] only)."""

    def __init__(self, samp_rate, center_freq, channel_freq, bandwidth, decimation=1, uint32_offset=False):
        self.fs = float(samp_rate)
        self.decimation = int(decimation)
        self.taps = firdes_low_pass(1.0, samp_rate, float(int(bandwidth) // 2) + 15000.0, 10000.0)
        # d_freq_offset = channel_list[0] - center_freq with float arguments, stored in a uint32_t (channelizer_impl.cc:47,
        # channelizer_impl.h:39): float32 subtraction, truncation to whole Hz.  A negative offset wraps upstream (an
        # unsigned field; the float -> unsigned conversion of a negative value is undefined behaviour); here and in the
        # device library it keeps its sign, which is what the block is meant to do.
        self.freq = float(np.trunc(np.float32(channel_freq) - np.float32(center_freq)))
        # uint32_offset: upstream's arithmetic as it stands - the value lands in a uint32_t (a negative one wraps: the x86-64 conversion goes
        # through int64), and apply_cfo adds d_cfo to it in float (LORA_HIP_CHANNELIZER_FLAG_UINT32_OFFSET)
        self.uint32_offset = bool(uint32_offset)
        if self.uint32_offset:
            self.freq = float(int(np.trunc(np.float32(channel_freq) - np.float32(center_freq))) % (1 << 32))
        self.cfo = 0.0
        self._hist = np.zeros(len(self.taps) - 1, dtype=np.complex128)   # raw input history (filter delay line)
        self._n = 0            # absolute index of the next input item
        self._phase = 0.0      # rotator phase (turns) at self._n

    def apply_cfo(self, cfo):
        # d_cfo += cfo; set_center_freq(d_freq_offset + d_cfo) (:68-71): freq_xlating rebuilds its band-pass taps for
        # the new frequency (they apply to the delay line as it stands) and its rotator keeps its phase
        self.cfo += float(np.float32(cfo))
        if self.uint32_offset:   # d_freq_offset + d_cfo: uint32_t + float -> float
            self.cfo = float(np.float32(self.cfo))
            self._f_override = float(np.float32(np.float32(self.freq) + np.float32(self.cfo)))

    def work(self, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.complex64).astype(np.complex128)
        if x.size == 0:
            return np.zeros(0, dtype=np.complex128)
        tps = (getattr(self, "_f_override", None) if getattr(self, "_f_override", None) is not None else (self.freq + self.cfo)) / self.fs
        k = np.arange(len(self.taps), dtype=np.float64)
        bp = self.taps.astype(np.float64) * np.exp(2j * np.pi * tps * k)          # band-pass taps h[k] e^{+j theta k}
        buf = np.concatenate([self._hist, x])
        y = np.convolve(buf, bp, mode="valid")                                      # y[i] <-> absolute index self._n + i
        ph = self._phase + tps * np.arange(x.size, dtype=np.float64)
        y = y * np.exp(-2j * np.pi * (ph - np.floor(ph)))                           # rotator
        first = (-self._n) % self.decimation
        out = y[first::self.decimation]
        self._hist = buf[-(len(self.taps) - 1):]
        self._n += x.size
        p = self._phase + tps * x.size
        self._phase = p - np.floor(p)
        return out
