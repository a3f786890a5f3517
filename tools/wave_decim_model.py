#!/usr/bin/env python3
"""Lane-level model of the wave-per-symbol FFT demodulator at decimation D = 2 / 4 / 8 (lora_wave_decim.inc.hip, and - D = 8 - lora_wave_demod.inc.hip):
the dataflow of one 64-lane wavefront in numpy, register by register, against the pruned DFT it restates (get_shift_fft, lib/decoder_impl.cc:430-464).
What it pins is the INDEX arithmetic - which bin a register of a lane holds after the in-lane FFT, the cross-lane stages and the reduce-scatter - and
the twiddle tables the host builds from it; the CPU suite runs it (tests/test_wave_decim_model.py)."""
import numpy as np


def brev(v, bits):
    r = 0
    for b in range(bits):
        if v & (1 << b):
            r |= 1 << (bits - 1 - b)
    return r


def layout_bin(J, logj, ld, g, lane):
    """bin held by register g of `lane` after the cross-lane FFT (the lane's r = lane & (D - 1) does not enter)"""
    b5, b4 = (lane >> 5) & 1, (lane >> 4) & 1
    s, t, i = g // (J // 2), (g // (J // 4)) & 1, g % (J // 4)
    e = i + (J // 4) * b4 + (J // 2) * b5
    k2 = s + 2 * t
    for b in range(3, ld - 1, -1):          # lane bit b is output bit 5 - b of the LQ-point DIF
        k2 += ((lane >> b) & 1) << (5 - b)
    return brev(e, logj) + J * k2


def reference_bins(x, down, N, D):
    sps = N * D
    F = np.fft.fft(x * down)
    out = np.empty(N, dtype=np.complex128)
    for j in range(N):
        out[j] = F[j] if j < N // 2 else F[sps - N + j]
    out[N // 2] = F[sps - N // 2] + F[N // 2]   # the fold (:447-450)
    return out


def wave_network(x, down, sf, ld):
    N, D = 1 << sf, 1 << ld
    sps, LQ = N * D, 64 >> ld
    J = sps // 64
    logj = J.bit_length() - 1
    lane = np.arange(64)
    lq, r = lane >> ld, lane & (D - 1)
    a = np.empty((J, 64), dtype=np.complex128)
    for j in range(J):
        a[j] = x[64 * j + lane] * down[64 * j + lane]
    h = J // 2                                # in-lane DIF, natural in, bit-reversed out
    while h >= 1:
        for b in range(J // 2):
            off, blk = b % h, b // h
            i0 = blk * 2 * h + off
            i1 = i0 + h
            u, v = a[i0].copy(), a[i1].copy()
            a[i0] = u + v
            a[i1] = (u - v) * np.exp(-2j * np.pi * off * (J // 2 // h) / J)
        h //= 2
    for m in range(J):
        a[m] = a[m] * np.exp(-2j * np.pi * ((lq * brev(m, logj)) % N) / N)
    # stages 1 and 2: the register-pair swap over lane bits 5 and 4
    for bit, half, groups, span in ((5, J // 2, 1, LQ // 2), (4, J // 4, 2, LQ // 4)):
        w = np.exp(-2j * np.pi * (lq % span) / (2 * span))
        up = ((lane >> bit) & 1) == 1
        for hgrp in range(groups):
            for i in range(half):
                g = i + hgrp * 2 * half if groups == 2 else i
                d, s_ = a[g].copy(), a[g + half].copy()
                part_d, part_s = d[lane ^ (1 << bit)], s_[lane ^ (1 << bit)]
                lo = np.where(up, part_s, d)
                hi = np.where(up, s_, part_d)
                a[g] = lo + hi
                a[g + half] = (lo - hi) * w
    # the remaining stages, each lane its own output
    for bit in range(3, ld - 1, -1):
        span = 1 << (bit - ld)                 # lq values below this stage's bit
        up = ((lane >> bit) & 1) == 1
        w = np.where(up, np.exp(-2j * np.pi * (lq % span) / (2 * span)), 1.0) if bit > ld else np.ones(64)
        for m in range(J):
            p = a[m][lane ^ (1 << bit)]
            a[m] = np.where(up, p - a[m], a[m] + p) * w
    # polyphase combine + fold
    for g in range(J):
        jb = np.array([layout_bin(J, logj, ld, g, l) for l in range(64)])
        k = np.where(jb < N // 2, jb, jb - N)
        tw = np.exp(-2j * np.pi * ((k * r) % sps) / sps)
        tw = tw + np.where(jb == N // 2, np.exp(-2j * np.pi * (((N // 2) * r) % sps) / sps), 0.0)
        a[g] = a[g] * tw
    # reduce-scatter over the r bits
    cur, cnt = a, J
    gbase = np.zeros(64, dtype=int)
    for t in range(ld):
        bit = ld - 1 - t
        half = cnt // 2
        up = ((lane >> bit) & 1) == 1
        nxt = np.empty((half, 64), dtype=np.complex128)
        for i in range(half):
            first = cur[i] + cur[i][lane ^ (1 << bit)]
            second = cur[i + half] + cur[i + half][lane ^ (1 << bit)]
            nxt[i] = np.where(up, second, first)
        gbase += np.where(up, J >> (t + 1), 0)
        cur, cnt = nxt, half
    out = np.zeros(N, dtype=np.complex128)
    seen = np.zeros(N, dtype=int)
    for i in range(cnt):
        for l in range(64):
            jb = layout_bin(J, logj, ld, gbase[l] + i, l)
            out[jb] = cur[i][l]
            seen[jb] += 1
    assert (seen == 1).all(), seen
    return out


def check(sf, ld, seed=0):
    rng = np.random.default_rng(seed)
    N, D = 1 << sf, 1 << ld
    sps = N * D
    x = rng.standard_normal(sps) + 1j * rng.standard_normal(sps)
    down = np.exp(-1j * np.pi * (np.arange(sps) ** 2) / (sps * D) * 1.0)
    got = wave_network(x, down, sf, ld)
    want = reference_bins(x, down, N, D)
    return float(np.abs(got - want).max() / np.abs(want).max())


if __name__ == "__main__":
    for ld in (1, 2, 3):
        for sf in (7, 8, 9):
            if (1 << sf << ld) // 64 > 64:
                continue
            print("D=%d SF%d: max rel err %.2e" % (1 << ld, sf, check(sf, ld)))
