"""Synthetic LoRa transmit model and workload generator.

The reference is receive-only (no modulator anywhere in its tree), so the
synthetic IQ used by the tests, the smoke check and bench.py comes from this
module.  It is the algebraic inverse of the reference's decode chain, derived
step by step from the decoder (paths relative to the reference tree):

  nibbles      <- loraphy_header_t layout (include/lora/loraphy.h:25-32) and the
                  byte assembly of hamming_decode / extract_data_only
                  (lib/decoder_impl.cc:661-664, 697-705)
  Hamming      <- hamming_encode_soft (include/lora/utilities.h:257-264)
  whitening    <- dewhiten (lib/decoder_impl.cc:579-580, 639-645; lib/tables.h)
  shuffle      <- deshuffle pattern {5,0,1,2,4,3,6,7} (lib/decoder_impl.cc:568, 611-621)
  interleave   <- deinterleave (lib/decoder_impl.cc:535-565)
  gray / rate  <- demodulate (lib/decoder_impl.cc:507-512)
  shift        <- max_frequency_gradient_idx returns (s-1) mod N for a chirp
                  advanced by s bins (lib/decoder_impl.cc:479-490)
  waveform     <- build_ideal_chirps (lib/decoder_impl.cc:141-175) and the
                  consumption pattern of FIND_SFD / PAUSE (:816, :822)

Nothing here runs on the product path; it only manufactures inputs.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_SHUFFLE = (5, 0, 1, 2, 4, 3, 6, 7)


def _load_whitening():
    """Parse the generated data include shared with the kernels (values only)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "whitening_data.inc")
    src = open(path).read()
    out = {}
    for name in ("LORA_WHITEN_HEADER", "LORA_WHITEN_CR56", "LORA_WHITEN_CR78"):
        m = re.search(name + r"\[\d+\]\s*=\s*\{([^}]*)\}", src)
        out[name] = np.array([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))], dtype=np.uint8)
    return out


_WHITEN = _load_whitening()


def hamming_encode(nibble: int) -> int:
    b = [(nibble >> i) & 1 for i in range(4)]
    p1 = b[1] ^ b[2] ^ b[3]
    p2 = b[0] ^ b[1] ^ b[2]
    p3 = b[0] ^ b[1] ^ b[3]
    p4 = b[0] ^ b[2] ^ b[3]
    return p1 | (b[0] << 1) | (b[1] << 2) | (b[2] << 3) | (p2 << 4) | (b[3] << 5) | (p3 << 6) | (p4 << 7)


_ENC = [hamming_encode(n) for n in range(16)]


def _shuffle_tx(pre: int) -> int:
    """Inverse of the receiver's deshuffle: rx computes out bit j = in bit pat[j]."""
    cw = 0
    for j, p in enumerate(_SHUFFLE):
        cw |= ((pre >> j) & 1) << p
    return cw


def _rotr(v: int, count: int, size: int) -> int:
    count %= size
    mask = (1 << size) - 1
    v &= mask
    return ((v >> count) | (v << (size - count))) & mask


def _gray_inverse(w: int) -> int:
    """b such that b ^ (b >> 1) == w (the receiver gray-codes the bin, :512)."""
    b = 0
    while w:
        b ^= w
        w >>= 1
    return b


def _interleave_block(codewords: Sequence[int], ppm: int, width: int) -> List[int]:
    """ppm codewords -> `width` words of ppm bits (inverse of deinterleave)."""
    words = []
    for i in range(width):
        wp = 0
        for x in range(ppm):
            wp |= ((codewords[x] >> i) & 1) << x
        words.append(_rotr(wp, i, ppm))
    return words


@dataclass
class TxConfig:
    sf: int = 7
    cr: int = 4
    bw: int = 125000
    samp_rate: float = 1e6
    crc: bool = True
    reduced_rate: bool = False
    implicit: bool = False
    preamble_len: int = 8
    sync_shifts: Optional[Tuple[int, int]] = None  # None -> (3N/16, N/4) = (24, 32) at SF7
    hdr_nibbles: Tuple[int, int] = (0, 4)  # (crc_msn, 5th header nibble); README vector has 0,4

    @property
    def nbins(self) -> int:
        return 1 << self.sf

    @property
    def sps(self) -> int:
        return int(np.uint32(np.float64(np.uint32(self.samp_rate)) / (np.float64(self.bw) / (1 << self.sf))))

    @property
    def decim(self) -> int:
        return self.sps // self.nbins


def payload_symbol_count(length_with_crc: int, sf: int, cr: int, reduced_rate: bool) -> int:
    """Restates lib/decoder_impl.cc:842-847 in float32 like the reference."""
    f = np.float32
    spb = cr + 4
    bits = f(length_with_crc) * f(8.0)
    symbols_needed = bits * (f(spb) / f(4.0)) / f(sf - (2 if reduced_rate else 0))
    blocks = int(np.ceil(symbols_needed / f(spb)))
    return blocks * spb


def encode_shifts(payload: bytes, cfg: TxConfig, crc_bytes: bytes = b"\x70\x0d") -> Tuple[List[int], List[int]]:
    """Returns (header_shifts[8], payload_shifts[...]) : cyclic advance in bins."""
    sf, cr, N = cfg.sf, cfg.cr, cfg.nbins
    if sf < 7 and not cfg.implicit:
        raise ValueError("transmit model needs sf >= 7 for an explicit header (the header block holds 5 codewords; SF6: implicit header only)")
    body = bytes(payload) + (bytes(crc_bytes[:2]) if cfg.crc else b"")
    nibbles = []
    for b in body:
        nibbles += [b & 15, b >> 4]  # low nibble first
    prng = _WHITEN["LORA_WHITEN_CR56"] if cr <= 2 else _WHITEN["LORA_WHITEN_CR78"]
    ppm_h = sf - 2
    ppm_p = sf - 2 if cfg.reduced_rate else sf
    n_blocks = payload_symbol_count(len(body), sf, cr, cfg.reduced_rate) // (cr + 4)
    n_slots = (ppm_h if cfg.implicit else ppm_h - 5) + n_blocks * ppm_p
    pay_cw = []
    for i in range(n_slots):
        nib = nibbles[i] if i < len(nibbles) else 0
        w = int(prng[i]) if i < len(prng) else 0
        pay_cw.append(_shuffle_tx(_ENC[nib] ^ w))
    if cfg.implicit:
        first = pay_cw[:ppm_h]
        rest = pay_cw[ppm_h:]
    else:
        hn = [len(payload) >> 4, len(payload) & 15, (cr << 1) | (1 if cfg.crc else 0), cfg.hdr_nibbles[0], cfg.hdr_nibbles[1]]
        first = [_shuffle_tx(_ENC[n]) for n in hn] + pay_cw[:ppm_h - 5]
        rest = pay_cw[ppm_h - 5:]
    hdr_words = _interleave_block(first, ppm_h, 8)
    hdr_shifts = [(4 * _gray_inverse(w) + 1) % N for w in hdr_words]
    pay_shifts = []
    mult = 4 if cfg.reduced_rate else 1
    for b in range(n_blocks):
        words = _interleave_block(rest[b * ppm_p:(b + 1) * ppm_p], ppm_p, cr + 4)
        pay_shifts += [(mult * _gray_inverse(w) + 1) % N for w in words]
    return hdr_shifts, pay_shifts


def base_upchirp(cfg: TxConfig, dtype=np.complex64) -> np.ndarray:
    """Unit-amplitude upchirp, -bw/2 -> +bw/2 over one symbol, phase 0 at n=0."""
    n = np.arange(cfg.sps, dtype=np.float64)
    t = n / float(cfg.samp_rate)
    sym_rate = cfg.bw / float(cfg.nbins)
    phase = 2.0 * np.pi * t * (-cfg.bw / 2.0 + 0.5 * cfg.bw * sym_rate * t)
    return np.exp(1j * phase).astype(dtype)


def frame_shift_plan(hdr_shifts: Sequence[int], pay_shifts: Sequence[int], cfg: TxConfig):
    """Symbol plan of one frame: list of (kind, shift, n_samples); kind 0 up, 1 down."""
    sps = cfg.sps
    plan = [(0, 0, sps)] * cfg.preamble_len
    s1, s2 = cfg.sync_shifts if cfg.sync_shifts is not None else (3 * cfg.nbins // 16, cfg.nbins // 4)
    plan += [(0, s1 % cfg.nbins, sps), (0, s2 % cfg.nbins, sps)]
    plan += [(1, 0, sps), (1, 0, sps), (1, 0, sps // 4)]
    plan += [(0, s, sps) for s in hdr_shifts]
    plan += [(0, s, sps) for s in pay_shifts]
    return plan


def modulate_frame(hdr_shifts, pay_shifts, cfg: TxConfig, amplitude: float = 1.0) -> np.ndarray:
    up = base_upchirp(cfg)
    down = np.conj(up)
    sps, D = cfg.sps, cfg.decim
    plan = frame_shift_plan(hdr_shifts, pay_shifts, cfg)
    out = np.empty(sum(p[2] for p in plan), dtype=np.complex64)
    pos = 0
    ar = np.arange(sps)
    for kind, s, n in plan:
        src = down if kind else up
        out[pos:pos + n] = src[(ar[:n] + s * D) % sps]
        pos += n
    if amplitude != 1.0:
        out *= np.float32(amplitude)
    return out


@dataclass
class SynthStream:
    iq: np.ndarray                      # complex64
    payloads: List[bytes]
    frame_starts: List[int]             # first preamble sample of each frame
    header_starts: List[int]            # first header-symbol sample of each frame
    shifts: List[Tuple[List[int], List[int]]] = field(default_factory=list)


def build_stream(payloads: Sequence[bytes], cfg: TxConfig, gaps: Optional[Sequence[int]] = None,
                 rng: Optional[np.random.Generator] = None, gap_symbols: Tuple[float, float] = (2.0, 6.0),
                 lead: Optional[int] = None, tail_symbols: float = 3.0, crc_bytes: bytes = b"\x70\x0d",
                 noise_sigma: float = 0.0, cfo_hz: float = 0.0, amplitude: float = 1.0) -> SynthStream:
    """Concatenate frames separated by zero gaps.

    gaps[i] = samples of silence before frame i (random in gap_symbols if None).
    The tail keeps >= 2 symbols after the last frame: the reference block only
    runs while 2*sps input items remain (set_output_multiple, decoder_impl.cc:91).
    """
    rng = rng or np.random.default_rng(0)
    sps = cfg.sps
    frames, fs, hs, sh = [], [], [], []
    pos = 0
    pieces = []
    for i, p in enumerate(payloads):
        if gaps is not None:
            g = int(gaps[i])
        elif i == 0 and lead is not None:
            g = int(lead)
        else:
            g = int(rng.integers(int(gap_symbols[0] * sps), int(gap_symbols[1] * sps) + 1))
        h, q = encode_shifts(p, cfg, crc_bytes)
        fr = modulate_frame(h, q, cfg, amplitude)
        pieces.append(np.zeros(g, dtype=np.complex64))
        pieces.append(fr)
        fs.append(pos + g)
        hs.append(pos + g + (cfg.preamble_len + 2) * sps + 2 * sps + sps // 4)
        sh.append((h, q))
        pos += g + len(fr)
    pieces.append(np.zeros(int(tail_symbols * sps), dtype=np.complex64))
    iq = np.concatenate(pieces)
    if cfo_hz != 0.0:
        n = np.arange(len(iq), dtype=np.float64)
        iq = (iq * np.exp(2j * np.pi * cfo_hz * n / cfg.samp_rate)).astype(np.complex64)
    if noise_sigma > 0.0:
        noise = rng.standard_normal((len(iq), 2)).astype(np.float32) * np.float32(noise_sigma / np.sqrt(2.0))
        iq = (iq + noise[:, 0] + 1j * noise[:, 1]).astype(np.complex64)
    return SynthStream(iq=iq, payloads=[bytes(p) for p in payloads], frame_starts=fs, header_starts=hs, shifts=sh)


def expected_frame_tail(payload: bytes, cfg: TxConfig, crc_bytes: bytes = b"\x70\x0d") -> bytes:
    """Bytes the reference publishes after the 15-byte loratap header."""
    b1 = (cfg.cr << 5) | ((1 if cfg.crc else 0) << 4) | (cfg.hdr_nibbles[0] & 15)
    b2 = (cfg.hdr_nibbles[1] & 15) << 4
    return bytes([len(payload), b1, b2]) + bytes(payload) + (bytes(crc_bytes[:2]) if cfg.crc else b"")


def awgn_sigma_for_snr(snr_db_inband: float, cfg: TxConfig, amplitude: float = 1.0) -> float:
    """Complex-noise sigma (per complex sample, full fs band) giving the requested
    SNR inside the LoRa bandwidth: noise power in-band = sigma^2 * bw / fs."""
    p_sig = amplitude ** 2
    p_noise_inband = p_sig / (10.0 ** (snr_db_inband / 10.0))
    return float(np.sqrt(p_noise_inband * cfg.samp_rate / cfg.bw))

# ---- header checksum / payload CRC as a LoRa transmitter computes them (the reference checks neither, README.md:12;
# lora_hip_check_frame does).  In the decoder's byte domain: the CRC field is not whitened on air but de-whitened like data
# by the reference (decoder_impl.cc:643), so the bytes that make a frame valid are the true CRC XOR the whitening bytes.
def whitening_bytes(n: int) -> bytes:
    r, out = 0xFF, bytearray()
    for _ in range(n):
        out.append(r)
        r = ((r << 1) | (((r >> 7) ^ (r >> 5) ^ (r >> 4) ^ (r >> 3)) & 1)) & 0xFF
    return bytes(out)


def lora_crc16(payload: bytes) -> int:
    crc = 0
    for b in payload[:-2] if len(payload) >= 2 else b"":
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    if len(payload) >= 1:
        crc ^= payload[-1]
    if len(payload) >= 2:
        crc ^= payload[-2] << 8
    return crc


def valid_crc_bytes(payload: bytes) -> bytes:
    """The two bytes to pass as `crc_bytes` so that the decoded frame carries a valid payload CRC."""
    c, w = lora_crc16(payload), whitening_bytes(len(payload) + 2)
    return bytes([(c & 0xFF) ^ w[len(payload)], (c >> 8) ^ w[len(payload) + 1]])


def valid_hdr_nibbles(length: int, cr: int, crc: bool) -> Tuple[int, int]:
    """`hdr_nibbles` (low nibble of PHY byte 1, high nibble of PHY byte 2) holding the valid 5-bit header checksum."""
    a = (length << 4) | (cr << 1) | (1 if crc else 0)
    bit = lambda i: (a >> (11 - i)) & 1
    c4 = bit(0) ^ bit(1) ^ bit(2) ^ bit(3)
    c3 = bit(0) ^ bit(4) ^ bit(5) ^ bit(6) ^ bit(11)
    c2 = bit(1) ^ bit(4) ^ bit(7) ^ bit(8) ^ bit(10)
    c1 = bit(2) ^ bit(5) ^ bit(7) ^ bit(9) ^ bit(10) ^ bit(11)
    c0 = bit(3) ^ bit(6) ^ bit(8) ^ bit(9) ^ bit(10) ^ bit(11)
    return c4, (c3 << 3) | (c2 << 2) | (c1 << 1) | c0
