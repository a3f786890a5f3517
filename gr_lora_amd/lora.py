"""GNU-Radio-free mirror of the reference's block API for the decoder path.

Same names, argument order and meaning as the reference module `lora`:
  lora.decoder(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction)
      (python/bindings/decoder_python.cc:36-66, include/lora/decoder.h:705)
  lora.lora_receiver(samp_rate, center_freq, channel_list, bandwidth, sf, implicit, cr, crc,
                     reduced_rate=False, conj=False, decimation=1, disable_channelization=False,
                     disable_drift_correction=False)            (python/lora_receiver.py:30)
  lora.message_socket_sink(ip, port, layer)   (lib/message_socket_sink_impl.cc:93-122)
  lora.message_file_sink(path)                (lib/message_file_sink_impl.cc)
Blocks exchange frames through message ports named as upstream ("frames",
"control"); `msg_connect(src, "frames", dst, "in")` wires them.  Samples are
pushed with `work(items)` the way the GNU Radio scheduler calls
decoder_impl::work (any chunking), and `stop()` plays the end of the flowgraph.

The decoder runs on the MI355X through the C ABI (include/lora_hip.h); there is
no CPU path here.  The channeliser of lora_receiver is host code (the step before
the hot path).
"""
from __future__ import annotations

import socket
import sys
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import capi

LORATAP_LEN = 15   # sizeof(loratap_header_t), include/lora/loratap.h:35-55
LORAPHY_LEN = 3    # sizeof(loraphy_header_t), include/lora/loraphy.h:25-32
MAC_CRC_SIZE = 2   # include/lora/utilities.h:29


class _MsgBlock:
    """Minimal message-port plumbing (gr::basic_block::message_port_pub/register)."""

    def __init__(self):
        self._out: Dict[str, List[Callable[[bytes], None]]] = {}
        self._in: Dict[str, Callable[[bytes], None]] = {}

    def message_port_register_out(self, name: str):
        self._out.setdefault(name, [])

    def message_port_register_in(self, name: str, handler: Callable[[bytes], None]):
        self._in[name] = handler

    def message_port_pub(self, name: str, blob: bytes):
        for cb in self._out.get(name, []):
            cb(blob)

    def subscribe(self, name: str, cb: Callable[[bytes], None]):
        if name not in self._out:
            raise KeyError("no message port '%s'" % name)
        self._out[name].append(cb)


def msg_connect(src: _MsgBlock, src_port: str, dst, dst_port: str = "in"):
    """top_block.msg_connect((src, port), (dst, port))"""
    if callable(dst) and not isinstance(dst, _MsgBlock):
        src.subscribe(src_port, dst)
    else:
        src.subscribe(src_port, dst._in[dst_port])


def _hex_line(data: bytes, endline: bool, ascii_part: bool) -> str:
    """print_vector_hex (include/lora/utilities.h:351-368)"""
    s = "".join(" %02x" % b for b in data)
    if ascii_part:
        s += " (" + "".join(chr(b) for b in data if 0x20 <= b <= 0x7e) + ")"
    return s + ("\n" if endline else "")


class decoder(_MsgBlock):
    """gr::lora::decoder on the MI355X.  demod: capi.DEMOD_FFT_COMPAT (default; dechirp x FFT x
    argmax, byte-identical to the upstream default path's bin convention), DEMOD_FFT, or
    DEMOD_GRAD (the upstream default estimator itself)."""

    def __init__(self, samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate=False,
                 disable_drift_correction=False, *, device=0, demod=capi.DEMOD_FFT_COMPAT, verbose=True,
                 batch_items=0, segment_symbols=0, cfo_estimates=False):
        super().__init__()
        self._cfo = bool(cfo_estimates)
        self._hist = None                              # cfo_estimates: ring buffer of the most recent input, for the preamble windows
        self._hist_end = 0                             # absolute item index one past the newest item in the ring
        self.cfo_dropped = 0                           # estimates not published because their window had left the ring
        if sf < 6 or sf > 12:  # decoder_impl.cc:57-61 -- the reference prints this and exit(1)s
            sys.stderr.write("[LoRa Decoder] ERROR : Spreading factor should be between 6 and 12 (inclusive)!\n"
                             "                       Other values are currently not supported.\n")
            raise SystemExit(1)
        self._verbose = verbose
        self._h = capi.Handle(samp_rate=samp_rate, bandwidth=bandwidth, sf=sf, implicit=implicit, cr=cr, crc=crc,
                              reduced_rate=reduced_rate, disable_drift_correction=disable_drift_correction,
                              device=device, demod=demod, batch_items=batch_items, segment_symbols=segment_symbols)
        self.samples_per_symbol = self._h.sps
        self.number_of_bins = self._h.nbins
        self.decim_factor = self._h.decim
        if verbose:  # decoder_impl.cc:93-103
            bits_per_symbol = float(sf) * (4.0 / (4.0 + cr))
            print("Bits (nominal) per symbol: \t%g" % bits_per_symbol)
            print("Bins per symbol: \t%d" % self.number_of_bins)
            print("Samples per symbol: \t%d" % self.samples_per_symbol)
            print("Decimation: \t\t%d" % self.decim_factor)
            if disable_drift_correction:
                print("Warning: clock drift correction disabled")
            if implicit:
                print("CR: \t\t%d" % cr)
                print("CRC: \t\t%d" % int(bool(crc)))
        if self._cfo:
            # A frame surfaces at most two device passes after its last sample (one being filled, one in flight) and its
            # preamble lies a whole packet further back: the ring holds 2 passes of the library's EFFECTIVE batch size
            # (lora_hip_stream_info; 2M items at SF12, not the constructor argument) + the longest packet + a margin.
            eff_batch = int(self._h.stream_info().batch_items)
            ppm = int(sf) - 2 if reduced_rate else int(sf)          # bits per payload symbol (decoder_impl.cc:842-847)
            longest = (14 + 8 + 8 * -(-(2 * 257 * 8) // (8 * ppm))) * self._h.sps   # preamble + header + ceil(257 B at CR 4/8 / block) blocks of 8
            self._hist = np.zeros(2 * eff_batch + longest + 8 * self._h.sps, dtype=np.complex64)
        self.message_port_register_out("frames")    # decoder_impl.cc:120
        self.message_port_register_out("control")   # :121 (registered, never published upstream)

    # scheduler-facing ------------------------------------------------------
    def output_multiple(self) -> int:
        return 2 * self.samples_per_symbol           # set_output_multiple (:91)

    def work(self, input_items) -> int:
        """Consumes every item handed in (buffers internally); publishes finished frames."""
        x = np.asarray(input_items)
        if self._cfo:   # preallocated ring: one copy of the new items, nothing re-copied
            xc = x.astype(np.complex64, copy=False).ravel()
            cap = self._hist.size
            if xc.size >= cap:
                xc = xc[-cap:]
                self._hist_end += int(x.size) - cap
            w = self._hist_end % cap
            first = min(xc.size, cap - w)
            self._hist[w:w + first] = xc[:first]
            self._hist[:xc.size - first] = xc[first:]
            self._hist_end += xc.size
        n = self._h.work(x)
        self._publish()
        return n

    def stop(self):
        """End of stream: decode what is buffered, like the scheduler draining its last buffers."""
        self._h.flush()
        self._publish()

    def _publish(self):
        for blob, _info in self._h.drain():
            if self._verbose:  # :832 and :872
                sys.stdout.write(_hex_line(blob[LORATAP_LEN:LORATAP_LEN + LORAPHY_LEN], False, False))
                sys.stdout.write(_hex_line(blob[LORATAP_LEN + LORAPHY_LEN:], True, True))
            self.message_port_pub("frames", blob)
            if self._cfo:
                self._publish_cfo(_info)

    def _publish_cfo(self, info):
        """What decoder_impl.cc:774-776 (commented out upstream) would publish: ("cfo", Hz) on the "control" port, from
        experimental_determine_cfo (:730-738) over a preamble upchirp.  The window is the last unmodulated upchirp but one
        in front of the frame (header - 6.25 symbols: 2.25 SFD + 2 sync words + 2), on the symbol clock the SFD search
        settled; the estimate is the mean over the window (mode 1), not upstream's single sample."""
        import torch
        sps = self.samples_per_symbol
        cap = self._hist.size
        a0 = int(info.header_pos) - (25 * sps) // 4           # absolute item index of the window
        if a0 < max(self._hist_end - cap, 0) or a0 + sps > self._hist_end:
            self.cfo_dropped += 1                                # (counted, not silent: the ring was sized too small for this traffic)
            return
        idx = (a0 + np.arange(sps)) % cap
        d = torch.from_numpy(np.ascontiguousarray(self._hist[idx]).view(np.float32)).to("cuda:%d" % self._h.device)
        hz = float(self._h.estimate_cfo_device(d.data_ptr(), sps, [0], mode=1)[0])
        self.message_port_pub("control", ("cfo", hz, a0))      # (the window's position lets a consumer ignore stale estimates)

    # decoder.h:708-709 ------------------------------------------------------
    def set_sf(self, sf):
        self._h.L.lora_hip_set_sf(self._h.h, int(sf))

    def set_samp_rate(self, samp_rate):
        self._h.L.lora_hip_set_samp_rate(self._h.h, float(samp_rate))

    def close(self):
        self._h.close()


def low_pass_taps(gain, fs, cutoff, transition):
    """gr::filter::firdes::low_pass(gain, fs, cutoff, transition, WIN_HAMMING) as GNU Radio 3.9 computes it:
    ntaps = int(53 * fs / (22 * transition)) made odd, float Hamming window, float taps, normalised so that the DC
    gain (taps[M] + 2 * sum of one side) is `gain`.  Same steps as firdes_low_pass in csrc/lora_channelizer.hip."""
    ntaps = int(53.0 * fs / (22.0 * transition))
    if ntaps % 2 == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    fw = 2.0 * np.pi * cutoff / fs
    w = (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(ntaps) / (ntaps - 1))).astype(np.float32)
    n = np.arange(-m, m + 1, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        taps = np.where(n == 0, fw / np.pi * w, np.sin(n * fw) / (n * np.pi) * w).astype(np.float32)
    fmax = float(taps[m]) + 2.0 * float(taps[m + 1:].astype(np.float64).sum())
    return (taps.astype(np.float64) * (gain / fmax)).astype(np.float32)


class channelizer:
    """gr::lora::channelizer: freq_xlating_fir_filter_ccf(decimation, low_pass(1, fs, bw/2 + 15 kHz,
    10 kHz, Hamming), channel_list[0] - center_freq, fs)  (lib/channelizer_impl.cc:46-57), on the GPU through
    lora_hip_channelizer_* (include/lora_hip_channelizer.h).  Like the reference only channel_list[0] is output.
    There is no host fall-back: without the HIP library and a device the constructor raises."""

    def __init__(self, samp_rate, center_freq, channel_list, bandwidth, decimation=1, device=0):
        from . import capi
        self.fs = float(samp_rate)
        self.center_freq = float(center_freq)
        self.channel_list = list(channel_list)
        self.bandwidth = int(bandwidth)
        self.decimation = int(decimation)
        self.device = device
        self._h = capi.Channelizer(samp_rate, center_freq, self.channel_list[:1], bandwidth, decimation, device)
        self.taps = self._h.taps()

    def work(self, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.complex64)
        if x.size == 0:
            return x
        return self._h.work(x)[0]

    def apply_cfo(self, cfo):                       # channelizer_impl.cc:68-71
        self._h.apply_cfo(cfo)

    def retune(self, center_freq):
        """A new capture centre frequency (filter history and oscillator phase start over)."""
        from . import capi
        self._h.close()
        self.center_freq = float(center_freq)
        self._h = capi.Channelizer(self.fs, self.center_freq, self.channel_list[:1], self.bandwidth, self.decimation, self.device)


class lora_receiver(_MsgBlock):
    """python/lora_receiver.py:26-89: [conjugate] o (channelizer | resampler) -> decoder."""

    def __init__(self, samp_rate, center_freq, channel_list, bandwidth, sf, implicit, cr, crc, reduced_rate=False,
                 conj=False, decimation=1, disable_channelization=False, disable_drift_correction=False, cfo_correction=False,
                 **decoder_kw):
        super().__init__()
        self.samp_rate = samp_rate
        self.center_freq = center_freq
        self.channel_list = channel_list
        self.bandwidth = bandwidth
        self.sf = sf
        self.implicit = implicit
        self.cr = cr
        self.crc = crc
        self.decimation = decimation
        self.conj = conj
        self.disable_channelization = disable_channelization
        self.disable_drift_correction = disable_drift_correction
        self.channelizer = None if disable_channelization else channelizer(samp_rate, center_freq, channel_list, bandwidth, decimation)
        self.decoder = decoder(samp_rate / decimation, bandwidth, sf, implicit, cr, crc, reduced_rate,
                               disable_drift_correction, cfo_estimates=bool(cfo_correction), **decoder_kw)
        self.cfo_log = []                             # estimates in Hz, as applied
        self._fed = 0                                 # items handed to the decoder so far
        self._cfo_valid_from = 0
        if cfo_correction and self.channelizer is not None:
            # msg_connect((self.decoder, 'control'), (self.channelizer, 'control')) (python/lora_receiver.py:67) with the
            # message upstream left commented out; conj flips the spectrum in front of the decoder, hence the sign
            def on_control(msg):
                if isinstance(msg, tuple) and msg[0] == "cfo":
                    if msg[2] < self._cfo_valid_from:   # measured on samples filtered before the last correction: already accounted for
                        return
                    hz = -msg[1] if self.conj else msg[1]
                    self.cfo_log.append(hz)
                    self.channelizer.apply_cfo(hz)
                    self._cfo_valid_from = self._fed    # decoder-input index from which the new tuning holds
            self.decoder.subscribe("control", on_control)
        self.message_port_register_out("frames")     # message_port_register_hier_out('frames')
        self.decoder.subscribe("frames", lambda blob: self.message_port_pub("frames", blob))

    def work(self, input_items) -> int:
        x = np.asarray(input_items, dtype=np.complex64)
        if self.disable_channelization:
            y = x[:: int(self.decimation)] if self.decimation != 1 else x   # fractional_resampler_cc(0, decimation)
        else:
            y = self.channelizer.work(x)
        if self.conj:
            y = np.conj(y)
        self._fed += int(np.asarray(y).size)
        self.decoder.work(y)
        return x.size

    def stop(self):
        self.decoder.stop()

    def get_sf(self):
        return self.sf

    def set_sf(self, sf):
        self.sf = sf
        self.decoder.set_sf(self.sf)

    def get_center_freq(self):
        return self.center_freq

    def set_center_freq(self, center_freq):
        # upstream calls a channelizer method that does not exist (lora_receiver.py:89); here it retunes
        self.center_freq = center_freq
        if self.channelizer is not None:
            self.channelizer.retune(center_freq)


class message_socket_sink(_MsgBlock):
    """PMT blob -> UDP datagram; `layer` strips headers (lib/message_socket_sink_impl.cc:93-122):
    0 LORATAP: whole blob; 1 LORAPHY: drop loratap; 2 LORAMAC: drop loratap + PHY header and the 2 CRC bytes."""

    def __init__(self, ip="127.0.0.1", port=40868, layer=0):
        super().__init__()
        self.addr = (ip, int(port))
        self.layer = int(layer)
        self._sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.message_port_register_in("in", self.handle)

    def handle(self, blob: bytes):
        if self.layer == 0:
            data = blob
        elif self.layer == 1:
            data = blob[LORATAP_LEN:]
        else:
            phy = blob[LORATAP_LEN:LORATAP_LEN + LORAPHY_LEN]
            has_mac_crc = (phy[1] >> 4) & 1
            end = len(blob) - (MAC_CRC_SIZE if has_mac_crc else 0)
            data = blob[LORATAP_LEN + LORAPHY_LEN:end]
        self._sock.sendto(data, self.addr)

    def close(self):
        self._sock.close()


class message_file_sink(_MsgBlock):
    """PMT blob -> appended to a binary file, flushed (lib/message_file_sink_impl.cc)."""

    def __init__(self, path):
        super().__init__()
        self._f = open(path, "ab")
        self.message_port_register_in("in", self.handle)

    def handle(self, blob: bytes):
        self._f.write(blob)
        self._f.flush()

    def close(self):
        self._f.close()


class LoRaUDPServer:
    """python/lorasocket.py:18-34: collect n datagrams as hex strings."""

    def __init__(self, ip="127.0.0.1", port=40868, timeout=10):
        self.s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.s.bind((ip, port))
        self.s.settimeout(timeout)

    def get_payloads(self, number_of_payloads):
        out = []
        for _ in range(number_of_payloads):
            try:
                data = self.s.recvfrom(65535)[0]
                out.append(data.hex() if data else "")
            except socket.timeout:
                pass
        return out

    def close(self):
        self.s.close()
