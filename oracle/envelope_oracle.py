"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the segment planner's burst-envelope pre-pass
(gr_lora_amd/csrc/lora_kernels.hip: envelope_kernel, edges_kernel), used by tests/ to check the device kernels.

The reference has no such stage (its decoder is one serial state machine, lib/decoder_impl.cc:740-903); the pre-pass only
decides where the MI355X scheduler cuts a stream into speculation segments, never what is decoded.  Definition:
  E[b]      = sum |x|^2 over 64 items of block b (one symbol): four runs of 16 items starting at
              base + l * sps / 4, l = 0..3, base = (stream offset + b * sps) rounded down to an even item
  quiet(b)  = E[b] < 0.5 * max(E[b-8 .. b+8])      (within the stream)
  gap start = b >= 1 with quiet(b) and not quiet(b-1); reported as the item position b * sps."""
import numpy as np

REACH = 8


def block_energy(iq_all: np.ndarray, off: int, length: int, sps: int) -> np.ndarray:
    x = np.asarray(iq_all)
    nb = length // sps
    p = (np.abs(x.astype(np.complex128)) ** 2)
    E = np.zeros(nb)
    for b in range(nb):
        base = (off + b * sps) & ~1
        for l in range(4):
            s = base + l * (sps // 4)
            E[b] += p[s:s + 16].sum()
    return E


def gap_starts(iq_all: np.ndarray, off: int, length: int, sps: int, rel_tol: float = 0.0):
    """Gap starts (item positions).  With rel_tol > 0 returns (certain, possible): `certain` are the gap starts whose
    two quiet / not-quiet decisions hold for any relative error up to rel_tol in the energies (the device sums in
    float32, in another order), `possible` those that hold for some such error -- e.g. a block whose sampled runs fall
    exactly half into a gap sits at E = 0.5 max exactly."""
    E = block_energy(iq_all, off, length, sps)
    nb = E.size
    m = np.array([E[max(0, b - REACH):min(nb, b + REACH + 1)].max() for b in range(nb)])

    def edges(q_now, q_prev):  # q_now[b]: block b counts as quiet; q_prev[b]: block b counts as not quiet
        return np.asarray([b for b in range(1, nb) if q_now[b] and q_prev[b - 1]], dtype=np.int64) * sps
    if rel_tol <= 0.0:
        q = E < 0.5 * m
        return edges(q, ~q)
    sure_quiet, maybe_quiet = E * (1 + rel_tol) < 0.5 * m * (1 - rel_tol), E * (1 - rel_tol) < 0.5 * m * (1 + rel_tol)
    return edges(sure_quiet, ~maybe_quiet), edges(maybe_quiet, ~sure_quiet)
