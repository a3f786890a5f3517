"""The lane-level model of the wave-per-symbol FFT demodulator (tools/wave_decim_model.py) against the pruned DFT it restates
(get_shift_fft, decoder_impl.cc:430-464): pins the index arithmetic lora_wave_decim.inc.hip and its host-built tables follow, at
decimation 2, 4 and - the layout lora_wave_demod.inc.hip has used since round 2 - 8."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("wave_decim_model", os.path.join(os.path.dirname(__file__), "..", "tools", "wave_decim_model.py"))
model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(model)


@pytest.mark.parametrize("ld", [1, 2, 3])
@pytest.mark.parametrize("sf", [7, 8, 9])
def test_network_equals_pruned_dft(sf, ld):
    assert model.check(sf, ld, seed=sf * 10 + ld) < 1e-12


@pytest.mark.parametrize("ld", [1, 2, 3])
def test_every_bin_has_one_owner(ld):
    """after the reduce-scatter every bin sits in exactly one (register, lane) - and all lanes of a polyphase group agree on a register's bin"""
    sf = 8
    J = ((1 << sf) << ld) // 64
    logj = J.bit_length() - 1
    for g in range(J):
        for lane in range(64):
            assert model.layout_bin(J, logj, ld, g, lane) == model.layout_bin(J, logj, ld, g, lane & ~((1 << ld) - 1))
    bins = sorted(model.layout_bin(J, logj, ld, g, lane) for g in range(J) for lane in range(0, 64, 1 << ld))
    assert bins == list(range(1 << sf))
