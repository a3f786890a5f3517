"""The C++ host mirror (csrc/decoder_block.hpp) compiles against the C ABI with plain g++
and, on a GPU box, decodes a stream to the known answer."""
import os
import subprocess
import textwrap

import numpy as np
import pytest

from gr_lora_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include <cstdio>
    #include <fstream>
    #include <vector>
    #include "gr_lora_amd/csrc/decoder_block.hpp"
    int main(int argc, char **argv) {
        if (argc < 2) return 2;
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const std::complex<float> *iq = reinterpret_cast<const std::complex<float> *>(raw.data());
        const int n = (int)(raw.size() / sizeof(std::complex<float>));
        auto dec = lora_hip::decoder::make(1e6f, 125000, 7, false, 4, true, false, false);
        dec->subscribe_frames([](const std::vector<uint8_t> &b) {
            for (size_t i = 15; i < b.size(); i++) std::printf("%02x", b[i]);
            std::printf("\n");
        });
        const int chunk = 2 * (int)dec->output_multiple();
        for (int i = 0; i < n; i += chunk) dec->work(n - i < chunk ? n - i : chunk, iq + i);
        dec->stop();
        return 0;
    }
''')


def _build(tmp_path):
    from gr_lora_amd import build
    build.build_library()
    src = tmp_path / "host_main.cpp"
    src.write_text(SRC)
    exe = tmp_path / "host_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "gr_lora_amd"), "-llora_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "gr_lora_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_host_decodes_known_answer(tmp_path):
    exe = _build(tmp_path)
    iq_path = os.path.join(ROOT, "tests", "golden", "sf7_cr4_deadbeef_x2.cf32")
    out = subprocess.check_output([str(exe), iq_path], timeout=120).decode().split()
    assert out == ["049040deadbeef700d"] * 2
