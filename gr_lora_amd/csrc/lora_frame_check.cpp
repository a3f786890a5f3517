// lora_frame_check.cpp -- frame validity (beyond the reference): the PHY header checksum and the payload
// CRC of a published frame blob.  Host only, no device work, no handle; see include/lora_hip.h for the contract and the
// reference lines (README.md:12, include/lora/utilities.h:396-404, lib/decoder_impl.cc:643,839).
#include <cstdint>
#include <cstring>

#include "../../include/lora_hip.h"

namespace {
constexpr int kLoratapLen = 15; // sizeof(loratap_header_t), include/lora/loratap.h:35-55
}

extern "C" {

lora_hip_status lora_hip_check_frame(const uint8_t *blob, size_t len, lora_hip_frame_check_t *out)
{
    if (!blob || !out) return LORA_HIP_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    if (len < (size_t)kLoratapLen + 3u) return LORA_HIP_ERR_ARG;
    const uint8_t *ph = blob + kLoratapLen, *pl = ph + 3;
    const uint32_t length = ph[0], cr = ph[1] >> 5, has_crc = (ph[1] >> 4) & 1u;
    out->has_crc = (uint8_t)has_crc;
    out->has_header = len == (size_t)kLoratapLen + 3u + length + 2u * has_crc;
    // parity sets over a0..a7 = length (MSB first), a8..a10 = cr (MSB first), a11 = has_crc (utilities.h:398-402 in its own bit numbering)
    const uint32_t a = (length << 4) | (cr << 1) | has_crc; // a0 is bit 11
    auto bit = [&](int i) { return (a >> (11 - i)) & 1u; };
    const uint32_t c4 = bit(0) ^ bit(1) ^ bit(2) ^ bit(3), c3 = bit(0) ^ bit(4) ^ bit(5) ^ bit(6) ^ bit(11),
                   c2 = bit(1) ^ bit(4) ^ bit(7) ^ bit(8) ^ bit(10), c1 = bit(2) ^ bit(5) ^ bit(7) ^ bit(9) ^ bit(10) ^ bit(11),
                   c0 = bit(3) ^ bit(6) ^ bit(8) ^ bit(9) ^ bit(10) ^ bit(11);
    out->header_checksum_calc = (uint8_t)((c4 << 4) | (c3 << 3) | (c2 << 2) | (c1 << 1) | c0);
    out->header_checksum_rx = (uint8_t)((((uint32_t)ph[1] << 4) & 0x10u) | (ph[2] >> 4)); // d_phy_crc (:839), 5 bits
    out->header_checksum_ok = out->header_checksum_calc == out->header_checksum_rx;
    if (!out->has_header || !has_crc) return LORA_HIP_OK;
    auto whiten_at = [](uint32_t idx) { // byte idx of the payload whitening sequence
        uint8_t r = 0xff;
        for (uint32_t i = 0; i < idx; i++) r = (uint8_t)((r << 1) | (((r >> 7) ^ (r >> 5) ^ (r >> 4) ^ (r >> 3)) & 1u));
        return r;
    };
    uint16_t crc = 0;
    for (uint32_t i = 0; i + 2u < length; i++) {
        crc ^= (uint16_t)(pl[i] << 8);
        for (int k = 0; k < 8; k++) crc = (crc & 0x8000u) ? (uint16_t)((crc << 1) ^ 0x1021u) : (uint16_t)(crc << 1);
    }
    if (length >= 1u) crc ^= pl[length - 1u];
    if (length >= 2u) crc ^= (uint16_t)(pl[length - 2u] << 8);
    out->crc_calc = crc;
    out->crc_rx = (uint16_t)((pl[length] ^ whiten_at(length)) | ((pl[length + 1u] ^ whiten_at(length + 1u)) << 8));
    out->crc_ok = out->crc_calc == out->crc_rx;
    return LORA_HIP_OK;
}


} // extern "C"
