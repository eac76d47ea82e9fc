// Host-side helpers of the input pipeline (no device code): CRC-32C for the TFRecord framing that the reference's data
// files use (data/build_tf_records.py:128-133 writes them with tf.io.TFRecordWriter; data/input_pipeline.py:31 reads them).
#include "common.h"
#include <string.h>

namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;      // Castagnoli, reflected
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};
}  // namespace

// CRC-32C (iSCSI polynomial) of data[0..n), continuing from `crc` (0 for a fresh sum), slicing-by-8.
extern "C" uint32_t frcnn_crc32c(uint32_t crc, const void* data, size_t n) {
    static const Crc32cTables tb;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = tb.t[7][lo & 0xFFu] ^ tb.t[6][(lo >> 8) & 0xFFu] ^ tb.t[5][(lo >> 16) & 0xFFu] ^ tb.t[4][lo >> 24] ^
            tb.t[3][hi & 0xFFu] ^ tb.t[2][(hi >> 8) & 0xFFu] ^ tb.t[1][(hi >> 16) & 0xFFu] ^ tb.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ tb.t[0][(c ^ *p++) & 0xFFu];
    return ~c;
}

// Hash of the kernel sources this library was built from (csrc/build.py passes it: sha1 over csrc/*.hip, csrc/*.h and the header, first
// 12 hex digits -- the same hash bench.py stamps its profiles with).  __graft_entry__.build() rebuilds with --force when it differs
// from the tree's: a pushed tree can never run stale objects.
#ifndef FRCNN_SOURCE_HASH
#define FRCNN_SOURCE_HASH "unknown"
#endif
extern "C" const char* frcnn_source_hash(void) { return FRCNN_SOURCE_HASH; }
