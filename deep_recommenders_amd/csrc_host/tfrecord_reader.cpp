// Host-side input boundary (include/dr_input.h): TFRecord framing + a minimal tf.Example decoder.  No protobuf / TF
// dependency: the two wire formats are restated from TensorFlow's public specification (see the header).
#include "dr_input.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

uint32_t g_crc_table[8][256];
bool g_crc_ready = false;

void crc_init() {
    if (g_crc_ready) return;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xffu];
    g_crc_ready = true;
}

uint32_t crc32c(const uint8_t* p, int64_t n) {
    crc_init();
    uint32_t c = 0xffffffffu;
    while (n >= 8) {             // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_table[7][lo & 0xff] ^ g_crc_table[6][(lo >> 8) & 0xff] ^ g_crc_table[5][(lo >> 16) & 0xff] ^
            g_crc_table[4][lo >> 24] ^ g_crc_table[3][hi & 0xff] ^ g_crc_table[2][(hi >> 8) & 0xff] ^
            g_crc_table[1][(hi >> 16) & 0xff] ^ g_crc_table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = (c >> 8) ^ g_crc_table[0][(c ^ *p++) & 0xffu];
    return c ^ 0xffffffffu;
}

uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---- protobuf wire primitives --------------------------------------------------------------------------------
struct Span {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool empty() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    Span bytes() {              // length-delimited field
        const uint64_t len = varint();
        if (!ok || len > (uint64_t)(end - p)) {
            ok = false;
            return Span{end, end, false};
        }
        Span s{p, p + len};
        p += len;
        return s;
    }
    void skip(uint32_t wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: if (end - p >= 8) p += 8; else ok = false; break;
            case 2: bytes(); break;
            case 5: if (end - p >= 4) p += 4; else ok = false; break;
            default: ok = false;
        }
    }
};

// Finds the Feature message of `key` inside a serialized Example.  Returns 1 found, 0 missing, -1 malformed.  A key
// that appears twice keeps the LAST value (protobuf map semantics).
int find_feature(const uint8_t* rec, int64_t len, const char* key, size_t klen, Span* out) {
    Span ex{rec, rec + len};
    int found = 0;
    while (!ex.empty() && ex.ok) {
        const uint64_t tag = ex.varint();
        if (!ex.ok) return -1;
        if ((tag >> 3) == 1 && (tag & 7) == 2) {              // Example.features
            Span fs = ex.bytes();
            if (!ex.ok) return -1;
            while (!fs.empty() && fs.ok) {
                const uint64_t t2 = fs.varint();
                if (!fs.ok) return -1;
                if ((t2 >> 3) == 1 && (t2 & 7) == 2) {        // Features.feature map entry
                    Span entry = fs.bytes();
                    if (!fs.ok) return -1;
                    Span k{nullptr, nullptr}, v{nullptr, nullptr};
                    bool has_k = false, has_v = false;
                    while (!entry.empty() && entry.ok) {
                        const uint64_t t3 = entry.varint();
                        if (!entry.ok) return -1;
                        if ((t3 >> 3) == 1 && (t3 & 7) == 2) { k = entry.bytes(); has_k = true; }
                        else if ((t3 >> 3) == 2 && (t3 & 7) == 2) { v = entry.bytes(); has_v = true; }
                        else entry.skip((uint32_t)(t3 & 7));
                    }
                    if (!entry.ok) return -1;
                    if (has_k && (size_t)(k.end - k.p) == klen && memcmp(k.p, key, klen) == 0) {
                        if (has_v) *out = v; else *out = Span{rec, rec};      // present with an empty Feature
                        found = 1;
                    }
                } else {
                    fs.skip((uint32_t)(t2 & 7));
                }
            }
            if (!fs.ok) return -1;
        } else {
            ex.skip((uint32_t)(tag & 7));
        }
    }
    return ex.ok ? found : -1;
}

// Feature -> the list message of the requested kind (1 bytes_list, 3 int64_list).  1 ok, 0 other kind / unset, -1 malformed.
int feature_list(Span feat, uint32_t kind, Span* list) {
    int got = 0;
    while (!feat.empty() && feat.ok) {
        const uint64_t tag = feat.varint();
        if (!feat.ok) return -1;
        if ((tag & 7) == 2 && (tag >> 3) >= 1 && (tag >> 3) <= 3) {
            Span l = feat.bytes();
            if (!feat.ok) return -1;
            if ((tag >> 3) == kind) { *list = l; got = 1; }
            else got = 0;                                   // oneof: the last one set wins
        } else {
            feat.skip((uint32_t)(tag & 7));
        }
    }
    return feat.ok ? got : -1;
}

struct File {
    FILE* f;
    explicit File(const char* path) : f(fopen(path, "rb")) {}
    ~File() { if (f) fclose(f); }
};

}  // namespace

extern "C" const char* dri_version(void) { return "dr_input 1 (tfrecord + tf.Example, host)"; }

extern "C" uint32_t dri_crc32c(const uint8_t* data, int64_t n) { return (data && n > 0) ? crc32c(data, n) : crc32c(nullptr, 0); }

extern "C" int dri_tfrecord_index(const char* path, int32_t verify_crc, int64_t* offsets, int64_t* lengths, int64_t capacity,
                                  int64_t* count_out) {
    if (!path || !count_out || capacity < 0 || (capacity > 0 && (!offsets || !lengths))) return DRI_EINVAL;
    File file(path);
    if (!file.f) return DRI_EIO;
    int64_t count = 0, pos = 0;
    std::vector<uint8_t> buf;
    for (;;) {
        uint8_t hdr[12];
        const size_t got = fread(hdr, 1, 12, file.f);
        if (got == 0) break;                                   // clean end of file
        if (got != 12) return DRI_ECORRUPT;
        uint64_t len;
        uint32_t len_crc;
        memcpy(&len, hdr, 8);
        memcpy(&len_crc, hdr + 8, 4);
        if (verify_crc && masked(crc32c(hdr, 8)) != len_crc) return DRI_ECORRUPT;
        if (len > (1ull << 40)) return DRI_ECORRUPT;
        const int64_t payload = pos + 12;
        if (verify_crc) {
            buf.resize(len);
            if (len && fread(buf.data(), 1, len, file.f) != len) return DRI_ECORRUPT;
            uint32_t data_crc;
            if (fread(&data_crc, 1, 4, file.f) != 4) return DRI_ECORRUPT;
            if (masked(crc32c(buf.data(), (int64_t)len)) != data_crc) return DRI_ECORRUPT;
        } else {
            if (fseek(file.f, (long)(len + 4), SEEK_CUR) != 0) return DRI_ECORRUPT;
        }
        if (count < capacity) {
            offsets[count] = payload;
            lengths[count] = (int64_t)len;
        }
        ++count;
        pos = payload + (int64_t)len + 4;
    }
    if (!verify_crc) {                                         // a seek past the end does not fail: check the size
        if (fseek(file.f, 0, SEEK_END) != 0) return DRI_EIO;
        if (ftell(file.f) != pos) return DRI_ECORRUPT;
    }
    *count_out = count;
    return count > capacity ? DRI_ECAPACITY : DRI_OK;
}

extern "C" int dri_tfrecord_read(const char* path, const int64_t* offsets, const int64_t* lengths, int64_t n, uint8_t* out,
                                 int64_t out_capacity, int64_t* rec_offsets) {
    if (!path || n < 0 || (n > 0 && (!offsets || !lengths || !out)) || !rec_offsets) return DRI_EINVAL;
    File file(path);
    if (!file.f) return DRI_EIO;
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (lengths[i] < 0 || w + lengths[i] > out_capacity) return DRI_ECAPACITY;
        rec_offsets[i] = w;
        if (fseek(file.f, (long)offsets[i], SEEK_SET) != 0) return DRI_EIO;
        if (lengths[i] && fread(out + w, 1, (size_t)lengths[i], file.f) != (size_t)lengths[i]) return DRI_EIO;
        w += lengths[i];
    }
    rec_offsets[n] = w;
    return DRI_OK;
}

extern "C" int dri_example_int64(const uint8_t* records, const int64_t* rec_offsets, int64_t n, const char* key, int64_t* out) {
    if (n < 0 || !key || (n > 0 && (!records || !rec_offsets || !out))) return DRI_EINVAL;
    const size_t klen = strlen(key);
    for (int64_t i = 0; i < n; ++i) {
        Span feat{nullptr, nullptr}, list{nullptr, nullptr};
        const int64_t len = rec_offsets[i + 1] - rec_offsets[i];
        if (len < 0) return DRI_EINVAL;
        if (find_feature(records + rec_offsets[i], len, key, klen, &feat) != 1) return DRI_EPARSE;
        if (feature_list(feat, 3, &list) != 1) return DRI_EPARSE;
        int64_t nvals = 0, value = 0;
        while (!list.empty() && list.ok) {
            const uint64_t tag = list.varint();
            if (!list.ok) return DRI_EPARSE;
            if ((tag >> 3) == 1 && (tag & 7) == 2) {            // packed
                Span pk = list.bytes();
                if (!list.ok) return DRI_EPARSE;
                while (!pk.empty() && pk.ok) { value = (int64_t)pk.varint(); ++nvals; }
                if (!pk.ok) return DRI_EPARSE;
            } else if ((tag >> 3) == 1 && (tag & 7) == 0) {     // unpacked
                value = (int64_t)list.varint();
                ++nvals;
            } else {
                list.skip((uint32_t)(tag & 7));
            }
        }
        if (!list.ok || nvals != 1) return DRI_EPARSE;          // FixedLenFeature([]): exactly one value
        out[i] = value;
    }
    return DRI_OK;
}

extern "C" int dri_example_bytes(const uint8_t* records, const int64_t* rec_offsets, int64_t n, const char* key, int32_t varlen,
                                 uint8_t* blob, int64_t blob_capacity, int64_t* value_offsets, int64_t values_capacity,
                                 int64_t* row_splits, int64_t* totals_out) {
    if (n < 0 || !key || !totals_out || (n > 0 && (!records || !rec_offsets))) return DRI_EINVAL;
    if (blob != nullptr && (!value_offsets || !row_splits)) return DRI_EINVAL;
    const size_t klen = strlen(key);
    int64_t nvalues = 0, nbytes = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (blob != nullptr) row_splits[i] = nvalues;
        Span feat{nullptr, nullptr}, list{nullptr, nullptr};
        const int64_t len = rec_offsets[i + 1] - rec_offsets[i];
        if (len < 0) return DRI_EINVAL;
        const int ff = find_feature(records + rec_offsets[i], len, key, klen, &feat);
        if (ff < 0) return DRI_EPARSE;
        int64_t row_vals = 0;
        if (ff == 1) {
            const int fl = feature_list(feat, 1, &list);
            if (fl < 0) return DRI_EPARSE;
            if (fl == 0 && !(feat.empty())) return DRI_EPARSE;   // the key holds a list of another kind
            while (fl == 1 && !list.empty() && list.ok) {
                const uint64_t tag = list.varint();
                if (!list.ok) return DRI_EPARSE;
                if ((tag >> 3) == 1 && (tag & 7) == 2) {
                    Span v = list.bytes();
                    if (!list.ok) return DRI_EPARSE;
                    const int64_t vl = v.end - v.p;
                    if (blob != nullptr) {
                        if (nvalues >= values_capacity || nbytes + vl > blob_capacity) return DRI_ECAPACITY;
                        value_offsets[nvalues] = nbytes;
                        if (vl) memcpy(blob + nbytes, v.p, (size_t)vl);
                    }
                    ++nvalues;
                    ++row_vals;
                    nbytes += vl;
                } else {
                    list.skip((uint32_t)(tag & 7));
                }
            }
            if (fl == 1 && !list.ok) return DRI_EPARSE;
        }
        if (!varlen && row_vals != 1) return DRI_EPARSE;
    }
    if (blob != nullptr) {
        row_splits[n] = nvalues;
        if (nvalues < values_capacity + 1) value_offsets[nvalues] = nbytes; else return DRI_ECAPACITY;
    }
    totals_out[0] = nvalues;
    totals_out[1] = nbytes;
    return DRI_OK;
}
