// TFRecord framing + tf.train.Example decoding: the DeepMind MeshGraphNets datasets as the reference
// reads them through TFRecord.jl (`read(path; channel_size)` at reference src/dataset.jl:107-112, records consumed by
// parse_data src/dataset.jl:61-75).  Host code, no GPU, no third-party library:
//   record   = uint64 length | uint32 masked_crc32c(length) | data[length] | uint32 masked_crc32c(data)
//   Example  = { features: { feature: map<string, Feature> } },  Feature = oneof { bytes_list, float_list, int64_list }
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgn_hip.h"

namespace {

uint32_t g_crc_table[8][256];
bool g_crc_ready = false;

void crc_init() {
    if (g_crc_ready) return;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;   // CRC-32C (Castagnoli), reflected
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFF];
    g_crc_ready = true;
}

uint32_t crc32c(const uint8_t* p, size_t n) {
    crc_init();
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {   // slicing-by-8
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_table[7][lo & 0xFF] ^ g_crc_table[6][(lo >> 8) & 0xFF] ^ g_crc_table[5][(lo >> 16) & 0xFF] ^ g_crc_table[4][lo >> 24] ^
            g_crc_table[3][hi & 0xFF] ^ g_crc_table[2][(hi >> 8) & 0xFF] ^ g_crc_table[1][(hi >> 16) & 0xFF] ^ g_crc_table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ g_crc_table[0][(c ^ *p++) & 0xFF];
    return c ^ 0xFFFFFFFFu;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
};

bool varint(Span& s, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && s.n > 0; shift += 7) {
        const uint8_t b = *s.p++;
        --s.n;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// next field of a protobuf message: field number, wire type, payload (length-delimited) or value (varint / fixed)
bool next_field(Span& s, uint32_t& field, uint32_t& wt, Span& payload, uint64_t& value) {
    uint64_t key;
    if (!varint(s, key)) return false;
    field = (uint32_t)(key >> 3);
    wt = (uint32_t)(key & 7);
    payload = Span();
    value = 0;
    switch (wt) {
        case 0: return varint(s, value);
        case 1: if (s.n < 8) return false; memcpy(&value, s.p, 8); s.p += 8; s.n -= 8; return true;
        case 5: if (s.n < 4) return false; { uint32_t v32; memcpy(&v32, s.p, 4); value = v32; } s.p += 4; s.n -= 4; return true;
        case 2: {
            uint64_t len;
            if (!varint(s, len) || len > s.n) return false;
            payload.p = s.p;
            payload.n = (size_t)len;
            s.p += len;
            s.n -= (size_t)len;
            return true;
        }
        default: return false;   // groups are not used by tf.train.Example
    }
}

struct Feature {
    std::string name;
    int32_t kind = 0;            // 1 bytes_list, 2 float_list, 3 int64_list
    Span bytes;                  // bytes_list: first value (what the reference reads: value[] of the bytes feature)
    std::vector<uint8_t> owned;  // float_list / int64_list gathered into one contiguous array
};

}  // namespace

struct mgn_tfrecord {
    FILE* f = nullptr;
    bool verify = true;
    std::vector<uint8_t> rec;
    std::vector<Feature> feats;
    std::string err;
    int64_t index = -1;
};

namespace {

bool parse_feature(Span body, Feature& ft, std::string& err) {
    uint32_t field, wt;
    Span pl;
    uint64_t val;
    while (body.n > 0) {
        if (!next_field(body, field, wt, pl, val)) { err = "malformed Feature"; return false; }
        if (wt != 2 || field < 1 || field > 3) continue;
        ft.kind = (int32_t)field;
        Span list = pl;          // BytesList / FloatList / Int64List: repeated field 1
        bool first_bytes = true;
        while (list.n > 0) {
            uint32_t f2, w2;
            Span p2;
            uint64_t v2;
            if (!next_field(list, f2, w2, p2, v2)) { err = "malformed value list"; return false; }
            if (f2 != 1) continue;
            if (field == 1) {            // bytes
                if (w2 != 2) { err = "bytes_list value is not length-delimited"; return false; }
                if (first_bytes) ft.bytes = p2;
                first_bytes = false;
            } else if (field == 2) {     // float: packed (wire type 2) or one fixed32 per entry
                if (w2 == 2) ft.owned.insert(ft.owned.end(), p2.p, p2.p + p2.n);
                else if (w2 == 5) { const uint32_t v32 = (uint32_t)v2; const uint8_t* q = reinterpret_cast<const uint8_t*>(&v32); ft.owned.insert(ft.owned.end(), q, q + 4); }
                else { err = "float_list value has a wrong wire type"; return false; }
            } else {                     // int64: packed varints or one varint per entry
                auto push = [&](uint64_t v) { const uint8_t* q = reinterpret_cast<const uint8_t*>(&v); ft.owned.insert(ft.owned.end(), q, q + 8); };
                if (w2 == 2) {
                    Span pv = p2;
                    while (pv.n > 0) {
                        uint64_t v;
                        if (!varint(pv, v)) { err = "malformed packed int64"; return false; }
                        push(v);
                    }
                } else if (w2 == 0) push(v2);
                else { err = "int64_list value has a wrong wire type"; return false; }
            }
        }
    }
    return true;
}

bool parse_example(mgn_tfrecord* r) {
    r->feats.clear();
    Span ex{r->rec.data(), r->rec.size()};
    uint32_t field, wt;
    Span pl;
    uint64_t val;
    while (ex.n > 0) {
        if (!next_field(ex, field, wt, pl, val)) { r->err = "malformed Example"; return false; }
        if (field != 1 || wt != 2) continue;          // Example.features
        Span feats = pl;
        while (feats.n > 0) {
            uint32_t f1, w1;
            Span entry;
            uint64_t v1;
            if (!next_field(feats, f1, w1, entry, v1)) { r->err = "malformed Features"; return false; }
            if (f1 != 1 || w1 != 2) continue;         // map entry
            Feature ft;
            Span body;
            while (entry.n > 0) {
                uint32_t f2, w2;
                Span p2;
                uint64_t v2;
                if (!next_field(entry, f2, w2, p2, v2)) { r->err = "malformed map entry"; return false; }
                if (f2 == 1 && w2 == 2) ft.name.assign(reinterpret_cast<const char*>(p2.p), p2.n);
                else if (f2 == 2 && w2 == 2) body = p2;
            }
            if (!parse_feature(body, ft, r->err)) return false;
            r->feats.push_back(std::move(ft));
        }
    }
    return true;
}

}  // namespace

extern "C" {

uint32_t mgn_crc32c(const void* data, size_t n) { return crc32c(static_cast<const uint8_t*>(data), n); }

int mgn_tfrecord_open(const char* path, int32_t verify_crc, mgn_tfrecord** out) {
    if (!path || !out) return MGN_E_ARG;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return MGN_E_ARG;
    mgn_tfrecord* r = new (std::nothrow) mgn_tfrecord();
    if (!r) { fclose(f); return MGN_E_OOM; }
    r->f = f;
    r->verify = verify_crc != 0;
    *out = r;
    return MGN_OK;
}

int mgn_tfrecord_next(mgn_tfrecord* r) {
    if (!r || !r->f) return MGN_E_ARG;
    uint8_t head[12];
    const size_t got = fread(head, 1, 12, r->f);
    if (got == 0) return 0;                               // clean end of file
    if (got != 12) { r->err = "truncated record header"; return MGN_E_ARG; }
    uint64_t len;
    uint32_t lcrc;
    memcpy(&len, head, 8);
    memcpy(&lcrc, head + 8, 4);
    if (r->verify && mask_crc(crc32c(head, 8)) != lcrc) { r->err = "length CRC mismatch at record " + std::to_string(r->index + 1); return MGN_E_ARG; }
    if (len > ((uint64_t)1 << 34)) { r->err = "implausible record length"; return MGN_E_ARG; }
    r->rec.resize((size_t)len);
    uint32_t dcrc;
    if (fread(r->rec.data(), 1, (size_t)len, r->f) != (size_t)len || fread(&dcrc, 1, 4, r->f) != 4) { r->err = "truncated record"; return MGN_E_ARG; }
    if (r->verify && mask_crc(crc32c(r->rec.data(), r->rec.size())) != dcrc) { r->err = "data CRC mismatch at record " + std::to_string(r->index + 1); return MGN_E_ARG; }
    ++r->index;
    if (!parse_example(r)) return MGN_E_ARG;
    return 1;
}

int mgn_tfrecord_feature_count(const mgn_tfrecord* r) { return r ? (int)r->feats.size() : MGN_E_ARG; }

const char* mgn_tfrecord_feature_name(const mgn_tfrecord* r, int32_t i) {
    return (r && i >= 0 && (size_t)i < r->feats.size()) ? r->feats[i].name.c_str() : nullptr;
}

int mgn_tfrecord_feature(const mgn_tfrecord* r, const char* key, int32_t* kind, const void** data, int64_t* nbytes) {
    if (!r || !key) return MGN_E_ARG;
    for (const Feature& ft : r->feats)
        if (ft.name == key) {
            if (kind) *kind = ft.kind;
            if (ft.kind == 1) {
                if (data) *data = ft.bytes.p;
                if (nbytes) *nbytes = (int64_t)ft.bytes.n;
            } else {
                if (data) *data = ft.owned.data();
                if (nbytes) *nbytes = (int64_t)ft.owned.size();
            }
            return MGN_OK;
        }
    return MGN_E_ARG;   // KeyError on the Julia side (data.features.feature[key], src/dataset.jl:64)
}

const char* mgn_tfrecord_error(const mgn_tfrecord* r) { return r ? r->err.c_str() : "null reader"; }

void mgn_tfrecord_close(mgn_tfrecord* r) {
    if (!r) return;
    if (r->f) fclose(r->f);
    delete r;
}

}  // extern "C"
