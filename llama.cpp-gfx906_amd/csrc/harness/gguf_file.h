// gguf_file.h — a read-only, memory-mapped GGUF (v2 / v3) file: key/value metadata and tensor placement.
// File layout as the reference reads it (gguf-py/gguf/gguf_reader.py:132-190, gguf-py/gguf/constants.py:10-12,2791-2804; the C reader the
// model loader uses is ggml/src/gguf.cpp, absent from the reference tree):
//   u32 magic "GGUF" | u32 version | u64 n_tensors | u64 n_kv
//   n_kv x { string key | u32 type | value }           string = u64 length + bytes; array = u32 item type + u64 count + items
//   n_tensors x { string name | u32 n_dims | u64 ne[n_dims] | u32 ggml type | u64 offset }
//   padding to general.alignment (default 32) | tensor data (offsets are relative to this point)
// Little-endian files only (the reader's byte-swapped variant is for big-endian hosts' files).
#pragma once

#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace mi355x {

enum gguf_vtype : uint32_t { GV_U8 = 0, GV_I8, GV_U16, GV_I16, GV_U32, GV_I32, GV_F32, GV_BOOL, GV_STR, GV_ARR, GV_U64, GV_I64, GV_F64 };

struct gguf_value {
    uint32_t type = GV_U32, item_type = GV_U32;      // item_type: for arrays
    uint64_t count = 1;                               // array length (1 for scalars)
    const uint8_t * data = nullptr;                   // scalars / arrays of scalars: the little-endian bytes in the mapping
    std::vector<std::string> strs;                    // a string, or an array of strings
};

struct gguf_tensor_info {
    std::string name;
    uint32_t n_dims = 0, type = 0;
    int64_t ne[4] = { 1, 1, 1, 1 };
    uint64_t offset = 0;                              // relative to data_offset
};

static inline size_t gguf_scalar_size(uint32_t t) {
    switch (t) {
        case GV_U8: case GV_I8: case GV_BOOL: return 1;
        case GV_U16: case GV_I16: return 2;
        case GV_U32: case GV_I32: case GV_F32: return 4;
        case GV_U64: case GV_I64: case GV_F64: return 8;
        default: return 0;
    }
}

struct gguf_file {
    int fd = -1;
    const uint8_t * base = nullptr; size_t size = 0;
    uint32_t version = 0; uint64_t alignment = 32, data_offset = 0;
    std::vector<std::string> keys;                    // in file order
    std::map<std::string, gguf_value> kv;
    std::vector<gguf_tensor_info> tensors;
    std::map<std::string, size_t> tensor_index;

    gguf_file() = default;
    gguf_file(const gguf_file &) = delete;
    ~gguf_file() { if (base) munmap((void *) base, size); if (fd >= 0) close(fd); }

    // ---- cursor over the mapping; every read is bounds-checked: a truncated or corrupt file is an error, never a wild read
    size_t pos = 0;
    const uint8_t * take(size_t n) {
        if (n > size || pos > size - n) throw std::runtime_error("gguf: file truncated at offset " + std::to_string(pos));
        const uint8_t * p = base + pos; pos += n; return p;
    }
    template <typename T> T rd() { T v; memcpy(&v, take(sizeof(T)), sizeof(T)); return v; }
    std::string rd_str() {
        const uint64_t n = rd<uint64_t>();
        if (n > size) throw std::runtime_error("gguf: string length out of range");
        const uint8_t * p = take((size_t) n);
        return std::string((const char *) p, (size_t) n);
    }

    void open(const char * path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) throw std::runtime_error(std::string("gguf: cannot open ") + path);
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 24) throw std::runtime_error("gguf: file too small");
        size = (size_t) st.st_size;
        void * p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) { base = nullptr; throw std::runtime_error("gguf: mmap failed"); }
        base = (const uint8_t *) p;

        if (rd<uint32_t>() != 0x46554747u) throw std::runtime_error("gguf: bad magic");
        version = rd<uint32_t>();
        if (version != 2 && version != 3) throw std::runtime_error("gguf: unsupported version " + std::to_string(version));
        const uint64_t n_tensors = rd<uint64_t>(), n_kv = rd<uint64_t>();
        if (n_tensors > size/24 || n_kv > size/12) throw std::runtime_error("gguf: implausible counts");

        for (uint64_t i = 0; i < n_kv; i++) {
            std::string key = rd_str();
            gguf_value v;
            v.type = rd<uint32_t>();
            auto scalars = [&](uint32_t t, uint64_t cnt) {
                const size_t es = gguf_scalar_size(t);
                if (!es) throw std::runtime_error("gguf: unknown value type " + std::to_string(t) + " for key " + key);
                if (cnt > size/es) throw std::runtime_error("gguf: array length out of range");
                v.data = take((size_t) cnt*es);
            };
            if (v.type == GV_STR) {
                v.strs.push_back(rd_str());
            } else if (v.type == GV_ARR) {
                v.item_type = rd<uint32_t>(); v.count = rd<uint64_t>();
                if (v.item_type == GV_STR) {
                    if (v.count > size/8) throw std::runtime_error("gguf: array length out of range");
                    v.strs.reserve((size_t) v.count);
                    for (uint64_t j = 0; j < v.count; j++) v.strs.push_back(rd_str());
                } else if (v.item_type == GV_ARR) {
                    throw std::runtime_error("gguf: nested arrays are not supported (key " + key + ")");
                } else {
                    scalars(v.item_type, v.count);
                }
            } else {
                scalars(v.type, 1);
            }
            if (kv.count(key)) throw std::runtime_error("gguf: duplicate key " + key);
            keys.push_back(key);
            kv.emplace(std::move(key), std::move(v));
        }
        if (has("general.alignment")) alignment = get_u64("general.alignment");
        if (alignment == 0 || (alignment & (alignment - 1))) throw std::runtime_error("gguf: alignment is not a power of two");

        tensors.resize((size_t) n_tensors);
        for (auto & t : tensors) {
            t.name = rd_str();
            t.n_dims = rd<uint32_t>();
            if (t.n_dims > 4) throw std::runtime_error("gguf: tensor " + t.name + " has more than 4 dimensions");
            for (uint32_t d = 0; d < t.n_dims; d++) {
                const uint64_t ne = rd<uint64_t>();
                if (ne > (uint64_t) INT64_MAX) throw std::runtime_error("gguf: tensor " + t.name + " has an implausible shape");
                t.ne[d] = (int64_t) ne;
            }
            t.type = rd<uint32_t>();
            t.offset = rd<uint64_t>();
            if (tensor_index.count(t.name)) throw std::runtime_error("gguf: duplicate tensor " + t.name);
            tensor_index[t.name] = (size_t)(&t - tensors.data());
        }
        data_offset = (pos + alignment - 1)/alignment*alignment;
        if (data_offset > size) throw std::runtime_error("gguf: no data section");
    }

    bool has(const std::string & k) const { return kv.count(k) != 0; }
    const gguf_value & at(const std::string & k) const {
        auto it = kv.find(k);
        if (it == kv.end()) throw std::runtime_error("gguf: key not found: " + k);
        return it->second;
    }
    // element i of a scalar or scalar array as the widest type of its class
    static double num(const gguf_value & v, uint64_t i, uint32_t t) {
        const uint8_t * p = v.data + i*gguf_scalar_size(t);
        switch (t) {
            case GV_U8:  return *p;                              case GV_I8:  return *(const int8_t *) p;
            case GV_BOOL: return *p != 0;
            case GV_U16: { uint16_t x; memcpy(&x, p, 2); return x; } case GV_I16: { int16_t x; memcpy(&x, p, 2); return x; }
            case GV_U32: { uint32_t x; memcpy(&x, p, 4); return x; } case GV_I32: { int32_t x; memcpy(&x, p, 4); return x; }
            case GV_F32: { float x; memcpy(&x, p, 4); return x; }
            case GV_U64: { uint64_t x; memcpy(&x, p, 8); return (double) x; } case GV_I64: { int64_t x; memcpy(&x, p, 8); return (double) x; }
            case GV_F64: { double x; memcpy(&x, p, 8); return x; }
            default: throw std::runtime_error("gguf: not a number");
        }
    }
    uint64_t get_u64(const std::string & k) const {
        const gguf_value & v = at(k);
        if (v.type == GV_STR || v.type == GV_ARR) throw std::runtime_error("gguf: key " + k + " is not a scalar");
        if (v.type == GV_U64) { uint64_t x; memcpy(&x, v.data, 8); return x; }
        const double d = num(v, 0, v.type);
        if (d < 0) throw std::runtime_error("gguf: key " + k + " is negative");
        return (uint64_t) d;
    }
    double get_f64(const std::string & k) const {
        const gguf_value & v = at(k);
        if (v.type == GV_STR || v.type == GV_ARR) throw std::runtime_error("gguf: key " + k + " is not a scalar");
        return num(v, 0, v.type);
    }
    // a scalar, or element i of an array (per-layer hyper-parameters may be either: llama_model_loader::get_key_or_arr)
    double get_f64_at(const std::string & k, uint64_t i) const {
        const gguf_value & v = at(k);
        if (v.type != GV_ARR) return get_f64(k);
        if (v.item_type == GV_STR || i >= v.count) throw std::runtime_error("gguf: key " + k + ": no numeric element " + std::to_string(i));
        return num(v, i, v.item_type);
    }
    const std::string & get_str(const std::string & k) const {
        const gguf_value & v = at(k);
        if (v.type != GV_STR) throw std::runtime_error("gguf: key " + k + " is not a string");
        return v.strs[0];
    }
    const gguf_tensor_info * find(const std::string & name) const {
        auto it = tensor_index.find(name);
        return it == tensor_index.end() ? nullptr : &tensors[it->second];
    }
    // the tensor's bytes in the mapping; nbytes is what ggml_nbytes gives for its type and shape (checked against the file size)
    const uint8_t * tensor_data(const gguf_tensor_info & t, size_t nbytes) const {
        if (t.offset % alignment) throw std::runtime_error("gguf: tensor " + t.name + " is not aligned");
        if (t.offset > size - data_offset || nbytes > size - data_offset - t.offset) throw std::runtime_error("gguf: tensor " + t.name + " lies outside the file");
        return base + data_offset + t.offset;
    }
};

// gguf_tools.cpp: one row of n elements of ggml type `type` -> f32 on the host; false = no decoder for the type
bool dequant_row_host(int type, const uint8_t * src, float * dst, int64_t n);

} // namespace mi355x
