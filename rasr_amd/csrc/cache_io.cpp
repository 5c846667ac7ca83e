// cache_io.cpp -- RASR feature caches: the SP_ARC1 file archive and the Flow cache entries stored in it.
//
// Host-side IO only: this is how features reach / leave the device path when the front-end and
// the scorers run as separate jobs (feature-extraction writes a cache, the trainer / recognizer reads it).
//
// Container: Core::FileArchive (Core/FileArchive.cc:27-85 format comment; the code is authoritative where the comment
// differs -- file-info names are length-prefixed strings, and the on-disk order is uncompressed size, compressed size):
//   "SP_ARC1\0"  u8 has_table
//   entries:  u32 0xaa55aa55 | string name | u32 size | u32 compressed (0 = stored) | u32 checksum (always 0) | data | u32 0x55aa55aa
//   removed entries keep the tags with an empty name: u32 0 | u32 len | u32 0 | u32 0 | len bytes
//   table (has_table != 0):  u32 n { string name, u64 pos, u32 size, u32 compressed }  u32 n_empty { u64 pos, u32 size }
//                            u64 empty_table_pos  u64 table_pos           (pos = offset of the entry's size field)
//   strings are u32 length + bytes (Core/BinaryStream.cc:174-179); everything little endian.
// Compressed entries are a gzip member assembled by Core::Archive::writeFile (Core/Archive.cc:142-222): 10-byte header
// 1f 8b 08 00 00000000 00 03, raw deflate at Z_DEFAULT_COMPRESSION, crc32, isize.
//
// Payload: Flow::CacheWriter (Flow/Cache.cc:81-120) writes, per block, the datatype name ("vector-f32") followed by
// Datatype::writeGatheredData (Flow/Datatype.cc:43-52): u32 n, then n x Flow::Vector<f32>::write (Flow/Vector.hh:101-106):
// u32 size, f32 x size, f64 start, f64 end.  A block is flushed once it holds more than `gather` packets (so gather+1
// per block) and at the end of the segment.  "<segment>.attribs" holds the stream attributes as XML.
#include <zlib.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace {

const char     kHeader[8] = {'S', 'P', '_', 'A', 'R', 'C', '1', 0};
const uint32_t kStartTag  = 0xaa55aa55u;
const uint32_t kEndTag    = 0x55aa55aau;
const char     kVectorF32[] = "vector-f32";

struct FileInfo {
    std::string name;
    uint64_t    pos = 0;  // offset of the u32 size field
    uint32_t    size = 0, compressed = 0;
    bool        removed = false;
};

struct EmptyInfo {
    uint64_t pos;
    uint32_t size;
};

// little-endian scalar IO on a FILE*
template<class T>
bool rd(FILE* f, T* v) {
    unsigned char b[sizeof(T)];
    if (fread(b, 1, sizeof(T), f) != sizeof(T))
        return false;
    uint64_t x = 0;
    for (size_t i = 0; i < sizeof(T); ++i)
        x |= (uint64_t)b[i] << (8 * i);
    *v = (T)x;
    return true;
}
template<class T>
bool wr(FILE* f, T v) {
    unsigned char b[sizeof(T)];
    for (size_t i = 0; i < sizeof(T); ++i)
        b[i] = (unsigned char)((uint64_t)v >> (8 * i));
    return fwrite(b, 1, sizeof(T), f) == sizeof(T);
}
bool rd_str(FILE* f, std::string* s, uint64_t limit) {
    uint32_t n;
    s->clear();
    if (!rd(f, &n) || n > limit)
        return false;
    s->resize(n);
    return n == 0 || fread(&(*s)[0], 1, n, f) == n;
}
bool wr_str(FILE* f, const std::string& s) {
    return wr(f, (uint32_t)s.size()) && (s.empty() || fwrite(s.data(), 1, s.size(), f) == s.size());
}

// Core::normalizePath would change the name -> FileArchive::file() rejects it (Core/FileArchive.cc:146-162)
bool name_ok(const char* name) {
    if (!name || !*name)
        return false;
    return !strstr(name, "//") && !strstr(name, "/./") && strncmp(name, "./", 2) != 0;
}

// memory-buffer serialisation of the payload
struct Out {
    std::string b;
    template<class T>
    void put(T v) {
        for (size_t i = 0; i < sizeof(T); ++i)
            b.push_back((char)((uint64_t)v >> (8 * i)));
    }
    void put_f64(double d) {
        uint64_t u;
        memcpy(&u, &d, 8);
        put(u);
    }
    void put_str(const char* s) {
        put((uint32_t)strlen(s));
        b.append(s);
    }
    void put_raw(const void* p, size_t n) {
        b.append((const char*)p, n);
    }
};

struct In {
    const unsigned char* p;
    size_t               n, at = 0;
    bool                 ok = true;
    template<class T>
    T get() {
        if (at + sizeof(T) > n) {
            ok = false;
            at = n;
            return T(0);
        }
        uint64_t x = 0;
        for (size_t i = 0; i < sizeof(T); ++i)
            x |= (uint64_t)p[at + i] << (8 * i);
        at += sizeof(T);
        return (T)x;
    }
    double get_f64() {
        uint64_t u = get<uint64_t>();
        double   d;
        memcpy(&d, &u, 8);
        return d;
    }
    std::string get_str() {
        uint32_t len = get<uint32_t>();
        if (!ok || at + len > n) {
            ok = false;
            return std::string();
        }
        std::string s((const char*)p + at, len);
        at += len;
        return s;
    }
};

}  // namespace

struct amx_archive {
    std::string                     path;
    FILE*                           f        = nullptr;
    bool                            writable = false, changed = false;
    uint64_t                        end_of_archive = 9;
    std::vector<FileInfo>           files;
    std::map<std::string, uint32_t> index;
    std::vector<EmptyInfo>          empties;

    bool add(const FileInfo& fi) {
        if (index.count(fi.name))
            return false;
        index[fi.name] = (uint32_t)files.size();
        files.push_back(fi);
        return true;
    }
    const FileInfo* find(const std::string& name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : &files[it->second];
    }

    // Core/FileArchive.cc:300-346
    bool read_table(uint64_t file_size) {
        uint64_t pos;
        if (file_size < 9 + 8 || fseeko(f, -8, SEEK_END) != 0 || !rd(f, &pos) || pos > file_size)
            return false;
        end_of_archive = pos;
        if (fseeko(f, (off_t)pos, SEEK_SET) != 0)
            return false;
        uint32_t count;
        if (!rd(f, &count))
            return false;
        for (uint32_t i = 0; i < count; ++i) {
            FileInfo fi;
            if (!rd_str(f, &fi.name, file_size) || !rd(f, &fi.pos) || !rd(f, &fi.size) || !rd(f, &fi.compressed))
                return false;
            // an entry is [u32 size][u32 compressed][u32 checksum][stored bytes][u32 tag] at pos: it must lie inside the file
            const uint64_t stored = fi.compressed ? fi.compressed : fi.size;
            if (fi.pos > file_size || stored > file_size || fi.pos + 12 + stored + 4 > file_size)
                return false;
            add(fi);
        }
        if (!rd(f, &count))
            return false;
        if ((uint64_t)count * 12 > file_size)
            return false;
        for (uint32_t i = 0; i < count; ++i) {
            EmptyInfo e;
            if (!rd(f, &e.pos) || !rd(f, &e.size) || e.pos > file_size)
                return false;
            empties.push_back(e);
        }
        return true;
    }

    // Core/FileArchive.cc:348-404: walk the recovery tags
    bool scan(uint64_t file_size) {
        files.clear();
        index.clear();
        empties.clear();
        if (fseeko(f, 9, SEEK_SET) != 0)
            return false;
        end_of_archive = 9;
        for (;;) {
            uint32_t tag = 0;
            if (!rd(f, &tag))
                break;
            if (tag != kStartTag)
                continue;
            FileInfo fi;
            uint32_t checksum;
            if (!rd_str(f, &fi.name, file_size))
                break;
            fi.pos = (uint64_t)ftello(f);
            if (!rd(f, &fi.size) || !rd(f, &fi.compressed) || !rd(f, &checksum))
                break;
            uint32_t skip = fi.name.empty() ? fi.size : (fi.compressed ? fi.compressed : fi.size);
            if ((uint64_t)ftello(f) + skip + 4 > file_size || fseeko(f, skip, SEEK_CUR) != 0 || !rd(f, &tag))
                break;
            if (fi.name.empty())
                empties.push_back({fi.pos, fi.size});
            else
                add(fi);
            if (tag == kEndTag)
                end_of_archive = (uint64_t)ftello(f);
        }
        return true;
    }

    void set_changed() {
        if (!changed) {
            fseeko(f, 8, SEEK_SET);
            wr(f, (uint8_t)0);
            changed = true;
        }
    }

    // Core/FileArchive.cc:406-458
    bool write_table() {
        if (!changed || !writable)
            return true;
        if (fseeko(f, (off_t)end_of_archive, SEEK_SET) != 0)
            return false;
        uint64_t table = end_of_archive;
        uint32_t live  = 0;
        for (const FileInfo& fi : files)
            live += !fi.removed;
        bool ok = wr(f, live);
        for (const FileInfo& fi : files)
            if (!fi.removed)
                ok = ok && wr_str(f, fi.name) && wr(f, fi.pos) && wr(f, fi.size) && wr(f, fi.compressed);
        uint64_t empty_table = (uint64_t)ftello(f);
        ok                   = ok && wr(f, (uint32_t)empties.size());
        for (const EmptyInfo& e : empties)
            ok = ok && wr(f, e.pos) && wr(f, e.size);
        ok             = ok && wr(f, empty_table) && wr(f, table);
        uint64_t total = (uint64_t)ftello(f);
        ok             = ok && fseeko(f, 8, SEEK_SET) == 0 && wr(f, (uint8_t)1) && fflush(f) == 0;
        if (ok && ftruncate(fileno(f), (off_t)total) != 0)
            ok = false;
        changed = !ok;
        return ok;
    }

    // Core/FileArchive.cc:243-297
    bool remove(const std::string& name) {
        auto it = index.find(name);
        if (it == index.end())
            return false;
        FileInfo&      fi    = files[it->second];
        const uint64_t begin = fi.pos - (4 + fi.name.size() + 4);
        uint32_t       size  = fi.compressed ? fi.compressed : fi.size;
        set_changed();
        if (fi.pos + 12 + size + 4 == end_of_archive) {
            end_of_archive = begin;
        }
        else {
            fseeko(f, (off_t)(begin + 4), SEEK_SET);
            wr(f, (uint32_t)0);
            uint64_t pos = (uint64_t)ftello(f);
            size += (uint32_t)fi.name.size();
            wr(f, size);
            wr(f, (uint32_t)0);
            wr(f, (uint32_t)0);
            empties.push_back({pos, size});
        }
        fi.removed = true;
        fi.name.clear();
        fi.pos = 0;
        index.erase(it);
        return true;
    }

    // Core/FileArchive.cc:504-563
    bool write_raw(const std::string& name, const void* data, size_t stored, uint32_t size, uint32_t compressed) {
        if (find(name))
            remove(name);
        set_changed();
        const uint32_t needed = (uint32_t)(stored + name.size());
        bool           append = true;
        uint64_t       at     = end_of_archive;
        for (size_t i = 0; i < empties.size(); ++i)
            if (empties[i].size == needed) {
                at     = empties[i].pos - 8;
                append = false;
                empties.erase(empties.begin() + i);
                break;
            }
        if (fseeko(f, (off_t)at, SEEK_SET) != 0)
            return false;
        bool     ok = wr(f, kStartTag) && wr_str(f, name);
        FileInfo fi;
        fi.name       = name;
        fi.pos        = (uint64_t)ftello(f);
        fi.size       = size;
        fi.compressed = compressed;
        ok            = ok && wr(f, size) && wr(f, compressed) && wr(f, (uint32_t)0) &&
             (stored == 0 || fwrite(data, 1, stored, f) == stored) && wr(f, kEndTag);
        if (!ok)
            return false;
        if (append)
            end_of_archive = (uint64_t)ftello(f);
        add(fi);
        return true;
    }
};

namespace {

// Core/Archive.cc:162-215
bool gzip_pack(const void* data, size_t len, std::string* out) {
    uLongf      cap = compressBound((uLong)len) + 16;
    std::string z(cap, '\0');
    if (compress2((Bytef*)&z[0], &cap, (const Bytef*)data, (uLong)len, Z_DEFAULT_COMPRESSION) != Z_OK || cap < 6)
        return false;
    static const unsigned char head[10] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0x03};
    out->assign((const char*)head, 10);
    out->append(z.data() + 2, cap - 6);  // drop the zlib header and the adler32
    uint32_t crc = (uint32_t)crc32(0L, (const Bytef*)data, (uInt)len), n = (uint32_t)len;
    for (int i = 0; i < 4; ++i)
        out->push_back((char)(crc >> (8 * i)));
    for (int i = 0; i < 4; ++i)
        out->push_back((char)(n >> (8 * i)));
    return true;
}

// Core/Archive.cc:82-131: skip the gzip header fields, inflate the raw deflate stream
bool gzip_unpack(const std::string& z, size_t size, unsigned char* out) {
    if (z.size() < 18 || (unsigned char)z[0] != 0x1f || (unsigned char)z[1] != 0x8b)
        return false;
    const unsigned char flags = (unsigned char)z[3];
    size_t              base  = 10;
    if (flags & 0x04) {
        if (base + 2 > z.size())
            return false;
        base += 2 + ((unsigned char)z[base] | ((unsigned char)z[base + 1] << 8));
    }
    for (int bit : {0x08, 0x10})
        if (flags & bit) {
            while (base < z.size() && z[base])
                ++base;
            ++base;
        }
    if (flags & 0x02)
        base += 2;
    if (base + 8 > z.size())
        return false;
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (inflateInit2(&s, -15) != Z_OK)
        return false;
    s.next_in   = (Bytef*)z.data() + base;
    s.avail_in  = (uInt)(z.size() - base - 8);
    s.next_out  = out;
    s.avail_out = (uInt)size;
    int rc      = inflate(&s, Z_FINISH);
    size_t got  = s.total_out;
    inflateEnd(&s);
    return rc == Z_STREAM_END && got == size;
}

int read_file(amx_archive* a, const char* name, std::string* out) {
    const FileInfo* fi = a->find(name);
    if (!fi) {
        amx::set_error("archive '%s' has no file '%s'", a->path.c_str(), name);
        return AMX_ERR_INVALID;
    }
    const size_t stored = fi->compressed ? fi->compressed : fi->size;
    if (fi->pos + 12 + stored + 4 > a->end_of_archive ||
        (fi->compressed && (uint64_t)fi->size > 1032ull * stored + 65536)) {  // entries end before the table; deflate expands at most ~1032 : 1
        amx::set_error("archive '%s': file '%s' is truncated or corrupt", a->path.c_str(), name);
        return AMX_ERR_INVALID;
    }
    std::string  raw(stored, '\0');
    bool         ok = fseeko(a->f, (off_t)(fi->pos + 12), SEEK_SET) == 0 && (stored == 0 || fread(&raw[0], 1, stored, a->f) == stored);
    if (ok && fi->compressed) {
        out->assign(fi->size, '\0');
        ok = gzip_unpack(raw, fi->size, (unsigned char*)&(*out)[0]);
    }
    else
        out->swap(raw);
    if (!ok) {
        amx::set_error("archive '%s': file '%s' is truncated or corrupt", a->path.c_str(), name);
        return AMX_ERR_INVALID;
    }
    return AMX_OK;
}

int write_file(amx_archive* a, const char* name, const void* data, size_t len, int compress) {
    AMX_REQUIRE(a->writable, AMX_ERR_STATE, "archive '%s' is open read-only", a->path.c_str());
    AMX_REQUIRE(name_ok(name), AMX_ERR_INVALID, "archive file name '%s' is empty or not normalised", name ? name : "");
    AMX_REQUIRE(len < (1ull << 32), AMX_ERR_UNSUPPORTED, "archive entries are limited to 4 GiB (u32 sizes)");
    std::string z;
    bool        ok;
    if (compress && gzip_pack(data, len, &z))
        ok = a->write_raw(name, z.data(), z.size(), (uint32_t)len, (uint32_t)z.size());
    else
        ok = a->write_raw(name, data, len, (uint32_t)len, 0);
    AMX_REQUIRE(ok, AMX_ERR_INVALID, "write to archive '%s' failed", a->path.c_str());
    return AMX_OK;
}

void* dup_bytes(const void* p, size_t n) {
    void* d = malloc(std::max<size_t>(n, 1));
    if (d && n)
        memcpy(d, p, n);
    return d;
}

void xml_escape(std::string* o, const char* s) {
    for (; *s; ++s) switch (*s) {
            case '&': o->append("&amp;"); break;
            case '<': o->append("&lt;"); break;
            case '>': o->append("&gt;"); break;
            case '"': o->append("&quot;"); break;
            case '\'': o->append("&apos;"); break;
            default: o->push_back(*s);
        }
}

}  // namespace

extern "C" {

int amx_archive_open(const char* path, int mode, amx_archive** out) {
    try {
        AMX_REQUIRE(path && out && (mode == AMX_ARCHIVE_READ || mode == AMX_ARCHIVE_WRITE), AMX_ERR_INVALID, "amx_archive_open: bad argument");
        *out = nullptr;
        struct stat st;
        const bool  exists = stat(path, &st) == 0 && st.st_size > 0;
        AMX_REQUIRE(exists || mode == AMX_ARCHIVE_WRITE, AMX_ERR_INVALID, "archive file '%s' does not exist", path);
        AMX_REQUIRE(!exists || S_ISREG(st.st_mode), AMX_ERR_UNSUPPORTED, "'%s' is not a file archive (directory and bundle archives are not supported)", path);
        FILE* f = fopen(path, exists ? (mode == AMX_ARCHIVE_WRITE ? "r+b" : "rb") : "w+b");
        AMX_REQUIRE(f, AMX_ERR_INVALID, "cannot open archive '%s'", path);
        amx_archive* a = new amx_archive;
        a->path        = path;
        a->f           = f;
        a->writable    = mode == AMX_ARCHIVE_WRITE;
        if (!exists) {
            bool ok           = fwrite(kHeader, 1, 8, f) == 8 && wr(f, (uint8_t)0);
            a->changed        = true;
            a->end_of_archive = 9;
            if (!ok) {
                fclose(f);
                delete a;
                amx::set_error("failed to create archive file '%s'", path);
                return AMX_ERR_INVALID;
            }
        }
        else {
            char    head[8];
            uint8_t has_table = 0;
            bool    ok        = fread(head, 1, 8, f) == 8 && memcmp(head, kHeader, 8) == 0 && rd(f, &has_table);
            if (ok)
                ok = has_table ? a->read_table((uint64_t)st.st_size) : a->scan((uint64_t)st.st_size);
            if (!ok) {
                fclose(f);
                delete a;
                amx::set_error("'%s' is not a readable SP_ARC1 file archive", path);
                return AMX_ERR_INVALID;
            }
        }
        *out = a;
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_archive_open", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_archive_close(amx_archive* a) {
    try {
        if (!a)
            return AMX_OK;
        bool ok = a->write_table();
        ok      = (fclose(a->f) == 0) && ok;
        std::string path = a->path;
        delete a;
        AMX_REQUIRE(ok, AMX_ERR_INVALID, "failed to finish archive '%s'", path.c_str());
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_archive_close", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_archive_n_files(amx_archive* a) {
    if (!a)
        return 0;
    int n = 0;
    for (const FileInfo& fi : a->files)
        n += !fi.removed;
    return n;
}

int amx_archive_file_info(amx_archive* a, int i, const char** name, uint32_t* size, uint32_t* compressed) {
    AMX_REQUIRE(a && i >= 0, AMX_ERR_INVALID, "amx_archive_file_info: bad argument");
    for (const FileInfo& fi : a->files) {
        if (fi.removed || i-- > 0)
            continue;
        if (name)
            *name = fi.name.c_str();
        if (size)
            *size = fi.size;
        if (compressed)
            *compressed = fi.compressed;
        return AMX_OK;
    }
    amx::set_error("amx_archive_file_info: index out of range");
    return AMX_ERR_INVALID;
}

int amx_archive_has_file(amx_archive* a, const char* name) {
    return a && name && a->find(name) ? 1 : 0;
}

int amx_archive_read_file(amx_archive* a, const char* name, void** data, size_t* len) {
    try {
        AMX_REQUIRE(a && name && data && len, AMX_ERR_INVALID, "amx_archive_read_file: NULL argument");
        *data = nullptr;
        *len  = 0;
        std::string b;
        int         rc = read_file(a, name, &b);
        if (rc != AMX_OK)
            return rc;
        *data = dup_bytes(b.data(), b.size());
        *len  = b.size();
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_archive_read_file", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_archive_write_file(amx_archive* a, const char* name, const void* data, size_t len, int compress) {
    try {
        AMX_REQUIRE(a && (data || len == 0), AMX_ERR_INVALID, "amx_archive_write_file: NULL argument");
        return write_file(a, name, data, len, compress);
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_archive_write_file", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_archive_remove_file(amx_archive* a, const char* name) {
    try {
        AMX_REQUIRE(a && name, AMX_ERR_INVALID, "amx_archive_remove_file: NULL argument");
        AMX_REQUIRE(a->writable, AMX_ERR_STATE, "archive '%s' is open read-only", a->path.c_str());
        AMX_REQUIRE(a->remove(name), AMX_ERR_INVALID, "archive '%s' has no file '%s'", a->path.c_str(), name);
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_archive_remove_file", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_feature_cache_write(amx_archive* a, const char* segment, int n, int dim, const float* feats, const double* times,
                            unsigned gather, int compress) {
    try {
        AMX_REQUIRE(a && segment && n >= 0 && dim >= 0 && (feats || (size_t)n * dim == 0) && (times || n == 0), AMX_ERR_INVALID,
                    "amx_feature_cache_write: bad argument");
        Out o;
        o.b.reserve((size_t)n * (dim * 4 + 20) + 64);
        // CacheWriter::putData flushes a block when it holds MORE than `gather` packets (Flow/Cache.cc:114-119)
        const uint64_t per_block = (uint64_t)gather + 1;
        for (int at = 0; at < n;) {
            const int cnt = (int)std::min<uint64_t>(per_block, (uint64_t)(n - at));
            o.put_str(kVectorF32);
            o.put((uint32_t)cnt);
            for (int i = at; i < at + cnt; ++i) {
                o.put((uint32_t)dim);
                o.put_raw(feats + (size_t)i * dim, (size_t)dim * 4);
                o.put_f64(times[2 * i]);
                o.put_f64(times[2 * i + 1]);
            }
            at += cnt;
        }
        return write_file(a, segment, o.b.data(), o.b.size(), compress);
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_feature_cache_write", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_feature_cache_read(amx_archive* a, const char* segment, int* n_out, int* dim_out, float** feats, double** times) {
    try {
        AMX_REQUIRE(a && segment && n_out && dim_out && feats, AMX_ERR_INVALID, "amx_feature_cache_read: NULL argument");
        *feats = nullptr;
        if (times)
            *times = nullptr;
        *n_out = *dim_out = 0;
        std::string b;
        int         rc = read_file(a, segment, &b);
        if (rc != AMX_OK)
            return rc;
        In                  in{(const unsigned char*)b.data(), b.size()};
        std::vector<float>  x;
        std::vector<double> t;
        long                n = 0;
        int                 dim = -1;
        while (in.at < in.n) {  // CacheReader::getData keeps calling readData until the entry is exhausted
            std::string type = in.get_str();
            AMX_REQUIRE(in.ok, AMX_ERR_INVALID, "feature cache entry '%s' is truncated", segment);
            AMX_REQUIRE(type == kVectorF32, AMX_ERR_UNSUPPORTED, "feature cache entry '%s' holds '%s' packets; only vector-f32 is supported",
                        segment, type.c_str());
            uint32_t cnt = in.get<uint32_t>();
            for (uint32_t i = 0; in.ok && i < cnt; ++i) {
                uint32_t d = in.get<uint32_t>();
                if (dim < 0)
                    dim = (int)d;
                AMX_REQUIRE((int)d == dim, AMX_ERR_UNSUPPORTED, "feature cache entry '%s' mixes vector sizes %d and %u", segment, dim, d);
                if (!in.ok || in.at + (size_t)d * 4 + 16 > in.n) {
                    in.ok = false;
                    break;
                }
                x.insert(x.end(), (const float*)(in.p + in.at), (const float*)(in.p + in.at) + d);  // LE host
                in.at += (size_t)d * 4;
                t.push_back(in.get_f64());
                t.push_back(in.get_f64());
                ++n;
            }
            AMX_REQUIRE(in.ok, AMX_ERR_INVALID, "feature cache entry '%s' is truncated", segment);
        }
        *n_out   = (int)n;
        *dim_out = dim < 0 ? 0 : dim;
        *feats   = (float*)dup_bytes(x.data(), x.size() * sizeof(float));
        if (times)
            *times = (double*)dup_bytes(t.data(), t.size() * sizeof(double));
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_feature_cache_read", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_feature_cache_write_attributes(amx_archive* a, const char* segment, int n, const char* const* names,
                                       const char* const* values, int compress) {
    try {
        AMX_REQUIRE(a && segment && n >= 0 && (n == 0 || (names && values)), AMX_ERR_INVALID, "amx_feature_cache_write_attributes: bad argument");
        // Core::XmlWriter output of Flow::Attributes (Flow/Attributes.hh:67-70,132-138) on an unformatted stream
        // (CacheWriter::~CacheWriter, Flow/Cache.cc:78-85): no declaration, no line breaks, the five XML escapes
        std::string x = "<flow-attributes>";
        for (int i = 0; i < n; ++i) {
            x += "<flow-attribute name=\"";
            xml_escape(&x, names[i]);
            x += "\" value=\"";
            xml_escape(&x, values[i]);
            x += "\"/>";
        }
        x += "</flow-attributes>";
        std::string name = std::string(segment) + ".attribs";
        return write_file(a, name.c_str(), x.data(), x.size(), compress);
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_feature_cache_write_attributes", e.what());
        return AMX_ERR_INVALID;
    }
}

int amx_feature_cache_read_attributes(amx_archive* a, const char* segment, char** xml) {
    try {
        AMX_REQUIRE(a && segment && xml, AMX_ERR_INVALID, "amx_feature_cache_read_attributes: NULL argument");
        *xml             = nullptr;
        std::string name = std::string(segment) + ".attribs", b;
        int         rc   = read_file(a, name.c_str(), &b);
        if (rc != AMX_OK)
            return rc;
        *xml = (char*)dup_bytes(b.c_str(), b.size() + 1);
        return AMX_OK;
    }
    catch (const std::exception& e) {  // std::bad_alloc / length_error on a corrupt size field must not cross the C boundary
        amx::set_error("%s: %s", "amx_feature_cache_read_attributes", e.what());
        return AMX_ERR_INVALID;
    }
}

}  // extern "C"
