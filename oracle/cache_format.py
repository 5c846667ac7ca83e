"""oracle/cache_format.py -- byte-level restatement of RASR's feature-cache files.  TEST INFRASTRUCTURE.

Only tests/ may import this (same rule as the rest of oracle/).  It is an independent second
implementation (struct + zlib, whole file in memory) of what rasr_amd/csrc/cache_io.cpp does with
FILE* streams, written from the reference sources:

  container   Core::FileArchive         src/Core/FileArchive.cc:27-85 (format), :165-225 (open), :243-297 (remove),
                                        :300-346 (table), :348-404 (scan), :406-458 (write table), :504-563 (write)
  compression Core::Archive::writeFile  src/Core/Archive.cc:142-222; readFile :52-139
  payload     Flow::CacheWriter         src/Flow/Cache.cc:81-120; Datatype gathered IO src/Flow/Datatype.cc:28-52;
                                        Flow::Vector<f32> src/Flow/Vector.hh:88-106; Timestamp src/Flow/Timestamp.cc:43-53
  strings     Core::BinaryOutputStream  src/Core/BinaryStream.cc:174-179 (u32 length + bytes, little endian)
  attributes  Flow::Attributes          src/Flow/Attributes.hh:67-70,132-138 on Core::XmlWriter (src/Core/XmlStream.cc)

Pinning: the payload block bytes and the attribute XML are checked against the reference's own
classes compiled unmodified into oracle/_ref/libref.so (ref_cache_block_write / ref_cache_block_read /
ref_attribs_xml; tests/test_cache.py, fixture tests/golden/ref_cache.json).  The CONTAINER layer is
PARITY UNPINNED: Core/FileArchive.cc needs Core/Configuration.hh (boost) and cannot be built here,
and the reference ships no archive fixture; it is restated from the source lines above only.
"""
import struct
import zlib

HEADER = b"SP_ARC1\x00"
START_TAG = 0xAA55AA55
END_TAG = 0x55AA55AA
VECTOR_F32 = b"vector-f32"


def _str(b):
    return struct.pack("<I", len(b)) + b


def block_bytes(feats, times):
    """one gathered block: name, u32 n, n x (u32 dim, f32 x dim, f64 start, f64 end)"""
    import numpy as np
    x = np.ascontiguousarray(feats, dtype="<f4")
    t = np.ascontiguousarray(times, dtype="<f8")
    out = [_str(VECTOR_F32), struct.pack("<I", x.shape[0])]
    for i in range(x.shape[0]):
        out.append(struct.pack("<I", x.shape[1]))
        out.append(x[i].tobytes())
        out.append(t[i].tobytes())
    return b"".join(out)


def entry_payload(feats, times, gather=0xFFFFFFFF):
    """CacheWriter: a block is flushed when it holds MORE than `gather` packets, and at the end"""
    per = gather + 1
    n = len(feats)
    return b"".join(block_bytes(feats[a:a + per], times[a:a + per]) for a in range(0, n, per))


def parse_payload(b):
    """-> (list of f32 vectors, list of (start, end)); raises on truncation / foreign datatypes"""
    import numpy as np
    at, vecs, times = 0, [], []
    while at < len(b):
        (ln,) = struct.unpack_from("<I", b, at)
        name = b[at + 4:at + 4 + ln]
        at += 4 + ln
        if name != VECTOR_F32:
            raise ValueError("datatype %r" % name)
        (n,) = struct.unpack_from("<I", b, at)
        at += 4
        for _ in range(n):
            (d,) = struct.unpack_from("<I", b, at)
            at += 4
            if at + 4 * d + 16 > len(b):
                raise ValueError("truncated")
            vecs.append(np.frombuffer(b, "<f4", d, at).copy())
            at += 4 * d
            times.append(struct.unpack_from("<dd", b, at))
            at += 16
    return vecs, times


def gzip_member(data, level=zlib.Z_DEFAULT_COMPRESSION):
    """Archive::writeFile: fixed 10-byte gzip header, raw deflate (zlib stream minus 2-byte header and adler32), crc32, size"""
    z = zlib.compress(data, level)
    assert z[:2] == b"\x78\x9c"
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + z[2:-4] + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def gunzip_member(z, size):
    """Archive::readFile: skip optional gzip header fields, inflate"""
    flags, base = z[3], 10
    if flags & 0x04:
        base += 2 + (z[base] | (z[base + 1] << 8))
    for bit in (0x08, 0x10):
        if flags & bit:
            base = z.index(b"\x00", base) + 1
    if flags & 0x02:
        base += 2
    out = zlib.decompressobj(-15).decompress(z[base:-8])
    assert len(out) == size
    return out


def entry_bytes(name, data, compress=False):
    """(bytes of one file entry, offset of its size field inside them, size, compressed)"""
    name = name.encode() if isinstance(name, str) else name
    stored = gzip_member(data) if compress else data
    comp = len(stored) if compress else 0
    head = struct.pack("<I", START_TAG) + _str(name)
    body = struct.pack("<III", len(data), comp, 0) + stored + struct.pack("<I", END_TAG)
    return head + body, len(head), len(data), comp


def empty_entry_bytes(length):
    """a removed file: empty name, `length` junk bytes"""
    head = struct.pack("<II", START_TAG, 0)
    return head + struct.pack("<III", length, 0, 0) + b"\xee" * length + struct.pack("<I", END_TAG), len(head), length


def archive_bytes(entries, with_table=True, empties=()):
    """entries: [(name, data, compress)]; empties: {index: junk_length} inserts a removed entry BEFORE entries[index]"""
    empties = dict(empties)
    body, infos, empty_infos = bytearray(HEADER + (b"\x01" if with_table else b"\x00")), [], []
    for i, (name, data, compress) in enumerate(entries):
        if i in empties:
            e, off, ln = empty_entry_bytes(empties[i])
            empty_infos.append((len(body) + off, ln))
            body += e
        e, off, size, comp = entry_bytes(name, data, compress)
        infos.append((name.encode() if isinstance(name, str) else name, len(body) + off, size, comp))
        body += e
    if with_table:
        table = len(body)
        body += struct.pack("<I", len(infos))
        for name, pos, size, comp in infos:
            body += _str(name) + struct.pack("<QII", pos, size, comp)
        empty_table = len(body)
        body += struct.pack("<I", len(empty_infos))
        for pos, ln in empty_infos:
            body += struct.pack("<QI", pos, ln)
        body += struct.pack("<QQ", empty_table, table)
    return bytes(body)


def parse_archive(b):
    """-> {name: bytes (uncompressed)} via the table when flagged, else via the recovery-tag scan; plus the raw infos"""
    assert b[:8] == HEADER
    infos = []
    if b[8]:
        (table,) = struct.unpack_from("<Q", b, len(b) - 8)
        at = table
        (n,) = struct.unpack_from("<I", b, at)
        at += 4
        for _ in range(n):
            (ln,) = struct.unpack_from("<I", b, at)
            name = b[at + 4:at + 4 + ln]
            at += 4 + ln
            pos, size, comp = struct.unpack_from("<QII", b, at)
            at += 16
            infos.append((name, pos, size, comp))
    else:
        at = 9
        while at + 4 <= len(b):
            (tag,) = struct.unpack_from("<I", b, at)
            at += 4
            if tag != START_TAG:
                continue
            (ln,) = struct.unpack_from("<I", b, at)
            name = b[at + 4:at + 4 + ln]
            at += 4 + ln
            pos = at
            size, comp, _ = struct.unpack_from("<III", b, at)
            at += 12 + (comp if (comp and name) else size) + 4
            if at > len(b):
                break
            if name:
                infos.append((name, pos, size, comp))
    files = {}
    for name, pos, size, comp in infos:
        raw = b[pos + 12:pos + 12 + (comp or size)]
        files[name.decode()] = gunzip_member(raw, size) if comp else raw
    return files, infos


def attribs_xml(attrs):
    def esc(s):
        return (s.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;").replace('"', "&quot;").replace("'", "&apos;"))
    return ("<flow-attributes>" + "".join('<flow-attribute name="%s" value="%s"/>' % (esc(k), esc(str(v))) for k, v in attrs.items()) +
            "</flow-attributes>")
