"""Feature-cache files: rasr_amd/csrc/cache_io.cpp against the byte-level restatement in
oracle/cache_format.py, the reference-generated fixture tests/golden/ref_cache.json and, where the reference tree is
mounted, the reference's own Flow::Vector / Datatype reader (oracle/_ref/libref.so).  Host-side IO: runs without a GPU."""
import gzip
import io
import json
import os
import struct

import numpy as np
import pytest

import rasr_amd
from oracle import cache_format as cf
from oracle.binding import load_ref

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_cache.json")))


def _case(c):
    x = np.frombuffer(bytes.fromhex(c["feats_hex"]), "<f4").reshape(c["n"], c["dim"])
    t = np.frombuffer(bytes.fromhex(c["times_hex"]), "<f8").reshape(c["n"], 2)
    return x, t, bytes.fromhex(c["block_hex"])


def _feats(rng, n, dim):
    x = (rng.standard_normal((n, dim)) * 5).astype(np.float32)
    s = np.arange(n) * 0.01
    return x, np.stack([s, s + 0.025], 1)


# ---------------------------------------------------------------- oracle vs the reference's bytes
@pytest.mark.parametrize("i", range(len(GOLD["blocks"])))
def test_oracle_block_matches_reference_bytes(i):
    x, t, want = _case(GOLD["blocks"][i])
    assert cf.block_bytes(x, t) == want


@pytest.mark.parametrize("i", range(len(GOLD["attributes"])))
def test_oracle_attribs_match_reference_xml(i):
    a = GOLD["attributes"][i]
    assert cf.attribs_xml(a["attrs"]) == a["xml"]


def test_oracle_live_against_libref():
    if load_ref() is None:
        pytest.skip("reference tree not mounted")
    from oracle.binding import ref_attribs_xml, ref_cache_block, ref_cache_block_parse
    rng = np.random.default_rng(5)
    for n, dim in ((7, 33), (64, 40), (2, 1)):
        x, t = _feats(rng, n, dim)
        b = ref_cache_block(x, t)
        assert cf.block_bytes(x, t) == b
        rx, rt, used, name = ref_cache_block_parse(b + b"tail", dim, n)
        assert name == "vector-f32" and used == len(b)
        assert rx.tobytes() == x.tobytes() and rt.tobytes() == t.tobytes()
    attrs = {"a&b": "<1>", "q": "'\""}
    assert cf.attribs_xml(attrs) == ref_attribs_xml(attrs)


# ---------------------------------------------------------------- the library writes what the reference writes
@pytest.mark.parametrize("i", range(len(GOLD["blocks"])))
def test_written_entry_is_reference_block(tmp_path, i):
    x, t, want = _case(GOLD["blocks"][i])
    p = str(tmp_path / "f.cache")
    with rasr_amd.FileArchive(p, "w") as a:
        a.write_features("corpus/rec/seg", x, t)
        got = a.read_file("corpus/rec/seg")
    # an empty segment writes no block at all (CacheWriter::~CacheWriter: `if (data_.size())`)
    assert got == (want if len(x) else b"")
    raw = open(p, "rb").read()
    files, _ = cf.parse_archive(raw)
    assert files == {"corpus/rec/seg": got}


@pytest.mark.parametrize("compress", [False, True])
def test_archive_bytes_match_oracle(tmp_path, compress):
    rng = np.random.default_rng(11)
    segs = {"c/r/%d" % i: _feats(rng, n, 16) for i, n in enumerate((5, 1, 40))}
    attrs = {"sample-rate": "100", "datatype": "vector-f32"}
    p = str(tmp_path / "a.cache")
    with rasr_amd.FileArchive(p, "w") as a:
        for k, (x, t) in segs.items():
            a.write_features(k, x, t, compress=compress, attributes=attrs)
    entries = []
    for k, (x, t) in segs.items():
        entries.append((k + ".attribs", cf.attribs_xml(attrs).encode(), compress))
        entries.append((k, cf.entry_payload(x, t), compress))
    want = cf.archive_bytes(entries, with_table=True)
    got = open(p, "rb").read()
    if compress:  # same structure and contents; deflate bytes may differ between zlib builds
        fw, iw = cf.parse_archive(want)
        fg, ig = cf.parse_archive(got)
        assert fw == fg and [i[0] for i in iw] == [i[0] for i in ig] and [i[2] for i in iw] == [i[2] for i in ig]
        assert all(i[3] > 0 for i in ig)
        for name, pos, size, comp in ig:  # every stored member is a plain gzip file (Archive.cc:191-213)
            assert gzip.decompress(got[pos + 12:pos + 12 + comp]) == fg[name.decode()]
    else:
        assert got == want


def test_gather_blocks(tmp_path):
    rng = np.random.default_rng(3)
    x, t = _feats(rng, 23, 8)
    p = str(tmp_path / "g.cache")
    with rasr_amd.FileArchive(p, "w") as a:
        a.write_features("s", x, t, gather=4)  # blocks of gather+1 = 5 packets (Flow/Cache.cc:114)
        raw = a.read_file("s")
        rx, rt = a.read_features("s")
    assert raw == cf.entry_payload(x, t, gather=4)
    assert raw.count(b"vector-f32") == 5
    assert rx.tobytes() == x.tobytes() and rt.tobytes() == t.tobytes()
    if load_ref() is not None:  # the reference reader walks the same blocks
        from oracle.binding import ref_cache_block_parse
        at, got = 0, []
        while at < len(raw):
            bx, _, used, _ = ref_cache_block_parse(raw[at:], 8, 23)
            got.append(bx)
            at += used
        assert [len(g) for g in got] == [5, 5, 5, 5, 3]
        assert np.concatenate(got).tobytes() == x.tobytes()


# ---------------------------------------------------------------- the library reads what the reference layout holds
@pytest.mark.parametrize("with_table", [True, False])
@pytest.mark.parametrize("compress", [False, True])
def test_read_oracle_built_archive(tmp_path, with_table, compress):
    rng = np.random.default_rng(21)
    segs = {"corpus/a/%d" % i: _feats(rng, n, 40) for i, n in enumerate((3, 17, 1, 9))}
    entries = [(k, cf.entry_payload(x, t), compress) for k, (x, t) in segs.items()]
    entries.insert(2, ("corpus/a/1.attribs", cf.attribs_xml({"sample-rate": "100"}).encode(), compress))
    raw = cf.archive_bytes(entries, with_table=with_table, empties={1: 13, 3: 0})
    p = str(tmp_path / "o.cache")
    open(p, "wb").write(raw)
    with rasr_amd.FileArchive(p) as a:
        assert [f[0] for f in a.files()] == [e[0] for e in entries]
        for k, (x, t) in segs.items():
            assert k in a
            rx, rt = a.read_features(k)
            assert rx.dtype == np.float32 and rx.tobytes() == x.tobytes() and rt.tobytes() == t.tobytes()
        assert a.read_attributes("corpus/a/1") == {"sample-rate": "100"}
        assert "corpus/a/none" not in a
        with pytest.raises(rasr_amd.AmxError):
            a.read_features("corpus/a/none")
        with pytest.raises(rasr_amd.AmxError):  # opened read-only
            a.write_file("x", b"1")
    assert open(p, "rb").read() == raw  # reading never modifies the file


def test_gzip_header_fields_are_skipped(tmp_path):
    """Archive::readFile skips FEXTRA / FNAME / FCOMMENT / FHCRC (Archive.cc:91-106) -- members written by gzip(1)"""
    data = cf.entry_payload(*_feats(np.random.default_rng(2), 6, 12))
    buf = io.BytesIO()
    with gzip.GzipFile(filename="seg.bin", mode="wb", fileobj=buf, mtime=0) as g:
        g.write(data)
    member = buf.getvalue()
    assert member[3] & 0x08
    e = (struct.pack("<I", cf.START_TAG) + struct.pack("<I", 1) + b"s" + struct.pack("<III", len(data), len(member), 0) + member +
         struct.pack("<I", cf.END_TAG))
    p = str(tmp_path / "z.cache")
    open(p, "wb").write(cf.HEADER + b"\x00" + e)
    with rasr_amd.FileArchive(p) as a:
        assert a.read_file("s") == data


# ---------------------------------------------------------------- archive maintenance
def test_remove_overwrite_append(tmp_path):
    p = str(tmp_path / "m.cache")
    with rasr_amd.FileArchive(p, "w") as a:
        a.write_file("a", b"A" * 10)
        a.write_file("b", b"B" * 20)
        a.write_file("c", b"C" * 5)
    with rasr_amd.FileArchive(p, "w") as a:
        assert [f[0] for f in a.files()] == ["a", "b", "c"]
        a.remove_file("b")          # middle entry -> empty slot of 20 + 1 bytes
        assert "b" not in a and [f[0] for f in a.files()] == ["a", "c"]
        a.write_file("dd", b"D" * 19)   # name + data == 21 -> reuses the slot (FileArchive.cc:529-537)
        a.write_file("c", b"c" * 7)     # overwrite of the last entry shrinks, then appends
        a.write_file("e", b"")
    files, infos = cf.parse_archive(open(p, "rb").read())
    assert files == {"a": b"A" * 10, "dd": b"D" * 19, "c": b"c" * 7, "e": b""}
    assert [i[0] for i in infos] == [b"a", b"c", b"dd", b"e"] or sorted(i[0] for i in infos) == [b"a", b"c", b"dd", b"e"]
    # the same content is found without the table (flag cleared -> recovery scan)
    raw = bytearray(open(p, "rb").read())
    (table,) = struct.unpack_from("<Q", raw, len(raw) - 8)
    raw[8] = 0
    scan_files, _ = cf.parse_archive(bytes(raw[:table]))
    assert scan_files == files
    q = str(tmp_path / "m2.cache")
    open(q, "wb").write(bytes(raw[:table]))
    with rasr_amd.FileArchive(q) as a:
        assert sorted(f[0] for f in a.files()) == sorted(files)
        assert a.read_file("dd") == b"D" * 19
    with rasr_amd.FileArchive(p) as a:
        assert a.read_file("c") == b"c" * 7 and a.read_file("e") == b""


def test_unclosed_archive_is_recovered(tmp_path):
    """a writer that died before ~FileArchive leaves flag 0 and no table: the reader scans the recovery tags"""
    p = str(tmp_path / "u.cache")
    x, t = _feats(np.random.default_rng(4), 8, 16)
    a = rasr_amd.FileArchive(p, "w")
    a.write_features("s1", x, t)
    a.write_features("s2", x[:3], t[:3])
    import ctypes
    ctypes.CDLL(None).fflush(None)
    raw = open(p, "rb").read()
    assert raw[8] == 0
    q = str(tmp_path / "u2.cache")
    open(q, "wb").write(raw + b"\x55\xaa")  # plus a torn tail
    a.close()
    with rasr_amd.FileArchive(q) as b:
        assert [f[0] for f in b.files()] == ["s1", "s2"]
        assert b.read_features("s2")[0].tobytes() == x[:3].tobytes()


def test_errors(tmp_path):
    p = str(tmp_path / "e.cache")
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.FileArchive(p)  # missing, read mode
    open(p, "wb").write(b"not an archive at all")
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.FileArchive(p)
    os.remove(p)
    with rasr_amd.FileArchive(p, "w") as a:
        with pytest.raises(rasr_amd.AmxError):
            a.write_file("a//b", b"x")  # FileArchive::file requires a normalised name
        with pytest.raises(rasr_amd.AmxError):
            a.write_file("", b"x")
        a.write_file("ints", cf._str(b"vector-u32") + struct.pack("<I", 0))
        a.write_file("ragged", cf.block_bytes(np.zeros((1, 3), np.float32), np.zeros((1, 2))) +
                     cf.block_bytes(np.zeros((1, 4), np.float32), np.zeros((1, 2))))
        a.write_file("torn", cf.block_bytes(np.zeros((2, 3), np.float32), np.zeros((2, 2)))[:-5])
        for name, status in (("ints", rasr_amd._lib.AMX_ERR_UNSUPPORTED), ("ragged", rasr_amd._lib.AMX_ERR_UNSUPPORTED),
                             ("torn", rasr_amd._lib.AMX_ERR_INVALID)):
            with pytest.raises(rasr_amd.AmxError) as e:
                a.read_features(name)
            assert e.value.status == status
    # a truncated table falls over cleanly
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-3])
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.FileArchive(p)


def test_large_segment_round_trip(tmp_path):
    rng = np.random.default_rng(9)
    x, t = _feats(rng, 30000, 45)  # a 5-minute recording at 10 ms
    p = str(tmp_path / "big.cache")
    with rasr_amd.FileArchive(p, "w") as a:
        a.write_features("big", x, t, compress=True)
        a.write_features("raw", x, t)
    with rasr_amd.FileArchive(p) as a:
        info = {f[0]: f for f in a.files()}
        assert info["big"][1] == info["raw"][1] == len(cf.entry_payload(x[:1], t[:1])) - 14 - 4 + 14 + 4 + (30000 - 1) * (4 + 45 * 4 + 16)
        assert 0 < info["big"][2] < info["big"][1] and info["raw"][2] == 0
        for k in ("big", "raw"):
            rx, rt = a.read_features(k)
            assert rx.tobytes() == x.tobytes() and rt.tobytes() == t.tobytes()


# ---------------------------------------------------------------- mixture-set estimator ("accumulator") files
def test_accumulator_file_round_trip_and_layout(tmp_path):
    """amx_gmm_accumulator_write / _read (host-only handle: no GPU needed) against oracle/acc_format.py"""
    from oracle import acc_format as af
    from tests import synth
    model = synth.gmm_cart(9, 1, 5, 6, seed=3, pooled=False)
    sc = rasr_amd.GmmFeatureScorer(None, model)
    n = sc.accumulator_size()
    rng = np.random.default_rng(7)
    acc = rng.standard_normal(n) * 100
    nk = int(model["mix_offsets"][-1])
    acc[:nk] = rng.integers(0, 50, nk)
    p = str(tmp_path / "acc.1")
    sc.write_accumulator(acc, p)
    raw = open(p, "rb").read()
    assert raw == af.write(model, acc)
    parsed = af.read(raw)
    assert parsed["version"] == 2 and parsed["dim"] == 6 and len(parsed["means"]) == model["means"].shape[0]
    assert [d for d, _ in parsed["mixtures"][2]] == list(model["dens_index"][model["mix_offsets"][2]:model["mix_offsets"][3]])
    back = sc.read_accumulator(p)
    assert back.tobytes() == acc.tobytes()
    # a file from another topology / a truncated file / a foreign file are refused
    other = rasr_amd.GmmFeatureScorer(None, synth.gmm_cart(9, 1, 5, 6, seed=4, pooled=False))
    with pytest.raises(rasr_amd.AmxError):
        other.read_accumulator(p)
    open(p, "wb").write(raw[:-5])
    with pytest.raises(rasr_amd.AmxError):
        sc.read_accumulator(p)
    open(p, "wb").write(b"SP_ARC1\x00" + raw[8:])
    with pytest.raises(rasr_amd.AmxError) as e:
        sc.read_accumulator(p)
    assert "MIXSET" in str(e.value)
