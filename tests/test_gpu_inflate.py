"""Device BGZF / DEFLATE decompression (csrc/inflate.hip) against zlib: stored, fixed-code and dynamic-code blocks, literal-heavy and
match-heavy data, overlapping copies, maximum-size blocks, many blocks per launch, malformed input."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    assert len(data) <= 65280      # what BGZF writers put into one block (BSIZE is 16 bits)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = co.compress(data) + co.flush()
    bsize = len(payload) + 25 + 1
    assert bsize <= 65536
    head = b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + payload + struct.pack("<II", zlib.crc32(data), len(data))


def cases():
    rng = np.random.default_rng(7)
    out = []
    out.append(b"")                                                        # the EOF marker's kind
    out.append(b"A")
    out.append(b"hello, hello, hello, hello!")                             # fixed codes, overlapping copy
    out.append(bytes(rng.integers(0, 256, 60000, dtype=np.uint8)))         # incompressible: stored or literal-only
    out.append(bytes(rng.integers(33, 74, 65280, dtype=np.uint8)))         # quality-string like, a full block
    out.append(b"\0" * 65280)                                              # one long run: distance 1, length 258
    out.append((b"ACGT" * 20000)[:65000])                                  # short period
    out.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 50000)))
    text = (b"read%07d\t99\tchr1\t%d\t60\t100M\t=\t%d\t300\t" % (1, 2, 3)) * 900
    out.append(text[:64000])
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(300)]
    out.append(b" ".join(words[int(i)] for i in rng.integers(0, 300, 9000))[:65000])   # far matches, many lengths
    period = bytes(rng.integers(0, 256, 32768, dtype=np.uint8))
    out.append(period + period[:3000])                                     # distance 32768: the far edge of the window
    return out


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_matches_zlib(dev, level):
    from clairs_to_amd.bgzf import inflate_bytes
    data = cases()
    raw = b"".join(bgzf_block(d, level) for d in data)
    got = inflate_bytes(raw, dev)
    assert len(got) == len(data)
    for i, (g, d) in enumerate(zip(got, data)):
        assert g == d, "block %d (level %d, %d bytes) differs at byte %d" % (
            i, level, len(d), next((k for k in range(min(len(g), len(d))) if g[k] != d[k]), min(len(g), len(d))))


@pytest.mark.parametrize("strategy", [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
def test_inflate_strategies(dev, strategy):
    from clairs_to_amd.bgzf import inflate_bytes
    data = cases()
    raw = b"".join(bgzf_block(d, 6, strategy) for d in data)
    got = inflate_bytes(raw, dev)
    assert got == data


def test_inflate_many_blocks(dev):
    """a launch of a few thousand blocks of mixed sizes (the shape of a chunk's byte range)"""
    from clairs_to_amd.bgzf import inflate_bytes
    rng = np.random.default_rng(11)
    data = []
    for _ in range(1500):
        n = int(rng.integers(1, 65280))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            d = bytes(rng.integers(33, 74, n, dtype=np.uint8))
        elif kind == 1:
            d = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n))
        else:
            d = (bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)) * (n // 1 + 1))[:n]
        data.append(d)
    raw = b"".join(bgzf_block(d, int(rng.integers(1, 10))) for d in data)
    assert inflate_bytes(raw, dev) == data


def test_inflate_rejects_malformed(dev):
    """corrupt payloads end with a status code (CtoError), never with a hang or a wrong answer"""
    from clairs_to_amd._lib import CtoError
    from clairs_to_amd.bgzf import inflate_bytes
    rng = np.random.default_rng(3)
    good_text = bytes(rng.integers(33, 74, 40000, dtype=np.uint8))
    good = bgzf_block(good_text, 6)
    # literal-heavy (quality-string like), match-heavy (tab-separated records: the hand-written match path, far and near distances) and
    # fixed-code streams, one to four flipped bits each - several damaged blocks per launch, next to intact ones that must still come out right
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(300)]
    texts = [bytes(rng.integers(33, 74, 40000, dtype=np.uint8)),
             b" ".join(words[int(i)] for i in rng.integers(0, 300, 9000))[:60000],
             (b"read%07d\t99\tchr1\t%d\t60\t100M\t=\t%d\t300\t" % (1, 2, 3)) * 800,
             bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 30000))]
    goods = [bgzf_block(t, lvl, strat) for t in texts for lvl, strat in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED))]
    n_err = n_trials = 0
    for trial in range(60):
        blocks = []
        for g in goods:
            b = bytearray(g)
            lo, hi = 18, len(b) - 8
            for _ in range(1 + trial % 4):
                b[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
            blocks.append(bytes(b))
        for k in range(0, len(blocks), 4):            # four damaged blocks in front of an intact one per launch
            import torch
            from clairs_to_amd import bgzf
            raw = b"".join(blocks[k:k + 4]) + good
            host = np.zeros(len(raw) + bgzf.BGZF_PAD, dtype=np.uint8)
            host[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            tbl, out_bytes = bgzf.scan(host, len(raw))
            d_out, d_status = bgzf.inflate_device(torch.from_numpy(host).to(dev), tbl, out_bytes, dev)
            torch.cuda.synchronize(dev)
            st, out = d_status.cpu().numpy(), d_out.cpu().numpy()
            assert len(tbl) == 5 and st[4] == 0
            # a damaged block may run at most CTO_BGZF_SLOT_PAD bytes past its size (a literal run's bound is checked per refill):
            # the intact block behind four damaged ones is untouched
            assert out[int(tbl[4]["out_off"]):int(tbl[4]["out_off"]) + int(tbl[4]["isize"])].tobytes() == good_text
            for j in range(4):
                n_trials += 1
                got = out[int(tbl[j]["out_off"]):int(tbl[j]["out_off"]) + int(tbl[j]["isize"])].tobytes()
                n_err += int(st[j] != 0 or zlib.crc32(got) != int(tbl[j]["crc32"]))
    assert n_err == n_trials   # a flipped bit either breaks the stream (status code) or decodes to other bytes (CRC-32)
    assert inflate_bytes(good + goods[3], dev) == [good_text, texts[1]]
    # a stored block that claims more bytes than the payload holds
    payload = b"\x01" + struct.pack("<HH", 60000, 60000 ^ 0xffff) + b"x" * 100
    bsize = len(payload) + 26
    lying = (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + payload +
             struct.pack("<II", 0, 60000))
    with pytest.raises(CtoError):
        inflate_bytes(lying, dev)
    garbage = bytearray(good)
    garbage[18:len(garbage) - 8] = bytes(rng.integers(0, 256, len(garbage) - 26, dtype=np.uint8))
    with pytest.raises(CtoError):
        inflate_bytes(bytes(garbage), dev)


def test_inflate_refuses_a_slot_table_without_the_pad(dev):
    """the literal path may store up to 32 bytes behind a block's size (its bound is checked once per refill): a table whose slots
    lie closer than CTO_BGZF_SLOT_PAD to each other - the `isize + 4` rule of an older layout - is refused block by block (status 9)
    instead of letting a block write into its neighbour's output"""
    import torch
    from clairs_to_amd import bgzf
    rng = np.random.default_rng(5)
    datas = [bytes(rng.integers(33, 74, 30000, dtype=np.uint8)) for _ in range(3)]
    raw = b"".join(bgzf_block(d, 6) for d in datas)
    host = np.zeros(len(raw) + bgzf.BGZF_PAD, dtype=np.uint8)
    host[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    tbl, out_bytes = bgzf.scan(host, len(raw))
    assert all(int(tbl[i + 1]["out_off"]) >= int(tbl[i]["out_off"]) + int(tbl[i]["isize"]) + 64 for i in range(2))   # CTO_BGZF_SLOT_PAD
    tight = tbl.copy()
    off = 0
    for i in range(3):
        tight[i]["out_off"] = off
        off += int(tight[i]["isize"]) + 4
    d_out, d_status = bgzf.inflate_device(torch.from_numpy(host).to(dev), tight, out_bytes, dev)
    torch.cuda.synchronize(dev)
    st = d_status.cpu().numpy()
    assert st[0] == 9 and st[1] == 9 and st[2] == 0        # the last slot's pad is the caller's buffer size
    out = d_out.cpu().numpy()
    assert out[int(tight[2]["out_off"]):int(tight[2]["out_off"]) + 30000].tobytes() == datas[2]


def test_bam_pack_device_inflate_equals_host_inflate(dev, tmp_path):
    """the pack of a multi-megabyte BAM region: BGZF blocks inflated on the device (bgzf.inflate_span) vs on the host, every array"""
    from clairs_to_amd.bgzf import inflate_span
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth_run import make_bam_run
    run = make_bam_run(str(tmp_path / "run"), region_kb=400, n_chunks=2)
    from clairs_to_amd.fasta import read_region
    ref = read_region(run["ref_fn"], "chr1", 1, 400000, as_bytes=True)
    for (lo, hi) in ((1, 400000), (150001, 260000)):
        bed = [(p - 17, p + 16) for p in range(max(lo, 500), hi - 500, 250)]
        host = ColumnPack.from_bam(run["bam_fn"], "chr1", lo, hi, ref, 1, bed=bed)
        inflated = inflate_span(run["bam_fn"], None, "chr1", lo, hi, dev)
        assert len(inflated[1]) > 10
        gpu = ColumnPack.from_bam(run["bam_fn"], "chr1", lo, hi, ref, 1, bed=bed, inflated=inflated)
        a, b = host.numpy(), gpu.numpy()
        assert len(a["col_pos"]) > 1000
        for k in a:
            assert np.array_equal(a[k], b[k]), k
