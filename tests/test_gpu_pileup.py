"""Reads -> columns on the device (csrc/pileup.hip): the pack built in HBM from device-inflated BGZF blocks must
equal cto_pack_from_bam's, array for array and key string for key string, on BAM + BAI files written by tests/bamutil.py.
PARITY UNPINNED against samtools (absent from both boxes); the host reader - itself held to an independent naive pileup and to
hand-derived SAM-specification vectors - is this path's specification."""
import ctypes as C

import numpy as np
import pytest

from bamutil import write_bam
from test_bam_reader import _random_reads, _pack_arrays

pytestmark = pytest.mark.gpu


def _device_arrays(pv, lite):
    from clairs_to_amd._lib import lib, check

    def grab(ptr, n, dt):
        host = np.zeros(int(n), dtype=dt)
        if n:
            check(lib.cto_device_read(ptr, host.ctypes.data, host.nbytes))
        return host
    a = dict(col_pos=grab(pv.col_pos, pv.n_cols, np.int32), col_ref=grab(pv.col_ref, pv.n_cols, np.uint8),
             col_off=grab(pv.col_off, pv.n_cols + 1, np.int64), key_off=grab(pv.key_off, pv.n_cols + 1, np.int32),
             entries=grab(pv.entries, pv.n_entries, np.uint32), key_meta=grab(pv.key_meta, pv.n_keys, np.uint8),
             key_group=grab(pv.key_group, pv.n_keys, np.int32))
    keys = []
    for k in range(pv.n_keys):
        s = C.c_char_p()
        n = lib.cto_pack_key_string(lite, k, C.byref(s))
        keys.append(C.string_at(s, n).decode())
    a["keys"] = keys
    return a


def _unpaired_reads(rng, n, ref_lens, skips=False):
    reads = _random_reads(rng, n, ref_lens, paired_frac=0.0)
    if not skips:
        for r in reads:
            r["cigar"] = [(("D" if op == "N" else op), ln) for op, ln in r["cigar"]]
    return reads


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_pileup_equals_the_host_reader(tmp_path, seed):
    import torch
    from clairs_to_amd._lib import lib
    from clairs_to_amd.bgzf import DevicePileup
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(seed)
    ref_lens = [40000, 3000]
    refs = [("chrA", ref_lens[0]), ("chrB", ref_lens[1])]
    ref_seqs = ["".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=L)) for L in ref_lens]
    reads = _unpaired_reads(rng, 1500, ref_lens)
    bam = str(tmp_path / "t.bam")
    write_bam(bam, refs, reads, block_payload=1500 if seed != 3 else 60000)
    cases = [(0, 1, ref_lens[0], None), (0, 5000, 9000, None), (0, 16380, 16400, None), (1, 1, ref_lens[1], None),
             (1, 700, 2400, [(650, 720), (900, 934), (2000, 2500)]), (0, 39000, 40000, [(38990, 39010)]),
             (0, 2000, 38000, [(k, k + 33) for k in range(2100, 37000, 211)])]
    dp = DevicePileup()
    dev = torch.device("cuda:0")
    done = 0
    for ref_i, start, end, bed in cases:
        name = refs[ref_i][0]
        want = _pack_arrays(ColumnPack.from_bam(bam, name, start, end, ref_seqs[ref_i], 1, bed=bed))
        pv, lite, fallback = dp.pileup(bam, None, name, start, end, ref_seqs[ref_i], 1, dev, bed=bed)
        assert not fallback
        got = _device_arrays(pv, lite)
        lib.cto_pack_free(lite)
        for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
            np.testing.assert_array_equal(got[k], want[k], err_msg="%s %s:%d-%d" % (k, name, start, end))
        assert got["keys"] == want["keys"]
        done += len(want["col_pos"]) > 0
    assert done >= 6


def test_device_pileup_depth_cap_only_where_it_can_bite(tmp_path):
    """--max-depth is a cap on the reads OPEN at a read's start, not on the reads of the region: a region holding many times
    max_depth reads at a depth below it stays on the device (and equals the host reader, which applies the cap itself); the same
    region with a cap below the depth falls back."""
    import torch
    from clairs_to_amd._lib import lib
    from clairs_to_amd.bgzf import DevicePileup
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(21)
    L = 60000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = _unpaired_reads(rng, 2500, [L])
    bam = str(tmp_path / "d.bam")
    write_bam(bam, [("chrA", L)], reads, block_payload=20000)
    dev = torch.device("cuda:0")
    dp = DevicePileup()
    full = _pack_arrays(ColumnPack.from_bam(bam, "chrA", 1, L, ref, 1, max_depth=0))
    deepest = int(np.diff(full["col_off"]).max())
    assert len(reads) > 8 * deepest                      # far more reads in the region than are ever open together
    cap = deepest + 8
    want = _pack_arrays(ColumnPack.from_bam(bam, "chrA", 1, L, ref, 1, max_depth=cap))
    pv, lite, fallback = dp.pileup(bam, None, "chrA", 1, L, ref, 1, dev, max_depth=cap)
    assert not fallback
    got = _device_arrays(pv, lite)
    lib.cto_pack_free(lite)
    for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["keys"] == want["keys"]
    pv, lite, fallback = dp.pileup(bam, None, "chrA", 1, L, ref, 1, dev, max_depth=max(2, deepest // 2))
    assert fallback


def test_device_pileup_reports_what_it_leaves_to_the_host(tmp_path):
    """paired reads, reference skips and a cap that bites come back as `fallback`, never as a different pack"""
    import torch
    from clairs_to_amd.bgzf import DevicePileup
    rng = np.random.default_rng(7)
    refs = [("chrA", 20000)]
    ref = "".join(rng.choice(list("ACGT"), size=20000))
    dev = torch.device("cuda:0")
    dp = DevicePileup()
    for kind in ("paired", "skips", "depth"):
        reads = _random_reads(rng, 300, [20000], paired_frac=0.6 if kind == "paired" else 0.0)
        if kind != "skips":
            for r in reads:
                r["cigar"] = [(("D" if op == "N" else op), ln) for op, ln in r["cigar"]]
        bam = str(tmp_path / (kind + ".bam"))
        write_bam(bam, refs, reads, block_payload=3000)
        pv, lite, fallback = dp.pileup(bam, None, "chrA", 1, 20000, ref, 1, dev, max_depth=3 if kind == "depth" else 8000)
        assert fallback, kind


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_device_pileup_random_regions_and_beds(tmp_path, seed):
    """more shapes of the same comparison: random regions (contig ends, single positions, regions without reads), random BED interval
    sets (touching, one base wide, outside the region), tiny and large BGZF blocks, reads with and without qualities / CG-tag CIGARs"""
    import torch
    from clairs_to_amd._lib import lib
    from clairs_to_amd.bgzf import DevicePileup
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(seed)
    L = int(rng.integers(6000, 30000))
    refs = [("c0", 2000), ("chrQ", L), ("c2", 1500)]
    ref_seqs = ["".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=n)) for _, n in refs]
    reads = _unpaired_reads(rng, int(rng.integers(200, 1800)), [2000, L, 1500])
    write_bam(str(tmp_path / "t.bam"), refs, reads, block_payload=int(rng.choice([400, 2000, 9000, 60000])))
    bam = str(tmp_path / "t.bam")
    dp = DevicePileup()
    dev = torch.device("cuda:0")
    nonempty = 0
    for case in range(10):
        ri = int(rng.choice([0, 1, 1, 1, 2]))
        n = refs[ri][1]
        a = int(rng.integers(1, n + 1))
        b = int(min(n, a + rng.choice([0, 1, 50, 700, 5000, 40000])))
        bed = None
        if rng.random() < 0.6:
            k = int(rng.integers(1, 40))
            starts = np.sort(rng.integers(max(0, a - 200), b + 200, size=k))
            bed, last = [], -1
            for s0 in starts.tolist():
                s0 = max(s0, last)
                e0 = s0 + int(rng.choice([1, 2, 33, 34, 120]))
                bed.append((s0, e0))
                last = e0 if rng.random() < 0.7 else e0 + int(rng.integers(0, 50))
        want = _pack_arrays(ColumnPack.from_bam(bam, refs[ri][0], a, b, ref_seqs[ri], 1, bed=bed))
        pv, lite, fallback = dp.pileup(bam, None, refs[ri][0], a, b, ref_seqs[ri], 1, dev, bed=bed)
        if fallback and pv is None:                    # nothing in the index for the region: the host reader says "no columns" too
            assert len(want["col_pos"]) == 0
            continue
        assert not fallback
        got = _device_arrays(pv, lite)
        lib.cto_pack_free(lite)
        for key in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
            np.testing.assert_array_equal(got[key], want[key], err_msg="%s seed %d case %d %s:%d-%d" % (key, seed, case, refs[ri][0], a, b))
        assert got["keys"] == want["keys"]
        nonempty += len(want["col_pos"]) > 0
    assert nonempty >= 3


def test_device_pileup_checks_the_bgzf_crc(tmp_path):
    """one flipped bit inside a block's payload: the DEFLATE stream breaks (inflate status) or the inflated bytes fail the gzip
    trailer's CRC-32 (k_crc32_blocks) - the device path never succeeds with different bytes, like the host reader"""
    import torch
    from clairs_to_amd._lib import CtoError, lib
    from clairs_to_amd.bgzf import DevicePileup
    rng = np.random.default_rng(5)
    reads = _unpaired_reads(rng, 150, [5000])
    bam = str(tmp_path / "ok.bam")
    write_bam(bam, [("chrA", 5000)], reads, block_payload=4000)
    ref = "".join(rng.choice(list("ACGT"), size=5000))
    dev = torch.device("cuda:0")
    dp = DevicePileup()
    pv, lite, fb = dp.pileup(bam, None, "chrA", 1, 5000, ref, 1, dev)
    good = _device_arrays(pv, lite)["entries"]
    lib.cto_pack_free(lite)
    raw = open(bam, "rb").read()
    n_err = n_same = 0
    for off in range(18 + 8 + 200, len(raw) - 60, max(1, (len(raw) - 300) // 40)):
        b = bytearray(raw)
        b[off] ^= 0x08
        p = tmp_path / "flip.bam"
        p.write_bytes(bytes(b))
        (tmp_path / "flip.bam.bai").write_bytes(open(bam + ".bai", "rb").read())
        try:
            pv, lite, fb = dp.pileup(str(p), None, "chrA", 1, 5000, ref, 1, dev)
            if pv is not None and not fb:
                assert np.array_equal(_device_arrays(pv, lite)["entries"], good)      # only a flip in bytes nobody reads may pass
                lib.cto_pack_free(lite)
            n_same += 1
        except CtoError:
            n_err += 1
    assert n_err >= 25 and n_same <= 10
