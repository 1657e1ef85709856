"""Oracle generator restatement vs (a) the committed outputs of the reference generator
(tests/golden, made by tests/golden/make_golden.py) and (b) the reference generator run live when
oracle/_ref/refgen exists (this container only; skipped on the GPU box)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as o


def _load(golden_dir, name):
    return np.fromfile(os.path.join(golden_dir, name), dtype=np.int32)


def _replay(m):
    """Re-create a manifest case with the oracle restatement; returns {filename: array}."""
    files = sorted(m["files"])
    if m["mode"] == "unique":
        return {files[0]: o.random_unique_gen(m["n"], m["maxid"], m["time_seed"])}
    if m["mode"] == "nonuniq":
        o.seed_generator(m["seed"])
        return {files[0]: o.random_gen(m["n"], m["maxid"])}
    if m["mode"] == "zipf":
        o.seed_generator(m["seed"])
        return {files[0]: o.gen_zipf(m["n"], m["alphabet"], m["theta"])}
    if m["mode"] == "fkpk":
        o.seed_generator(m["seed"])
        pk = o.random_gen(m["npk"], m["maxid"])
        fk = o.fk_from_pk(pk, m["nfk"])
        return {[f for f in files if f.startswith("pk_")][0]: pk,
                [f for f in files if f.startswith("fk_")][0]: fk}
    if m["mode"] == "repeat":
        return None
    raise AssertionError(m["mode"])


def test_golden_files_intact(manifest, golden_dir):
    for m in manifest:
        for f, digest in m["files"].items():
            assert hashlib.sha256(open(os.path.join(golden_dir, f), "rb").read()).hexdigest() == digest


def test_oracle_reproduces_reference_outputs(manifest, golden_dir):
    n = 0
    for m in manifest:
        got = _replay(m)
        if got is None:
            continue
        for f, arr in got.items():
            assert np.array_equal(arr, _load(golden_dir, f)), f
            n += 1
    assert n >= 9


def test_create_relation_n(golden_dir):
    base = _load(golden_dir, "unique_16.bin")
    assert np.array_equal(o.create_relation_n(base, 3), _load(golden_dir, "unique_16_x3.bin"))


def test_unique_generator_structure(golden_dir):
    # gen.cu:137-144: 0 once, then 1..maxid cycling; R of N tuples with maxid=N is perm(0..N-1)
    r = _load(golden_dir, "unique_4096.bin")
    assert np.array_equal(np.sort(r), np.arange(4096))
    s = np.sort(_load(golden_dir, "unique_fk40_max16.bin"))
    vals, cnt = np.unique(s, return_counts=True)
    assert vals.tolist() == list(range(17))
    assert cnt[0] == 1 and cnt[16] == 2 and cnt[1:8].tolist() == [3] * 7 and cnt[8:16].tolist() == [2] * 8


def test_zipf_range(golden_dir):
    z = _load(golden_dir, "zipf_S20000_a4096_t1.0_seed42.bin")
    assert z.min() >= 1 and z.max() <= 4096  # alphabet is i+1: no zeros (gen.cu:245)


def test_bin_roundtrip(tmp_path):
    a = o.random_unique_gen(1000, 1000, 99)
    p = str(tmp_path / "x.bin")
    o.write_bin(p, a)
    assert os.path.getsize(p) == 4000  # raw int32, no header (gen.cu:48,65)
    assert np.array_equal(o.read_bin(p, 1000), a)
    with pytest.raises(IOError):
        o.read_bin(p, 1001)  # short read is an error here (reference ignores it, D12)


@pytest.mark.skipif(o.refgen_path() is None, reason="oracle/_ref/refgen only exists where /root/reference does")
def test_live_reference_generator(tmp_path):
    """Run the reference generator now, under whatever time() seed it gets, and replay it."""
    refgen = o.refgen_path()

    def run(*args):
        p = subprocess.run([refgen] + [str(a) for a in args], stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, check=True)
        return json.loads(p.stderr.decode().strip().splitlines()[-1])

    out = str(tmp_path / "u.bin")
    for n, maxid in [(1, 1), (2, 2), (1000, 1000), (5000, 1234), (65536, 65536)]:
        m = run("unique", n, maxid, out)
        assert np.array_equal(np.fromfile(out, np.int32), o.random_unique_gen(n, maxid, m["time_seed"]))
    for seed, n, maxid in [(1, 100, 50), (12345, 30000, 2147483647)]:
        run("nonuniq", seed, n, maxid, out)
        o.seed_generator(seed)
        assert np.array_equal(np.fromfile(out, np.int32), o.random_gen(n, maxid))
    run("zipf", 5, 3000, 100, 0.75, out)
    o.seed_generator(5)
    assert np.array_equal(np.fromfile(out, np.int32), o.gen_zipf(3000, 100, 0.75))
