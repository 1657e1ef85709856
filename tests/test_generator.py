"""Product generator (csrc/gen_ethz.cpp, the generator_ETHZ drop-in) vs the reference generator's
committed outputs (tests/golden) and vs the oracle restatement; .bin cache semantics; `bench -b 8`."""
import os
import subprocess

import numpy as np
import pytest

from hjtest import pkg
from oracle import pyoracle as o


def _load(golden_dir, name):
    return np.fromfile(os.path.join(golden_dir, name), dtype=np.int32)


def test_product_generator_reproduces_reference_outputs(manifest, golden_dir):
    g = pkg().generator
    checked = 0
    for m in manifest:
        files = sorted(m["files"])
        if m["mode"] == "unique":
            g.seed_generator(m["time_seed"])
            got = g.create_relation_unique(None, m["n"], m["maxid"])
            assert np.array_equal(got, _load(golden_dir, files[0])), files[0]
        elif m["mode"] == "nonuniq":
            g.seed_generator(m["seed"])
            assert np.array_equal(g.create_relation_nonunique(None, m["n"], m["maxid"]), _load(golden_dir, files[0]))
        elif m["mode"] == "zipf":
            g.seed_generator(m["seed"])
            assert np.array_equal(g.create_relation_zipf(None, m["n"], m["alphabet"], m["theta"]),
                                  _load(golden_dir, files[0]))
        elif m["mode"] == "fkpk":
            g.seed_generator(m["seed"])
            pk = g.create_relation_nonunique(None, m["npk"], m["maxid"])
            fk = g.create_relation_fk_from_pk(None, m["nfk"], pk)
            assert np.array_equal(pk, _load(golden_dir, [f for f in files if f.startswith("pk_")][0]))
            assert np.array_equal(fk, _load(golden_dir, [f for f in files if f.startswith("fk_")][0]))
        elif m["mode"] == "repeat":
            base = _load(golden_dir, "unique_16.bin")
            assert np.array_equal(g.create_relation_n(base, m["times"]), _load(golden_dir, files[0]))
        checked += 1
    assert checked == len(manifest)


@pytest.mark.parametrize("seed", [1, 2, 12345, 0x7FFFFFFF, 0xFFFFFFFF])
def test_own_prngs_match_libc(seed):
    """GlibcRand / Rand48 (csrc/gen_ethz.cpp) against libc through the oracle, on other seeds."""
    g = pkg().generator
    g.seed_generator(seed)
    o.seed_generator(seed)
    assert np.array_equal(g.create_relation_nonunique(None, 5000, 1 << 30), o.random_gen(5000, 1 << 30))
    g.seed_generator(seed)
    assert np.array_equal(g.create_relation_unique(None, 3000, 777), o.random_unique_gen(3000, 777, seed))
    g.seed_generator(seed)
    o.seed_generator(seed)
    assert np.array_equal(g.create_relation_zipf(None, 4000, 300, 0.9), o.gen_zipf(4000, 300, 0.9))


def test_bin_cache_semantics(tmp_path):
    g = pkg().generator
    f = str(tmp_path / "unique_1000.bin")
    g.seed_generator(5)
    a = g.create_relation_unique(f, 1000, 1000)
    assert os.path.getsize(f) == 4000                      # raw int32, no header (gen.cu:48,65)
    g.seed_generator(6)                                     # a different seed …
    b = g.create_relation_unique(f, 1000, 1000)             # … but the cache file wins (gen.cu:86-94)
    assert np.array_equal(a, b)
    assert np.array_equal(g.readFromFile(f, 1000), a)
    with pytest.raises(IOError):
        g.readFromFile(f, 1001)                             # short file is an error (reference: D12)
    with pytest.raises(IOError):
        g.readFromFile(str(tmp_path / "missing.bin"), 1)


def test_bench_cli_generate_only(tmp_path):
    """`bench -b 8` = generate + cache the relations, no join (src/main.cu:264) — runs without a GPU."""
    bench = pkg()._lib.BENCH_PATH
    r = subprocess.run([bench, "-b", "8", "-R", "4096", "-S", "10000", "--seed", "77"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "INPUT: option = 8" in r.stdout and "||R|| = 4096" in r.stdout and "||S|| = 10000" in r.stdout
    assert "Creating relation R with 4096 tuples (0 MB) using unique keys" in r.stdout
    R = np.fromfile(tmp_path / "unique_4096.bin", np.int32)   # main.cu:135
    S = np.fromfile(tmp_path / "unique_10000.bin", np.int32)  # main.cu:143
    assert np.array_equal(R, o.random_unique_gen(4096, 4096, 77))
    assert np.array_equal(S, o.random_unique_gen(10000, 4096, 77))
    # same sizes: S is re-read from R's cache file → S ≡ R
    r = subprocess.run([bench, "-b", "8", "-R", "4096", "-S", "4096", "--seed", "78"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and len(os.listdir(tmp_path)) == 2
    # other modes name their cache files as main.cu:121-158 does
    for extra, names in [(["--non-unique"], ["nonUnique_R300.bin", "nonUnique_S500.bin"]),
                         (["--full-range"], ["pk_R300.bin", "fk_S500_pk_R300.bin"]),
                         (["-s", "0.75"], ["unique_300.bin", "unique_skew0.75_S500.bin"])]:
        r = subprocess.run([bench, "-b", "8", "-R", "300", "-S", "500", "--seed", "9"] + extra, cwd=tmp_path,
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        for n in names:
            assert os.path.exists(tmp_path / n), n
    z = np.fromfile(tmp_path / "unique_skew0.75_S500.bin", np.int32)
    o.seed_generator(9)
    assert np.array_equal(z, o.gen_zipf(500, 300, 0.75))


def test_bench_cli_rejects_bad_input(tmp_path):
    bench = pkg()._lib.BENCH_PATH
    assert subprocess.run([bench, "-b", "3"], cwd=tmp_path, capture_output=True).returncode == 1
    assert subprocess.run([bench, "-b", "7", "-a", "NLJ", "-R", "8", "-S", "8"], cwd=tmp_path,
                          capture_output=True).returncode == 1          # unknown algorithm (reference: UB, D8)
    assert subprocess.run([bench, "-b", "7", "-R", "8", "-S", "8"], cwd=tmp_path,
                          capture_output=True).returncode == 1          # -a omitted
