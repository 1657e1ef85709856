#!/usr/bin/env python3
"""Regenerate tests/golden/*.bin + manifest.json by running the REFERENCE's generator.

Runs only in the build container: needs oracle/_ref/refgen, which oracle/Makefile compiles from
/root/reference/src/generator_ETHZ.cu (unmodified).  The outputs are data (raw int32 relations in
the reference's .bin format + the seeds they were made under), never reference source.

The unique-key generator of the reference is seeded with time(NULL) (gen.cu:133-135); refgen
reports the second it ran in, and that seed is stored in the manifest so the oracle restatement can
be replayed against the committed bytes.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFGEN = os.path.join(ROOT, "oracle", "_ref", "refgen")


def run_refgen(*args):
    p = subprocess.run([REFGEN] + [str(a) for a in args], stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, check=True)
    return json.loads(p.stderr.decode().strip().splitlines()[-1])


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


REFJOIN = os.path.join(ROOT, "oracle", "_ref", "refjoin")

# (R file, S file) pairs joined by the REFERENCE's own CPU join
JOIN_PAIRS = [
    ("unique_4096.bin", "unique_4096.bin"),
    ("unique_4096.bin", "unique_fk10000_max4096.bin"),
    ("unique_16.bin", "unique_16.bin"),
    ("unique_16.bin", "unique_fk40_max16.bin"),
    ("unique_16.bin", "unique_16_x3.bin"),
    ("unique_16_x3.bin", "unique_fk40_max16.bin"),
    ("unique_4096.bin", "zipf_S20000_a4096_t1.0_seed42.bin"),
    ("unique_4096.bin", "zipf_S5000_a512_t0.5_seed43.bin"),
    ("nonuniq_R6000_seed7.bin", "nonuniq_S9000_seed8.bin"),
    ("nonuniq_S9000_seed8.bin", "nonuniq_R6000_seed7.bin"),
    ("pk_R3000_seed11.bin", "fk_S7000_pk_R3000_seed11.bin"),
    ("unique_65536.bin", "unique_65536.bin"),
    ("unique_65536.bin", "zipf_S100000_a65536_t1.0_seed44.bin"),
    ("nonuniq_R65536_seed21.bin", "nonuniq_S131072_seed22.bin"),
    ("unique_65536.bin", "nonuniq_S131072_seed22.bin"),
]


def join_answers():
    """Run the reference's joinCpu (hash_join_clustered_probe.cu:2013-2059, compiled from where it lies:
    oracle/_ref/refjoin) on every pair and record what it prints: s = matching pairs, g = sum of the
    matching S keys mod 2^32, c = build + probe tuple count.  OMP_NUM_THREADS=1: `s` is not in the
    reduction clause of the reference's probe loop."""
    if not os.path.exists(REFJOIN):
        sys.exit("oracle/_ref/refjoin missing: run `make -C oracle ref` where /root/reference exists")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = []
    for r, s in JOIN_PAIRS:
        p = subprocess.run([REFJOIN, r, s], stdout=subprocess.PIPE, check=True, env=env, cwd=HERE)
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("===")]
        build = int(lines[0][3:])
        sv, cv, gv = (int(x) for x in lines[1][3:].split())
        out.append({"R": r, "S": s, "s": sv, "c": cv, "g": gv, "build": build})
    json.dump({"source": "joinCpu, /root/reference/src/hash_join_clustered_probe.cu:2013-2059, via oracle/_ref/refjoin, "
                         "OMP_NUM_THREADS=1", "joins": out},
              open(os.path.join(HERE, "join_answers.json"), "w"), indent=1, sort_keys=True)
    print("wrote %d reference join answers" % len(out))


def main():
    if not os.path.exists(REFGEN):
        sys.exit("oracle/_ref/refgen missing: run `make -C oracle ref` where /root/reference exists")
    os.chdir(HERE)
    man = []
    old = json.load(open("manifest.json")) if os.path.exists("manifest.json") else []
    kept = {tuple(sorted(m["files"])): m for m in old
            if all(os.path.exists(f) and sha(f) == h for f, h in m["files"].items())}

    def add(gen, *files):
        """Cases whose committed files are intact are kept as they are (the unique-key generator is
        time-seeded: regenerating would change them); only missing cases run the reference generator."""
        key = tuple(sorted(files))
        if key in kept:
            man.append(kept[key])
            return
        meta = gen()
        meta["files"] = {f: sha(f) for f in files}
        man.append(meta)

    def run(*args):  # deferred: only executed for cases that are not already committed
        return lambda: run_refgen(*args)

    # bench -R N -S N, unique keys: R = perm(0..N-1), S re-read from the same file (main.cu:135,143)
    add(run("unique", 4096, 4096, "unique_4096.bin"), "unique_4096.bin")
    # bench -R 4096 -S 10000: S cycles 0,1..4096,1..4096,... shuffled (gen.cu:137-144)
    add(run("unique", 10000, 4096, "unique_fk10000_max4096.bin"), "unique_fk10000_max4096.bin")
    # tiny known-answer cases
    add(run("unique", 16, 16, "unique_16.bin"), "unique_16.bin")
    add(run("unique", 40, 16, "unique_fk40_max16.bin"), "unique_fk40_max16.bin")
    # --non-unique: uniform in [0,|R|/2) via rand() (main.cu:251-261)
    add(run("nonuniq", 7, 6000, 3000, "nonuniq_R6000_seed7.bin"), "nonuniq_R6000_seed7.bin")
    add(run("nonuniq", 8, 9000, 3000, "nonuniq_S9000_seed8.bin"), "nonuniq_S9000_seed8.bin")
    # -s 1.0: Zipf foreign keys over alphabet 1..4096 (gen.cu:299-348)
    add(run("zipf", 42, 20000, 4096, 1.0, "zipf_S20000_a4096_t1.0_seed42.bin"),
        "zipf_S20000_a4096_t1.0_seed42.bin")
    add(run("zipf", 43, 5000, 512, 0.5, "zipf_S5000_a512_t0.5_seed43.bin"),
        "zipf_S5000_a512_t0.5_seed43.bin")
    # --full-range: PK uniform in [0,INT_MAX), FK = PK repeated + Knuth shuffle (main.cu:190-201)
    add(run("fkpk", 11, 3000, 2147483647, 7000, "pk_R3000_seed11.bin", "fk_S7000_pk_R3000_seed11.bin"),
        "pk_R3000_seed11.bin", "fk_S7000_pk_R3000_seed11.bin")
    # -y 3: create_relation_n (gen.cu:97-110)
    add(run("repeat", 16, 3, "unique_16.bin", "unique_16_x3.bin"), "unique_16_x3.bin")

    # medium cases (multi-pass radix paths on the GPU): 2^16 unique keys, Zipf / non-unique probe sides
    add(run("unique", 65536, 65536, "unique_65536.bin"), "unique_65536.bin")
    add(run("zipf", 44, 100000, 65536, 1.0, "zipf_S100000_a65536_t1.0_seed44.bin"),
        "zipf_S100000_a65536_t1.0_seed44.bin")
    add(run("nonuniq", 21, 65536, 20000, "nonuniq_R65536_seed21.bin"), "nonuniq_R65536_seed21.bin")
    add(run("nonuniq", 22, 131072, 20000, "nonuniq_S131072_seed22.bin"), "nonuniq_S131072_seed22.bin")

    json.dump(man, open("manifest.json", "w"), indent=1, sort_keys=True)
    join_answers()
    print("wrote %d cases, %d bytes of .bin" % (len(man), sum(os.path.getsize(f) for m in man for f in m["files"])))


if __name__ == "__main__":
    main()
