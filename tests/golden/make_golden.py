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


def run(*args):
    p = subprocess.run([REFGEN] + [str(a) for a in args], stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, check=True)
    return json.loads(p.stderr.decode().strip().splitlines()[-1])


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    if not os.path.exists(REFGEN):
        sys.exit("oracle/_ref/refgen missing: run `make -C oracle ref` where /root/reference exists")
    os.chdir(HERE)
    man = []

    def add(meta, *files):
        meta["files"] = {f: sha(f) for f in files}
        man.append(meta)

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

    json.dump(man, open("manifest.json", "w"), indent=1, sort_keys=True)
    print("wrote %d cases, %d bytes of .bin" % (len(man), sum(os.path.getsize(f) for m in man for f in m["files"])))


if __name__ == "__main__":
    main()
